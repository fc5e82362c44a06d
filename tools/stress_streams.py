"""Determinism stress of the sub-batch stream split (GPU box): repeated match() with 2 HIP streams against the
single-stream result of the same handle; reports the runs / pairs / bounding boxes that differ.

    python tools/stress_streams.py --pairs 3 --runs 400 [--fuse 0|1] [--amp bf16|f16|mixed|f32] [--seed S] [--trace]
Kernel-selection switches come from the environment (ROMA_GEMM8P, ROMA_LC_MODE, ROMA_RI_VEC, ROMA_STREAMS_SERIAL ...),
so one GPU visit can run a matrix of configurations (tools/r02_visit2.sh)."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from roma_amd import _lib, roma_model, synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=3)
ap.add_argument("--runs", type=int, default=400)
ap.add_argument("--fuse", type=int, default=1)
ap.add_argument("--amp", default="bf16")
ap.add_argument("--res", type=int, nargs=2, default=[112, 168])
ap.add_argument("--seed", type=int, default=7, help="seed of the synthetic image pairs")
ap.add_argument("--trace", action="store_true", help="per-stage checksums: name the first stage that deviates from the majority")
args = ap.parse_args()

lib = _lib.load()
sd, dsd = synthetic.make_matcher_state_dict(0), synthetic.make_dinov2_state_dict(0)
NB = args.pairs
amp = {"bf16": torch.bfloat16, "f32": torch.float32, "f16": torch.float16, "mixed": torch.bfloat16}[args.amp]
env = {k: v for k, v in os.environ.items() if k.startswith("ROMA_")}
inp = {k: v.cuda() for k, v in synthetic.make_inputs(NB, args.res[0], args.res[1], seed=args.seed).items()}
m = roma_model((args.res[0],) * 2, True, device="cuda:0", weights=sd, dinov2_weights=dsd, amp_dtype=amp, symmetric=True,
               upsample_res=(args.res[1],) * 2, max_batch=NB, decoder_dtype=torch.float16 if args.amp == "mixed" else None)
lib = m._lib  # the library of this mode
_lib.check(lib.roma_set_option(m._handle, b"fuse_refiner_blocks", args.fuse))
kw = dict(im_A_high_res=inp["im_A_high_res"], im_B_high_res=inp["im_B_high_res"])
m.dual_stream = False
w1, c1 = m.match(inp["im_A"], inp["im_B"], **kw)
w1b, c1b = m.match(inp["im_A"], inp["im_B"], **kw)
torch.cuda.synchronize()
single_repro = bool(torch.equal(w1, w1b) and torch.equal(c1, c1b))
m.dual_stream = True
m.trace = args.trace
bad = []
traces = []
t0 = time.time()
for i in range(args.runs):
    w, c = m.match(inp["im_A"], inp["im_B"], **kw)
    if args.trace:
        traces.append([m.debug_trace(s) for s in (0, 1)])
    ne = c != c1
    if bool(ne.any()) or not torch.equal(w, w1):
        per = []
        for b in range(NB):
            nw = (w[b] != w1[b]).any(-1)
            bad_px = ne[b] | nw
            if bool(bad_px.any()):
                idx = bad_px.nonzero()
                per.append((b, int(bad_px.sum()), idx[:, 0].min().item(), idx[:, 0].max().item(), idx[:, 1].min().item(),
                            idx[:, 1].max().item(), float((c[b] - c1[b]).abs().max()), float((w[b] - w1[b]).abs().max())))
        bad.append((i, per))
print(f"amp={args.amp} fuse={args.fuse} pairs={NB} env={env}: single-stream reproducible={single_repro}; "
      f"mismatching dual-stream runs {len(bad)}/{args.runs} in {time.time() - t0:.1f}s; "
      f"(run, [(pair, npix, r0, r1, c0, c1, dcert, dwarp)]): {bad[:5]}", flush=True)

if args.trace and traces:
    import collections
    for slot in (0, 1):
        names = traces[0][slot][0]
        sums = [tr[slot][1] for tr in traces]
        n = min(len(s) for s in sums)
        major = []
        for k in range(n):
            major.append(collections.Counter(int(s[k]) for s in sums).most_common(1)[0][0])
        first = collections.Counter()
        ndev = 0
        for ri, s in enumerate(sums):
            dev = [k for k in range(n) if int(s[k]) != major[k]]
            if dev:
                ndev += 1
                first[names[dev[0]]] += 1
                if ndev <= 4:  # the whole chain of the first few deviating runs: does the deviation propagate or heal?
                    print(f"  slot {slot} run {ri}: {len(dev)} deviating stages: {[names[k] for k in dev[:12]]}", flush=True)
        print(f"trace slot {slot}: {n} stages; runs deviating from the per-stage majority: {ndev}/{len(sums)}; first deviating stage: {dict(first)}", flush=True)

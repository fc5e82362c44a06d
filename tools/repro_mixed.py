"""Round-4 debugging aid: the ROMA_MIXED handle was not reproducible inside tests/test_gpu_parity.py (2 of 3 runs of the
five-call sequence dual, dual, single, dual, single) but is in tools/stress_streams.py.  Replays the test's sequence with
options: which other handles exist, seed, per-stage checksum trace.

    python tools/repro_mixed.py [--others 0|1] [--seed 1] [--rounds 10] [--trace] [--amp mixed]"""
import argparse
import collections
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from roma_amd import roma_model, synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--others", type=int, default=1)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--rounds", type=int, default=10)
ap.add_argument("--trace", action="store_true")
ap.add_argument("--amp", default="mixed")
ap.add_argument("--fuse", type=int, default=1)
ap.add_argument("--dual-only", action="store_true", help="every call on two streams (after one single-stream reference call)")
args = ap.parse_args()
sd, dsd = synthetic.make_matcher_state_dict(0), synthetic.make_dinov2_state_dict(0)


def build(name):
    amp = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16, "mixed": torch.bfloat16}[name]
    return roma_model((560, 560), True, device="cuda:0", weights=sd, dinov2_weights=dsd, amp_dtype=amp, symmetric=True,
                      upsample_res=(864, 864), max_batch=8, decoder_dtype=torch.float16 if name == "mixed" else None)


models = {}
if args.others:
    for n in ("f32", "bf16", "f16"):
        if n != args.amp:
            models[n] = build(n)
m = build(args.amp)
inp = {k: v.cuda() for k, v in synthetic.make_inputs(8, 560, 864, seed=args.seed).items()}
kw = dict(im_A_high_res=inp["im_A_high_res"], im_B_high_res=inp["im_B_high_res"])
m.dual_stream = False
_w0, _c0 = m.match(inp["im_A"], inp["im_B"], **kw)
torch.cuda.synchronize()
m.trace = args.trace
from roma_amd import _lib  # noqa: E402
_lib.check(m._lib.roma_set_option(m._handle, b"fuse_refiner_blocks", args.fuse))
env = {k: v for k, v in os.environ.items() if k.startswith("ROMA_")}
ref = (_w0.cpu().numpy(), _c0.cpu().numpy())  # the single-stream result
bad = []
traces = collections.defaultdict(list)
k = 0
for r in range(args.rounds):
    for dual in ((False, True, True, True, True) if args.dual_only else (True, True, False, True, False)):
        m.dual_stream = dual
        w, c = m.match(inp["im_A"], inp["im_B"], **kw)
        torch.cuda.synchronize()
        w, c = w.cpu().numpy(), c.cpu().numpy()
        if args.trace:
            traces[dual].append([m.debug_trace(s) for s in ((0, 1) if dual else (0,))])
        if ref is None:
            ref = (w, c)
        elif not (np.array_equal(w, ref[0]) and np.array_equal(c, ref[1])):
            dc = np.abs(c - ref[1])
            pairs = [int(b) for b in np.nonzero(dc.reshape(8, -1).max(axis=1) > 0)[0]]
            bad.append((k, dual, float(np.abs(w - ref[0]).max()), float(dc.max()), pairs))
        k += 1
print(f"amp={args.amp} others={args.others} seed={args.seed} trace={args.trace} fuse={args.fuse} env={env}: {len(bad)}/{k} calls differ from the single-stream result: {bad[:4]}", flush=True)
if args.trace:
    for dual, trs in traces.items():
        for slot in range(2 if dual else 1):
            names = trs[0][slot][0]
            sums = [t[slot][1] for t in trs]
            n = min(len(s) for s in sums)
            major = [collections.Counter(int(s[i]) for s in sums).most_common(1)[0][0] for i in range(n)]
            first = collections.Counter()
            shown = 0
            for s in sums:
                dev = [i for i in range(n) if int(s[i]) != major[i]]
                if dev:
                    first[names[dev[0]]] += 1
                    if shown < 6:
                        shown += 1
                        print(f"    chain ({len(dev)} stages): {[names[i] for i in dev if 'blk' not in names[i]][:14]}", flush=True)
            print(f"  dual={dual} slot {slot}: first deviating stage over {len(sums)} calls: {dict(first)}", flush=True)

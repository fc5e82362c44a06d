"""Determinism stress for the sub-batch stream split (GPU box): repeated match() with 2 streams vs the single-stream
result; reports the pairs / bounding boxes that differ.  Configs: bf16 default, bf16 without the fused refiner blocks,
f32."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from roma_amd import roma_model, synthetic, _lib

lib = _lib.load()
sd, dsd = synthetic.make_matcher_state_dict(0), synthetic.make_dinov2_state_dict(0)
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 3
NR = int(sys.argv[2]) if len(sys.argv) > 2 else 700
for tag, amp, fuse, n in ((f"bf16 unfused B={NB} RI_VEC={os.environ.get('ROMA_RI_VEC', '1')}", torch.bfloat16, 0, NR),):
    inp = {k: v.cuda() for k, v in synthetic.make_inputs(NB, 112, 168, seed=7).items()}
    m = roma_model((112, 112), True, device="cuda:0", weights=sd, dinov2_weights=dsd, amp_dtype=amp, symmetric=True,
                   upsample_res=(168, 168), max_batch=NB)
    _lib.check(lib.roma_set_option(m._handle, b"fuse_refiner_blocks", fuse))
    kw = dict(im_A_high_res=inp["im_A_high_res"], im_B_high_res=inp["im_B_high_res"])
    m.dual_stream = False
    w1, c1 = m.match(inp["im_A"], inp["im_B"], **kw)
    torch.cuda.synchronize()
    m.dual_stream = True
    bad = []
    t0 = time.time()
    for i in range(n):
        w, c = m.match(inp["im_A"], inp["im_B"], **kw)
        ne = c != c1
        if bool(ne.any()) or not torch.equal(w, w1):
            per = []
            for b in range(NB):
                if bool(ne[b].any()):
                    idx = ne[b].nonzero()
                    per.append((b, int(ne[b].sum()), idx[:, 0].min().item(), idx[:, 0].max().item(), idx[:, 1].min().item(),
                                idx[:, 1].max().item(), float((c[b] - c1[b]).abs().max()), float((w[b] - w1[b]).abs().max())))
            bad.append((i, per))
    print(f"{tag:14s} mismatching runs {len(bad)}/{n} in {time.time() - t0:.1f}s; (run, [(pair, n, r0, r1, c0, c1, dcert, dwarp)]):", bad[:6], flush=True)

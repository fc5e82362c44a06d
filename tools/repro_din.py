"""Round-4 debugging aid: capture the stride-1 refiner input d (p1_din1) of sub-batch 1 in TWO-stream mode
(ROMA_DEBUG_DUAL_SLOT=1) over repeated calls and show what differs between calls: which channels (x | x_hat | emb), which
pixels, by how much, and whether the input flow differs."""
import os
import sys

os.environ.setdefault("ROMA_DEBUG_DUAL_SLOT", "1")
os.environ.setdefault("ROMA_DEBUG_ONLY", "p1_din1,p1_flowin1")
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from roma_amd import roma_model, synthetic  # noqa: E402

amp = sys.argv[1] if len(sys.argv) > 1 else "f16"
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 40
sd, dsd = synthetic.make_matcher_state_dict(0), synthetic.make_dinov2_state_dict(0)
dt = {"bf16": torch.bfloat16, "f16": torch.float16, "mixed": torch.bfloat16}[amp]
m = roma_model((560, 560), True, device="cuda:0", weights=sd, dinov2_weights=dsd, amp_dtype=dt, symmetric=True,
               upsample_res=(864, 864), max_batch=8, decoder_dtype=torch.float16 if amp == "mixed" else None)
inp = {k: v.cuda() for k, v in synthetic.make_inputs(8, 560, 864, seed=1).items()}
kw = dict(im_A_high_res=inp["im_A_high_res"], im_B_high_res=inp["im_B_high_res"])
m.debug = True
m.dual_stream = True


def h16(u):
    return u.view(np.float16).astype(np.float32) if m._lib.h16 == "f16" else (u.astype(np.uint32) << 16).view(np.float32)


ref = None
nbad = 0
for i in range(runs):
    m.match(inp["im_A"], inp["im_B"], **kw)
    torch.cuda.synchronize()
    din = m.debug_fetch("p1_din1", dtype=np.uint16).reshape(8, 560, 560, 24)
    fl = m.debug_fetch("p1_flowin1").reshape(8, 560, 560, 2)
    if ref is None:
        ref = (din.copy(), fl.copy())
        continue
    ne = din != ref[0]
    if ne.any():
        nbad += 1
        idx = np.argwhere(ne)
        ch = np.bincount(idx[:, 3], minlength=24)
        a, b = h16(din[ne]), h16(ref[0][ne])
        print(f"call {i}: {int(ne.sum())} elements differ; per channel {ch.tolist()}; pairs {sorted(set(idx[:, 0].tolist()))}; "
              f"rows {idx[:, 1].min()}..{idx[:, 1].max()} cols {idx[:, 2].min()}..{idx[:, 2].max()}; max |d| {np.abs(a - b).max():.4g}; "
              f"first: {idx[0].tolist()} {a[0]:.6g} vs {b[0]:.6g}; flow differs: {bool((fl != ref[1]).any())} "
              f"({int((fl != ref[1]).sum())} elements, max {np.abs(fl - ref[1]).max():.3g})", flush=True)
        if nbad <= 3:
            px = np.unique(idx[:, :3], axis=0)
            print("   pixels:", px[:12].tolist(), "... total", len(px), flush=True)
print(f"amp={amp}: {nbad}/{runs - 1} calls with a different p1_din1 (sub-batch 1)", flush=True)

import os, sys, subprocess, numpy as np
code = r'''
import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
from roma_amd import synthetic, roma_outdoor
sd, dsd = synthetic.make_matcher_state_dict(0), synthetic.make_dinov2_state_dict(0)
m = roma_outdoor(device="cuda", weights=sd, dinov2_weights=dsd, coarse_res=112, upsample_res=168, amp_dtype=torch.float32)
inp = synthetic.make_inputs(1, 112, 168, seed=1)
d = {k: v.cuda() for k, v in inp.items()}
m.debug = True
m.match(d["im_A"], d["im_B"], im_A_high_res=d["im_A_high_res"], im_B_high_res=d["im_B_high_res"])
torch.cuda.synchronize()
for s in (16, 8, 4, 2):
    np.save(f"gpurun_out/din{s}_{os.environ.get('ROMA_RI_VEC','1')}.npy", m.debug_fetch(f"p1_din{s}"))
'''
for v in ("1", "0"):
    subprocess.run([sys.executable, "-c", code], env=dict(os.environ, ROMA_RI_VEC=v), check=True)
for s in (16, 8, 4, 2):
    a = np.load(f"gpurun_out/din{s}_1.npy"); b = np.load(f"gpurun_out/din{s}_0.npy")
    a = a.view(np.float32); b = b.view(np.float32)
    diff = np.abs(a - b)
    print(s, a.shape, "max diff", diff.max(), "n>1e-6:", int((diff > 1e-6).sum()), "first idx", np.argwhere(diff > 1e-6)[:5].ravel())

"""Stage-by-stage comparison of the device Tiny RoMa path with the CPU oracle on the golden features (GPU box)."""
import ctypes as C
import math
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import tiny_oracle as T  # noqa: E402
from roma_amd import TinyRoMa, _lib, synthetic  # noqa: E402

g = np.load(os.path.join(ROOT, "tests", "golden", "tiny_reference.npz"))
tag = sys.argv[1] if len(sys.argv) > 1 else "a"
sd = synthetic.make_tiny_state_dict(0)
m = TinyRoMa(xfeat=synthetic.XFeatStandIn(0), weights=sd, device="cuda:0")
lib = _lib.load()
n = g[tag + "_im_A"].shape[0]
ff, fc = torch.from_numpy(g[tag + "_feat_fine"]), torch.from_numpy(g[tag + "_feat_coarse"])
f0c, f1c = fc[:n], fc[n:]
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
st = None
a_c, b_c = m._nhwc(f0c.cuda(), st), m._nhwc(f1c.cuda(), st)
print("nhwc", float((a_c.cpu() - f0c.permute(0, 2, 3, 1)).abs().max()))
B, Hc, Wc, Cc = a_c.shape
n0 = n1 = Hc * Wc
cv = torch.empty((B, n1, n0), device="cuda")
_lib.check(lib.roma_op_gemm(P(b_c), Cc, P(a_c), Cc, P(cv), n0, n1, n0, Cc, B, n1 * Cc, n0 * Cc, n1 * n0, None, None, None, 0, 0,
                            1.0 / math.sqrt(Cc), 0, 0, st))
cv_ref = T.corr_volume(f0c, f1c).reshape(B, n1, n0)
print("cv", float((cv.cpu() - cv_ref).abs().max()), float(cv_ref.abs().max()))
cw = torch.empty((B, Hc, Wc, 2), device="cuda")
_lib.check(lib.roma_op_tiny_pos_embed(P(cv), P(cw), B, Hc, Wc, Hc, Wc, 0, st))  # exact_softmax = 0
cw_ref = T.pos_embed(cv_ref.reshape(B, Hc, Wc, Hc, Wc)).permute(0, 2, 3, 1)
print("pos_embed", float((cw.cpu() - cw_ref).abs().max()))
cp = 160
d = torch.empty((B, Hc, Wc, cp), device="cuda")
_lib.check(lib.roma_op_tiny_matcher_input(P(a_c), P(b_c), P(cw), 2, P(d), B, Hc, Wc, Hc, Wc, Cc, cp, st))
cm0 = torch.cat((cw_ref.permute(0, 3, 1, 2), torch.zeros(B, 1, Hc, Wc)), dim=1)
f1w = F.grid_sample(f1c, cm0.permute(0, 2, 3, 1)[..., :2], mode="bilinear", align_corners=False)
d_ref = torch.cat((f0c, f1w, cw_ref.permute(0, 3, 1, 2)), dim=1).permute(0, 2, 3, 1)
print("matcher input", float((d.cpu()[..., :130] - d_ref).abs().max()), "pad", float(d.cpu()[..., 130:].abs().max()))
x = d_ref.permute(0, 3, 1, 2)
cur = d
for i, (wt, b, cin_p, cout) in enumerate(m._w["coarse_matcher"]["layers"]):
    nxt = torch.empty((B, Hc, Wc, cout), device="cuda")
    _lib.check(lib.roma_op_conv3x3(P(cur), P(wt), P(b), P(nxt), B, Hc, Wc, cin_p, cout, 1, 0, st))
    x = F.relu(F.batch_norm(F.conv2d(x, sd[f"coarse_matcher.{i}.layer.0.weight"], None, padding=1), sd[f"coarse_matcher.{i}.layer.1.running_mean"],
                            sd[f"coarse_matcher.{i}.layer.1.running_var"], None, None, False, 0.1, 1e-5))
    print(f"conv {i}", float((nxt.cpu() - x.permute(0, 2, 3, 1)).abs().max()), float(x.abs().max()))
    cur = nxt
delta = m._matcher("coarse_matcher", d, B, Hc, Wc, st)
dref = T.matcher(d_ref.permute(0, 3, 1, 2), sd, "coarse_matcher").permute(0, 2, 3, 1)
print("delta", float((delta.cpu()[:, :3].reshape(B, Hc, Wc, 3) - dref).abs().max()))

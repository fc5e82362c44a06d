import ctypes as C, sys, os, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from roma_amd import _lib
lib = _lib.load()
P = lambda t: C.c_void_p(t.data_ptr())
torch.manual_seed(0)
Cp, B, H, W = 144, 1, 13, 10
x = torch.randn(B, Cp, H, W).to(torch.bfloat16)
w, b = torch.randn(Cp, 1, 5, 5) * 0.2, torch.randn(Cp)
pw = (torch.randn(Cp, Cp) * Cp ** -0.5).to(torch.bfloat16)
pb = torch.randn(Cp)
mid = F.relu(F.conv2d(x.double(), w.double(), b.double(), padding=2, groups=Cp)).to(torch.bfloat16)
ref = (F.conv2d(mid.double(), pw.double()[:, :, None, None], pb.double())).permute(0, 2, 3, 1)
out = torch.full((B, H, W, Cp), float("nan"), device="cuda", dtype=torch.bfloat16)
wp = w.reshape(Cp, 25).T.contiguous().cuda()
xc, bc, pwc, pbc = x.permute(0, 2, 3, 1).contiguous().cuda(), b.cuda(), pw.cuda(), pb.cuda()
rc = lib.roma_op_refiner_block(P(xc), P(out), P(wp), P(bc), P(pwc), P(pbc), B, H, W, Cp, 1, None)
torch.cuda.synchronize()
err = (out.cpu().double() - ref).abs()[0]
print("rc", rc, "max", err.max().item())
print("per-channel max err (blocks of 16):", [round(err[:, :, i:i+16].max().item(), 3) for i in range(0, Cp, 16)])
print("per-row max err:", [round(err[y].max().item(), 3) for y in range(H)])
print("per-col max err:", [round(err[:, xx].max().item(), 3) for xx in range(W)])
# try identity pw to isolate dw
pwi = torch.eye(Cp).to(torch.bfloat16).cuda()
pb0 = torch.zeros(Cp).cuda()
rc = lib.roma_op_refiner_block(P(xc), P(out), P(wp), P(bc), P(pwi), P(pb0), B, H, W, Cp, 1, None)
torch.cuda.synchronize()
e2 = (out.cpu().double() - mid.double().permute(0, 2, 3, 1)).abs()[0]
print("identity-pw: max", e2.max().item(), "per-ch16:", [round(e2[:, :, i:i+16].max().item(), 3) for i in range(0, Cp, 16)])

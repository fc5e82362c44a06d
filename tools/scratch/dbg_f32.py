import ctypes as C, os, sys, torch
sys.path.insert(0, os.getcwd())
from roma_amd import _lib
lib = _lib.load()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
torch.manual_seed(0)
for (M, N, K) in [(9000, 64, 32), (9000, 64, 64), (9000, 64, 72), (9000, 64, 128), (8200, 128, 64), (8192, 64, 256)]:
    A = torch.randn(M, K); W = torch.randn(N, K)
    out = torch.empty(M, N, device="cuda")
    Ad, Wd = A.cuda(), W.cuda()
    rc = lib.roma_op_gemm(P(Ad), K, P(Wd), K, P(out), N, M, N, K, 1, 0, 0, 0, None, None, None, 0, 0, 1.0, 0, 0, None)
    torch.cuda.synchronize()
    ref = A.double() @ W.double().T
    err = (out.cpu().double() - ref).abs()
    # per k-group contribution check: which k ranges are missing?
    miss = []
    for g0 in range(0, K, 8):
        part = A[:, g0:g0+8].double() @ W[:, g0:g0+8].double().T
        e2 = (out.cpu().double() + part - ref).abs().max().item()   # if this group was counted twice
        e3 = (out.cpu().double() - (ref - part)).abs().max().item()  # if this group is missing
        if e3 < 1e-3: miss.append(("missing", g0))
        if e2 < 1e-3: miss.append(("double", g0))
    print(M, N, K, "rc", rc, "max err", err.max().item(), "rows bad", int((err.max(1).values > 1e-3).sum()), miss[:6])

import ctypes as C, os, sys, torch
sys.path.insert(0, os.getcwd())
from roma_amd import _lib
lib = _lib.load()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
def timeit(fn, iters=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
M, N, K = 25616, 4096, 1024
A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
W = (torch.randn(N, K, device="cuda") * 0.03).to(torch.bfloat16)
b = torch.randn(N, device="cuda")
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for act in (0, 1, 2):
    def f():
        rc = lib.roma_op_gemm(P(A), K, P(W), K, P(out), N, M, N, K, 1, 0, 0, 0, P(b), None, None, 0, act, 1.0, 1, 1, None); assert rc == 0
    t = timeit(f); print(f"fc1 GEMM act={act}: {t:.3f} ms {2.0*M*N*K/t*1e-9:.0f} TF/s")

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
B, N, heads, hd, K = 16, 1601, 16, 64, 1024
npad = (N + 127) // 128 * 128
A = torch.randn(B * N, K, device="cuda").to(torch.bfloat16)
W = (torch.randn(3 * heads * hd, K, device="cuda") * 0.03).to(torch.bfloat16)
bias = torch.randn(3 * heads * hd, device="cuda")
q = torch.zeros(B * heads * npad * hd, device="cuda", dtype=torch.bfloat16); k = torch.zeros_like(q); vt = torch.zeros_like(q)
out = torch.empty(B * N, heads * hd, device="cuda", dtype=torch.bfloat16)
def qkv():
    rc = lib.roma_op_qkv_scatter_gemm(P(A), P(W), P(bias), P(q), P(k), P(vt), B, N, npad, heads, hd, K, 1, 1, None); assert rc == 0, lib.roma_last_error()
def att():
    rc = lib.roma_op_attention(P(q), P(k), P(vt), P(out), B, heads, N, npad, hd, 1, 1, None); assert rc == 0, lib.roma_last_error()
t = timeit(qkv); print(f"qkv scatter gemm M={B*N} N=3072 K=1024: {t:.3f} ms  {2.0*B*N*3072*K/t*1e-9:.0f} TF/s")
t = timeit(att); print(f"attention B={B} heads={heads} N={N} hd={hd}: {t:.3f} ms  {4.0*B*heads*N*N*hd/t*1e-9:.0f} TF/s")

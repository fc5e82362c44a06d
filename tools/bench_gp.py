"""The GP's blocked Cholesky solve alone (n = 1600 tokens, d = 512 right-hand sides, 8 images = one sub-batch stream of the
benchmark): right-looking launch chain (roma_tuning("gp_col", 0)) against the left-looking block-column kernel (chol_col.hip),
one stream and two concurrent streams (the benchmark's regime: both sub-batch streams are inside the chain at the same time)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from roma_amd import _lib  # noqa: E402

lib = _lib.load()


def P(t):
    return C.c_void_p(t.data_ptr())


def make(batch, n, d, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    y = torch.randn(batch, n, 64, generator=g)
    yn = y / y.norm(dim=-1, keepdim=True)
    A = torch.exp((yn @ yn.transpose(1, 2) - 1.0) / 0.2) + 0.1 * torch.eye(n)
    Ft = torch.randn(batch, d, n, generator=g)
    src = torch.cat([A.reshape(batch, -1), Ft.reshape(batch, -1)], dim=1).cuda()
    return A, Ft, src


def main():
    n, d, batch = 1600, 512, 8
    A, Ft, src = make(batch, n, d, 1)
    ref = torch.cholesky_solve(Ft[:1].transpose(1, 2).double(), torch.linalg.cholesky(A[:1].double()))
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    bufs = [torch.empty_like(src) for _ in streams]
    LT = [torch.empty((batch, n, n), device="cuda") for _ in streams]
    Li = [torch.empty((batch, n // 64, 64, 64), device="cuda") for _ in streams]
    LiT = [torch.empty_like(x) for x in Li]

    def solve(i):
        st = streams[i]
        with torch.cuda.stream(st):
            bufs[i].copy_(src, non_blocking=True)
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record(st)
            rc = lib.roma_op_cholesky_solve_t(P(bufs[i]), C.c_void_p(bufs[i].data_ptr() + n * n * 4), P(LT[i]), P(Li[i]), P(LiT[i]),
                                              n, d, batch, C.c_void_p(st.cuda_stream))
            assert rc == 0, _lib.last_error(lib)
            e1.record(st)
        return e0, e1

    for col, leader in ((0, 1), (1, 0), (1, 1), (0, 1), (1, 0), (1, 1)):
        assert lib.roma_tuning(b"gp_col", col) == 0
        assert lib.roma_tuning(b"gp_col_leader", leader) == 0
        for nst in (1, 2):
            for _ in range(2):
                for i in range(nst):
                    solve(i)
            torch.cuda.synchronize()
            ts = []
            for _ in range(10):
                evs = [solve(i) for i in range(nst)]
                torch.cuda.synchronize()
                ts.append(max(e0.elapsed_time(e1) for e0, e1 in evs))
            X = bufs[0][0, n * n:].reshape(d, n).cpu().t().double()
            err = float((X - ref[0]).abs().max())
            ts.sort()
            print(f"gp_col={col} leader={leader} streams={nst}: median {ts[len(ts) // 2] * 1e3:8.1f} us  min {ts[0] * 1e3:8.1f} us   max|X - X_f64| = {err:.2e}", flush=True)
    lib.roma_tuning(b"gp_col", -1)
    lib.roma_tuning(b"gp_col_leader", -1)


if __name__ == "__main__":
    main()

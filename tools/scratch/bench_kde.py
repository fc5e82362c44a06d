import os, sys, torch, numpy as np
sys.path.insert(0, os.getcwd())
import roma_amd
g = np.random.Generator(np.random.PCG64(5))
n = 40000
x = torch.from_numpy(g.random((n, 4), dtype=np.float32) * 2 - 1).cuda()
for _ in range(3): roma_amd.kde(x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): roma_amd.kde(x)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"kde n={n}: {ms:.3f} ms  {n*n/ms*1e-6:.1f} Gpair/s  ({10.0*n*n/ms*1e-9:.1f} TFLOP-eq/s of 78.6 TF/s non-packed f32 VALU)")
# the reference formulation on the same device (torch ops), for scale
xh = x.half()
def ref():
    return (-torch.cdist(xh, xh) ** 2 / (2 * 0.1 ** 2)).exp().sum(-1)
ref(); torch.cuda.synchronize()
e0.record()
for _ in range(5): ref()
e1.record(); torch.cuda.synchronize()
print(f"torch fp16 cdist formulation (reference, same GPU): {e0.elapsed_time(e1)/5:.3f} ms")

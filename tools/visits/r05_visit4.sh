#!/bin/bash
# round 5, visit 4: where the fused C = 576 block's time goes - ablations of refiner_block_wide v2 (ROMA_RBW_DBG: 1 no stencil,
# 2 no MFMA, 4 no DMA after the prologue, 8 no tap reads, 16 no epilogue) at 16 x 216 x 216
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/v4; mkdir -p "$OUT"
for d in 0 1 2 4 8 16 3 7 19 23 31 5 6; do
ROMA_RBW_DBG=$d timeout 120 python - <<P 2>&1 | tee -a "$OUT/ablation.log"
import ctypes as C, os, sys, torch
sys.path.insert(0, os.getcwd())
from roma_amd import _lib
lib = _lib.load()
P = lambda t: C.c_void_p(t.data_ptr())
B, H, W, Cp = 16, 216, 216, 576
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(B, H, W, Cp, device="cuda", generator=g).to(torch.bfloat16)
w = torch.randn(25, Cp, device="cuda", generator=g) * 0.1
b = torch.randn(Cp, device="cuda", generator=g) * 0.1
pw = (torch.randn(Cp, Cp, device="cuda", generator=g) * Cp ** -0.5).to(torch.bfloat16)
pb = torch.randn(Cp, device="cuda", generator=g)
y = torch.empty_like(x)
f = lambda: lib.roma_op_refiner_block(P(x), P(y), P(w), P(b), P(pw), P(pb), B, H, W, Cp, 1, None)
for _ in range(3): assert f() == 0
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): f()
e1.record(); torch.cuda.synchronize()
print(f"dbg=$d: {e0.elapsed_time(e1) * 100:8.1f} us")
P
done
echo "== done"

import os, subprocess, sys
# ROMA_GEMM_DBG bits: 1 skip stores, 2 one K slab only, 4 no DMA after the first slab (MFMA+LDS only), 8 DMA only
code = r'''
import sys, os
sys.path.insert(0, os.getcwd()); sys.argv=["x"]
import tools.bench_gemm as b
for (M,N,K) in [(65536,1024,8192),(65536,1024,1024),(25616,4096,1024),(65536,576,576)]:
    b.run(M,N,K,b.BF16,b.BF16)
'''
for d in (0, 4, 8, 12):
    print("== ROMA_GEMM_DBG=%d" % d, flush=True)
    env = dict(os.environ, ROMA_GEMM_DBG=str(d))
    subprocess.run([sys.executable, "-c", code], env=env)

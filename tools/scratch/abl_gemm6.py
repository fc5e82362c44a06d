import os, subprocess, sys
code = r'''
import sys, os
sys.path.insert(0, os.getcwd()); sys.argv=["x"]
import tools.bench_gemm as b
for (M,N,K) in [(25616,4096,1024),(25616,3072,1024),(186624,1152,1152),(78400,1152,1152),(65536,2048,2048)]:
    b.run(M,N,K,b.BF16,b.BF16)
'''
for d in (0, 64, 0, 64):
    print("== ROMA_GEMM_DBG=%d (64 = old n-fastest order)" % d, flush=True)
    subprocess.run([sys.executable, "-c", code], env=dict(os.environ, ROMA_GEMM_DBG=str(d)))

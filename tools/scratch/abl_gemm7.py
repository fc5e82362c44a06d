import os, subprocess, sys
code = r'''
import sys, os
sys.path.insert(0, os.getcwd()); sys.argv=["x"]
import tools.bench_gemm as b
for (M,N,K) in [(65536,1024,1024),(25616,4096,1024),(25616,3072,1024),(25616,1024,4096),(25616,1024,1024),(65536,1024,8192)]:
    b.run(M,N,K,b.BF16,b.BF16)
    b.run(M,N,K,b.BF16,b.F32)
'''
for d in ("1", "0", "1", "0"):
    print("== ROMA_GEMM_PP=%s" % d, flush=True)
    subprocess.run([sys.executable, "-c", code], env=dict(os.environ, ROMA_GEMM_PP=d))

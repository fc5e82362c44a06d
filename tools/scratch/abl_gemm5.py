import os, subprocess, sys
code = r'''
import sys, os
sys.path.insert(0, os.getcwd()); sys.argv=["x"]
import tools.bench_gemm as b
for (M,N,K) in [(65536,1024,1024),(25616,4096,1024),(25616,3072,1024),(25616,1024,4096),(65536,1024,8192),(186624,1152,1152),(746496,576,576)]:
    b.run(M,N,K,b.BF16,b.BF16)
'''
for d in (0, 32, 0, 32):
    print("== ROMA_GEMM_DBG=%d (32 = single DMA burst)" % d, flush=True)
    env = dict(os.environ, ROMA_GEMM_DBG=str(d))
    subprocess.run([sys.executable, "-c", code], env=env)

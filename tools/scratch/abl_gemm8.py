import os, subprocess, sys
code = r'''
import sys, os
sys.path.insert(0, os.getcwd()); sys.argv=["x"]
import tools.bench_gemm as b
for (M,N,K) in [(65536,1024,1024),(25616,4096,1024),(25616,3072,1024),(25616,1024,4096),(25616,1024,1024),(65536,1024,8192),(186624,1152,1152),(746496,576,576)]:
    b.run(M,N,K,b.BF16,b.BF16)
    b.run(M,N,K,b.BF16,b.F32)
for (M,N,K) in [(65536,1152,1152),(25616,1024,4096)]:
    b.run(M,N,K,b.F32,b.F32)
'''
subprocess.run([sys.executable, "-c", code], env=dict(os.environ, ROMA_GEMM_PP="0"))

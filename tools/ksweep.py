import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import bench_gemm as bg
for K in (64, 1024, 4096):
    bg.run(65536, 1024, K, 1, 1)
bg.run(65536, 1152, 1152, 1, 1)
bg.run(25616, 3072, 1024, 1, 1)
bg.run(25616, 1024, 4096, 1, 0)

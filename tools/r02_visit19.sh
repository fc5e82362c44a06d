#!/bin/bash
# visit 19: conv64 with the conflict-free swizzle
cd /root/repo
mkdir -p gpurun_out/v19
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "conv3x3" 2>&1 | tail -3
timeout 300 python tools/bench_conv64.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/v19/bench_conv64.log

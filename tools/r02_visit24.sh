#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/v24
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "first_conv" 2>&1 | tail -3
timeout 300 python tools/bench_conv64.py 2>&1 | grep "first layer" | tee gpurun_out/v24/first_layer.log

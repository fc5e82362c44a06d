#!/bin/bash
set -u
echo "== attention tests"
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "attention" 2>&1 | tail -4
echo "== full-size two-stream stress, default kernels"
timeout 400 python tools/stress_streams.py --pairs 8 --runs 80 --fuse 1 --res 560 864 --trace 2>&1 | grep -v amdgpu.ids | grep -v "run [0-9]*: 1[0-9][0-9] dev" | cut -c1-500 | head -8

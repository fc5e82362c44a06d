#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/v28
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "refiner_block" 2>&1 | tail -4
for v in 0 1; do ROMA_RB_V2=$v timeout 120 python tools/bench_refiner_block.py 2>&1 | grep "dbg=" | sed "s/^/v2=$v /"; done | tee gpurun_out/v28/refiner_block_v2.log

#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/v36
for sy in 36 0 72 108; do echo "== ROMA_RB_SY=$sy"; ROMA_RB_SY=$sy timeout 120 python tools/bench_refiner_block.py 2>&1 | grep "dbg="; done | tee gpurun_out/v36/rb_sy.log
timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -k "refiner_block" 2>&1 | tail -2

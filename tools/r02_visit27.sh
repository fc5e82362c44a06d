#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/v27
for d in 0 1 2 4 8 3 6 7 15; do ROMA_RB_DBG=$d timeout 120 python tools/bench_refiner_block.py 2>&1 | grep "dbg="; done | tee gpurun_out/v27/refiner_block_ablation.log

#!/bin/bash
# visit 20: where conv64's time goes - stores off / row DMA off / strip length
cd /root/repo
mkdir -p gpurun_out/v20
for dbg in 0 1 2 3; do
  echo "== dbg $dbg"; ROMA_CONV64_DBG=$dbg timeout 200 python tools/bench_conv64.py 2>&1 | grep "conv64=1"
done | tee gpurun_out/v20/conv64_dbg.log
for sy in 16 64 108; do
  echo "== SY $sy"; ROMA_CONV64_SY=$sy timeout 200 python tools/bench_conv64.py 2>&1 | grep "conv64=1"
done | tee -a gpurun_out/v20/conv64_dbg.log

#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/v21
for nt in 0 1; do
  echo "== NT $nt"; ROMA_CONV64_NT=$nt timeout 200 python tools/bench_conv64.py 2>&1 | grep "conv64=1"
done | tee gpurun_out/v21/conv64_nt.log

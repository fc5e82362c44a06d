#!/bin/bash
set -u
mkdir -p gpurun_out/v24
for v in 2 1; do
  echo "---- ROMA_ATTN_V=$v"
  ROMA_ATTN_V=$v timeout 400 python tools/stress_streams.py --pairs 8 --runs 60 --fuse 1 --res 560 864 --trace > gpurun_out/v24/stress_attn_v$v.log 2>&1
  grep -v amdgpu.ids gpurun_out/v24/stress_attn_v$v.log | grep -v "run [0-9]*: 1[0-9][0-9] dev" | cut -c1-600 | head -12
done

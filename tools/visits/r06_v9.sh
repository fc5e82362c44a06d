#!/bin/bash
# Round 6, visit 9: conv_patch v1 (weights as two 128-row halves, DMA 2 + 2 pieces in P1 / P4) against v2 (sub-tiles by k-pair, one
# piece per phase) on ONE box, alternating processes.
set -u
OUT=$PWD/gpurun_out/v9; rm -rf "$OUT"; mkdir -p "$OUT"
for i in 1 2 3; do
  echo "-- v1 (two halves)"; ROMA_LIB_DIR=$PWD/tools/scratch/ab_v1 timeout 300 python tools/bench_conv_patch.py 2>&1 | grep -v amdgpu | tee -a "$OUT/v1.log"
  echo "-- v2 (k-pair sub-tiles)"; timeout 300 python tools/bench_conv_patch.py 2>&1 | grep -v amdgpu | tee -a "$OUT/v2.log"
done
echo "== step A/B"
for i in 1 2 3; do
  ROMA_LIB_DIR=$PWD/tools/scratch/ab_v1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-other-configs --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('v1', round(d['value'],2), 'pairs/s', round(d['ms_per_step'],2), 'ms')" | tee -a "$OUT/bench_ab.log"
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-other-configs --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('v2', round(d['value'],2), 'pairs/s', round(d['ms_per_step'],2), 'ms')" | tee -a "$OUT/bench_ab.log"
done
echo "== done"

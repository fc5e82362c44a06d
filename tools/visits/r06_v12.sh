#!/bin/bash
# Round 6, visit 12: conv_patch with the de-correlated patch swizzle (key = (row >> 1) - row / PW) against the previous build, one box.
set -u
OUT=$PWD/gpurun_out/v12; rm -rf "$OUT"; mkdir -p "$OUT"
echo "== operator tests"
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "conv3x3_patch" 2>&1 | tail -4 | tee "$OUT/pytest_ops.log"
for i in 1 2; do
  echo "-- before (key = (row >> 1) & 7)"; ROMA_LIB_DIR=$PWD/tools/scratch/ab_v1 timeout 300 python tools/bench_conv_patch.py 2>&1 | grep -v amdgpu | tee -a "$OUT/before.log"
  echo "-- after"; timeout 300 python tools/bench_conv_patch.py 2>&1 | grep -v amdgpu | tee -a "$OUT/after.log"
done
echo "== step A/B"
for i in 1 2 3; do
  ROMA_LIB_DIR=$PWD/tools/scratch/ab_v1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-other-configs --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('before', round(d['value'],2), 'pairs/s', round(d['ms_per_step'],2), 'ms')" | tee -a "$OUT/bench_ab.log"
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-other-configs --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('after ', round(d['value'],2), 'pairs/s', round(d['ms_per_step'],2), 'ms')" | tee -a "$OUT/bench_ab.log"
done
echo "== parity"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "tiny or small or f32_full8_vs or mixed" 2>&1 | tail -4 | tee "$OUT/pytest_parity.log"
echo "== done"

#!/bin/bash
# Round 6, visit 14: (a) what a tile's epilogue costs with 256 / 128 / 64 / 32 workgroups storing at the same time
# (tools/bench_gemm_burst.py); (b) gemm8p K loop split into a STEADY copy and a general copy (gemm8p_ktile.inc), prio / stagger
# compile-time: A/B against the previous build (tools/scratch/ab_v2) on one box - bit-identity, per-GEMM time, step time, tests.
set -u
OUT=$PWD/gpurun_out/v14; rm -rf "$OUT"; mkdir -p "$OUT"
echo "== burst"; timeout 600 python tools/bench_gemm_burst.py 2>&1 | grep -v amdgpu | tee "$OUT/burst.log"
for i in 1 2; do
  echo "-- before"; ROMA_LIB_DIR=$PWD/tools/scratch/ab_v2 timeout 300 python tools/bench_gemm_epilogue.py 2>&1 | grep -v amdgpu | tee -a "$OUT/before.log"
  echo "-- after"; timeout 300 python tools/bench_gemm_epilogue.py 2>&1 | grep -v amdgpu | tee -a "$OUT/after.log"
done
echo "== step A/B"
for i in 1 2 3; do
  ROMA_LIB_DIR=$PWD/tools/scratch/ab_v2 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-other-configs --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('before', round(d['value'],2), 'pairs/s', round(d['ms_per_step'],2), 'ms')" | tee -a "$OUT/bench_ab.log"
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-other-configs --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('after ', round(d['value'],2), 'pairs/s', round(d['ms_per_step'],2), 'ms')" | tee -a "$OUT/bench_ab.log"
done
echo "== operator tests (gemm)"
timeout 1500 python -m pytest tests/test_gpu_ops.py -q -x -k "gemm or conv or qkv or linear" 2>&1 | tail -4 | tee "$OUT/pytest_ops.log"
echo "== parity"
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -4 | tee "$OUT/pytest_parity.log"
echo "== done"

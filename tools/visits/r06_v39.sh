#!/bin/bash
# Round 6, visit 39: refiner out_conv kernel (wide scales) - row groups software-pipelined, the weight prologue as 16-byte loads,
# 8 row groups per wave as before (16 and 32 measured beside it).  A/B against tools/scratch/ab_v39 (HEAD 49f591d + DESIGN row) on one box.
set -u
OUT=$PWD/gpurun_out/v39b; rm -rf "$OUT"; mkdir -p "$OUT"
for i in 1 2; do
  echo "-- before"; ROMA_LIB_DIR=$PWD/tools/scratch/ab_v39 timeout 300 python tools/bench_refiner_out.py 2>&1 | grep -v amdgpu | tee -a "$OUT/before.log"
  echo "-- after"; timeout 300 python tools/bench_refiner_out.py 2>&1 | grep -v amdgpu | tee -a "$OUT/after.log"
done
for r in 16 32; do echo "-- after, ROMA_OUT_ROWS_IT=$r"; ROMA_OUT_ROWS_IT=$r timeout 300 python tools/bench_refiner_out.py 2>&1 | grep -v amdgpu | grep -v Radeon | tee -a "$OUT/rows_it.log"; done
echo "== operator tests (refiner)"
timeout 1500 python -m pytest tests/test_gpu_ops.py -q -x -k "refiner" 2>&1 | tail -3 | tee "$OUT/pytest_ops.log"
echo "== parity"
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -3 | tee "$OUT/pytest_parity.log"
echo "== step A/B"
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-other-configs --no-roofline"
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["value"],2), "pairs/s", round(d["ms_per_step"],2), "ms")'
for i in 1 2 3; do
  ROMA_LIB_DIR=$PWD/tools/scratch/ab_v39 timeout 300 $B 2>/dev/null | python -c "$P" "before(ab_v39)" | tee -a "$OUT/bench_ab.log"
  timeout 300 $B 2>/dev/null | python -c "$P" "after" | tee -a "$OUT/bench_ab.log"
done
echo "== two-stream determinism (short)"
timeout 900 python tools/stress_streams.py --pairs 8 --res 560 864 --amp mixed --runs 60 2>&1 | grep -v amdgpu | tail -2 | cut -c1-260 | tee "$OUT/stress.log"
echo "== done"

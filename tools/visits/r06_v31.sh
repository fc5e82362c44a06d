#!/bin/bash
# Round 6, visit 31: C = 144 fused block with the wave index wave-uniform in a form the compiler sees (the DMA issue sat behind
# exec masks) and the row DMA requested behind the row stores - A/B against tools/scratch/ab_v10 (HEAD 2517430) on one box.
set -u
OUT=$PWD/gpurun_out/v31; rm -rf "$OUT"; mkdir -p "$OUT"
for i in 1 2 3; do
  echo "-- before"; ROMA_LIB_DIR=$PWD/tools/scratch/ab_v10 timeout 300 python tools/bench_refiner_block.py 2>&1 | grep -v amdgpu | tee -a "$OUT/rb_before.log"
  echo "-- after"; timeout 300 python tools/bench_refiner_block.py 2>&1 | grep -v amdgpu | tee -a "$OUT/rb_after.log"
done
echo "== step A/B"
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-other-configs --no-roofline"
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["value"],2), "pairs/s", round(d["ms_per_step"],2), "ms")'
for i in 1 2 3; do
  ROMA_LIB_DIR=$PWD/tools/scratch/ab_v10 timeout 300 $B 2>/dev/null | python -c "$P" "before(ab_v10)" | tee -a "$OUT/bench_ab.log"
  timeout 300 $B 2>/dev/null | python -c "$P" "after" | tee -a "$OUT/bench_ab.log"
done
echo "== operator tests (refiner)"
timeout 1500 python -m pytest tests/test_gpu_ops.py -q -x -k "refiner or block or dwconv" 2>&1 | tail -4 | tee "$OUT/pytest_ops.log"
echo "== parity"
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -4 | tee "$OUT/pytest_parity.log"
echo "== two-stream determinism (short)"
timeout 900 python tools/stress_streams.py --pairs 8 --res 560 864 --amp mixed --runs 60 2>&1 | grep -v amdgpu | tail -2 | cut -c1-260 | tee "$OUT/stress.log"
echo "== done"

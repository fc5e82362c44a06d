#!/bin/bash
# Round 6, visit 25: buffer-descriptor LDS-DMA in the ring kernels (dwconv5x5_ring, C = 24 and C = 144 fused blocks) and ws1x1 - A/B against
# tools/scratch/ab_v6 (HEAD f56f0b1) on one box: bit-identity, per-kernel time, step time, tests, stress.
set -u
OUT=$PWD/gpurun_out/v25; rm -rf "$OUT"; mkdir -p "$OUT"
for i in 1 2; do
  echo "-- dwconv before"; ROMA_LIB_DIR=$PWD/tools/scratch/ab_v6 timeout 300 python tools/bench_dwconv.py 2>&1 | grep -v amdgpu | tee -a "$OUT/dw_before.log"
  echo "-- dwconv after"; timeout 300 python tools/bench_dwconv.py 2>&1 | grep -v amdgpu | tee -a "$OUT/dw_after.log"
  echo "-- gemm before"; ROMA_LIB_DIR=$PWD/tools/scratch/ab_v6 timeout 300 python tools/bench_gemm_epilogue.py 2>&1 | grep -v amdgpu | grep "refiner\|lib dir" | tee -a "$OUT/gemm_before.log"
  echo "-- gemm after"; timeout 300 python tools/bench_gemm_epilogue.py 2>&1 | grep -v amdgpu | grep "refiner\|lib dir" | tee -a "$OUT/gemm_after.log"
done
echo "== step A/B"
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-other-configs --no-roofline"
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["value"],2), "pairs/s", round(d["ms_per_step"],2), "ms")'
for i in 1 2 3; do
  ROMA_LIB_DIR=$PWD/tools/scratch/ab_v6 timeout 300 $B 2>/dev/null | python -c "$P" "before(ab_v6)" | tee -a "$OUT/bench_ab.log"
  timeout 300 $B 2>/dev/null | python -c "$P" "after" | tee -a "$OUT/bench_ab.log"
done
echo "== operator tests"
timeout 1500 python -m pytest tests/test_gpu_ops.py -q -x 2>&1 | tail -4 | tee "$OUT/pytest_ops.log"
echo "== parity"
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -4 | tee "$OUT/pytest_parity.log"
echo "== two-stream determinism (short)"
timeout 900 python tools/stress_streams.py --pairs 8 --res 560 864 --amp mixed --runs 60 2>&1 | grep -v amdgpu | tail -6 | tee "$OUT/stress.log"
echo "== done"

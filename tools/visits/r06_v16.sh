#!/bin/bash
# Round 6, visit 16: the STEADY / general split of the K loop for gemm6p (gemm6p_ktile.inc) and, with the tap a constant, for the
# patch-resident convolution (conv_patch_ktile.inc: nine STEADY copies + the general copy) - A/B against tools/scratch/ab_v2 on
# one box: bit-identity, per-kernel time, step time, tests.
set -u
OUT=$PWD/gpurun_out/v16; rm -rf "$OUT"; mkdir -p "$OUT"
for i in 1 2; do
  echo "-- conv before"; ROMA_LIB_DIR=$PWD/tools/scratch/ab_v2 timeout 300 python tools/bench_conv_patch.py 2>&1 | grep -v amdgpu | tee -a "$OUT/conv_before.log"
  echo "-- conv after"; timeout 300 python tools/bench_conv_patch.py 2>&1 | grep -v amdgpu | tee -a "$OUT/conv_after.log"
  echo "-- gemm before"; ROMA_LIB_DIR=$PWD/tools/scratch/ab_v2 timeout 300 python tools/bench_gemm_epilogue.py 2>&1 | grep -v amdgpu | tee -a "$OUT/gemm_before.log"
  echo "-- gemm after"; timeout 300 python tools/bench_gemm_epilogue.py 2>&1 | grep -v amdgpu | tee -a "$OUT/gemm_after.log"
done
echo "== step A/B"
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-other-configs --no-roofline"
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["value"],2), "pairs/s", round(d["ms_per_step"],2), "ms")'
for i in 1 2 3; do
  ROMA_LIB_DIR=$PWD/tools/scratch/ab_v2 timeout 300 $B 2>/dev/null | python -c "$P" "before(ab_v2)" | tee -a "$OUT/bench_ab.log"
  timeout 300 $B 2>/dev/null | python -c "$P" "after" | tee -a "$OUT/bench_ab.log"
done
echo "== operator tests"
timeout 1500 python -m pytest tests/test_gpu_ops.py -q -x 2>&1 | tail -4 | tee "$OUT/pytest_ops.log"
echo "== parity"
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -4 | tee "$OUT/pytest_parity.log"
echo "== done"

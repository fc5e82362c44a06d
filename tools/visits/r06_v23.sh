#!/bin/bash
# Round 6, visit 23: gemm8p's dense operands through buffer descriptors (buffer_load_dwordx4 ... lds: SGPR base + 32-bit lane offset
# + SGPR K offset, hardware zeros outside M / N; -DROMA_R8_BUFLDS, library in tools/scratch/ab_buf) against the flat
# global_load_lds form of the tree, one box: bit-identity, per-GEMM time, step time, tests.
set -u
OUT=$PWD/gpurun_out/v23; rm -rf "$OUT"; mkdir -p "$OUT"
for i in 1 2 3; do
  echo "-- flat"; timeout 300 python tools/bench_gemm_epilogue.py 2>&1 | grep -v amdgpu | tee -a "$OUT/flat.log"
  echo "-- buffer"; ROMA_LIB_DIR=$PWD/tools/scratch/ab_buf timeout 300 python tools/bench_gemm_epilogue.py 2>&1 | grep -v amdgpu | tee -a "$OUT/buffer.log"
done
echo "== step A/B"
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-other-configs --no-roofline"
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["value"],2), "pairs/s", round(d["ms_per_step"],2), "ms")'
for i in 1 2 3; do
  timeout 300 $B 2>/dev/null | python -c "$P" "flat" | tee -a "$OUT/bench_ab.log"
  ROMA_LIB_DIR=$PWD/tools/scratch/ab_buf timeout 300 $B 2>/dev/null | python -c "$P" "buffer" | tee -a "$OUT/bench_ab.log"
done
echo "== operator tests with the buffer build"
ROMA_LIB_DIR=$PWD/tools/scratch/ab_buf timeout 1500 python -m pytest tests/test_gpu_ops.py -q -x 2>&1 | tail -4 | tee "$OUT/pytest_ops.log"
echo "== parity with the buffer build"
ROMA_LIB_DIR=$PWD/tools/scratch/ab_buf timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -4 | tee "$OUT/pytest_parity.log"
echo "== done"

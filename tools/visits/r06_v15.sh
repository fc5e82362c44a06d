#!/bin/bash
# Round 6, visit 15 (REJECTED experiments: the start-delay code and tools/bench_gemm_dephase.py were removed afterwards; the kernel snippet is quoted in profiles/r06_v15_*.log): (a) gemm8p start-delay classes ("de-phasing" the store bursts): sweep of delay and mode per GEMM, then the
# step time per setting (environment); (b) attention row sums with v_pk_add_f32: A/B against tools/scratch/ab_v2; tests.
set -u
OUT=$PWD/gpurun_out/v15; rm -rf "$OUT"; mkdir -p "$OUT"
echo "== dephase sweep"; timeout 900 python tools/bench_gemm_dephase.py 2>&1 | grep -v amdgpu | tee "$OUT/dephase.log"
echo "== attention A/B"
for i in 1 2; do
  echo "-- before"; ROMA_LIB_DIR=$PWD/tools/scratch/ab_v2 timeout 300 python tools/bench_attention.py 2>&1 | grep -v amdgpu | tee -a "$OUT/attn_before.log"
  echo "-- after"; timeout 300 python tools/bench_attention.py 2>&1 | grep -v amdgpu | tee -a "$OUT/attn_after.log"
done
echo "== step time per dephase setting"
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-other-configs --no-roofline"
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["value"],2), "pairs/s", round(d["ms_per_step"],2), "ms")'
for rep in 1 2; do
  for s in "0 2" "200 2" "300 2" "500 2" "300 1" "500 1"; do
    set -- $s
    ROMA_GEMM8P_DEPHASE=$1 ROMA_GEMM8P_DEPHASE_MODE=$2 timeout 300 $B 2>/dev/null | python -c "$P" "dephase=$1,mode=$2" | tee -a "$OUT/bench_dephase.log"
  done
  ROMA_LIB_DIR=$PWD/tools/scratch/ab_v2 timeout 300 $B 2>/dev/null | python -c "$P" "ab_v2(before)" | tee -a "$OUT/bench_dephase.log"
done
echo "== attention tests + determinism"
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "attention" 2>&1 | tail -4 | tee "$OUT/pytest_attn.log"
timeout 600 python tools/attn_determinism.py 2>&1 | grep -v amdgpu | tee "$OUT/attn_determinism.log"
echo "== parity"
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -4 | tee "$OUT/pytest_parity.log"
echo "== done"

#!/bin/bash
# Round 6, visit 19: gemm8p with the younger wave group at static priority 1 and no per-phase s_setprio flips (-DROMA_R8_PRIO_STATIC,
# MI355X_MICROARCH.md "static priority for the younger half") against the per-phase flips (tools/scratch/ab_v5 = HEAD f9230cb).
set -u
OUT=$PWD/gpurun_out/v19; rm -rf "$OUT"; mkdir -p "$OUT"
for i in 1 2 3; do
  echo "-- before"; ROMA_LIB_DIR=$PWD/tools/scratch/ab_v5 timeout 300 python tools/bench_gemm_epilogue.py 2>&1 | grep -v amdgpu | tee -a "$OUT/before.log"
  echo "-- after"; timeout 300 python tools/bench_gemm_epilogue.py 2>&1 | grep -v amdgpu | tee -a "$OUT/after.log"
done
echo "== step A/B"
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-other-configs --no-roofline"
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["value"],2), "pairs/s", round(d["ms_per_step"],2), "ms")'
for i in 1 2 3; do
  ROMA_LIB_DIR=$PWD/tools/scratch/ab_v5 timeout 300 $B 2>/dev/null | python -c "$P" "before(ab_v5)" | tee -a "$OUT/bench_ab.log"
  timeout 300 $B 2>/dev/null | python -c "$P" "after(static prio)" | tee -a "$OUT/bench_ab.log"
done
echo "== done"

#!/bin/bash
# Round 6, visit 29: attention key-tile loop in pairs with the LDS buffer index a compile-time constant (immediate LDS
# addressing) - A/B against tools/scratch/ab_v8 (HEAD) on one box: bit-identity, time, tests, determinism.
set -u
OUT=$PWD/gpurun_out/v29; rm -rf "$OUT"; mkdir -p "$OUT"
for i in 1 2 3 4; do
  echo "-- before"; ROMA_LIB_DIR=$PWD/tools/scratch/ab_v8 timeout 300 python tools/bench_attention.py 2>&1 | grep -v amdgpu | tee -a "$OUT/attn_before.log"
  echo "-- after"; timeout 300 python tools/bench_attention.py 2>&1 | grep -v amdgpu | tee -a "$OUT/attn_after.log"
done
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-other-configs --no-roofline"
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["value"],2), "pairs/s", round(d["ms_per_step"],2), "ms")'
for i in 1 2 3; do
  ROMA_LIB_DIR=$PWD/tools/scratch/ab_v8 timeout 300 $B 2>/dev/null | python -c "$P" "before(ab_v8)" | tee -a "$OUT/bench_ab.log"
  timeout 300 $B 2>/dev/null | python -c "$P" "after" | tee -a "$OUT/bench_ab.log"
done
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "attention or qkv" 2>&1 | tail -3 | tee "$OUT/pytest_attn.log"
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -3 | tee "$OUT/pytest_parity.log"
timeout 600 python tools/attn_determinism.py 2>&1 | grep -v amdgpu | tail -12 | tee "$OUT/attn_determinism.log"
echo "== done"

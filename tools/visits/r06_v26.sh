#!/bin/bash
# Round 6, visit 26: buffer-descriptor loads in conv64.hip (row DMA) and attention.hip (K / V^T tile fetch) - A/B against
# tools/scratch/ab_v7 (HEAD 3d82f0c) on one box: bit-identity, per-kernel time, step time, tests, attention determinism.
set -u
OUT=$PWD/gpurun_out/v26; rm -rf "$OUT"; mkdir -p "$OUT"
for i in 1 2 3; do
  echo "-- attention before"; ROMA_LIB_DIR=$PWD/tools/scratch/ab_v7 timeout 300 python tools/bench_attention.py 2>&1 | grep -v amdgpu | tee -a "$OUT/attn_before.log"
  echo "-- attention after"; timeout 300 python tools/bench_attention.py 2>&1 | grep -v amdgpu | tee -a "$OUT/attn_after.log"
done
for i in 1 2; do
  echo "-- conv64 before"; ROMA_LIB_DIR=$PWD/tools/scratch/ab_v7 timeout 300 python tools/bench_conv64.py 2>&1 | grep -v amdgpu | tee -a "$OUT/c64_before.log"
  echo "-- conv64 after"; timeout 300 python tools/bench_conv64.py 2>&1 | grep -v amdgpu | tee -a "$OUT/c64_after.log"
done
echo "== step A/B"
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-other-configs --no-roofline"
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["value"],2), "pairs/s", round(d["ms_per_step"],2), "ms")'
for i in 1 2 3; do
  ROMA_LIB_DIR=$PWD/tools/scratch/ab_v7 timeout 300 $B 2>/dev/null | python -c "$P" "before(ab_v7)" | tee -a "$OUT/bench_ab.log"
  timeout 300 $B 2>/dev/null | python -c "$P" "after" | tee -a "$OUT/bench_ab.log"
done
echo "== operator tests"
timeout 1500 python -m pytest tests/test_gpu_ops.py -q -x 2>&1 | tail -4 | tee "$OUT/pytest_ops.log"
echo "== parity"
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -4 | tee "$OUT/pytest_parity.log"
timeout 600 python tools/attn_determinism.py 2>&1 | grep -v amdgpu | tail -12 | tee "$OUT/attn_determinism.log"
echo "== done"

#!/bin/bash
# Round 6, visit 35: attention - the reference-move test taken from the row sums (sum of a tile's P <= 2^14) instead of a 32-deep
# max tree on every tile; the rare path re-does the tile.  A/B against tools/scratch/ab_v35 (HEAD 6e87f01) on one box.
set -u
OUT=$PWD/gpurun_out/v35; rm -rf "$OUT"; mkdir -p "$OUT"
for i in 1 2 3; do
  echo "-- before"; ROMA_LIB_DIR=$PWD/tools/scratch/ab_v35 timeout 300 python tools/bench_attention.py 2>&1 | grep -v amdgpu | tee -a "$OUT/attn_before.log"
  echo "-- after"; timeout 300 python tools/bench_attention.py 2>&1 | grep -v amdgpu | tee -a "$OUT/attn_after.log"
done
echo "== operator tests (attention, vit)"
timeout 1500 python -m pytest tests/test_gpu_ops.py -q -x -k "attention or attn or vit or qkv" 2>&1 | tail -4 | tee "$OUT/pytest_ops.log"
echo "== attention determinism (both exponent forms, spiky operands too)"
timeout 600 python tools/attn_determinism.py 2>&1 | grep -v amdgpu | tail -12 | cut -c1-240 | tee "$OUT/attn_determinism.log"
echo "== parity"
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -4 | tee "$OUT/pytest_parity.log"
echo "== step A/B"
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-other-configs --no-roofline"
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["value"],2), "pairs/s", round(d["ms_per_step"],2), "ms")'
for i in 1 2 3; do
  ROMA_LIB_DIR=$PWD/tools/scratch/ab_v35 timeout 300 $B 2>/dev/null | python -c "$P" "before(ab_v35)" | tee -a "$OUT/bench_ab.log"
  timeout 300 $B 2>/dev/null | python -c "$P" "after" | tee -a "$OUT/bench_ab.log"
done
echo "== two-stream determinism (short)"
timeout 900 python tools/stress_streams.py --pairs 8 --res 560 864 --amp mixed --runs 60 2>&1 | grep -v amdgpu | tail -2 | cut -c1-260 | tee "$OUT/stress.log"
echo "== done"

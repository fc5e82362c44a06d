#!/bin/bash
# Round 6, visit 30: conv_patch with the next tap's fragment addresses formed under the current K tile's MFMAs (they sat at the
# head of the P1 / P2 load blocks) - A/B against tools/scratch/ab_v9 (HEAD dc9a770) on one box.
set -u
OUT=$PWD/gpurun_out/v30; rm -rf "$OUT"; mkdir -p "$OUT"
for i in 1 2 3; do
  echo "-- conv before"; ROMA_LIB_DIR=$PWD/tools/scratch/ab_v9 timeout 300 python tools/bench_conv_patch.py 2>&1 | grep -v amdgpu | tee -a "$OUT/conv_before.log"
  echo "-- conv after"; timeout 300 python tools/bench_conv_patch.py 2>&1 | grep -v amdgpu | tee -a "$OUT/conv_after.log"
done
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-other-configs --no-roofline"
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["value"],2), "pairs/s", round(d["ms_per_step"],2), "ms")'
for i in 1 2 3; do
  ROMA_LIB_DIR=$PWD/tools/scratch/ab_v9 timeout 300 $B 2>/dev/null | python -c "$P" "before(ab_v9)" | tee -a "$OUT/bench_ab.log"
  timeout 300 $B 2>/dev/null | python -c "$P" "after" | tee -a "$OUT/bench_ab.log"
done
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "conv" 2>&1 | tail -3 | tee "$OUT/pytest_conv.log"
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -3 | tee "$OUT/pytest_parity.log"
echo "== done"

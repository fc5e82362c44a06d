#!/bin/bash
# Round 6, visit 34: gemm8p's walk order inside an XCD band (groups of g tile rows; row-major for the model's shapes, g = 8 from
# 24 tile columns on) - the model's shapes must not move (A/B against tools/scratch/ab_v33 = HEAD 03942ea), 8192^3 gains.
set -u
OUT=$PWD/gpurun_out/v34; rm -rf "$OUT"; mkdir -p "$OUT"
for i in 1 2; do
  echo "-- before"; ROMA_LIB_DIR=$PWD/tools/scratch/ab_v33 timeout 300 python tools/bench_gemm_walk.py 2>&1 | grep -v amdgpu | cut -c1-330 | tee -a "$OUT/walk_before.log"
  echo "-- after"; timeout 300 python tools/bench_gemm_walk.py 2>&1 | grep -v amdgpu | cut -c1-520 | tee -a "$OUT/walk_after.log"
done
echo "== step A/B"
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-other-configs --no-roofline"
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["value"],2), "pairs/s", round(d["ms_per_step"],2), "ms")'
for i in 1 2 3; do
  ROMA_LIB_DIR=$PWD/tools/scratch/ab_v33 timeout 300 $B 2>/dev/null | python -c "$P" "before(ab_v33)" | tee -a "$OUT/bench_ab.log"
  timeout 300 $B 2>/dev/null | python -c "$P" "after" | tee -a "$OUT/bench_ab.log"
done
echo "== operator tests (gemm)"
timeout 1500 python -m pytest tests/test_gpu_ops.py -q -x -k "gemm or qkv or vit or conv3x3" 2>&1 | tail -4 | tee "$OUT/pytest_ops.log"
echo "== parity"
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -4 | tee "$OUT/pytest_parity.log"
echo "== two-stream determinism (short)"
timeout 900 python tools/stress_streams.py --pairs 8 --res 560 864 --amp mixed --runs 60 2>&1 | grep -v amdgpu | tail -2 | cut -c1-260 | tee "$OUT/stress.log"
echo "== done"

#!/bin/bash
# Round-4 visit 13: weight-stationary 1x1 (ws1x1.hip) - bitwise vs the tile kernels, yardstick, whole-model A/B
set -u
OUT=$PWD/gpurun_out/v13
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== ws1x1 tests"
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "ws1x1" 2>&1 | tail -6
echo "== yardstick ws1x1 on / off"
for w in 1 0; do ROMA_WS1X1=$w timeout 300 python tools/bench_vendor_gemm.py 2>&1 | grep "stride 4" | cut -c1-170; done
echo "== bench A/B ws1x1 1, 0, 1"
for w in 1 0 1; do
  ROMA_WS1X1=$w timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-parity > "$OUT/bench_ws$w.json" 2> "$OUT/bench_ws$w.err"
  python - "$OUT/bench_ws$w.json" $w <<'PY'
import json,sys
r=json.load(open(sys.argv[1]))
print("ws1x1",sys.argv[2],"pairs/s",round(r["value"],2),"ms",round(r["ms_per_step"],3))
for k,v in r["kernels"].items():
    if "gemm6p" in k or "ws1x1" in k: print("   ",k,round(v["ms_per_step"],3),v["calls_per_step"],{a:round(b,1) for a,b in v.items() if a not in("ms_per_step","calls_per_step")})
PY
done
echo "== done"

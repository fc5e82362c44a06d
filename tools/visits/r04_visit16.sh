#!/bin/bash
# round 4, visit 16: the exp2-polynomial GELU epilogue + the sub-batch schedule skew (late VGG on the odd stream): tests, A/B
set -u
OUT=$PWD/gpurun_out/v16
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== tests: GEMM epilogues, stream split, full-size parity"
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_f16.py -q -x -k "gemm or gelu or epilogue" 2>&1 | tail -3
timeout 1500 python -m pytest tests/test_gpu_match.py tests/test_gpu_parity.py -q -x -k "stream or full8 or reproducible" 2>&1 | tail -3
echo "== schedule skew A/B (bf16, 20 steps each)"
for k in 0 1 0 1; do
  timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-parity --no-roofline --stream-skew $k > "$OUT/bench_skew${k}.json" 2> "$OUT/bench_skew${k}.err"
  python - "$OUT/bench_skew${k}.json" $k <<'PY'
import json,sys
r=json.load(open(sys.argv[1]))
print("skew",sys.argv[2],"pairs/s",round(r["value"],2),"ms",round(r["ms_per_step"],3))
PY
done
echo "== one stream, for the same box"
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-parity --no-roofline --streams 1 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('1 stream pairs/s',round(r['value'],2),'ms',round(r['ms_per_step'],3))"
echo "== mixed, skew 1 / 0"
for k in 1 0; do
timeout 400 python bench.py --steps 20 --warmup 5 --dtype mixed --no-cpu-baseline --no-other-configs --no-parity --no-roofline --stream-skew $k 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mixed skew $k pairs/s',round(r['value'],2),'ms',round(r['ms_per_step'],3))"
done
echo "== kernel table with the new GELU (instrumented pass)"
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-parity > "$OUT/bench_full.json" 2> "$OUT/bench_full.err"
python - "$OUT/bench_full.json" <<'PY'
import json,sys
r=json.load(open(sys.argv[1]))
print("pairs/s",round(r["value"],2),"ms",round(r["ms_per_step"],3))
for k,v in list(r["kernels"].items())[:14]:
    print("   ",k,round(v["ms_per_step"],3),v["calls_per_step"],{a:round(b,1) for a,b in v.items() if a not in("ms_per_step","calls_per_step")})
PY
echo "== done"

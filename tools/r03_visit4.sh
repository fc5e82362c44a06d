#!/bin/bash
# Round-3 visit 4: small-M GEMM tile rule A/B on BASELINE config 2 (and a check that it does not touch B = 8).
set -u
OUT=$PWD/gpurun_out/v4
mkdir -p "$OUT"
export TMPDIR=/tmp
for sm in 1 0; do
  ROMA_GEMM_SMALLM=$sm timeout 300 python bench.py --config coarse --steps 40 --warmup 5 --no-cpu-baseline --no-parity > "$OUT/coarse_smallm$sm.json" 2> "$OUT/coarse_smallm$sm.err"
  python - "$OUT/coarse_smallm$sm.json" $sm <<'PY'
import json,sys
r=json.load(open(sys.argv[1]))
print("smallm",sys.argv[2],"pairs/s",round(r["value"],2),"ms",round(r["ms_per_step"],3))
for k,v in list(r["kernels"].items())[:10]:
    print("   ",k,round(v["ms_per_step"],3),v["calls_per_step"],{a:round(b,1) for a,b in v.items() if a not in("ms_per_step","calls_per_step")})
PY
done
for sm in 1 0; do
  ROMA_GEMM_SMALLM=$sm timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-roofline --no-other-configs > "$OUT/full_smallm$sm.json" 2> "$OUT/full_smallm$sm.err"
  python -c "import json,sys; r=json.load(open('$OUT/full_smallm$sm.json')); print('full smallm $sm', round(r['value'],2), round(r['ms_per_step'],2))"
done
echo "== done"

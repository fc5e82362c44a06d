#!/bin/bash
# Round-3 visit 5: wave-private ring dwconv (dwconv_ring.hip): bitwise tests, microbench A/B, bench A/B.
set -u
OUT=$PWD/gpurun_out/v5
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_f16.py -q -k "dwconv" 2>&1 | tail -8
for r in 0 1; do
  echo "== ROMA_DW_RING=$r microbench"
  ROMA_DW_RING=$r timeout 300 python tools/bench_refiner.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/bench_refiner_ring$r.log" | grep "dw " | cut -c1-110
done
for r in 1 0 1 0; do
  ROMA_DW_RING=$r timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-other-configs > "$OUT/bench_ring$r.json" 2> "$OUT/bench_ring$r.err"
  python - "$OUT/bench_ring$r.json" $r <<'PY'
import json,sys
r=json.load(open(sys.argv[1])); k=r["kernels"]
dw=[v for n,v in k.items() if n.startswith("dwconv")][0]
print("ring",sys.argv[2],round(r["value"],2),"pairs/s",round(r["ms_per_step"],2),"ms  dwconv",round(dw["ms_per_step"],2),"ms",round(dw["GB/s"]),"GB/s")
PY
done
echo "== done"

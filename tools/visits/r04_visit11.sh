#!/bin/bash
# Round-4 visit 11: after the packed-f32 / SGPR fix (no SLP, no packed f32 outside the stencil sources): determinism of the
# three 16-bit modes with two streams, gemm race screens, bench.
set -u
OUT=$PWD/gpurun_out/v11
mkdir -p "$OUT"
export TMPDIR=/tmp
run() { echo "== $*"; env "$@" 2>&1 | grep -v "amdgpu.ids" | cut -c1-700; }
for amp in mixed f16 bf16; do run timeout 900 python tools/repro_mixed.py --others 0 --rounds 40 --amp $amp; done
echo "== reproducibility + mixed parity tests"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "reproducible or mixed" 2>&1 | tail -5
echo "== bench bf16 / mixed"
for dt in bf16 mixed; do
  timeout 400 python bench.py --dtype $dt --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > "$OUT/bench_$dt.json" 2> "$OUT/bench_$dt.err"
  python - "$OUT/bench_$dt.json" $dt <<'PY'
import json,sys
r=json.load(open(sys.argv[1]))
print(sys.argv[2],"pairs/s",round(r["value"],2),"ms",round(r["ms_per_step"],3), "instrumented sum", round(sum(v["ms_per_step"] for v in r["kernels"].values()),2))
for k,v in list(r["kernels"].items())[:14]: print("   ",k,round(v["ms_per_step"],3),v["calls_per_step"])
p=r.get("parity") or {}
print("   parity injected:", json.dumps(p.get("outputs_with_reference_coarse_match_injected"))[:400], "flips", (p.get("coarse_argmax") or {}).get("flips"))
PY
done
echo "== done"

#!/bin/bash
# Round-3 visit 3: suite (new: odd resolutions, mega geometry, handle cache, on-device XFeat backbone, exact softmax), default bench with the new legs.
set -u
OUT=$PWD/gpurun_out/v3
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== GPU suite"
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -40 > "$OUT/pytest_gpu.log"; tail -25 "$OUT/pytest_gpu.log"
cp gpurun_out/parity_report.json "$OUT/parity_report.json" 2>/dev/null
echo "== default bench"
timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; tail -3 "$OUT/bench_default.err" | cut -c1-300
python - "$OUT/bench_default.json" <<'PY'
import json,sys
r=json.load(open(sys.argv[1]))
print("value",r["value"],"ms",r["ms_per_step"])
print("roofline",json.dumps(r.get("roofline"))[:600])
print("coherent",json.dumps(r.get("kernels_coherent"))[:1500])
for k,v in (r.get("other_configs") or {}).items():
    print(k, v["value"], v["ms_per_step"], json.dumps(v["roofline"])[:300], json.dumps(v.get("parity"))[:500])
print("cpu",json.dumps(r.get("cpu_baseline"))[:600])
PY
echo "== done"

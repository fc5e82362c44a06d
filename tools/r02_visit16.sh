#!/bin/bash
# Round-2 GPU visit 16: experimental four-wave GEMM (gemm4w): correctness against the classic loop, then timing
set -u
OUT=$PWD/gpurun_out/v16
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_ops.py -q -x -k "gemm4w" 2>&1 | tail -6
timeout 300 python tools/bench_gemm_overhead.py > "$OUT/bench_gemm_overhead.log" 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/v16/bench_gemm_overhead.log"):
    if l.startswith("{"):
        r = json.loads(l)
        g8, g4 = r["gemm8p"], r.get("gemm4w", {})
        print(r["shape"], "classic", r["classic"]["dbg0_us"], "| 8p/6p", g8["dbg0_us"], f"({g8['TFLOPs']} TF) no-epi", g8["dbg256_us"], "| 4w", g4.get("dbg0_us"), f"({g4.get('TFLOPs')} TF) no-stores", g4.get("dbg1_us"), "no-epi", g4.get("dbg256_us"))
PY
tail -3 "$OUT/bench_gemm_overhead.log" | cut -c1-300
echo "== done"

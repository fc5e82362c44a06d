#!/bin/bash
# Round-2 GPU visit 14: branch-free rows in the per-query gather kernels of the local correlation
set -u
OUT=$PWD/gpurun_out/v14
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== op tests"
timeout 500 python -m pytest tests/test_gpu_ops.py -q -x -k "local_corr" 2>&1 | tail -3
echo "== local correlation regimes"
timeout 300 python tools/bench_local_corr.py > "$OUT/bench_local_corr.log" 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/v14/bench_local_corr.log"):
    if l.startswith("{"):
        r = json.loads(l)
        print(f"r={r['r']} C={r['C']} hw={r['hw']} {r['dtype']} {r['warp']:10s} tiled {r['tiled']['ms']:.3f} ({r['tiled']['algorithmic_GBs']:.0f} GB/s)  list-only {r['all_to_gather_list']['ms']:.3f}  per-pixel {r['per_pixel']['ms']:.3f} ms  diff {r['max_abs_diff_between_forms']:.1e}")
PY
echo "== bench x2"
for i in 1 2; do timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-roofline > "$OUT/bench$i.json" 2> "$OUT/bench$i.err"; cut -c1-230 "$OUT/bench$i.json"; done
echo "== done"

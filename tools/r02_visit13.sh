#!/bin/bash
# Round-2 GPU visit 13: branch-free tile kernel of the local correlation, list-kernel head loads, tile-grid cap A/B,
# non-temporal stores in refiner_block A/B.
set -u
OUT=$PWD/gpurun_out/v13
REPO=$PWD
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== op tests"
timeout 500 python -m pytest tests/test_gpu_ops.py -q -x -k "local_corr or refiner" 2>&1 | tail -3
lc() {
python - "$1" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        r = json.loads(l)
        print(f"r={r['r']} C={r['C']} hw={r['hw']} {r['dtype']} {r['warp']:10s} tiled {r['tiled']['ms']:.3f} ({r['tiled']['algorithmic_GBs']:.0f} GB/s)  list-only {r['all_to_gather_list']['ms']:.3f}  per-pixel {r['per_pixel']['ms']:.3f} ms  diff {r['max_abs_diff_between_forms']:.1e}")
PY
}
echo "== local correlation regimes"
timeout 300 python tools/bench_local_corr.py > "$OUT/bench_local_corr.log" 2>&1; lc "$OUT/bench_local_corr.log"
echo "== same, tile kernel launched with at most 1024 workgroups"
ROMA_LC_TILE_GRID=1024 timeout 300 python tools/bench_local_corr.py > "$OUT/bench_local_corr_grid1024.log" 2>&1; lc "$OUT/bench_local_corr_grid1024.log"
echo "== refiner blocks: default / non-temporal output stores"
timeout 200 python tools/bench_refiner.py > "$OUT/bench_refiner.log" 2>&1; grep fused "$OUT/bench_refiner.log" | cut -c1-120
ROMA_RB_DBG=32 timeout 200 python tools/bench_refiner.py > "$OUT/bench_refiner_nt.log" 2>&1; grep fused "$OUT/bench_refiner_nt.log" | cut -c1-120
echo "== bench"
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-roofline > "$OUT/bench.json" 2> "$OUT/bench.err"; cut -c1-230 "$OUT/bench.json"
ROMA_RB_DBG=32 timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-roofline > "$OUT/bench_rbnt.json" 2> "$OUT/bench_rbnt.err"; cut -c1-230 "$OUT/bench_rbnt.json"
echo "== done"

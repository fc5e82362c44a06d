#!/bin/bash
# Round-2 GPU visit 6: graph replay with the host running ahead (cause + fix), which commit removed the stream-split
# deviations (stress on the libraries of two earlier commits), local-corr classifier with wave-aggregated appends.
set -u
OUT=$PWD/gpurun_out/v6
REPO=$PWD
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== graph replay, 40 calls without host synchronisation: guard off (expected to fault), guard on"
ROMA_GRAPH_NOSYNC=1 timeout 200 python tools/debug_graph.py --res 560 --batch 1 --calls 40 --nosync > "$OUT/graph_nosync_guard_off.log" 2>&1; echo "guard off rc=$? $(grep -h 'fault\|GRAPH_OK\|all equal' "$OUT/graph_nosync_guard_off.log" | head -2 | cut -c1-200)"
timeout 200 python tools/debug_graph.py --res 560 --batch 1 --calls 40 --nosync > "$OUT/graph_nosync_guard_on.log" 2>&1; echo "guard on  rc=$? $(grep -h 'fault\|GRAPH_OK\|all equal' "$OUT/graph_nosync_guard_on.log" | head -2 | cut -c1-200)"
echo "== op / graph tests"
timeout 500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_match.py -q -x -k "local_corr or graph" 2>&1 | tail -4
echo "== bench coarse (graph on / off), full (graph on)"
for g in 0 1; do
  timeout 400 python bench.py --config coarse --steps 30 --warmup 5 --no-cpu-baseline --graph $g > "$OUT/bench_coarse_g$g.json" 2> "$OUT/bench_coarse_g$g.err"; tail -1 "$OUT/bench_coarse_g$g.err" | cut -c1-200; cut -c1-330 "$OUT/bench_coarse_g$g.json"
done
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-roofline --graph 1 > "$OUT/bench_graph.json" 2> "$OUT/bench_graph.err"; tail -1 "$OUT/bench_graph.err" | cut -c1-200; cut -c1-330 "$OUT/bench_graph.json"
echo "== stream-split stress on the libraries of earlier commits (visit 2 saw 15/300 deviating runs, un-fused blocks)"
for c in 7c54dd5 e83660d; do
  ( cd tools/scratch/bisect/$c && timeout 300 python tools/stress_streams.py --pairs 3 --runs 400 --fuse 0 ) > "$OUT/stress_$c.log" 2>&1; echo "$c: $(tail -1 "$OUT/stress_$c.log" | cut -c1-330)"
done
timeout 300 python tools/stress_streams.py --pairs 3 --runs 400 --fuse 0 > "$OUT/stress_head.log" 2>&1; echo "HEAD: $(tail -1 "$OUT/stress_head.log" | cut -c1-330)"
echo "== local correlation regimes"
timeout 300 python tools/bench_local_corr.py > "$OUT/bench_local_corr.log" 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/v6/bench_local_corr.log"):
    if l.startswith("{"):
        r = json.loads(l)
        print(f"r={r['r']} C={r['C']} hw={r['hw']} {r['dtype']} {r['warp']:10s} tiled {r['tiled']['ms']:.3f}  list-only {r['all_to_gather_list']['ms']:.3f}  per-pixel {r['per_pixel']['ms']:.3f} ms")
PY
echo "== bench (default)"
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity > "$OUT/bench.json" 2> "$OUT/bench.err"; cut -c1-330 "$OUT/bench.json"
python - <<'PY'
import json
r = json.load(open("gpurun_out/v6/bench.json"))
for k, v in r["kernels"].items():
    if "local_corr" in k: print(k, v)
PY
echo "== done"

#!/bin/bash
# Round-2 GPU visit 5: graph-replay fault bisection, stream-split stress without the trace, local-corr list kernel
# against the per-pixel kernel on identical work, HBM traffic (PMC) of the bench.
set -u
OUT=$PWD/gpurun_out/v5
REPO=$PWD
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== graph replay bisection"
for cfg in "112 0 1" "224 0 1" "560 0 1" "224 336 1" "224 336 2"; do
  set -- $cfg
  for e in "ROMA_GEMM8P=1" "ROMA_GEMM8P=0"; do
    env $e timeout 200 python tools/debug_graph.py --res $1 --up $2 --batch $3 > "$OUT/graph_$1_$2_$3_$e.log" 2>&1
    echo "res=$1 up=$2 batch=$3 $e rc=$? : $(grep -c 'equal to eager = True' "$OUT/graph_$1_$2_$3_$e.log") ok calls; $(grep -h 'fault\|GRAPH_OK\|rror' "$OUT/graph_$1_$2_$3_$e.log" | head -2 | cut -c1-160)"
  done
done
echo "== graph replay with the runtime's launch log (smallest failing size)"
AMD_LOG_LEVEL=3 timeout 300 python tools/debug_graph.py --res 224 --up 0 --batch 1 2>&1 | grep -i "ShaderName\|fault\|graph call\|eager ok\|error" | tail -40 | cut -c1-260 > "$OUT/graph_amdlog_tail.log"
tail -25 "$OUT/graph_amdlog_tail.log"
echo "== stream-split stress, no trace"
timeout 400 python tools/stress_streams.py --pairs 3 --runs 400 --fuse 0 > "$OUT/stress_unfused.log" 2>&1; tail -1 "$OUT/stress_unfused.log" | cut -c1-400
timeout 400 python tools/stress_streams.py --pairs 3 --runs 400 --fuse 1 > "$OUT/stress_fused.log" 2>&1; tail -1 "$OUT/stress_fused.log" | cut -c1-400
ROMA_GEMM8P=0 timeout 400 python tools/stress_streams.py --pairs 3 --runs 400 --fuse 0 > "$OUT/stress_unfused_classic.log" 2>&1; tail -1 "$OUT/stress_unfused_classic.log" | cut -c1-400
echo "== local correlation regimes (0 tiled / 1 all tiles to the gather list / 2 per-pixel)"
timeout 300 python tools/bench_local_corr.py > "$OUT/bench_local_corr.log" 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/v5/bench_local_corr.log"):
    if l.startswith("{"):
        r = json.loads(l)
        print(f"r={r['r']} C={r['C']} hw={r['hw']} {r['dtype']} {r['warp']:10s} tiled {r['tiled']['ms']:.3f}  list-only {r['all_to_gather_list']['ms']:.3f}  per-pixel {r['per_pixel']['ms']:.3f} ms")
PY
echo "== PMC passes (HBM traffic per kernel)"
bash tools/pmc_round.sh > "$OUT/pmc_round.log" 2>&1; tail -14 "$OUT/pmc_round.log" | cut -c1-200
cp gpurun_out/pmc_summary.json "$OUT/" 2>/dev/null
echo "== done"

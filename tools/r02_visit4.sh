#!/bin/bash
# Round-2 GPU visit 4: 256 x 192 8-phase-style kernel (gemm6p), classic-kernel spill fix, residual double buffering,
# graph replay on an own stream, non-persistent local-corr work list; stream-split race trace.
set -u
OUT=$PWD/gpurun_out/v4
REPO=$PWD
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== gemm op tests"
timeout 500 python -m pytest tests/test_gpu_ops.py -q -x -k "gemm or conv3x3 or qkv or attention or local_corr" 2>&1 | tail -12 > "$OUT/pytest_gemm.log"; tail -4 "$OUT/pytest_gemm.log"
echo "== gemm overhead (dbg bits)"
timeout 300 python tools/bench_gemm_overhead.py > "$OUT/bench_gemm_overhead.log" 2>&1; cut -c1-640 "$OUT/bench_gemm_overhead.log"
echo "== gemm A/B microbench"
ROUNDS=5 RACE=10 timeout 600 python tools/bench_gemm8p.py all > "$OUT/bench_gemm8p.log" 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/v4/bench_gemm8p.log"):
    if l.startswith("{") and "speedup" in l:
        r = json.loads(l)
        print(f"{r['name'][:58]:58s} classic {r['classic']['TFLOPs']:7.1f} 8p {r['gemm8p']['TFLOPs']:7.1f} TF x{r['speedup']:.3f} bit={r['bitwise_equal_to_classic']} race={r['race_screen_diff_runs']}")
PY
echo "== full GPU suite"
timeout 1000 python -m pytest tests -m gpu -q 2>&1 | tail -30 > "$OUT/pytest_gpu.log"; tail -6 "$OUT/pytest_gpu.log"
cp gpurun_out/parity_report.json "$OUT/" 2>/dev/null
echo "== bench"
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench_8p.json" 2> "$OUT/bench_8p.err"; tail -2 "$OUT/bench_8p.err"; cut -c1-330 "$OUT/bench_8p.json"
ROMA_GEMM8P=0 timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity > "$OUT/bench_classic.json" 2> "$OUT/bench_classic.err"; cut -c1-330 "$OUT/bench_classic.json"
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-roofline --streams 2 > "$OUT/bench_streams2.json" 2> "$OUT/bench_streams2.err"; cut -c1-330 "$OUT/bench_streams2.json"
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-roofline --graph 1 > "$OUT/bench_graph.json" 2> "$OUT/bench_graph.err"; tail -1 "$OUT/bench_graph.err"; cut -c1-330 "$OUT/bench_graph.json"
for g in 0 1; do
  timeout 400 python bench.py --config coarse --steps 30 --warmup 5 --no-cpu-baseline --graph $g > "$OUT/bench_coarse_g$g.json" 2> "$OUT/bench_coarse_g$g.err"; tail -1 "$OUT/bench_coarse_g$g.err"; cut -c1-420 "$OUT/bench_coarse_g$g.json"
done
echo "== local correlation regimes"
timeout 200 python tools/bench_local_corr.py > "$OUT/bench_local_corr.log" 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/v4/bench_local_corr.log"):
    if l.startswith("{"):
        r = json.loads(l)
        print(f"r={r['r']} C={r['C']} hw={r['hw']} {r['dtype']} {r['warp']:10s} tiled {r['tiled']['ms']:.3f} ms ({r['tiled']['algorithmic_GBs']:.0f} GB/s)  per-pixel {r['per_pixel']['ms']:.3f} ms  diff {r['max_abs_diff_between_forms']:.2e}")
PY
echo "== stream-split race: per-stage checksum trace (unfused refiner blocks: the configuration that deviated 15/300)"
timeout 500 python tools/stress_streams.py --pairs 3 --runs 300 --fuse 0 --trace > "$OUT/stress_trace_unfused.log" 2>&1; tail -25 "$OUT/stress_trace_unfused.log" | cut -c1-300
echo "== kernel trace"
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o bench -- python "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-parity > "$OUT/prof.log" 2>&1
cd "$REPO"
for f in $(find "$OUT/prof" -name "*kernel_stats.csv"); do head -26 "$f" | cut -c1-200; done
find "$OUT/prof" -name "*kernel_trace.csv" -delete; find "$OUT/prof" -name "*agent_info.csv" -delete
echo "== done"

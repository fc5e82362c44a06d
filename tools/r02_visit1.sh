#!/bin/bash
# Round-2 GPU visit 1: validate the 8-phase GEMM (stand-alone draft, op tests, A/B microbench), run the whole GPU
# suite incl. the new parity gates, bench with / without gemm8p, co-run stress, kernel trace.
#   gpurun --timeout 1500 -- 'bash tools/r02_visit1.sh'
set -u
OUT=$PWD/gpurun_out/v1
REPO=$PWD
mkdir -p "$OUT"
export TMPDIR=/tmp
( nproc; free -g | head -2; rocm-smi --showclocks 2>/dev/null | head -20 ) > "$OUT/host.txt" 2>&1

echo "== gemm op tests"
timeout 420 python -m pytest tests/test_gpu_ops.py -q -x -k "gemm or conv3x3 or qkv" 2>&1 | tail -15 > "$OUT/pytest_gemm.log"; tail -6 "$OUT/pytest_gemm.log"

echo "== gemm A/B microbench"
timeout 500 python tools/bench_gemm8p.py all > "$OUT/bench_gemm8p.log" 2>&1; grep -c speedup "$OUT/bench_gemm8p.log"
python - <<'PY'
import json
for l in open("gpurun_out/v1/bench_gemm8p.log"):
    if l.startswith("{") and "speedup" in l:
        r = json.loads(l)
        print(f"{r['name'][:58]:58s} classic {r['classic']['TFLOPs']:7.1f} 8p {r['gemm8p']['TFLOPs']:7.1f} TF x{r['speedup']:.3f} bit={r['bitwise_equal_to_classic']} race={r['race_screen_diff_runs']}")
PY

echo "== full GPU suite"
timeout 1000 python -m pytest tests -m gpu -q 2>&1 | tail -40 > "$OUT/pytest_gpu.log"; tail -12 "$OUT/pytest_gpu.log"
cp gpurun_out/parity_report.json "$OUT/" 2>/dev/null

echo "== bench (gemm8p on / off)"
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench_8p.json" 2> "$OUT/bench_8p.err"; tail -2 "$OUT/bench_8p.err"; cut -c1-600 "$OUT/bench_8p.json"
ROMA_GEMM8P=0 timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity > "$OUT/bench_classic.json" 2> "$OUT/bench_classic.err"; cut -c1-400 "$OUT/bench_classic.json"

echo "== local correlation regimes"
timeout 200 python tools/bench_local_corr.py > "$OUT/bench_local_corr.log" 2>&1; tail -12 "$OUT/bench_local_corr.log" | cut -c1-330

echo "== co-run stress"
timeout 300 python tools/corun_stress.py 10 > "$OUT/corun_stress.log" 2>&1; grep -v "   0/" "$OUT/corun_stress.log" | tail -30

echo "== kernel trace"
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o bench -- python "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-parity > "$OUT/prof.log" 2>&1
cd "$REPO"
for f in $(find "$OUT/prof" -name "*kernel_stats.csv"); do head -30 "$f"; done
find "$OUT/prof" -name "*kernel_trace.csv" -delete
find "$OUT/prof" -name "*agent_info.csv" -delete
echo "== done"

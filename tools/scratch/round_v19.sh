#!/bin/bash
# v19 validation visit: GPU tests, bench (bf16 residual stream on / off), kernel trace, single-rank RCCL smoke.
set -u
OUT=$PWD/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -s 2>&1 | tail -60 > "$OUT/pytest_gpu.log"; tail -12 "$OUT/pytest_gpu.log"
timeout 600 python bench.py --steps 5 --warmup 2 > "$OUT/bench_bf16.json" 2> "$OUT/bench_bf16.err"; tail -3 "$OUT/bench_bf16.err"; cut -c1-900 "$OUT/bench_bf16.json"
ROMA_VIT_RES_F32=1 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > "$OUT/bench_bf16_resf32.json" 2> "$OUT/bench_bf16_resf32.err"; cut -c1-300 "$OUT/bench_bf16_resf32.json"
timeout 120 python tools/scratch/rccl_smoke.py > "$OUT/rccl_smoke.log" 2>&1; tail -2 "$OUT/rccl_smoke.log"
REPO=$PWD; cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_bf16" -o bench -- python "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > "$OUT/prof_bf16.log" 2>&1
cd "$REPO"
for f in $(find "$OUT/prof_bf16" -name "*kernel_stats.csv"); do head -8 "$f" | cut -c1-160; done
find "$OUT/prof_bf16" -name "*kernel_trace.csv" -size +20M -delete

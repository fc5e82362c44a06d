#!/bin/bash
# v20 validation visit: GPU tests, bench bf16 + f32 (2 sub-batch streams), kernel traces with the split on and off
set -u
OUT=$PWD/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -s 2>&1 | tail -60 > "$OUT/pytest_gpu.log"; tail -5 "$OUT/pytest_gpu.log"
timeout 600 python bench.py --steps 5 --warmup 2 > "$OUT/bench_bf16.json" 2> "$OUT/bench_bf16.err"; tail -3 "$OUT/bench_bf16.err"; cut -c1-1100 "$OUT/bench_bf16.json"
timeout 300 python bench.py --steps 3 --warmup 1 --dtype f32 --no-cpu-baseline > "$OUT/bench_f32.json" 2> "$OUT/bench_f32.err"; cut -c60-200 "$OUT/bench_f32.json"
REPO=$PWD; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_bf16" -o bench -- python "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > "$OUT/prof_bf16.log" 2>&1
ROMA_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_bf16_s1" -o bench -- python "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > "$OUT/prof_bf16_s1.log" 2>&1
cd "$REPO"
for f in $(find "$OUT/prof_bf16" "$OUT/prof_bf16_s1" -name "*kernel_stats.csv"); do echo $f; head -4 "$f" | cut -c1-140; done
find "$OUT" -name "*kernel_trace.csv" -size +20M -delete

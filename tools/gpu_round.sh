#!/bin/bash
# One GPU-box visit: parity tests, bench (bf16 + f32), rocprofv3 kernel trace.  Usage (from the repo root):
#   gpurun --timeout 2400 -- 'bash tools/gpu_round.sh [tests|bench|prof|all]'
set -u
WHAT=${1:-all}
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
nproc > "$OUT/host.txt"; free -g | head -2 >> "$OUT/host.txt"
if [[ $WHAT == tests || $WHAT == all ]]; then
  timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | tail -80 > "$OUT/pytest_gpu.log"
  tail -40 "$OUT/pytest_gpu.log"
fi
if [[ $WHAT == bench || $WHAT == all ]]; then
  timeout 900 python bench.py --steps 5 --warmup 2 > "$OUT/bench_bf16.json" 2> "$OUT/bench_bf16.err"; tail -3 "$OUT/bench_bf16.err"; cat "$OUT/bench_bf16.json"
  timeout 900 python bench.py --steps 3 --warmup 1 --dtype f32 --no-cpu-baseline > "$OUT/bench_f32.json" 2> "$OUT/bench_f32.err"; tail -3 "$OUT/bench_f32.err"; cat "$OUT/bench_f32.json"
fi
if [[ $WHAT == prof || $WHAT == all ]]; then
  REPO=$PWD
  cd /tmp
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_bf16" -o bench -- python "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > "$OUT/prof_bf16.log" 2>&1
  cd "$REPO"
  find "$OUT/prof_bf16" -name "*stats*" | head;
  for f in $(find "$OUT/prof_bf16" -name "*kernel_stats.csv"); do head -25 "$f"; done
  # keep only the summaries (traces are large)
  find "$OUT/prof_bf16" -name "*kernel_trace.csv" -size +20M -delete
fi

#!/bin/bash
# Round-6 closing GPU visit: the ONE script that regenerates the committed r06_final_* artefacts.
#   full GPU suite, default bench (ROMA_MIXED = the reference timing script's precision policy; two streams; parity; configs 2 / 5 / f16 / all-bf16 as other_configs
#   of the same JSON line; coherent-warp leg; checked CPU baseline),
#   single-stream bench, rocprofv3 kernel stats of the bench (one stream) + a two-stream kernel trace (concurrency analysis),
#   HBM traffic (PMC) and SQ counter passes, the full-size determinism stress (bf16 x 3 input seeds, mixed, f32).
set -u
OUT=$PWD/gpurun_out/final6
REPO=$PWD
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== full GPU suite"
timeout 1700 python -m pytest tests -m gpu -q 2>&1 | tail -30 > "$OUT/pytest_gpu.log"; tail -5 "$OUT/pytest_gpu.log"
cp gpurun_out/parity_report.json "$OUT/" 2>/dev/null
echo "== bench (default)"
timeout 900 python bench.py --steps 20 --warmup 3 > "$OUT/bench_mixed.json" 2> "$OUT/bench_mixed.err"; tail -1 "$OUT/bench_mixed.err" | cut -c1-200; cut -c1-300 "$OUT/bench_mixed.json"
echo "== bench, one stream"
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-other-configs --streams 1 > "$OUT/bench_mixed_1stream.json" 2> "$OUT/bench_mixed_1stream.err"; cut -c1-240 "$OUT/bench_mixed_1stream.json"
echo "== kernel trace of the bench (one stream: per-kernel stats)"
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o bench -- python "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-parity --no-other-configs --streams 1 > "$OUT/prof.log" 2>&1
cd "$REPO"
for f in $(find "$OUT/prof" -name "*kernel_stats.csv"); do head -14 "$f" | cut -c1-170; cp "$f" "$OUT/bench_mixed_kernel_stats.csv"; done
find "$OUT/prof" -name "*kernel_trace.csv" -delete; find "$OUT/prof" -name "*agent_info.csv" -delete
echo "== kernel trace of the bench (two streams: who runs next to whom)"
cd /tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d "$OUT/prof2" -o bench2 -- python "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-parity --no-other-configs > "$OUT/prof2.log" 2>&1
cd "$REPO"
for f in $(find "$OUT/prof2" -name "*kernel_trace.csv"); do python tools/stream_overlap.py "$f" > "$OUT/stream_overlap.txt" 2>&1; gzip -c "$f" > "$OUT/bench_mixed_2stream_kernel_trace.csv.gz"; done
tail -25 "$OUT/stream_overlap.txt"
rm -rf "$OUT/prof2"
echo "== PMC: HBM traffic"
bash tools/pmc_round.sh > "$OUT/pmc_round.log" 2>&1; tail -8 "$OUT/pmc_round.log" | cut -c1-200
cp gpurun_out/pmc_summary.json "$OUT/" 2>/dev/null
echo "== PMC: SQ"
bash tools/pmc_sq_round.sh > "$OUT/pmc_sq_round.log" 2>&1; tail -14 "$OUT/pmc_sq_round.log" | cut -c1-200
cp gpurun_out/pmc_sq_summary.json "$OUT/" 2>/dev/null
echo "== determinism stress at the benchmark size (two streams vs one, bit-exact)"
for seed in 7 8 9; do timeout 300 python tools/stress_streams.py --pairs 8 --res 560 864 --runs 100 --amp mixed --seed $seed 2>&1 | tail -1 | cut -c1-260; done | tee "$OUT/stress.log"
timeout 300 python tools/stress_streams.py --pairs 8 --res 560 864 --runs 100 --amp bf16 2>&1 | tail -1 | cut -c1-260 | tee -a "$OUT/stress.log"
timeout 300 python tools/stress_streams.py --pairs 8 --res 560 864 --runs 20 --amp f32 2>&1 | tail -1 | cut -c1-260 | tee -a "$OUT/stress.log"
echo "== done"

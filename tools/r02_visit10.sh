#!/bin/bash
# Round-2 GPU visit 10: full GPU suite (bit-identical stream split x 2 000 runs, graph replay), SQ counters of the
# bench, deviation rate of the pre-rewrite library, kernel trace of the coarse-only configuration.
set -u
OUT=$PWD/gpurun_out/v10
REPO=$PWD
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== full GPU suite"
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -30 > "$OUT/pytest_gpu.log"; tail -6 "$OUT/pytest_gpu.log"
cp gpurun_out/parity_report.json "$OUT/" 2>/dev/null
echo "== stream-split stress on the library of the commit before the epilogue rewrite (7c54dd5), 2 x 1000 runs"
for f in 0 1; do
  ( cd tools/scratch/bisect/7c54dd5 && timeout 300 python tools/stress_streams.py --pairs 3 --runs 1000 --fuse $f ) > "$OUT/stress_7c54dd5_fuse$f.log" 2>&1; echo "7c54dd5 fuse=$f: $(tail -1 "$OUT/stress_7c54dd5_fuse$f.log" | cut -c1-260)"
done
echo "== SQ counters"
bash tools/pmc_sq_round.sh > "$OUT/pmc_sq_round.log" 2>&1; tail -15 "$OUT/pmc_sq_round.log" | cut -c1-220
cp gpurun_out/pmc_sq_summary.json "$OUT/" 2>/dev/null
echo "== kernel trace, coarse-only B = 1"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_coarse" -o bench -- python "$REPO/bench.py" --config coarse --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --no-parity > "$OUT/prof_coarse.log" 2>&1
cd "$REPO"
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/v10/prof_coarse/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("coarse-only: kernel time per step (12 traced steps): %.3f ms, launches per step %.0f" % (tot / 12e6, sum(int(r["Calls"]) for r in rows) / 12))
PY
tail -1 "$OUT/prof_coarse.log" | cut -c1-300
find "$OUT/prof_coarse" -name "*kernel_trace.csv" -delete; find "$OUT/prof_coarse" -name "*agent_info.csv" -delete
echo "== bench with two streams"
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-roofline --streams 2 > "$OUT/bench_streams2.json" 2> "$OUT/bench_streams2.err"; cut -c1-330 "$OUT/bench_streams2.json"
echo "== done"

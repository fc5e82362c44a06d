#!/bin/bash
# Round 6, visit 2: block-column Cholesky with the leader / follower hand-off.
set -u
OUT=$PWD/gpurun_out/v2; rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== operator tests"
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "cholesky or gp_posterior" 2>&1 | tail -5 | tee "$OUT/pytest_ops.log"
echo "== the chain alone"
timeout 300 python tools/bench_gp.py 2>&1 | tee "$OUT/bench_gp.log"
echo "== kernel trace of the chain"
REPO=$PWD
cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/prof" -o gp -- python "$REPO/tools/bench_gp.py" > "$OUT/prof.log" 2>&1; cd "$REPO"
for f in $(find "$OUT/prof" -name "*kernel_trace.csv"); do python - "$f" <<'P' | tee "$OUT/gp_trace_summary.txt"
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Grid_Size_X"]) for r in rows))
# last solve with chol_col kernels on a single stream: print the per-launch durations of the last 25 chol_col launches
cc = [e for e in ev if "chol_col_kernel" in e[2]]
print("chol_col launches", len(cc))
last = cc[-25:]
for s, e, n, g in last: print(f"  start {(s - last[0][0]) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f} us  grid_x {g}")
by = collections.defaultdict(list)
for s, e, n, g in ev: by[n.split("(")[0][:60]].append((e - s) / 1e3)
for n, v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:12]: print(f"{n:62s} n={len(v):5d} total {sum(v) / 1e3:8.2f} ms avg {sum(v) / len(v):7.1f} us")
P
done
rm -rf "$OUT/prof"
echo "== bench A/B (mixed, two streams)"
for i in 1 2; do
for col in 0 1; do
  ROMA_GP_COL=$col timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-other-configs --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('gp_col=$col', d['dtype'], round(d['value'],2), 'pairs/s', round(d['ms_per_step'],2), 'ms')" | tee -a "$OUT/bench_ab.log"
done; done
echo "== done"

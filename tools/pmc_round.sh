#!/bin/bash
# HBM traffic of the dominant kernels via rocprofv3 PMC counters (separate passes, per MI355X_MICROARCH.md):
#   gpurun --timeout 1500 -- 'bash tools/pmc_round.sh'
set -u
OUT=$PWD/gpurun_out
REPO=$PWD
mkdir -p "$OUT"
export TMPDIR=/tmp
export ROMA_STREAMS=1   # per-launch traffic of the full-batch launches, like bench.py's instrumented roofline pass
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_$C" -o pmc -- python "$REPO/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-parity --no-other-configs > "$OUT/pmc_$C.log" 2>&1
  ls "$OUT/pmc_$C" | head
done
cd "$REPO"
python - <<'PY'
import csv, glob, collections, json, os
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(f"gpurun_out/pmc_{c}/*counter_collection.csv")
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != c:
                continue
            name = r["Kernel_Name"].split("(")[0]
            agg[name][0] += 1
            agg[name][1] += float(r["Counter_Value"])
    out[c] = {k: {"launches": v[0], "sum_kb": v[1]} for k, v in agg.items()}
json.dump(out, open("gpurun_out/pmc_summary.json", "w"), indent=1)
top = sorted(out.get("FETCH_SIZE", {}).items(), key=lambda kv: -kv[1]["sum_kb"])[:12]
for k, v in top:
    w = out.get("WRITE_SIZE", {}).get(k, {"sum_kb": 0})
    print(f"{k[:90]:90s} launches={v['launches']:5d} fetch_raw={v['sum_kb']/1e6:8.2f} GB(KB units) write_raw={w['sum_kb']/1e6:8.2f}")
PY
# the raw traces are large; keep the summary only
rm -rf "$OUT"/pmc_FETCH_SIZE/*kernel_trace.csv "$OUT"/pmc_WRITE_SIZE/*kernel_trace.csv

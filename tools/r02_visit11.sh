#!/bin/bash
# Round-2 GPU visit 11: visualize_warp, non-temporal store experiment, SQ counters of the tiled local correlation on
# coherent warps, default bench (two streams) with parity + CPU baseline.
set -u
OUT=$PWD/gpurun_out/v11
REPO=$PWD
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== op tests"
timeout 400 python -m pytest tests/test_gpu_ops.py -q -x -k "visualize or gemm8p or fb_consistency" 2>&1 | tail -3
echo "== gemm overhead (1024 = non-temporal stores)"
timeout 300 python tools/bench_gemm_overhead.py > "$OUT/bench_gemm_overhead.log" 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/v11/bench_gemm_overhead.log"):
    if l.startswith("{"):
        r = json.loads(l); g = r["gemm8p"]
        print(r["shape"], "8p/6p full", g["dbg0_us"], "nt-stores", g["dbg1024_us"], "no stores", g["dbg1_us"], "no epilogue", g["dbg256_us"])
PY
echo "== SQ counters: tiled local correlation, coherent warps"
cd /tmp
CNT="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
timeout 300 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d "$OUT/pmc_lc" -o pmc -- python "$REPO/tools/bench_local_corr.py" coherent > "$OUT/pmc_lc.log" 2>&1
cd "$REPO"
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("gpurun_out/v11/pmc_lc/*counter_collection.csv"):
    seen = set()
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0]
        if "local_corr" not in name: continue
        agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
        k = (r.get("Dispatch_Id"), name)
        if k not in seen: seen.add(k); n[name] += 1
for k, c in sorted(agg.items()):
    wc = c.get("SQ_WAVE_CYCLES", 1.0)
    print(f"{k[:80]:80s} n={n[k]:3d} wait_any={c.get('SQ_WAIT_ANY',0)/wc:.2f} wait_inst={c.get('SQ_WAIT_INST_ANY',0)/wc:.2f} wait_inst_lds={c.get('SQ_WAIT_INST_LDS',0)/wc:.2f} active={c.get('SQ_ACTIVE_INST_ANY',0)/wc:.2f} lds_conf={c.get('SQ_LDS_BANK_CONFLICT',0)/max(c.get('SQ_LDS_IDX_ACTIVE',1),1):.3f} lds_active_per_busy={c.get('SQ_LDS_IDX_ACTIVE',0)/max(c.get('SQ_BUSY_CYCLES',1),1):.3f}")
PY
rm -f "$OUT"/pmc_lc/*kernel_trace.csv
echo "== bench, default (two streams), with parity and the CPU baseline"
timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; tail -2 "$OUT/bench_default.err" | cut -c1-200; cut -c1-400 "$OUT/bench_default.json"
echo "== done"

#!/bin/bash
# Round-2 GPU visit 12: local-correlation bank-conflict fixes (16-byte f0 accesses in the gather kernels, rectangle row
# pitch = 8 mod 16 in the tile kernel), non-temporal GEMM stores as default (A/B on the bench).
set -u
OUT=$PWD/gpurun_out/v12
REPO=$PWD
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== op tests"
timeout 500 python -m pytest tests/test_gpu_ops.py -q -x -k "local_corr or gemm or conv3x3 or qkv" 2>&1 | tail -3
echo "== local correlation regimes"
timeout 300 python tools/bench_local_corr.py > "$OUT/bench_local_corr.log" 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/v12/bench_local_corr.log"):
    if l.startswith("{"):
        r = json.loads(l)
        print(f"r={r['r']} C={r['C']} hw={r['hw']} {r['dtype']} {r['warp']:10s} tiled {r['tiled']['ms']:.3f} ({r['tiled']['algorithmic_GBs']:.0f} GB/s)  list-only {r['all_to_gather_list']['ms']:.3f}  per-pixel {r['per_pixel']['ms']:.3f} ms  diff {r['max_abs_diff_between_forms']:.1e}")
PY
echo "== SQ counters, coherent"
cd /tmp
CNT="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
timeout 300 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d "$OUT/pmc_lc" -o pmc -- python "$REPO/tools/bench_local_corr.py" coherent > "$OUT/pmc_lc.log" 2>&1
cd "$REPO"
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("gpurun_out/v12/pmc_lc/*counter_collection.csv"):
    seen = set()
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0]
        if "local_corr" not in name or "classify" in name: continue
        agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
        k = (r.get("Dispatch_Id"), name)
        if k not in seen: seen.add(k); n[name] += 1
for k, c in sorted(agg.items()):
    wc = c.get("SQ_WAVE_CYCLES", 1.0)
    print(f"{k[:80]:80s} n={n[k]:3d} wait_any={c.get('SQ_WAIT_ANY',0)/wc:.2f} wait_inst={c.get('SQ_WAIT_INST_ANY',0)/wc:.2f} wait_inst_lds={c.get('SQ_WAIT_INST_LDS',0)/wc:.2f} active={c.get('SQ_ACTIVE_INST_ANY',0)/wc:.2f} lds_conf={c.get('SQ_LDS_BANK_CONFLICT',0)/max(c.get('SQ_LDS_IDX_ACTIVE',1),1):.3f}")
PY
rm -f "$OUT"/pmc_lc/*kernel_trace.csv
echo "== bench: non-temporal stores on (default) / off, two streams; then one stream with the kernel table"
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-roofline > "$OUT/bench_nt1.json" 2> "$OUT/bench_nt1.err"; cut -c1-230 "$OUT/bench_nt1.json"
ROMA_GEMM_NT=0 timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-roofline > "$OUT/bench_nt0.json" 2> "$OUT/bench_nt0.err"; cut -c1-230 "$OUT/bench_nt0.json"
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-roofline > "$OUT/bench_nt1b.json" 2> "$OUT/bench_nt1b.err"; cut -c1-230 "$OUT/bench_nt1b.json"
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --streams 1 > "$OUT/bench_1stream.json" 2> "$OUT/bench_1stream.err"; cut -c1-230 "$OUT/bench_1stream.json"
python - <<'PY'
import json
r = json.load(open("gpurun_out/v12/bench_1stream.json"))
print(json.dumps(r["roofline"])[:330])
for k, v in list(r["kernels"].items())[:16]: print(f"{k:52s} {v['ms_per_step']:7.2f} ms  {v.get('TFLOP/s', 0):7.1f} TF {v.get('GB/s', 0):6.0f} GB/s")
PY
echo "== done"

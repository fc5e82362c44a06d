#!/bin/bash
# SQ / GRBM counters of the dominant kernels (MFMA busy, issue stalls, LDS bank conflicts), one rocprofv3 --pmc pass
# (8 SQ slots + GRBM; never combined with runtime / HIP traces):
#   gpurun --timeout 1500 -- 'bash tools/pmc_sq_round.sh'
set -u
OUT=$PWD/gpurun_out
REPO=$PWD
mkdir -p "$OUT"
export TMPDIR=/tmp
export ROMA_STREAMS=1   # counters per full-batch launch, like the instrumented roofline pass
cd /tmp
CNT="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
timeout 900 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d "$OUT/pmc_sq" -o pmc -- python "$REPO/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-parity --no-other-configs > "$OUT/pmc_sq.log" 2>&1
tail -3 "$OUT/pmc_sq.log"
cd "$REPO"
python - <<'PY'
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for f in glob.glob("gpurun_out/pmc_sq/*counter_collection.csv"):
    seen = set()
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0]
        agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r.get("Dispatch_Id"), name)
        if key not in seen:
            seen.add(key); n[name] += 1
out = {}
for k, c in agg.items():
    d = dict(c); d["launches"] = n[k]
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    if gui:
        # SQ_VALU_MFMA_BUSY_CYCLES is summed over the chip's 1024 SIMDs (256 CUs x 4), GRBM_GUI_ACTIVE over the 8 XCDs
        # (calibration: the dominant GEMM's 741 TFLOP/s = 0.30 of peak reads 0.32 with this normalisation)
        d["mfma_busy_frac"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8.0 * 1024.0)
    if c.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_conflict_frac"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]
    if c.get("SQ_WAVE_CYCLES"):
        d["wait_any_frac"] = c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
        d["wait_inst_frac"] = c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
        d["active_inst_frac"] = c.get("SQ_ACTIVE_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
    out[k] = d
json.dump(out, open("gpurun_out/pmc_sq_summary.json", "w"), indent=1)
top = sorted(out.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0.0))[:14]
for k, d in top:
    print(f"{k[:84]:84s} n={d['launches']:4d} mfma_busy={d.get('mfma_busy_frac', 0):.3f} wait_any={d.get('wait_any_frac', 0):.2f} "
          f"wait_inst={d.get('wait_inst_frac', 0):.2f} active={d.get('active_inst_frac', 0):.2f} lds_conf={d.get('lds_conflict_frac', 0):.3f}")
PY
rm -rf "$OUT"/pmc_sq/*kernel_trace.csv

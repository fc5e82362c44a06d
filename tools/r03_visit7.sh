#!/bin/bash
# Round-3 visit 7: 8-phase GEMM for single-pair wide launches (config 2) A/B, ring dwconv microbench, kernel trace of config 2.
set -u
OUT=$PWD/gpurun_out/v7
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "gemm_big_m" 2>&1 | tail -4
for mm in 2048 8192; do
  ROMA_GEMM8P_MINM=$mm timeout 300 python bench.py --config coarse --steps 40 --warmup 5 --no-cpu-baseline --no-parity > "$OUT/coarse_minm$mm.json" 2> "$OUT/coarse_minm$mm.err"
  python - "$OUT/coarse_minm$mm.json" $mm <<'PY'
import json,sys
r=json.load(open(sys.argv[1]))
print("minm",sys.argv[2],"pairs/s",round(r["value"],2),"ms",round(r["ms_per_step"],3), "instrumented sum", round(sum(v["ms_per_step"] for v in r["kernels"].values()),2))
for k,v in list(r["kernels"].items())[:8]:
    print("   ",k,round(v["ms_per_step"],3),v["calls_per_step"],{a:round(b,1) for a,b in v.items() if a not in("ms_per_step","calls_per_step")})
PY
done
echo "== ring dwconv microbench (strip rule)"
ROMA_DW_RING=1 timeout 300 python tools/bench_refiner.py 2>&1 | grep "dw " | cut -c1-60
echo "== kernel trace, config 2"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_coarse" -o coarse -- python "$OLDPWD/bench.py" --config coarse --steps 20 --warmup 3 --no-cpu-baseline --no-parity --no-roofline > "$OUT/prof_coarse.log" 2>&1
cd "$OLDPWD"
for f in $(find "$OUT/prof_coarse" -name "*kernel_stats.csv"); do head -16 "$f" | cut -c1-150; cp "$f" "$OUT/coarse_kernel_stats.csv"; done
find "$OUT/prof_coarse" -name "*kernel_trace.csv" -delete; find "$OUT/prof_coarse" -name "*agent_info.csv" -delete
echo "== done"

#!/bin/bash
# Round-4 visit 1: K-loop schedules of gemm8p (k-half "KH" vs quadrant phases) - race screens, ablation + phase trace,
# vendor yardstick, whole-model A/B; depthwise-on-MFMA go/no-go microbenchmark.
set -u
OUT=$PWD/gpurun_out/v1
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== gemm race screens (both schedules)"
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "gemm8p or qkv_scatter or res_bf16 or gemm_big" 2>&1 | tail -5
echo "== depthwise 5x5 on v_mfma_f32_4x4x4_16b_bf16 (tools/scratch/dwmfma)"
timeout 120 tools/scratch/dwmfma/dwmfma > "$OUT/dwmfma.log" 2>&1; tail -25 "$OUT/dwmfma.log"
echo "== gemm8p ablation + phase trace"
timeout 600 python tools/bench_gemm_ablation.py > "$OUT/gemm_ablation.log" 2>&1; tail -8 "$OUT/gemm_ablation.log" | cut -c1-1500
echo "== vendor yardstick, schedule 1 (KH) then 0"
for s in 1 0; do
  ROMA_GEMM8P_SCHED=$s timeout 300 python tools/bench_vendor_gemm.py > "$OUT/vendor_sched$s.log" 2>&1
  cut -c1-110 "$OUT/vendor_sched$s.log" | tail -12
done
echo "== bench A/B (schedule 1, 0, 1)"
for s in 1 0 1; do
  ROMA_GEMM8P_SCHED=$s timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-parity > "$OUT/bench_sched$s.json" 2> "$OUT/bench_sched$s.err"
  python - "$OUT/bench_sched$s.json" $s <<'PY'
import json,sys
r=json.load(open(sys.argv[1]))
print("sched",sys.argv[2],"pairs/s",round(r["value"],2),"ms",round(r["ms_per_step"],3))
for k,v in r["kernels"].items():
    if "gemm8p" in k: print("   ",k,round(v["ms_per_step"],3),v["calls_per_step"],{a:round(b,1) for a,b in v.items() if a not in("ms_per_step","calls_per_step")})
PY
done
echo "== ROMA_MIXED (bf16 DINOv2 in libroma_hip.so + binary16 elsewhere): parity gates, bench"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "mixed" 2>&1 | tail -8
timeout 400 python bench.py --dtype mixed --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > "$OUT/bench_mixed.json" 2> "$OUT/bench_mixed.err"
python - "$OUT/bench_mixed.json" <<'PY'
import json,sys
r=json.load(open(sys.argv[1]))
print("mixed pairs/s",round(r["value"],2),"ms",round(r["ms_per_step"],3))
print(json.dumps(r.get("parity"))[:1500])
PY
cp gpurun_out/parity_report.json "$OUT/parity_report.json" 2>/dev/null
echo "== done"

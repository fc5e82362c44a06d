#!/bin/bash
# round 4, visit 18: chol_diag as a two-wave pipeline (inverse one row behind the factorisation): tests, timing
set -u
OUT=$PWD/gpurun_out/v18
REPO=$PWD
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== tests: Cholesky / GP operators (x3 for the flag hand-off), f32 + 16-bit full parity, stream split"
for i in 1 2 3; do timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "cholesky or gp" 2>&1 | tail -1; done
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_match.py -q -x 2>&1 | tail -3
echo "== bench (bf16, 20 steps) x2, config 2 x2"
for k in 1 2; do
  timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-parity --no-roofline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=8 pairs/s',round(r['value'],2),'ms',round(r['ms_per_step'],3))"
  timeout 400 python bench.py --config coarse --steps 40 --warmup 5 --no-cpu-baseline --no-other-configs --no-parity --no-roofline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config2 pairs/s',round(r['value'],2),'ms',round(r['ms_per_step'],3))"
done
echo "== kernel stats of config 2 (chol_diag average)"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o c2 -- python "$REPO/bench.py" --config coarse --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --no-parity --no-other-configs > "$OUT/prof.log" 2>&1
cd "$REPO"
for f in $(find "$OUT/prof" -name "*kernel_stats.csv"); do grep -E "chol_diag|gemm_kernel<float, float" "$f" | cut -c1-150; done
rm -rf "$OUT/prof"
echo "== done"

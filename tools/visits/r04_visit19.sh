#!/bin/bash
# round 4, visit 19: backward substitution with one launch per step (M_k = Linv_kk L[k,:k] precomputed, second batch level): tests, A/B
set -u
OUT=$PWD/gpurun_out/v19
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== tests: operators, parity, stream split (ROMA_GP_BWD2 default = 1)"
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "cholesky or test_gp or gemm" 2>&1 | tail -2
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_match.py tests/test_gpu_f16.py -q -x 2>&1 | tail -3
echo "== A/B: ROMA_GP_BWD2 = 1, 0, 1, 0 (B = 8), then config 2"
for k in 1 0 1 0; do
  ROMA_GP_BWD2=$k timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-parity --no-roofline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bwd2 $k B=8 pairs/s',round(r['value'],2),'ms',round(r['ms_per_step'],3))"
done
for k in 1 0 1 0; do
  ROMA_GP_BWD2=$k timeout 400 python bench.py --config coarse --steps 40 --warmup 5 --no-cpu-baseline --no-other-configs --no-parity --no-roofline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bwd2 $k config2 pairs/s',round(r['value'],2),'ms',round(r['ms_per_step'],3))"
done
echo "== done"

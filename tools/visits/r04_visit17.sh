#!/bin/bash
# round 4, visit 17: forward substitution folded into the blocked Cholesky (augmented storage): tests, A/B
set -u
OUT=$PWD/gpurun_out/v17
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== tests: Cholesky / GP operators, f32 + 16-bit full parity, stream split"
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "cholesky or gp or multinomial" 2>&1 | tail -3
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_match.py -q -x 2>&1 | tail -3
echo "== A/B (bf16, 20 steps each): ROMA_GP_AUG = 1, 0, 1, 0"
for k in 1 0 1 0; do
  ROMA_GP_AUG=$k timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-parity --no-roofline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('aug $k pairs/s',round(r['value'],2),'ms',round(r['ms_per_step'],3))"
done
echo "== coarse-only B = 1 (config 2): aug 1, 0"
for k in 1 0; do
  ROMA_GP_AUG=$k timeout 400 python bench.py --config coarse --steps 40 --warmup 5 --no-cpu-baseline --no-other-configs --no-parity --no-roofline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config2 aug $k pairs/s',round(r['value'],2),'ms',round(r['ms_per_step'],3))"
done
echo "== done"

#!/bin/bash
# round 4, visit 24: the GP's Gram matrices (EPI_COSK, f32 out) through the staged row writer instead of 32-byte scatter stores
set -u
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "test_gp or cholesky or cosk or gemm" 2>&1 | tail -2
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_match.py -q -x 2>&1 | tail -2
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-parity 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pairs/s',round(r['value'],2),'ms',round(r['ms_per_step'],3),{n:(round(v['ms_per_step'],3),v['calls_per_step'],round(v.get('TFLOP/s',0))) for n,v in r['kernels'].items() if 'f32,' in n and 'gemm_kernel<bf16' in n})"
timeout 400 python bench.py --config coarse --steps 40 --warmup 5 --no-cpu-baseline --no-other-configs --no-parity --no-roofline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config2 pairs/s',round(r['value'],2),'ms',round(r['ms_per_step'],3))"
echo "== done"

#!/bin/bash
# round 5, visit 7: FINAL form of the narrow fused refiner blocks (composed out_conv on the MFMA, per-pixel deltas): op tests,
# parity, and the whole match() against the previous build's numbers (compose wide-only = 84.5 ms in visit 6)
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/v7; mkdir -p "$OUT"
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "refiner_block_final or refiner_block_fused or refiner_block_repeated" 2>&1 | tail -5 | tee "$OUT/tests.log"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_match.py -q -x 2>&1 | tail -4 | tee -a "$OUT/tests.log"
for c in 1 0 1; do
ROMA_COMPOSE_OUT=$c timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
p=r['parity']
print('compose=$c pairs/s',round(r['value'],2),'ms',round(r['ms_per_step'],3),{n:(round(v['ms_per_step'],3),v['calls_per_step']) for n,v in r['kernels'].items() if 'refiner_block' in n or 'refiner_out' in n or 'apply' in n})
print('   parity injected:', {k:(round(v['max'],6) if isinstance(v,dict) else v) for k,v in p.get('outputs_with_reference_coarse_match_injected',{}).items()})" | tee -a "$OUT/bench_ab.log"
done
echo "== done"

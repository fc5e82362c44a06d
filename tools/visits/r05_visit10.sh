#!/bin/bash
# round 5, visit 10: C = 24 wave block with the lane table (ring + Xt conflict free, shifted Xt rows) and direct row stores (no Ot tile): bit-identity tests,
# parity, A/B against visit 7's 84.6 ms, and the SQ counter pass for SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/v10; mkdir -p "$OUT"
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "refiner_block" 2>&1 | tail -4 | tee "$OUT/tests.log"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_match.py -q -x 2>&1 | tail -4 | tee -a "$OUT/tests.log"
for c in 1 2; do
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('run $c pairs/s',round(r['value'],2),'ms',round(r['ms_per_step'],3),{n:(round(v['ms_per_step'],3),v['calls_per_step']) for n,v in r['kernels'].items() if 'refiner_block' in n})" | tee -a "$OUT/bench.log"
done
bash tools/pmc_sq_round.sh 2>&1 | tail -30 | tee "$OUT/pmc_sq.log"
cp gpurun_out/pmc_sq_summary.json "$OUT/" 2>/dev/null
echo "== done"

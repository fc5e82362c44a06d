#!/bin/bash
# round 5, visit 11: C = 144 block, LDS-DMA ring of 4 rows (the LDS the output tile freed) against 3: operator tests with the
# deeper ring, then the bench A/B (kernel time of refiner_block<144> and the whole step)
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/v11; mkdir -p "$OUT"
ROMA_RB1_NR=4 timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "refiner_block" 2>&1 | tail -3 | tee "$OUT/tests.log"
for nr in 3 4 3 4; do
ROMA_RB1_NR=$nr timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-parity 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('NR=$nr pairs/s',round(r['value'],2),'ms',round(r['ms_per_step'],3),{n:(round(v['ms_per_step'],3),v['calls_per_step']) for n,v in r['kernels'].items() if 'refiner_block' in n})" | tee -a "$OUT/bench.log"
done
echo "== done"

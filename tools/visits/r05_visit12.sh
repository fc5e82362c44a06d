#!/bin/bash
# round 5, visit 12: VALU diet of the three rolling-window stencil kernels (new row seeded from the bias inside the first FMA
# instead of 16 v_mov per row; ReLU on the packed 16-bit pair, one v_pk_max_i16 instead of two v_max_f32): operator tests
# (bit-identity against the generic kernels), parity, bench
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/v12; mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "refiner_block or dwconv" 2>&1 | tail -3 | tee "$OUT/tests.log"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_match.py -q -x 2>&1 | tail -3 | tee -a "$OUT/tests.log"
for c in 1 2; do
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('run $c pairs/s',round(r['value'],2),'ms',round(r['ms_per_step'],3),{n:(round(v['ms_per_step'],3),v['calls_per_step']) for n,v in r['kernels'].items() if 'refiner_block' in n or 'dwconv' in n})" | tee -a "$OUT/bench.log"
done
echo "== done"

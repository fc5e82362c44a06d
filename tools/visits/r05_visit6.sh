#!/bin/bash
# round 5, visit 6: out_conv composed with the last block's 1x1 (linear o linear): parity + A/B of the whole match()
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/v6; mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -4 | tee "$OUT/tests.log"
for c in 1 0 1; do
ROMA_COMPOSE_OUT=$c timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
p=r['parity']
print('compose=$c pairs/s',round(r['value'],2),'ms',round(r['ms_per_step'],3),{n:(round(v['ms_per_step'],3),v['calls_per_step']) for n,v in r['kernels'].items() if 'dwconv' in n or 'ws1x1' in n or 'gemm6p' in n or 'refiner_block' in n or 'refiner_out' in n})
print('   parity injected:', {k:(round(v,6) if isinstance(v,float) else v) for k,v in p.get('outputs_with_reference_coarse_match_injected',{}).items()})
print('   flips:', p['coarse_argmax'])" | tee -a "$OUT/bench_ab.log"
done
ROMA_COMPOSE_OUT=1 timeout 300 python bench.py --dtype f32 --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-roofline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f32 composed: pairs/s',round(r['value'],2), r['parity']['outputs'])" | tee -a "$OUT/bench_ab.log"
echo "== done"

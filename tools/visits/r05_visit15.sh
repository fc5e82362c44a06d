#!/bin/bash
# round 5, visit 15: slab-major against tap-major conv K order once more, alternating three times on one box (the closing visit's
# box read the conv kernel 6 % slower than the previous box with the other order - box or order?)
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/v15; mkdir -p "$OUT"
for ko in 1 0 1 0 1 0; do
ROMA_CONV_KORDER=$ko timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-parity 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('korder=$ko pairs/s',round(r['value'],2),'ms',round(r['ms_per_step'],3),{n:(round(v['ms_per_step'],3),round(v.get('TFLOP/s',0),1)) for n,v in r['kernels'].items() if 'conv3x3,relu' in n or 'dense,gelu' in n})" | tee -a "$OUT/bench.log"
done
echo "== done"

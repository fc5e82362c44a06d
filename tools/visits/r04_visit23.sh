#!/bin/bash
# round 4, visit 23: attention hd = 64 at four workgroups per CU (128-register cap, ~22 spilled registers) against three
set -u
export TMPDIR=/tmp
timeout 600 env ROMA_ATTN_OCC4=1 python -m pytest tests/test_gpu_ops.py -q -x -k "attention" 2>&1 | tail -2
run() { timeout 400 env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-parity 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=r['kernels']
print('$*','pairs/s',round(r['value'],2),'ms',round(r['ms_per_step'],3),{n:(round(v['ms_per_step'],3),round(v.get('TFLOP/s',0))) for n,v in k.items() if 'attn' in n})"; }
run ROMA_ATTN_OCC4=0
run ROMA_ATTN_OCC4=1
run ROMA_ATTN_OCC4=0
run ROMA_ATTN_OCC4=1
echo "== done"

#!/bin/bash
# attention v2 with V^T fragments prefetched under the softmax (hd = 64, two workgroups per CU): tests + A/B against v1 and the row-sum variants
set -u
OUT=$PWD/gpurun_out/v11
mkdir -p "$OUT"
echo "== attention tests"
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "attention or qkv" 2>&1 | tail -5
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-other-configs > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"
  python -c "
import json; r=json.load(open('$OUT/bench_$name.json')); k=r['kernels']
print('$name', round(r['value'],2), 'pairs/s', {n:(round(x['ms_per_step'],3), round(x.get('TFLOP/s',0))) for n,x in k.items() if n.startswith('attn')})"
}
run v1 ROMA_ATTN_V=1
run v2 ROMA_ATTN_V=2
run v2_novpre ROMA_ATTN_V=2 ROMA_ATTN_VPRE=0
run v1b ROMA_ATTN_V=1
run v2b ROMA_ATTN_V=2
echo "== done"

#!/bin/bash
# one-barrier fused refiner block for C = 144: tests + A/B against the two-barrier kernel
set -u
OUT=$PWD/gpurun_out/v16
mkdir -p "$OUT"
echo "== refiner_block tests"
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_f16.py -m gpu -q -k "refiner_block" 2>&1 | tail -8
echo "== kernel alone, 16 x 432^2 x 144"
for v in 0 1 0 1; do echo "rb144_1b=$v"; ROMA_RB144_1B=$v timeout 120 python tools/bench_refiner_block.py 2>&1 | grep "C=144"; done
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-other-configs > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"
  python -c "
import json; r=json.load(open('$OUT/bench_$name.json')); k=r['kernels']
print('$name', round(r['value'],2), 'pairs/s', {n:(round(x['ms_per_step'],3), round(x.get('GB/s',0))) for n,x in k.items() if n.startswith('refiner_block')})"
}
run two ROMA_RB144_1B=0
run one ROMA_RB144_1B=1
run two_b ROMA_RB144_1B=0
run one_b ROMA_RB144_1B=1
echo "== done"

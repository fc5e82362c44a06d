#!/bin/bash
# round 5, visit 3: the fused C = 576 ConvRefiner block (refiner_block_wide.hip), v2 (two wave groups, stencil and MFMA de-phased; taps by LDS-DMA): correctness,
# A/B against dwconv5x5 + ws1x1 at the model's shapes, then the whole match()
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/v3; mkdir -p "$OUT"
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "refiner_block_wide" 2>&1 | tail -8 | tee "$OUT/tests.log"
timeout 300 python tools/bench_refiner_wide.py 2>&1 | tee "$OUT/bench_refiner_wide.log"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_match.py -q -x 2>&1 | tail -4 | tee -a "$OUT/tests.log"
for rb in 1 0; do
ROMA_RB_WIDE=$rb timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-parity 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rb_wide=$rb pairs/s',round(r['value'],2),'ms',round(r['ms_per_step'],3),{n:(round(v['ms_per_step'],3),v['calls_per_step']) for n,v in r['kernels'].items() if 'refiner_block_wide' in n or 'dwconv' in n or 'ws1x1' in n})" | tee -a "$OUT/bench_ab.log"
done
echo "== done"

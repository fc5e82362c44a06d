#!/bin/bash
# Round 6, visit 4: fused max-pool + proj head (pool_proj.hip); full GPU suite over everything of the round so far.
set -u
OUT=$PWD/gpurun_out/v4; rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== operator test"
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "pool_proj or local_corr" 2>&1 | tail -8 | tee "$OUT/pytest_ops.log"
echo "== bench A/B (mixed, two streams)"
for i in 1 2; do
for v in 0 1; do
  ROMA_POOL_PROJ=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-other-configs --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pool_proj=$v', d['dtype'], round(d['value'],2), 'pairs/s', round(d['ms_per_step'],2), 'ms')" | tee -a "$OUT/bench_ab.log"
done; done
echo "== full GPU suite"
timeout 1700 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 | tee "$OUT/pytest_gpu.log"
echo "== kernels (one instrumented pass)"
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-other-configs > "$OUT/bench_kernels.json" 2>/dev/null
python - "$OUT/bench_kernels.json" <<'P'
import json, sys
d = json.load(open(sys.argv[1]))
print(d["value"], d["ms_per_step"])
for k, v in list(d["kernels"].items())[:45]:
    print(f"{v['ms_per_step']:7.3f} ms  x{v['calls_per_step']:5.1f}  {k}  {v.get('TFLOP/s', v.get('GB/s')):.0f}")
P
echo "== done"

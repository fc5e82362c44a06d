#!/bin/bash
# Round 6, visit 11: fused backward substitution of the GP solve.
set -u
OUT=$PWD/gpurun_out/v11; rm -rf "$OUT"; mkdir -p "$OUT"
echo "== operator tests"
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "cholesky or gp_posterior" 2>&1 | tail -8 | tee "$OUT/pytest_ops.log"
echo "== the chain alone"
timeout 300 python tools/bench_gp.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/bench_gp.log"
echo "== parity f32"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "f32" 2>&1 | tail -4 | tee "$OUT/pytest_parity.log"
echo "== bench A/B (mixed, two streams)"
for i in 1 2 3; do
for v in 0 1; do
  ROMA_GP_BWD=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-other-configs --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('gp_bwd=$v', round(d['value'],2), 'pairs/s', round(d['ms_per_step'],2), 'ms')" | tee -a "$OUT/bench_ab.log"
done; done
echo "== done"

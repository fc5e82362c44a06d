#!/bin/bash
# Round 6, visit 7: register-resident inverse wave of the diagonal block; 128 x 128 tiles for small exact-f32 launches (GP posterior mean).
set -u
OUT=$PWD/gpurun_out/v7; rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== operator tests"
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "cholesky or gp_posterior or gemm" 2>&1 | tail -8 | tee "$OUT/pytest_ops.log"
echo "== the chain alone"
timeout 300 python tools/bench_gp.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/bench_gp.log"
echo "== parity f32"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_match.py -q -x 2>&1 | tail -5 | tee "$OUT/pytest_parity.log"
echo "== bench A/B: f32 fill rule (mixed, two streams)"
for i in 1 2; do
for v in 0 1; do
  ROMA_GEMM_F32_FILL=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-other-configs --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('f32_fill=$v', d['dtype'], round(d['value'],2), 'pairs/s', round(d['ms_per_step'],2), 'ms')" | tee -a "$OUT/bench_ab.log"
done; done
echo "== config 2 (coarse-only, B = 1)"
timeout 300 python bench.py --config coarse --batch 1 --steps 30 --warmup 5 --no-cpu-baseline --no-parity --no-other-configs --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('coarse', round(d['ms_per_step'],2), 'ms')" | tee -a "$OUT/bench_ab.log"
echo "== done"

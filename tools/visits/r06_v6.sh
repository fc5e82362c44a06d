#!/bin/bash
# Round 6, visit 6: band-blocked register-resident diagonal-block factorisation (chol_diag.h).
set -u
OUT=$PWD/gpurun_out/v6; rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== operator tests"
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "cholesky or gp_posterior" 2>&1 | tail -8 | tee "$OUT/pytest_ops.log"
echo "== the chain alone"
timeout 300 python tools/bench_gp.py 2>&1 | tee "$OUT/bench_gp.log"
echo "== parity f32"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "f32" 2>&1 | tail -5 | tee "$OUT/pytest_parity.log"
echo "== bench (mixed, two streams) x3"
for i in 1 2 3; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-other-configs --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['dtype'], round(d['value'],2), 'pairs/s', round(d['ms_per_step'],2), 'ms')" | tee -a "$OUT/bench.log"
done
echo "== done"

#!/bin/bash
# Round 6, visit 3: local correlation - the queries of incoherent tiles sorted by target bin, served by the LIST form of the tile kernel.
set -u
OUT=$PWD/gpurun_out/v3; rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== operator tests"
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "local_corr" 2>&1 | tail -15 | tee "$OUT/pytest_ops.log"
echo "== the kernel alone"
timeout 600 python tools/bench_local_corr.py 2>&1 | tee "$OUT/bench_local_corr.log" | cut -c1-900
echo "== bench A/B (mixed, two streams)"
for i in 1 2; do
for v in 0 1; do
  ROMA_LC_BIN=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-other-configs --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lc_bin=$v', d['dtype'], round(d['value'],2), 'pairs/s', round(d['ms_per_step'],2), 'ms')" | tee -a "$OUT/bench_ab.log"
done; done
echo "== parity (f32, small configs + full8)"
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "f32" 2>&1 | tail -8 | tee "$OUT/pytest_parity.log"
echo "== done"

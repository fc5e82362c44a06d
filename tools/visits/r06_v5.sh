#!/bin/bash
# Round 6, visit 5: patch-resident VGG convolution (conv_patch.hip).
set -u
OUT=$PWD/gpurun_out/v5; rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== operator tests"
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "conv3x3" 2>&1 | tail -12 | tee "$OUT/pytest_ops.log"
echo "== the kernel alone"
timeout 600 python tools/bench_conv_patch.py 2>&1 | tee "$OUT/bench_conv_patch.log"
echo "== bench A/B (mixed, two streams)"
for i in 1 2; do
for v in 0 1; do
  ROMA_CONV_PATCH=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-other-configs --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('conv_patch=$v', d['dtype'], round(d['value'],2), 'pairs/s', round(d['ms_per_step'],2), 'ms')" | tee -a "$OUT/bench_ab.log"
done; done
echo "== parity"
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_match.py tests/test_gpu_f16.py -q -x 2>&1 | tail -8 | tee "$OUT/pytest_parity.log"
echo "== done"

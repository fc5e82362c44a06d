#!/bin/bash
# Round 6, visit 10: local correlation r = 7 on 512-thread workgroups; full suite.
set -u
OUT=$PWD/gpurun_out/v10; rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== operator tests"
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "local_corr" 2>&1 | tail -5 | tee "$OUT/pytest_ops.log"
echo "== the kernel alone"
timeout 600 python tools/bench_local_corr.py 2>&1 | grep -v amdgpu | python -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{'): print(l.rstrip()); continue
    d = json.loads(l)
    print(d['warp'], 'r', d['r'], d['hw'], d['dtype'], {k: (v['ms'], v['algorithmic_GBs']) for k, v in d.items() if isinstance(v, dict) and 'ms' in v})
" | tee "$OUT/bench_local_corr.log"
echo "== full GPU suite"
timeout 1700 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 | tee "$OUT/pytest_gpu.log"
echo "== bench x2"
for i in 1 2; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-other-configs --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2), 'pairs/s', round(d['ms_per_step'],2), 'ms')" | tee -a "$OUT/bench.log"
done
echo "== done"

#!/bin/bash
# Round-4 visit 2: which configuration is not reproducible at the benchmark size? (mixed failed in visit 1)
set -u
OUT=$PWD/gpurun_out/v2
mkdir -p "$OUT"
export TMPDIR=/tmp
for s in 0 1; do
  echo "== reproducibility at 560 -> 864, B = 8: ROMA_GEMM8P_SCHED=$s"
  ROMA_GEMM8P_SCHED=$s timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "reproducible" 2>&1 | grep -E "passed|failed|FAILED|AssertionError|assert|^E " | head -20
done
echo "== done"

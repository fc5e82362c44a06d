#!/bin/bash
set -u
export TMPDIR=/tmp
for w in 1 3 5 9 7 13; do echo "== ROMA_WS1X1=$w (1 on, +2 no stores, +4 no DMA, +8 no MFMA)"; ROMA_WS1X1=$w timeout 300 python tools/bench_vendor_gemm.py 2>&1 | grep "stride 4" | cut -c1-110; done
echo "== done"

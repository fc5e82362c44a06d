#!/bin/bash
# Round-2 GPU visit 15: Tiny RoMa on the device against the reference golden
set -u
OUT=$PWD/gpurun_out/v15
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_tiny.py -q -x -s 2>&1 | tail -25
echo "== done"

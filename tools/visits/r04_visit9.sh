#!/bin/bash
set -u
export TMPDIR=/tmp
run() { echo "== $*"; env "$@" 2>&1 | grep -v "amdgpu.ids" | cut -c1-700; }
R="timeout 900 python tools/repro_mixed.py --others 0 --rounds 16"
run HIP_FORCE_DEV_KERNARG=0 $R
run HIP_FORCE_DEV_KERNARG=1 $R
run $R
run $R --amp bf16
echo "== done"

#!/bin/bash
set -u
export TMPDIR=/tmp
run() { echo "== $*"; env "$@" 2>&1 | grep -v "amdgpu.ids" | cut -c1-1500; }
run timeout 900 python tools/repro_mixed.py --others 0 --rounds 80 --trace
run timeout 900 python tools/repro_mixed.py --others 0 --rounds 40 --trace --amp f16
echo "== done"

#!/bin/bash
set -u
export TMPDIR=/tmp
run() { echo "== $*"; env "$@" 2>&1 | grep -v "amdgpu.ids" | cut -c1-1800; }
run timeout 1200 python tools/repro_mixed.py --others 0 --rounds 200 --trace
echo "== done"

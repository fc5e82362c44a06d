#!/bin/bash
set -u
export TMPDIR=/tmp
run() { echo "== $*"; env "$@" 2>&1 | grep -v "amdgpu.ids" | cut -c1-900; }
R="timeout 900 python tools/repro_mixed.py --others 0 --rounds 16"
run ROMA_RB24W=0 $R
run ROMA_RB144_1B=0 $R
run ROMA_RB24W=0 $R --amp f16
run ROMA_RB144_1B=0 $R --amp f16
echo "== done"

#!/bin/bash
set -u
export TMPDIR=/tmp
run() { echo "== $*"; env "$@" 2>&1 | grep -v "amdgpu.ids" | cut -c1-1500; }
R="timeout 900 python tools/repro_mixed.py --others 0"
run $R --rounds 100 --trace
run ROMA_STREAMS_SERIAL=1 $R --rounds 12
run $R --rounds 12 --fuse 0
run ROMA_CONV64=0 $R --rounds 12
run ROMA_DW_RING=0 $R --rounds 12
run ROMA_GEMM8P=0 $R --rounds 12
run ROMA_LC_MODE=2 $R --rounds 12
run ROMA_RB24W=0 ROMA_RB144_1B=0 $R --rounds 12
run ROMA_ATTN_XCD=0 $R --rounds 12
run $R --rounds 12 --amp f16
echo "== done"

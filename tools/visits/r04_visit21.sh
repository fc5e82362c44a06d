#!/bin/bash
# round 4, visit 21: long determinism stress at the benchmark size WITHOUT the stage trace (the round-4 hazard was 1 in 50
# two-stream calls in bf16, 19 in 49 in mixed mode): 1000 calls per 16-bit mode against the single-stream result, bit-exact
set -u
export TMPDIR=/tmp
for amp in bf16 mixed f16; do
  timeout 600 python tools/stress_streams.py --pairs 8 --res 560 864 --runs 1000 --amp $amp 2>&1 | tail -1 | cut -c1-260
done
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
echo "== done"

#!/bin/bash
# round 4, visit 22: uneven two-stream split (5 + 3, 6 + 2 pairs) and three streams against the balanced split (experiment switch)
set -u
export TMPDIR=/tmp
run() { timeout 400 env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-parity --no-roofline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', 'pairs/s',round(r['value'],2),'ms',round(r['ms_per_step'],3))"; }
for rep in 1 2; do
  run ROMA_STREAM_SPLIT0=0
  run ROMA_STREAM_SPLIT0=5
  run ROMA_STREAM_SPLIT0=6
  run ROMA_STREAMS=3
done
echo "== done"

#!/bin/bash
# round 4, visit 20: after removing the local-correlation A/B paths (lc_mode 3, 8-wave list): operator + model tests, bench
set -u
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "local_corr" 2>&1 | tail -2
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_match.py -q -x 2>&1 | tail -2
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-parity --no-roofline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=8 pairs/s',round(r['value'],2),'ms',round(r['ms_per_step'],3))"
echo "== done"

#!/bin/bash
# round 5, visit 1: the full GPU suite on the pruned build (gemm families in five objects, tools-only A/B variants gone, -Bsymbolic,
# roma_forward, path / PIL goldens, chol_diag release / acquire) + a baseline bench line of the same box for the round's A/Bs
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/v1; mkdir -p "$OUT"
timeout 1700 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 | tee "$OUT/pytest_gpu.log"
timeout 600 python -m pytest tests/test_gpu_match.py tests/test_gpu_tiny.py -q -s -k "paths or forward_apis or from_path" 2>&1 | grep -i "max|\|passed\|failed\|forward APIs" | cut -c1-400 | tee "$OUT/new_tests.log"
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-parity > "$OUT/bench.json" 2> "$OUT/bench.err"
python - <<'P'
import json
r=json.loads(open("gpurun_out/v1/bench.json").read().strip().splitlines()[-1])
print('pairs/s',round(r['value'],2),'ms',round(r['ms_per_step'],3))
for n,v in list(r['kernels'].items())[:40]:
    print(f"{n[:70]:70s} {v['ms_per_step']:8.3f} ms {v['calls_per_step']:6.1f} calls", {k:round(x,1) for k,x in v.items() if k in('TFLOP/s','GB/s')})
P
timeout 300 python bench.py --config coarse --steps 40 --warmup 5 --no-cpu-baseline --no-other-configs --no-parity --no-roofline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config2 pairs/s',round(r['value'],2),'ms',round(r['ms_per_step'],3))"
echo "== done"

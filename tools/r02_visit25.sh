#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/v25
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "cholesky or gp" 2>&1 | tail -3
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --streams 1 > gpurun_out/v25/bench_1s.json 2> gpurun_out/v25/bench_1s.err
python - <<'PY'
import json
l=[x for x in open("gpurun_out/v25/bench_1s.json") if x.startswith("{")]
d=json.loads(l[-1]); print("1 stream", round(d["value"],2), round(d["ms_per_step"],2))
ks=d.get("kernels") or []
for k in ks[:40]:
    print(k)
PY

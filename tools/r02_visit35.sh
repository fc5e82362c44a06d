#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/v35
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "refiner_input" 2>&1 | tail -2
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity --streams 1 > gpurun_out/v35/bench_1s.json 2> gpurun_out/v35/bench_1s.err
python - <<'PY'
import json
l=[x for x in open("gpurun_out/v35/bench_1s.json") if x.startswith("{")]
d=json.loads(l[-1]); print("1 stream", round(d["value"],2), round(d["ms_per_step"],2))
for k,v in d["kernels"].items():
    if "refiner_input" in k: print(k, v)
PY

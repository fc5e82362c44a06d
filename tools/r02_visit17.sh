#!/bin/bash
# visit 17: conv64 (weight-stationary 3x3, Cin = 64) - correctness, microbench, bench A/B
cd /root/repo
mkdir -p gpurun_out/v17
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "conv3x3" > gpurun_out/v17/pytest_conv.txt 2>&1
tail -5 gpurun_out/v17/pytest_conv.txt
timeout 300 python tools/bench_conv64.py > gpurun_out/v17/bench_conv64.log 2>&1
cat gpurun_out/v17/bench_conv64.log
for m in 0 1; do
  ROMA_CONV64=$m timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/v17/bench_conv64_$m.json 2> gpurun_out/v17/bench_conv64_$m.err
  python - <<PY
import json
l=[x for x in open("gpurun_out/v17/bench_conv64_$m.json") if x.startswith("{")]
d=json.loads(l[-1]); print("conv64=$m", d["value"], d["ms_per_step"], d.get("parity"))
PY
done

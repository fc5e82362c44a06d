#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/v26
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
for n in 2 3 4 2; do
  ROMA_STREAMS=$n timeout 400 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-parity --no-roofline > gpurun_out/v26/bench_s$n.json 2> gpurun_out/v26/bench_s$n.err
  python - <<PY
import json
l=[x for x in open("gpurun_out/v26/bench_s$n.json") if x.startswith("{")]
d=json.loads(l[-1]); print("streams=$n", round(d["value"],2), round(d["ms_per_step"],2))
PY
done

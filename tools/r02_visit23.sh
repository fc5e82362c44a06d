#!/bin/bash
# visit 23: VGG front-end kernels (conv64.hip) - op tests, microbench, whole-model A/B, match-level parity tests
cd /root/repo
mkdir -p gpurun_out/v23
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "conv" 2>&1 | tail -3
timeout 300 python tools/bench_conv64.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/v23/bench_conv64.log
for m in 0 7; do
  ROMA_CONV64=$m timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/v23/bench_conv64_$m.json 2> gpurun_out/v23/bench_conv64_$m.err
  python - <<PY
import json
l=[x for x in open("gpurun_out/v23/bench_conv64_$m.json") if x.startswith("{")]
d=json.loads(l[-1]); print("conv64=$m", round(d["value"],2), round(d["ms_per_step"],2), d.get("parity",{}).get("outputs",{}).get("flow",{}).get("p50"), d.get("parity",{}).get("outputs_with_reference_coarse_match_injected",{}).get("flow",{}))
PY
done
timeout 900 python -m pytest tests/test_gpu_match.py -x -q 2>&1 | tail -5

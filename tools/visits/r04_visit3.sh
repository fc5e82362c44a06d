#!/bin/bash
# Round-4 visit 3: the mixed mode is not reproducible at the benchmark size (2 of 3 five-call sequences) - how often, and
# which stage deviates first?  (per-stage checksum trace, tools/stress_streams.py)
set -u
OUT=$PWD/gpurun_out/v3
mkdir -p "$OUT"
export TMPDIR=/tmp
for amp in mixed f16 bf16; do
  echo "== $amp, 560 -> 864, B = 8, 40 two-stream runs against the single-stream result, stage trace"
  timeout 600 python tools/stress_streams.py --pairs 8 --res 560 864 --runs 40 --amp $amp --trace 2>&1 | grep -v "amdgpu.ids" | cut -c1-900
done
echo "== mixed, streams serialised (same streams and arenas, no overlap)"
ROMA_STREAMS_SERIAL=1 timeout 600 python tools/stress_streams.py --pairs 8 --res 560 864 --runs 40 --amp mixed --trace 2>&1 | grep -v "amdgpu.ids" | cut -c1-900
echo "== done"

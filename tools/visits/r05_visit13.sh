#!/bin/bash
# round 5, visit 13: extended determinism stress of the closing build at the benchmark size (two streams vs one, bit-exact):
# the fused-block store path and lane maps changed after the round's first 420-call stress
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/v13; mkdir -p "$OUT"
for seed in 11 12 13 14; do timeout 600 python tools/stress_streams.py --pairs 8 --res 560 864 --runs 500 --amp bf16 --seed $seed 2>&1 | tail -1 | cut -c1-260; done | tee "$OUT/stress.log"
timeout 600 python tools/stress_streams.py --pairs 8 --res 560 864 --runs 500 --amp mixed 2>&1 | tail -1 | cut -c1-260 | tee -a "$OUT/stress.log"
timeout 600 python tools/stress_streams.py --pairs 8 --res 560 864 --runs 500 --amp f16 2>&1 | tail -1 | cut -c1-260 | tee -a "$OUT/stress.log"
echo "== done"

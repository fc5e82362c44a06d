#!/bin/bash
# Round 6, visit 22: TIMING-ONLY ablation - gemm8p with its steady-state counted vmcnt waits turned into no-ops (tools/scratch/abl_relax,
# built from gemm8p.hip with R8_WAIT_VM(4) -> vmcnt(12); results are garbage): how much of the K tile's 1.82 us on a full chip is
# spent waiting for the operand DMA (i.e. would more look-ahead buy anything)?
set -u
OUT=$PWD/gpurun_out/v22; rm -rf "$OUT"; mkdir -p "$OUT"
for i in 1 2; do
  echo "-- shipped waits"; timeout 600 python tools/bench_gemm_burst.py 2>&1 | grep -v amdgpu | tee -a "$OUT/burst_tree.log"
  echo "-- no steady-state waits"; ROMA_LIB_DIR=$PWD/tools/scratch/abl_relax timeout 600 python tools/bench_gemm_burst.py 2>&1 | grep -v amdgpu | tee -a "$OUT/burst_relax.log"
done
echo "== done"

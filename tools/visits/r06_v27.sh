#!/bin/bash
# Round 6, visit 27: gemm8p STEADY copy with the phase's two DMA pieces issued AHEAD of its fragment reads (-DROMA_R8_DMA_FIRST,
# tools/scratch/ab_df) against reads-first (tree) - now that the DMA issue carries no VALU.
set -u
OUT=$PWD/gpurun_out/v27; rm -rf "$OUT"; mkdir -p "$OUT"
for i in 1 2 3; do
  echo "-- reads first"; timeout 300 python tools/bench_gemm_epilogue.py 2>&1 | grep -v amdgpu | tee -a "$OUT/reads_first.log"
  echo "-- dma first"; ROMA_LIB_DIR=$PWD/tools/scratch/ab_df timeout 300 python tools/bench_gemm_epilogue.py 2>&1 | grep -v amdgpu | tee -a "$OUT/dma_first.log"
done
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-other-configs --no-roofline"
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["value"],2), "pairs/s", round(d["ms_per_step"],2), "ms")'
for i in 1 2 3; do
  timeout 300 $B 2>/dev/null | python -c "$P" "reads-first" | tee -a "$OUT/bench_ab.log"
  ROMA_LIB_DIR=$PWD/tools/scratch/ab_df timeout 300 $B 2>/dev/null | python -c "$P" "dma-first" | tee -a "$OUT/bench_ab.log"
done
echo "== done"

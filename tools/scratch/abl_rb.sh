#!/bin/bash
# ablation of the fused refiner block kernel (ROMA_RB_DBG bits: 1 no dw FMAs, 2 no MFMA, 4 no stores, 8 no DMA refill)
for d in 0 1 2 4 8 3 7 15; do
  echo "== ROMA_RB_DBG=$d"
  ROMA_RB_DBG=$d timeout 200 python tools/bench_refiner.py 2>&1 | grep fused
done

// Ping-pong (half-slab staggered wave rows) variant of the 256 x 256 bf16 MFMA GEMM - see gemm_pp.hip.
#pragma once
#include "gemm.h"

namespace roma {
// true when `a` (after gemm_launch's normalisation) can run on the ping-pong kernel: bf16 dense A, big M, an N that
// tiles by 256, and a plain / QKV epilogue that the LDS-staged epilogues cover.
bool gemm_pp_eligible(const GemmArgs& a);
int gemm_pp_launch(const GemmArgs& a, hipStream_t stream);
// gemm_launch with the ping-pong kernel in front: what model.hip / api.hip call.
int gemm_dispatch(const GemmArgs& a, hipStream_t stream);
}  // namespace roma

// One instantiation family of gemm_kernel (gemm_kernel.inc; dispatch in gemm.hip): <bf16_t, bf16_t, CONV = false>.
#include "gemm_kernel.inc"

namespace roma {
int gemm_family_h16(const GemmArgs& a, hipStream_t stream) { return launch_shape<bf16_t, bf16_t, false>(a, stream); }
}  // namespace roma

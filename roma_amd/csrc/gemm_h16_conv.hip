// One instantiation family of gemm_kernel (gemm_kernel.inc; dispatch in gemm.hip): <bf16_t, bf16_t, CONV = true>.
#include "gemm_kernel.inc"

namespace roma {
int gemm_family_h16_conv(const GemmArgs& a, hipStream_t stream) { return launch_shape<bf16_t, bf16_t, true>(a, stream); }
}  // namespace roma

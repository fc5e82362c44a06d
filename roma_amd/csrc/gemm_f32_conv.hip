// One instantiation family of gemm_kernel (gemm_kernel.inc; dispatch in gemm.hip): <float, float, CONV = true>.
#include "gemm_kernel.inc"

namespace roma {
int gemm_family_f32_conv(const GemmArgs& a, hipStream_t stream) { return launch_shape<float, float, true>(a, stream); }
}  // namespace roma

// One instantiation family of gemm_kernel (gemm_kernel.inc; dispatch in gemm.hip): <float, float, CONV = false>.
#include "gemm_kernel.inc"

namespace roma {
int gemm_family_f32(const GemmArgs& a, hipStream_t stream) { return launch_shape<float, float, false>(a, stream); }
}  // namespace roma

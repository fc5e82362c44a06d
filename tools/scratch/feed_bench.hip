// Per-CU operand-feed microbenchmark: how fast can one 512-thread workgroup per CU pull L2-resident 64 KB slabs
// into LDS?  V0: global_load_lds (8 rows x 128 B per instruction, the GEMM pattern)  V1: global_load_lds contiguous
// 1 KiB  V2: global_load_dwordx4 -> VGPR -> ds_write_b128 (8-row pattern)  V3: global_load_dwordx4 -> VGPR only
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

template <int V>
__global__ __launch_bounds__(512, 1) void feed(const char* __restrict__ src, long region_bytes, int nslab, long ld_bytes, unsigned* sink) {
  __shared__ __attribute__((aligned(1024))) char lds[2][65536];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const char* base = src + (long)(blockIdx.x % 8) * region_bytes;  // one region per XCD (blocks b -> XCD b%8)
  unsigned acc = 0;
  const int r8 = lane >> 3, slot = lane & 7;
  for (int s = 0; s < nslab; ++s) {
    const int buf = s & 1;
    // slab s: 512 rows x 128 B; row pitch ld_bytes; slab advances 128 B along the row (K direction), wraps in region
    const long koff = ((long)s * 128) % ld_bytes;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int piece = wave * 8 + j;  // 64 pieces of 1 KiB per slab
      if (V == 0) {
        const int row = piece * 8 + r8;
        const char* p = base + (long)row * ld_bytes + koff + slot * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                         (__attribute__((address_space(3))) void*)(&lds[buf][piece * 1024]), 16, 0, 0);
      } else if (V == 1) {
        const char* p = base + ((long)(s % 32) * 65536 + piece * 1024 + lane * 16);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                         (__attribute__((address_space(3))) void*)(&lds[buf][piece * 1024]), 16, 0, 0);
      }
    }
    if (V == 2 || V == 3) {
      uint4 r[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int piece = wave * 8 + j;
        const int row = piece * 8 + r8;
        r[j] = *reinterpret_cast<const uint4*>(base + (long)row * ld_bytes + koff + slot * 16);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (V == 2) *reinterpret_cast<uint4*>(&lds[buf][(wave * 8 + j) * 1024 + lane * 16]) = r[j];
        else acc ^= r[j].x ^ r[j].y ^ r[j].z ^ r[j].w;
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (V != 3) acc ^= *reinterpret_cast<const unsigned*>(&lds[buf][tid * 4]);
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int V> static void run(const char* name, const char* d, long region, int nslab, long ld, unsigned* sink) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(feed<V>, dim3(256), dim3(512), 0, 0, d, region, nslab, ld, sink);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(feed<V>, dim3(256), dim3(512), 0, 0, d, region, nslab, ld, sink);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
  const double bytes = 256.0 * nslab * 65536.0;
  printf("%-44s ld=%6ld: %7.3f ms  %7.2f TB/s aggregate  %6.1f GB/s per CU  (%5.1f B/clk @2.4GHz)\n", name, ld, ms,
         bytes / ms * 1e-9, bytes / 256 / ms * 1e-6, bytes / 256 / (ms * 1e-3) / 2.4e9);
}

int main() {
  const long region = 512l * 16384;  // 8 MB per XCD region (512 rows x 16 KB pitch max)
  char* d; unsigned* sink;
  CK(hipMalloc(&d, region * 8 + 65536)); CK(hipMemset(d, 1, region * 8 + 65536)); CK(hipMalloc(&sink, 64));
  for (long ld : {2048l, 16384l}) {
    run<0>("V0 global_load_lds, 8 rows x 128 B / instr", d, region, 512, ld, sink);
    run<1>("V1 global_load_lds, contiguous 1 KiB / instr", d, region, 512, ld, sink);
    run<2>("V2 global_load_dwordx4 -> ds_write_b128", d, region, 512, ld, sink);
    run<3>("V3 global_load_dwordx4 -> VGPR only", d, region, 512, ld, sink);
  }
  return 0;
}

// Issue rate of v_pk_fma_f32 against v_fma_f32 under different operand patterns (gfx950).  One or two waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x2 acc[16];
  f32x2 a[8], b[8];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = f32x2{(float)threadIdx.x, 1.f + i};
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = f32x2{1.0001f + i * 1e-6f, 0.9999f}; b[i] = f32x2{1e-7f * (i + 1), 2e-7f}; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (MODE == 0) {  // packed, all operands distinct registers (a, b cycle over 8)
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[(i + r) & 7]), "v"(b[i & 7]));
        } else if (MODE == 1) {  // packed, shared multiplier b[0]
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[(i + r) & 7]), "v"(b[0]));
        } else if (MODE == 2) {  // packed, both multiplicands shared
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[0]), "v"(b[0]));
        } else if (MODE == 3) {  // two scalar FMAs
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i][0]) : "v"(a[(i + r) & 7][0]), "v"(b[i & 7][0]));
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i][1]) : "v"(a[(i + r) & 7][1]), "v"(b[i & 7][1]));
        } else if (MODE == 4) {  // packed multiply by a shared register pair, add distinct (pk_fma with src2 = acc)
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i & 7]), "v"(a[i & 7]));
        } else if (MODE == 5) {  // v_pk_mul_f32 + v_pk_add_f32
          f32x2 t;
          asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(a[(i + r) & 7]), "v"(b[i & 7]));
          asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(acc[i]) : "v"(t));
        } else if (MODE == 6) {  // scalar FMA with an SGPR multiplier (weights uniform?) - reference for operand ports
          asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i][0]) : "v"(a[(i + r) & 7][0]), "v"(b[i & 7][0]));
          asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i][1]) : "v"(a[(i + r) & 7][1]), "v"(b[i & 7][1]));
        }
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE> void run(float* out, int wgs, const char* name) {
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), 0, 0, out, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  // MAC-pairs (2 MACs each) per wave: iters * 64 ; waves per SIMD = wgs * 4 / 1024
  const double pairs = (double)iters * 64;
  const double wps = wgs * 4 / 1024.0;
  const double ns_per_pair_per_simd = ms * 1e6 / (pairs * (wps < 1 ? 1 : wps));
  printf("%-44s wgs %4d: %8.3f ms  %6.2f ns per 2-MAC wave-op per SIMD  (= %5.2f cycles at 2.4 GHz)  %6.1f TFLOP/s\n", name, wgs, ms,
         ns_per_pair_per_simd, ns_per_pair_per_simd * 2.4, pairs * 64 * 4 * wgs * 4 / (ms * 1e-3) / 1e12);
}

int main() {
  float* out;
  hipMalloc(&out, 4096 * 256 * 4);
  for (int wgs : {256, 512, 1024}) {
    run<0>(out, wgs, "pk_fma distinct a,b");
    run<1>(out, wgs, "pk_fma shared b");
    run<2>(out, wgs, "pk_fma shared a,b");
    run<3>(out, wgs, "2 x v_fma_f32");
    run<4>(out, wgs, "pk_fma a*a");
    run<5>(out, wgs, "pk_mul + pk_add");
    run<6>(out, wgs, "2 x v_fmac_f32");
  }
  return 0;
}

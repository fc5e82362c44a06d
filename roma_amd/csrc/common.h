// Shared device/host helpers for libroma_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

namespace roma {

typedef unsigned short bf16_t;  // raw bfloat16 bits

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short bf16x8;

// ---- error plumbing (C-ABI returns negative codes; text via roma_last_error) ----
void set_error(const std::string& msg);
#define ROMA_CHECK_HIP(expr)                                                              \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) {                                                               \
      ::roma::set_error(std::string(#expr) + ": " + hipGetErrorString(_e) + " @" + __FILE__ + ":" + \
                        std::to_string(__LINE__));                                        \
      return -2;                                                                          \
    }                                                                                     \
  } while (0)
#define ROMA_REQUIRE(cond, msg)                   \
  do {                                            \
    if (!(cond)) {                                \
      ::roma::set_error(std::string(msg));        \
      return -1;                                  \
    }                                             \
  } while (0)
#define ROMA_LAUNCH_CHECK() ROMA_CHECK_HIP(hipGetLastError())

// ---- bf16 <-> f32 (round-to-nearest-even), usable on host and device ----
__host__ __device__ inline float bf16_to_f32(bf16_t v) {
  union { uint32_t u; float f; } x;
  x.u = ((uint32_t)v) << 16;
  return x.f;
}
__host__ __device__ inline bf16_t f32_to_bf16(float f) {
  union { uint32_t u; float f; } x;
  x.f = f;
  if ((x.u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((x.u >> 16) | 0x40);  // NaN
  uint32_t r = 0x7fffu + ((x.u >> 16) & 1u);
  return (bf16_t)((x.u + r) >> 16);
}

// two f32 -> packed bf16x2 (round-to-nearest-even) in ONE instruction: gfx950's v_cvt_pk_bf16_f32 (no builtin;
// the software rounding costs ~6 VALU ops per element and made the attention softmax / GEMM epilogues VALU-bound)
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
#if defined(__HIP_DEVICE_COMPILE__)
  // fptrunc <2 x float> -> <2 x bfloat> selects v_cvt_pk_bf16_f32 (RNE) on gfx950.  Deliberately NOT inline asm: the
  // compiler must see the instruction to insert the MFMA-result read wait states when this is the first consumer.
  typedef __attribute__((ext_vector_type(2))) float pk_f32x2;
  typedef __attribute__((ext_vector_type(2))) __bf16 pk_bf16x2;
  const pk_f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, pk_bf16x2));
#else
  return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
#endif
}

template <typename T> struct ElemIO;
template <> struct ElemIO<float> {
  __device__ static inline float ld(const float* p) { return *p; }
  __device__ static inline void st(float* p, float v) { *p = v; }
  __device__ static inline f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
  __device__ static inline void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
};
template <> struct ElemIO<bf16_t> {
  __device__ static inline float ld(const bf16_t* p) { return bf16_to_f32(*p); }
  __device__ static inline void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
  __device__ static inline f32x4 ld4(const bf16_t* p) {
    uint2 u = *reinterpret_cast<const uint2*>(p);
    f32x4 r;
    r[0] = __uint_as_float(u.x << 16);
    r[1] = __uint_as_float(u.x & 0xffff0000u);
    r[2] = __uint_as_float(u.y << 16);
    r[3] = __uint_as_float(u.y & 0xffff0000u);
    return r;
  }
  __device__ static inline void st4(bf16_t* p, f32x4 v) {
    uint2 u;
    u.x = pack_bf16x2(v[0], v[1]);
    u.y = pack_bf16x2(v[2], v[3]);
    *reinterpret_cast<uint2*>(p) = u;
  }
};

// ---- optional per-launch HIP-event timing (bench.py roofline pass; off by default) ----
// work = algorithmic FLOPs (unit "flop") or algorithmic HBM bytes (unit "byte") of this launch.
bool prof_enabled();
void prof_begin(const char* kernel, double work, const char* unit, hipStream_t s);
void prof_end(hipStream_t s);
struct ProfScope {
  hipStream_t s;
  bool on;
  ProfScope(const char* kernel, double work, const char* unit, hipStream_t st) : s(st), on(prof_enabled()) {
    if (on) prof_begin(kernel, work, unit, st);
  }
  ~ProfScope() {
    if (on) prof_end(s);
  }
};

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline long round_up(long a, long b) { return (a + b - 1) / b * b; }

}  // namespace roma

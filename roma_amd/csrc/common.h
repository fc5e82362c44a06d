// Shared device/host helpers for libroma_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

namespace roma {

// The library's 16-bit storage type ("h16"), as raw bits.  It is a BUILD-TIME property: libroma_hip.so stores bfloat16
// (torch.bfloat16: the reference's timing script, tests/test_roma_upsample_inference_time.py:45), libroma_hip_f16.so -
// the same sources compiled with -DROMA_H16_F16 - stores IEEE binary16 (torch.float16: the reference's DEFAULT amp_dtype,
// model_zoo/__init__.py:37, matcher.py:46,341, encoders.py:7).  Both run the same v_mfma_f32_32x32x16 rate; binary16 has
// 11 significand bits against 8 (8 x less storage rounding through the 45 refiner blocks) and overflows at 65504.
// `bf16_t`, DT_BF16 and "bf16" in kernel / function names are the historical names of this type: they mean "the h16 of
// this build".  Everything format specific lives in this header: conversions, packing, the MFMA and the dot product.
typedef unsigned short h16_t;
typedef h16_t bf16_t;
#ifdef ROMA_H16_F16
#define ROMA_H16_NAME "f16"
typedef _Float16 h16_native;
#else
#define ROMA_H16_NAME "bf16"
#if defined(__HIP_DEVICE_COMPILE__)
typedef __bf16 h16_native;
#else
typedef unsigned short h16_native;  // host side: only the bit pattern is ever touched
#endif
#endif

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

// f32 -> bfloat16 bits (round-to-nearest-even) whatever this build's own h16 is: ROMA_MIXED handles of the binary16 build pack
// their DINOv2 weights for the bfloat16 library with it (model.hip)
__host__ __device__ inline unsigned short f32_to_bfloat16_bits(float f) {
  union { uint32_t u; float f; } x;
  x.f = f;
  if ((x.u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((x.u >> 16) | 0x40);  // NaN
  const uint32_t r = 0x7fffu + ((x.u >> 16) & 1u);
  return (unsigned short)((x.u + r) >> 16);
}

// ---- h16 <-> f32 (round-to-nearest-even), usable on host and device ----
#ifdef ROMA_H16_F16
__host__ __device__ inline float bf16_to_f32(bf16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
__host__ __device__ inline bf16_t f32_to_bf16(float f) { return __builtin_bit_cast(bf16_t, (_Float16)f); }
// low / high half of a dword holding two h16
__device__ __forceinline__ float h16_lo(uint32_t u) {
  typedef __attribute__((ext_vector_type(2))) _Float16 pk_h;
  return (float)__builtin_bit_cast(pk_h, u)[0];
}
__device__ __forceinline__ float h16_hi(uint32_t u) {
  typedef __attribute__((ext_vector_type(2))) _Float16 pk_h;
  return (float)__builtin_bit_cast(pk_h, u)[1];
}
// two f32 -> packed h16x2, round-to-nearest-even (v_cvt_pk_f16_f32 on gfx950; left to the compiler, see below)
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) float pk_f32x2;
  typedef __attribute__((ext_vector_type(2))) _Float16 pk_h;
  const pk_f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, pk_h));
}
#else
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
// low / high half of a dword holding two h16 (bf16: one shift / one mask)
__device__ __forceinline__ float h16_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float h16_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }

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
#endif
// ReLU + pack of two f32 into 16-bit storage: round first, then clamp the PACKED pair with one v_pk_max_i16 against zero (the
// sign bit of bfloat16 / binary16 is the sign bit of the 16-bit integer, so max(x, 0) as int16 is max(x, +0) as a float; -0 and
// values that round to -0 become +0 exactly as with fmaxf before the rounding - rounding is monotonic and keeps the sign).  One
// VALU instruction per pair instead of two v_max_f32 per pair in the VALU-bound stencil kernels (round 5).
__device__ __forceinline__ uint32_t pack_relu_h16x2(float lo, float hi) {
  typedef short pk_i16x2 __attribute__((ext_vector_type(2)));
  const pk_i16x2 z = {0, 0};
  const pk_i16x2 v = __builtin_bit_cast(pk_i16x2, pack_bf16x2(lo, hi));
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(v, z));
}

// D[32x32] += A[32 x 16] . B[16 x 32] on the matrix core, 16-bit operands of this build's format, f32 accumulate:
// v_mfma_f32_32x32x16_bf16 / v_mfma_f32_32x32x16_f16 (same rate, same register layout: 8 consecutive k per lane).
// a, b: any 16-byte register type holding 8 h16.
typedef __attribute__((ext_vector_type(8))) h16_native h16x8_t;
template <typename TA, typename TB>
__device__ __forceinline__ f32x16 mfma_h16_32x32x16(TA a, TB b, f32x16 c) {
  static_assert(sizeof(TA) == 16 && sizeof(TB) == 16, "8 h16 per operand");
#if !defined(__HIP_DEVICE_COMPILE__)
  return c;  // host pass of the single-source compile: never executed
#elif defined(ROMA_H16_F16)
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8_t, a), __builtin_bit_cast(h16x8_t, b), c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(h16x8_t, a), __builtin_bit_cast(h16x8_t, b), c, 0, 0, 0);
#endif
}
// D[16x16] += A[16 x 32] . B[32 x 16]: v_mfma_f32_16x16x32_{bf16,f16} (8 consecutive k per lane, k block = lane >> 4; C/D: column
// lane & 15, rows 4 (lane >> 4) + 0..3)
template <typename TA, typename TB>
__device__ __forceinline__ f32x4 mfma_h16_16x16x32(TA a, TB b, f32x4 c) {
  static_assert(sizeof(TA) == 16 && sizeof(TB) == 16, "8 h16 per operand");
#if !defined(__HIP_DEVICE_COMPILE__)
  return c;
#elif defined(ROMA_H16_F16)
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8_t, a), __builtin_bit_cast(h16x8_t, b), c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(h16x8_t, a), __builtin_bit_cast(h16x8_t, b), c, 0, 0, 0);
#endif
}
// acc + x.lo * y.lo + x.hi * y.hi on packed h16 pairs (v_dot2c_f32_bf16 / v_dot2_f32_f16)
__device__ __forceinline__ float dot2_h16(uint32_t x, uint32_t y, float acc) {
  typedef h16_native pk_h16x2 __attribute__((ext_vector_type(2)));
#if !defined(__HIP_DEVICE_COMPILE__)
  return acc;
#elif defined(ROMA_H16_F16)
  return __builtin_amdgcn_fdot2(__builtin_bit_cast(pk_h16x2, x), __builtin_bit_cast(pk_h16x2, y), acc, false);
#else
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(pk_h16x2, x), __builtin_bit_cast(pk_h16x2, y), acc, false);
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
    r[0] = h16_lo(u.x);
    r[1] = h16_hi(u.x);
    r[2] = h16_lo(u.y);
    r[3] = h16_hi(u.y);
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

int internal_h16_code();  // vit.hip: this build's 16-bit code, for roma_self_check (api.hip)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline long round_up(long a, long b) { return (a + b - 1) / b * b; }

}  // namespace roma

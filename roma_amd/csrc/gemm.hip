// MFMA GEMM for gfx950 (see gemm.h).  One 256-thread workgroup = 4 wave64s.
//
// Tile anatomy (CDNA4):
//   * K is staged through LDS in slabs of 8 x 16-byte chunks per row (32 f32 / 64 bf16) with one
//     16-byte pad chunk per row (row stride 144 B): a ds_read_b128 lane group {16 rows x one chunk}
//     then touches 16 distinct 16-B bank slots -> conflict free, and the 128-B row writes are linear.
//   * next slab's global loads are issued into registers before the MFMAs of the current slab
//     (issue-early / write-late staging), one LDS buffer, two barriers per slab.
//   * MFMA operand roles are SWAPPED: the weight tile feeds the A operand (rows = n) and the
//     activation tile the B operand (cols = m).  D[n][m] then puts 4 CONSECUTIVE n of one output
//     row m in each lane's register quad, so bias/scale/residual/output move as 16-byte vectors.
//   * f32 path: v_mfma_f32_32x32x2_f32; the k-order inside a slab is permuted (lane half h takes
//     k = 8g+4h+s) which is legal because A and B use the same permutation.
#include "gemm.h"
#include <stdio.h>

namespace roma {

constexpr int CHUNKS = 8;
constexpr int LDS_ROW = 9 * 16;

template <typename T> struct InTraits;
template <> struct InTraits<float> { static constexpr int CE = 4; };
template <> struct InTraits<bf16_t> { static constexpr int CE = 8; };

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;

__device__ inline float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }

template <typename TOUT> __device__ inline void store4(TOUT* p, f32x4 v, bool vec, int nvalid) {
  if (vec && nvalid >= 4) {
    ElemIO<TOUT>::st4(p, v);
  } else {
    for (int j = 0; j < 4; ++j)
      if (j < nvalid) ElemIO<TOUT>::st(p + j, v[j]);
  }
}

template <typename TIN, typename TOUT, int WM, int WN, int TM, int TN, bool CONV>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmArgs a) {
  constexpr int CE = InTraits<TIN>::CE;
  constexpr int BKE = CHUNKS * CE;
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int A_PER_T = (BM * CHUNKS + 255) / 256;
  constexpr int W_PER_T = (BN * CHUNKS + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* As = smem;
  char* Ws = smem + BM * LDS_ROW;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, h = lane >> 5;
  const int bz = blockIdx.z;
  const long m0 = (long)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  if (a.lower_only && n0 > m0 + BM - 1) return;

  const TIN* Ab = reinterpret_cast<const TIN*>(a.A) + (long)bz * a.sA;
  const TIN* Wb = reinterpret_cast<const TIN*>(a.W) + (long)bz * a.sW;

  // ---- per-thread staging descriptors
  long a_off[A_PER_T];
  bool a_ok[A_PER_T];
  int a_y[A_PER_T], a_x[A_PER_T];
#pragma unroll
  for (int i = 0; i < A_PER_T; ++i) {
    const int c = tid + 256 * i;
    const int row = c >> 3;
    const long gm = m0 + row;
    a_ok[i] = (row < BM) && (gm < a.M);
    if (CONV) {
      const long hw = (long)a.conv_h * a.conv_w;
      const long gmc = a_ok[i] ? gm : 0;
      const long b = gmc / hw;
      const int rem = (int)(gmc - b * hw);
      a_y[i] = rem / a.conv_w;
      a_x[i] = rem - a_y[i] * a.conv_w;
      a_off[i] = gmc * a.conv_c + (c & 7) * CE;  // pixel base (tap offset added per slab)
    } else {
      a_y[i] = a_x[i] = 0;
      a_off[i] = gm * a.lda + (c & 7) * CE;
    }
  }
  long w_off[W_PER_T];
  bool w_ok[W_PER_T];
#pragma unroll
  for (int i = 0; i < W_PER_T; ++i) {
    const int c = tid + 256 * i;
    const int row = c >> 3;
    w_ok[i] = (row < BN) && (n0 + row < a.N);
    w_off[i] = (long)(n0 + row) * a.ldw + (c & 7) * CE;
  }

  uint4 ra[A_PER_T], rw[W_PER_T];
  const uint4 zero4 = make_uint4(0, 0, 0, 0);

  auto load_slab = [&](int kt) {
    const int k0 = kt * BKE;
    if (CONV) {
      const int tap = k0 / a.conv_c;
      const int c0 = k0 - tap * a.conv_c;
      const int dy = tap / 3 - 1, dx = tap % 3 - 1;
      const long toff = ((long)dy * a.conv_w + dx) * a.conv_c + c0;
#pragma unroll
      for (int i = 0; i < A_PER_T; ++i) {
        const int yy = a_y[i] + dy, xx = a_x[i] + dx;
        const bool ok = a_ok[i] && yy >= 0 && yy < a.conv_h && xx >= 0 && xx < a.conv_w;
        ra[i] = ok ? *reinterpret_cast<const uint4*>(Ab + a_off[i] + toff) : zero4;
      }
    } else {
#pragma unroll
      for (int i = 0; i < A_PER_T; ++i) {
        const int c = tid + 256 * i;
        const bool ok = a_ok[i] && (k0 + (c & 7) * CE < a.K);
        ra[i] = ok ? *reinterpret_cast<const uint4*>(Ab + a_off[i] + k0) : zero4;
      }
    }
#pragma unroll
    for (int i = 0; i < W_PER_T; ++i) {
      const int c = tid + 256 * i;
      const bool ok = w_ok[i] && (k0 + (c & 7) * CE < a.K);
      rw[i] = ok ? *reinterpret_cast<const uint4*>(Wb + w_off[i] + k0) : zero4;
    }
  };
  auto store_slab = [&]() {
#pragma unroll
    for (int i = 0; i < A_PER_T; ++i) {
      const int c = tid + 256 * i;
      if (c < BM * CHUNKS) *reinterpret_cast<uint4*>(As + (c >> 3) * LDS_ROW + (c & 7) * 16) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < W_PER_T; ++i) {
      const int c = tid + 256 * i;
      if (c < BN * CHUNKS) *reinterpret_cast<uint4*>(Ws + (c >> 3) * LDS_ROW + (c & 7) * 16) = rw[i];
    }
  };

  f32x16 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = (a.K + BKE - 1) / BKE;
  load_slab(0);
  for (int kt = 0; kt < nk; ++kt) {
    store_slab();
    __syncthreads();
    if (kt + 1 < nk) load_slab(kt + 1);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      uint4 wv[TN], av[TM];
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
        wv[tn] = *reinterpret_cast<const uint4*>(Ws + ((wn * TN + tn) * 32 + l31) * LDS_ROW + (2 * g + h) * 16);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
        av[tm] = *reinterpret_cast<const uint4*>(As + ((wm * TM + tm) * 32 + l31) * LDS_ROW + (2 * g + h) * 16);
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
          if constexpr (sizeof(TIN) == 4) {
            acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(wv[tn].x), __uint_as_float(av[tm].x), acc[tn][tm], 0, 0, 0);
            acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(wv[tn].y), __uint_as_float(av[tm].y), acc[tn][tm], 0, 0, 0);
            acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(wv[tn].z), __uint_as_float(av[tm].z), acc[tn][tm], 0, 0, 0);
            acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(wv[tn].w), __uint_as_float(av[tm].w), acc[tn][tm], 0, 0, 0);
          } else {
            acc[tn][tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wv[tn]),
                                                                  __builtin_bit_cast(bf16x8_t, av[tm]), acc[tn][tm], 0, 0, 0);
          }
        }
    }
    __syncthreads();
  }

  // ---------------------------------------------------------------- epilogue
  TOUT* Cb = reinterpret_cast<TOUT*>(a.C) + (long)bz * a.sC;
  const float* Rb = a.res ? a.res + (long)bz * a.sR : nullptr;
  const bool vecC = ((a.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(Cb) & 15) == 0);
  const bool vecR = Rb && ((a.ldr & 3) == 0) && ((reinterpret_cast<uintptr_t>(Rb) & 15) == 0);
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const long m = m0 + (wm * TM + tm) * 32 + l31;
    if (m >= a.M) continue;
    float nxm = 0.f;
    if (a.mode == EPI_COSK) nxm = a.nx[(long)bz * a.sNx + m];
    int qb = 0, qt = 0;
    if (a.mode == EPI_QKV) {
      qb = (int)(m / a.ntok);
      qt = (int)(m - (long)qb * a.ntok);
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int n = n0 + (wn * TN + tn) * 32 + 8 * rg + 4 * h;
        const int nvalid = a.N - n;
        if (nvalid <= 0) continue;
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = a.alpha * acc[tn][tm][4 * rg + j];
        if (a.mode == EPI_COSK) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (j < nvalid) {
              const float nyn = a.ny[(long)bz * a.sNy + n + j];
              float c = v[j] / (nxm * nyn + 1e-6f);
              float kk = expf((c - 1.0f) * a.inv_t);
              if (a.diag_add != 0.f && m == n + j) kk += a.diag_add;
              v[j] = kk;
            }
          }
          store4<TOUT>(Cb + m * a.ldc + n, v, vecC, nvalid);
          continue;
        }
        if (a.bias) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (j < nvalid) v[j] += a.bias[n + j];
        }
        if (a.mode == EPI_QKV) {
          const int D = a.heads * a.hd;
          const int which = n / D;
          const int rem = n - which * D;
          const int head = rem / a.hd;
          const int d = rem - head * a.hd;
          const long bh = (long)qb * a.heads + head;
          if (which == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] *= a.qscale;
            ElemIO<TOUT>::st4(reinterpret_cast<TOUT*>(a.q) + (bh * a.npad + qt) * a.hd + d, v);
          } else if (which == 1) {
            ElemIO<TOUT>::st4(reinterpret_cast<TOUT*>(a.k) + (bh * a.npad + qt) * a.hd + d, v);
          } else {
            TOUT* vp = reinterpret_cast<TOUT*>(a.vt) + (bh * a.hd + d) * a.npad + qt;
#pragma unroll
            for (int j = 0; j < 4; ++j) ElemIO<TOUT>::st(vp + (long)j * a.npad, v[j]);
          }
          continue;
        }
        if (a.act == ACT_RELU) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
        } else if (a.act == ACT_GELU) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = gelu_erf(v[j]);
        }
        if (a.scale) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (j < nvalid) v[j] *= a.scale[n + j];
        }
        if (Rb) {
          if (vecR && nvalid >= 4) {
            f32x4 r = *reinterpret_cast<const f32x4*>(Rb + m * a.ldr + n);
            v += r;
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (j < nvalid) v[j] += Rb[m * a.ldr + n + j];
          }
        }
        store4<TOUT>(Cb + m * a.ldc + n, v, vecC, nvalid);
      }
    }
  }
}

template <typename TIN, typename TOUT, int WM, int WN, int TM, int TN, bool CONV>
static int launch_cfg(const GemmArgs& a, hipStream_t stream) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  dim3 grid((unsigned)((a.M + BM - 1) / BM), (unsigned)((a.N + BN - 1) / BN), (unsigned)a.batch);
  size_t lds = (size_t)(BM + BN) * LDS_ROW;
  char pname[96];
  snprintf(pname, sizeof pname, "gemm_kernel<%s,%s,%d,%d,%d,%d,%s>", sizeof(TIN) == 4 ? "f32" : "bf16",
           sizeof(TOUT) == 4 ? "f32" : "bf16", WM, WN, TM, TN, CONV ? "conv3x3" : "dense");
  ProfScope ps(pname, 2.0 * (double)a.M * a.N * a.K * a.batch * (a.lower_only ? 0.5 : 1.0), "flop", stream);
  hipLaunchKernelGGL((gemm_kernel<TIN, TOUT, WM, WN, TM, TN, CONV>), grid, dim3(256), lds, stream, a);
  ROMA_LAUNCH_CHECK();
  return 0;
}

template <typename TIN, typename TOUT, bool CONV>
static int launch_shape(const GemmArgs& a, hipStream_t stream) {
  if (a.N <= 32) return launch_cfg<TIN, TOUT, 4, 1, 2, 1, CONV>(a, stream);       // 256 x 32
  if (a.N <= 64 || (a.N % 128 != 0 && a.N <= 192))
    return launch_cfg<TIN, TOUT, 4, 1, 1, 2, CONV>(a, stream);                    // 128 x 64
  return launch_cfg<TIN, TOUT, 2, 2, 2, 2, CONV>(a, stream);                      // 128 x 128
}

int gemm_launch(const GemmArgs& a, hipStream_t stream) {
  const int ce = a.in_dt == DT_F32 ? 4 : 8;
  ROMA_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0 && a.batch > 0, "gemm: empty problem");
  ROMA_REQUIRE(a.K % ce == 0, "gemm: K must be a multiple of the 16-byte chunk");
  ROMA_REQUIRE(a.ldw % ce == 0 && (reinterpret_cast<uintptr_t>(a.W) & 15) == 0, "gemm: W not 16-byte aligned");
  ROMA_REQUIRE((reinterpret_cast<uintptr_t>(a.A) & 15) == 0, "gemm: A not 16-byte aligned");
  ROMA_REQUIRE(a.sA % ce == 0 && a.sW % ce == 0, "gemm: batch strides must keep 16-byte alignment");
  const bool conv = a.conv_c > 0;
  if (conv) {
    ROMA_REQUIRE(a.conv_c % (8 * ce) == 0, "gemm(conv3x3): Cin must be a multiple of the K slab");
    ROMA_REQUIRE(a.K == 9 * a.conv_c, "gemm(conv3x3): K != 9*Cin");
  } else {
    ROMA_REQUIRE(a.lda % ce == 0, "gemm: lda must keep 16-byte alignment");
  }
  if (a.mode == EPI_QKV) {
    ROMA_REQUIRE(a.hd % 4 == 0 && a.N == 3 * a.heads * a.hd, "gemm(qkv): bad head geometry");
  }
#define ROMA_GEMM_DISPATCH(TIN, TOUT)                                              \
  return conv ? launch_shape<TIN, TOUT, true>(a, stream) : launch_shape<TIN, TOUT, false>(a, stream)
  if (a.in_dt == DT_F32 && a.out_dt == DT_F32) { ROMA_GEMM_DISPATCH(float, float); }
  if (a.in_dt == DT_BF16 && a.out_dt == DT_BF16) { ROMA_GEMM_DISPATCH(bf16_t, bf16_t); }
  if (a.in_dt == DT_BF16 && a.out_dt == DT_F32) { ROMA_GEMM_DISPATCH(bf16_t, float); }
  if (a.in_dt == DT_F32 && a.out_dt == DT_BF16) { ROMA_GEMM_DISPATCH(float, bf16_t); }
#undef ROMA_GEMM_DISPATCH
  set_error("gemm: unsupported dtype combination");
  return -1;
}

}  // namespace roma

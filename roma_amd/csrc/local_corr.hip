// Fused local-window correlation for gfx950 (see local_corr.h).
//
// HBM/LDS-bound gather + dot, NOT reshaped into a GEMM.  One wave64 per query pixel.
//
// Integer-patch identity: the (2r+1)^2 window taps are exactly one f1 pixel apart
// (romatch/utils/local_correlation.py:93-108), so every tap shares the same fractional
// offset (fx,fy) and
//     corr[j][i] = (1-fy)(1-fx) D[j][i] + (1-fy)fx D[j][i+1] + fy(1-fx) D[j+1][i] + fy fx D[j+1][i+1]
// with D[a][b] = <f0, f1[y0-r+a][x0-r+b]>, a,b in [0, 2r+2): (2r+2)^2 dot products instead of
// 4(2r+1)^2 bilinear taps, and every f1 row of the patch is ONE contiguous (2r+2)*C run in
// channels-last memory.
//
// Lane mapping: lane = pos*S + s; `pos` walks the 2r+2 patch columns, the S lanes of a column
// split the channels in interleaved 16-byte chunks, so one wave load instruction touches
// (2r+2) fully used 64..128-byte segments.  f0 is staged once per wave in LDS as f32.
#include "local_corr.h"
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include "gemm.h"  // DT_*

namespace roma {

template <typename T> struct LcIO;
template <> struct LcIO<float> {
  static constexpr int CE = 4;
  __device__ static inline void ld(const float* p, float* v) {
    f32x4 x = *reinterpret_cast<const f32x4*>(p);
    v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3];
  }
};
template <> struct LcIO<bf16_t> {
  static constexpr int CE = 8;
  __device__ static inline void ld(const bf16_t* p, float* v) {
    uint4 u = *reinterpret_cast<const uint4*>(p);
    v[0] = h16_lo(u.x); v[1] = h16_hi(u.x);
    v[2] = h16_lo(u.y); v[3] = h16_hi(u.y);
    v[4] = h16_lo(u.z); v[5] = h16_hi(u.z);
    v[6] = h16_lo(u.w); v[7] = h16_hi(u.w);
  }
};

__device__ inline float unnormalize(float w, int size) {
  // grid_sample, align_corners=False: ((x + 1) * size - 1) / 2
  const float ix = ((w + 1.f) * size - 1.f) * 0.5f;
  return fminf(fmaxf(ix, -1.0e6f), 1.0e6f);  // also maps NaN to a finite (all-zero-padding) location
}
__device__ inline void unnormalize_floor(float w, int size, int& i0, float& frac) {
  const float ix = unnormalize(w, size);
  const float f = floorf(ix);
  i0 = (int)f;
  frac = ix - f;
}

// One query pixel by one wave64: gathers its own (2r+2)^2 patch (the form that serves scattered warps).  `myf0` is this
// wave's LDS slice of C floats; the caller provides the block-wide barriers around the staging of f0.
template <int R, typename T, typename TOUT>
__device__ __forceinline__ void lc_gather_stage_f0(const LocalCorrArgs& a, long pix, bool active, int lane, float* myf0) {
  constexpr int CE = LcIO<T>::CE;
  const long HW = (long)a.H * a.W;
  if (!active) return;
  const int b = (int)(pix / HW);
  const long p = pix - (long)b * HW;
  const T* f0p = reinterpret_cast<const T*>(a.f0) + ((long)b * HW + p) * a.ld0;
  // 16-byte LDS accesses on purpose (myf0 is 16-byte aligned: C % 4 == 0): with scalar indexing hipcc emitted
  // ds_write2_b32 / ds_read2_b32, which bank over 32 dwords - the channel slices s and s + 4 of the readers (32 bytes
  // apart) then collide: LDS bank-conflict share 0.47-0.49 of this kernel (profiles/r02_v11_local_corr_sq.txt)
  for (int c = lane * CE; c < a.C; c += 64 * CE) {
    float v[CE];
    LcIO<T>::ld(f0p + c, v);
#pragma unroll
    for (int j = 0; j < CE; j += 4) *reinterpret_cast<f32x4*>(myf0 + c + j) = f32x4{v[j], v[j + 1], v[j + 2], v[j + 3]};
  }
}

template <int R, typename T, typename TOUT>
__device__ __forceinline__ void lc_gather_pixel(const LocalCorrArgs& a, long pix, bool active, int lane, const float* myf0) {
  constexpr int P = 2 * R + 2;
  constexpr int PP = (P <= 8) ? 8 : 16;
  constexpr int S = 64 / PP;
  constexpr int CE = LcIO<T>::CE;
  constexpr int KW = 2 * R + 1;
  const long HW = (long)a.H * a.W;
  if (!active) pix = 0;
  const int b = (int)(pix / HW);
  const int pos = lane / S, s = lane % S;
  int x0, y0;
  float fx, fy;
  unnormalize_floor(a.warp[pix * 2 + 0], a.W, x0, fx);
  unnormalize_floor(a.warp[pix * 2 + 1], a.H, y0, fy);
  const int x = x0 - R + pos;
  const bool xok = active && pos < P && x >= 0 && x < a.W;
  const int simg = (b + a.f1_shift) % a.nimg;
  const T* f1p = reinterpret_cast<const T*>(a.f1) + (long)simg * HW * a.ld1;
  const int NI = a.C / (CE * S);

  // (A branch-free form of this loop - clamped addresses for out-of-image rows, sums discarded by a select - was measured
  //  SLOWER on incoherent warps, 1.20 -> 1.57 ms at r = 2, 216 x 216: the gather is bound by L2 / fabric traffic there,
  //  not by the serialisation of the rows, and the extra loads cost more than the overlap gains.)
  const f32x4* fq = reinterpret_cast<const f32x4*>(myf0 + CE * s);  // ds_read_b128 (see lc_gather_stage_f0)
  float D[P];
#pragma unroll
  for (int r = 0; r < P; ++r) {
    const int y = y0 - R + r;
    float sum = 0.f;
    if (y >= 0 && y < a.H && xok) {
      const T* src = f1p + ((long)y * a.W + x) * a.ld1 + CE * s;
#pragma unroll 4
      for (int i = 0; i < NI; ++i) {
        float v[CE];
        LcIO<T>::ld(src + (long)i * CE * S, v);
#pragma unroll
        for (int j = 0; j < CE; j += 4) {
          const f32x4 q = fq[(i * CE * S + j) >> 2];
          sum = fmaf(v[j], q[0], sum);
          sum = fmaf(v[j + 1], q[1], sum);
          sum = fmaf(v[j + 2], q[2], sum);
          sum = fmaf(v[j + 3], q[3], sum);
        }
      }
    }
    D[r] = sum;
  }
  // ---- reduce the S channel slices of each column, fetch the right-hand neighbour column
  float D1[P];
#pragma unroll
  for (int r = 0; r < P; ++r) {
    float v = D[r];
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    if (S == 8) v += __shfl_xor(v, 4);
    D[r] = v;
    D1[r] = __shfl_down(v, S);
  }
  if (active && pos < KW) {
    TOUT* o = reinterpret_cast<TOUT*>(a.out) + pix * a.ldo;
    const float w00 = (1.f - fy) * (1.f - fx), w01 = (1.f - fy) * fx, w10 = fy * (1.f - fx), w11 = fy * fx;
#pragma unroll
    for (int j = 0; j < KW; ++j) {
      if ((j % S) == s) {
        const float c = w00 * D[j] + w01 * D1[j] + w10 * D[j + 1] + w11 * D1[j + 1];
        ElemIO<TOUT>::st(o + j * KW + pos, c * a.scale);
      }
    }
  }
}

template <int R, typename T, typename TOUT>
__global__ __launch_bounds__(256) void local_corr_window_kernel(const LocalCorrArgs a) {
  extern __shared__ __attribute__((aligned(16))) float f0s[];  // [4 waves][C]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long total = (long)a.B * a.H * a.W;
  const long pix = (long)blockIdx.x * 4 + wave;
  const bool active = pix < total;
  float* myf0 = f0s + wave * a.C;
  lc_gather_stage_f0<R, T, TOUT>(a, pix, active, lane, myf0);
  __syncthreads();
  lc_gather_pixel<R, T, TOUT>(a, pix, active, lane, myf0);
}

// ---------------------------------------------------------------------------------------------------------------
// Tiled form (the default): a workgroup owns an 8 x 8 tile of query pixels.
//
// With a coherent warp (what a trained matcher produces: neighbouring queries look at neighbouring places) the 64
// windows of a tile overlap almost completely, so the union of their integer patches is a small rectangle of f1 -
// (8 + 2r + 1)^2 pixels for a unit-scale warp - and every f1 pixel of that rectangle is needed by up to (2r+2)^2
// queries.  The tile stages that rectangle ONCE, channels-last, in LDS (in chunks of 128 bytes of channels per pixel,
// 144-byte pitch so that the 16 queries of a ds_read_b128 lane group hit different banks), together with the 64
// f0 rows, and all (2r+2)^2 x 64 dot products are evaluated from LDS: HBM / L2 traffic per tile = the algorithmic
// bytes (f0 tile + f1 rectangle), instead of (2r+2)^2 x that for per-query gathers.  Lanes = queries, the 4 waves
// split the patch rows; bf16 dots use v_dot2c_f32_bf16 on packed pairs (f32 accumulate), f32 dots are fmaf chains.
//
// A tile whose windows do NOT overlap (its bounding rectangle exceeds the LDS stage: an incoherent warp, e.g. the
// random-weight benchmark model, or a strongly zooming one) needs (2r+2)^2 x the bytes whatever the kernel does; it is
// appended to a work list (one atomic per tile) and local_corr_list_kernel serves its 64 pixels with per-query gathers
// at the high occupancy that latency-bound form wants (8 KB of LDS per workgroup instead of 74 KB).  The choice is per
// tile, data dependent and made on the device; results do not depend on it beyond f32 summation order.
constexpr int LC_TQ = 8;                 // tile edge (queries)
constexpr int LC_PITCH = 144;            // bytes per staged pixel: 128 B of channels + 16 B pad (bank spread)
// f1 pixels the stage can hold (the 64 f0 rows take 64 more slots): r <= 3 -> 448 pixels = 73 728 B, two workgroups per
// CU (a unit-scale warp needs 13^2 / 15^2 pixels, so zoom factors up to ~2 (r = 2) / ~1.75 (r = 3) still fit; stronger
// zooms go to the gather list); r > 3 -> 704 pixels = 110 592 B, one per CU (r = 7: 23^2 = 529 pixels at unit scale).
// (Three per CU with a 288-pixel stage was tried: 168 VGPRs do not hold the prefetch registers, ~35 spilled.)
// MFMA form (16-bit features): the accumulators of the all-pairs product are 64 x PXMAX / 256 registers per lane, so with two
// workgroups per CU (r <= 3) the stage is capped at 320 slots = 10 blocks of 32 (80 accumulator registers next to the 60
// staging-prefetch registers); its rectangle rows keep their natural pitch (the matrix-core reads go down the slots, the
// 144-byte slot pitch alone makes them conflict free), so 320 slots hold zooms up to ~1.35 (r = 2) / ~1.2 (r = 3).
// Round 6: the large windows (r = 7: one 110 KB workgroup per CU) run with 512 threads - twice the waves to hide the stage /
// barrier / matrix-core chain of a chunk (two per SIMD instead of one), the all-pairs accumulators split 11 -> 6 blocks per
// wave.  (Until round 5: 256 threads - 283 GB/s on coherent warps.)
template <int R, bool MFMA = false> struct LcGeom {
  static constexpr int NTHR = R <= 3 ? 256 : 512;
  static constexpr int PXMAX = R <= 3 ? (MFMA ? 320 : 448) : 704;
  static constexpr int NSLOT = PXMAX + LC_TQ * LC_TQ;
  static constexpr int STAGE = NSLOT * LC_PITCH;
  static constexpr int NPRE = (NSLOT * 8 + NTHR - 1) / NTHR;  // 16-byte pieces per thread per chunk
  static constexpr bool PREFETCH = R <= 3;  // r = 7: with the 48 prefetch registers next to 96 accumulators hipcc spills 10-18 (measured: ISA)
};

// Row pitch (in staged pixel slots) of the f1 rectangle.  Lanes are queries: a 16-lane LDS group covers 4 runs of 4
// neighbouring queries on 4 tile rows - {row 0: cols 0-3}, {row 1: cols 4-7}, {row 2: cols 4-7}, {row 3: cols 0-3} - and
// with the 144-byte slot pitch the bank group of a slot is (9 * slot) mod 16, so the four runs fall on distinct banks
// exactly when the row pitch is 8 (mod 16).  With the natural pitch (the rectangle width, 15-17 at unit scale) the runs
// overlap: bank-conflict share 0.53-0.59 of the kernel's LDS cycles (profiles/r02_v11_local_corr_sq.txt).  The padded
// pitch is used when the padded rectangle still fits the stage; otherwise the natural one (slower, still correct).
__device__ __host__ inline int lc_row_pitch(int bw, int bh, int pxmax) {
  const int padded = ((bw + 7) / 16) * 16 + 8;  // smallest p >= bw with p % 16 == 8
  return (long)padded * bh <= pxmax ? padded : bw;
}

template <typename T> struct LcDot;
template <> struct LcDot<float> {  // 32 channels per 128-byte chunk
  static constexpr int CC = 32;
  __device__ static __forceinline__ float dot(const uint4 (&q)[8], const char* p, float acc) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint4 v = *reinterpret_cast<const uint4*>(p + 16 * i);
      acc = fmaf(__uint_as_float(v.x), __uint_as_float(q[i].x), acc);
      acc = fmaf(__uint_as_float(v.y), __uint_as_float(q[i].y), acc);
      acc = fmaf(__uint_as_float(v.z), __uint_as_float(q[i].z), acc);
      acc = fmaf(__uint_as_float(v.w), __uint_as_float(q[i].w), acc);
    }
    return acc;
  }
};
template <> struct LcDot<bf16_t> {  // 64 channels per 128-byte chunk, packed pairs
  static constexpr int CC = 64;
  __device__ static __forceinline__ float d2(unsigned x, unsigned y, float acc) { return dot2_h16(x, y, acc); }
  __device__ static __forceinline__ float dot(const uint4 (&q)[8], const char* p, float acc) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint4 v = *reinterpret_cast<const uint4*>(p + 16 * i);
      acc = d2(v.x, q[i].x, acc);
      acc = d2(v.y, q[i].y, acc);
      acc = d2(v.z, q[i].z, acc);
      acc = d2(v.w, q[i].w, acc);
    }
    return acc;
  }
};

// MFMA = true (16-bit features): instead of (2r+2)^2 dot products per query, each re-reading its f1 pixel from LDS (4 MB
// of LDS reads per tile at C = 512, r = 3), the tile computes the ALL-PAIRS product  D[rectangle slot][query] =
// sum_c f1[slot][c] f0[query][c]  on v_mfma_f32_32x32x16 straight from the same 144-byte-pitch stage (every staged row is
// read once per 32 queries: ~0.3 MB), and every lane - a query - then picks the (2r+2)^2 entries of its own window out of
// its accumulators.  Wave w owns the 32 queries of tile half w & 1 and the 32-slot blocks (w >> 1), (w >> 1) + 2, ...
// LIST = true (round 6): the work items are not 8 x 8 tiles of the query image but groups of up to 64 queries whose windows
// fall into one TS x TS bin of f1 (local_corr_bin_* kernels below): the queries of the tiles the classifier calls
// incoherent, sorted by where they look.  Everything behind the query identities is the same kernel - bounding rectangle,
// one staged copy of it, all-pairs on the matrix core - so an incoherent warp costs ~ (rectangle + 64 f0 rows) per 64 queries
// instead of 64 x (2r+2)^2 gathered f1 pixels.  The bin edge is chosen so that the rectangle always fits the stage.
template <int R, typename T, typename TOUT, bool MFMA, bool LIST = false>
__global__ __launch_bounds__((LcGeom<R, MFMA>::NTHR), 2) void local_corr_tile_kernel(const LocalCorrArgs a) {
  constexpr int P = 2 * R + 2, KW = 2 * R + 1, K = KW * KW;
  constexpr int NTHR = LcGeom<R, MFMA>::NTHR, NWV = NTHR / 64, NBG = NWV / 2;  // waves; block groups (wave >> 1) of the MFMA form
  constexpr int NR = (P + NWV - 1) / NWV;  // patch rows per wave
  constexpr int NBW = MFMA ? ((LcGeom<R, MFMA>::PXMAX + 31) / 32 + NBG - 1) / NBG : 1;  // 32-slot blocks per wave (5 ; 6)
  static_assert(!MFMA || sizeof(T) == 2, "the MFMA form takes 16-bit features");
  constexpr int CC = LcDot<T>::CC;
  constexpr int LC_STAGE = LcGeom<R, MFMA>::STAGE, LC_PXMAX = LcGeom<R, MFMA>::PXMAX, NPRE = LcGeom<R, MFMA>::NPRE;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  // [0, LC_STAGE): staged pixels (f1 rectangle, then the 64 f0 rows); later reused for the D exchange
  int* qx0 = reinterpret_cast<int*>(lds + LC_STAGE);       // [64]
  int* qy0 = qx0 + 64;                                      // [64]
  float* qfx = reinterpret_cast<float*>(qy0 + 64);          // [64]
  float* qfy = qfx + 64;                                    // [64]
  int* tinfo = reinterpret_cast<int*>(qfy + 64);            // bx0, by0, bw, bh
  int* qpx = tinfo + 8;                                     // [64] pixel of each query inside its image, -1 = none

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_x = (a.W + LC_TQ - 1) / LC_TQ, tiles_y = (a.H + LC_TQ - 1) / LC_TQ;
  const int tpi = tiles_x * tiles_y;
  const long HW = (long)a.H * a.W;
  const int* tlist = a.ws + 4 + a.B * tpi;  // coherent tiles (local_corr_classify_kernel)
  // One workgroup per listed tile; the launch covers every tile of the call and the surplus workgroups exit at once.
  // (Persistent workgroups pulling tiles from the list were measured 30 % slower on coherent warps: the two workgroups
  // of a CU then run their load and compute phases in lock-step instead of drifting apart like short-lived ones do.)
  for (int tidx = blockIdx.x; tidx < (LIST ? a.ws[2] : a.ws[1]); tidx += gridDim.x) {
  __syncthreads();  // previous tile's readers of tinfo / the stage are done (grid-stride repeat only)
  int b, qin;        // image of the work item; this lane's query as a pixel index inside the image (-1: none)
  if constexpr (LIST) {
    const int* ql = a.ws + a.ws_qlist + (long)tidx * 64;
    const int g = ql[lane];                 // global pixel index b * HW + y * W + x, or -1 (padding of the item)
    b = __shfl(g, 0) / (int)HW;             // slot 0 of an item is always a query (local_corr_bin_scatter_kernel)
    qin = g >= 0 ? g - b * (int)HW : -1;
  } else {
    const int tile = tlist[tidx];
    b = tile / tpi;
    const int trem = tile - b * tpi;
    const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
    const int qy = ty * LC_TQ + (lane >> 3), qx = tx * LC_TQ + (lane & 7);
    qin = (qy < a.H && qx < a.W) ? qy * a.W + qx : -1;
  }

  // ---- per-query window origin (lane = query), bounding rectangle of the item's integer patches
  const bool qvalid = qin >= 0;
  const long qpix = (long)b * HW + (qvalid ? qin : 0);
  int x0 = 0, y0 = 0;
  float fx = 0.f, fy = 0.f;
  if (qvalid) {
    unnormalize_floor(a.warp[qpix * 2 + 0], a.W, x0, fx);
    unnormalize_floor(a.warp[qpix * 2 + 1], a.H, y0, fy);
  }
  if (wave == 0) {
    int xlo = qvalid ? x0 : 0x3fffffff, xhi = qvalid ? x0 : -0x3fffffff;
    int ylo = qvalid ? y0 : 0x3fffffff, yhi = qvalid ? y0 : -0x3fffffff;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      xlo = min(xlo, __shfl_xor(xlo, off));
      xhi = max(xhi, __shfl_xor(xhi, off));
      ylo = min(ylo, __shfl_xor(ylo, off));
      yhi = max(yhi, __shfl_xor(yhi, off));
    }
    qx0[lane] = x0; qy0[lane] = y0; qfx[lane] = fx; qfy[lane] = fy; qpx[lane] = qin;
    if (lane == 0) {
      // rectangle of f1 pixels any query of the tile touches, clipped to the image (outside taps contribute zero)
      const int bx0 = max(xlo - R, 0), bx1 = min(xhi + R + 1, a.W - 1);
      const int by0 = max(ylo - R, 0), by1 = min(yhi + R + 1, a.H - 1);
      tinfo[0] = bx0; tinfo[1] = by0;
      tinfo[2] = max(bx1 - bx0 + 1, 0);
      tinfo[3] = max(by1 - by0 + 1, 0);
    }
  }
  __syncthreads();
  const int bx0 = tinfo[0], by0 = tinfo[1], bw = tinfo[2], bh = tinfo[3];
  const int bwp = MFMA ? bw : lc_row_pitch(bw, bh, LC_PXMAX);  // slots per staged rectangle row (>= bw, see lc_row_pitch)
  const long npx = (long)bwp * bh;
  const int simg = (b + a.f1_shift) % a.nimg;
  TOUT* outp = reinterpret_cast<TOUT*>(a.out);

  if (npx > LC_PXMAX) continue;  // (cannot happen: the classifier only lists tiles that fit)

  // ---- coherent tile: stage the rectangle + the f0 rows chunk by chunk, dots from LDS
  const T* f1p = reinterpret_cast<const T*>(a.f1) + (long)simg * HW * a.ld1;
  const T* f0p = reinterpret_cast<const T*>(a.f0) + (long)b * HW * a.ld0;
  float acc[MFMA ? 1 : NR][MFMA ? 1 : P];
  f32x16 macc[NBW];
  if constexpr (MFMA) {
#pragma unroll
    for (int i = 0; i < NBW; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) macc[i][e] = 0.f;
  } else {
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
      for (int j = 0; j < P; ++j) acc[i][j] = 0.f;
  }
  const int nslots = (int)npx + LC_TQ * LC_TQ;  // staged rows: f1 rectangle then the tile's f0 rows
  const int my_x = x0 - R - bx0, my_y = y0 - R - by0;  // patch origin inside the rectangle (may be negative: clipped)
  char* f0slot = lds + ((int)npx + lane) * LC_PITCH;

  // staging is split (issue early / write late): the global loads of chunk c + 1 are issued before the dots of chunk c
  // and written to LDS after them, so HBM latency hides under the LDS / VALU work of this workgroup too
  constexpr bool PREFETCH = LcGeom<R, MFMA>::PREFETCH;
  constexpr int NP = PREFETCH ? NPRE : 1;
  uint4 pre[NP];
  // per-piece source of chunk 0 as a 32-bit byte offset from its image base (one feature map of one image is far below
  // 4 GB): bit k of `from_f0` selects the base, 0xffffffff = zero fill.  Half the registers of 64-bit pointers.
  unsigned poff[NP];
  unsigned from_f0 = 0;
  const char* f1b = reinterpret_cast<const char*>(f1p);
  const char* f0b = reinterpret_cast<const char*>(f0p);
#pragma unroll
  for (int k = 0; k < (PREFETCH ? NPRE : 0); ++k) {
    const int i = tid + NTHR * k;
    const int slotp = i >> 3, piece = i & 7;
    unsigned off = 0xffffffffu;
    if (slotp < (int)npx) {
      const int py = slotp / bwp, px = slotp - py * bwp;
      if (px < bw) off = (unsigned)((((long)(by0 + py) * a.W + (bx0 + px)) * a.ld1) * (long)sizeof(T) + piece * 16);
    } else if (slotp < nslots) {
      const int qp = qpx[slotp - (int)npx];
      if (qp >= 0) {
        off = (unsigned)(((long)qp * a.ld0) * (long)sizeof(T) + piece * 16);
        from_f0 |= 1u << k;
      }
    }
    poff[k] = off;
    pre[k] = make_uint4(0, 0, 0, 0);
    if (off != 0xffffffffu) pre[k] = *reinterpret_cast<const uint4*>(((from_f0 >> k) & 1u ? f0b : f1b) + off);
  }
  for (int c0 = 0; c0 < a.C; c0 += CC) {
    if constexpr (PREFETCH) {
#pragma unroll
      for (int k = 0; k < NPRE; ++k) {
        const int i = tid + NTHR * k;
        if (i < nslots * 8) *reinterpret_cast<uint4*>(lds + (i >> 3) * LC_PITCH + (i & 7) * 16) = pre[k];
      }
    } else {  // plain staging loop: load and store the pieces of this chunk
      for (int i = tid; i < nslots * 8; i += NTHR) {
        const int slotp = i >> 3, piece = i & 7;
        const char* src = nullptr;
        if (slotp < (int)npx) {
          const int py = slotp / bwp, px = slotp - py * bwp;
          if (px < bw) src = reinterpret_cast<const char*>(f1p + ((long)(by0 + py) * a.W + (bx0 + px)) * a.ld1 + c0) + piece * 16;
        } else {
          const int qp = qpx[slotp - (int)npx];
          if (qp >= 0) src = reinterpret_cast<const char*>(f0p + (long)qp * a.ld0 + c0) + piece * 16;
        }
        uint4 v = make_uint4(0, 0, 0, 0);
        if (src) v = *reinterpret_cast<const uint4*>(src);
        *reinterpret_cast<uint4*>(lds + slotp * LC_PITCH + piece * 16) = v;
      }
    }
    __syncthreads();
    if constexpr (PREFETCH) {
      if (c0 + CC < a.C) {
        const long coff = (long)(c0 + CC) * (long)sizeof(T);
#pragma unroll
        for (int k = 0; k < NPRE; ++k)
          if (poff[k] != 0xffffffffu) pre[k] = *reinterpret_cast<const uint4*>(((from_f0 >> k) & 1u ? f0b : f1b) + poff[k] + coff);
      }
    }
    if constexpr (MFMA) {
      // all-pairs on the matrix core: first operand = 32 rectangle slots (rows), second = this wave's 32 queries (columns);
      // both fragments are the 16 bytes (8 channels) at (32 ks + 16 h) of a staged 128-byte row.  Blocks beyond the
      // rectangle are skipped (wave-uniform); the last block may run into the staged f0 rows: finite values, never extracted.
      const int l31 = lane & 31, hh = lane >> 5;
      const char* arow = lds + ((int)npx + 32 * (wave & 1) + l31) * LC_PITCH + hh * 16;
      const char* brow = lds + (32 * (wave >> 1) + l31) * LC_PITCH + hh * 16;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const uint4 af = *reinterpret_cast<const uint4*>(arow + ks * 32);
#pragma unroll
        for (int i = 0; i < NBW; ++i) {
          if ((NBG * i + (wave >> 1)) * 32 < (int)npx) {
            const uint4 bf = *reinterpret_cast<const uint4*>(brow + i * (NBG * 32 * LC_PITCH) + ks * 32);
            macc[i] = mfma_h16_32x32x16(bf, af, macc[i]);
          }
        }
      }
    } else if constexpr (sizeof(T) == 2 && R <= 3) {
      // Branch-free per lane: a patch position outside the rectangle reads slot 0 (always staged) and its dot is discarded
      // by a select.  With per-position `if`s every one of the 8 x 16-byte LDS reads of a dot sat in its own exec-masked
      // block and was waited for on its own (149 s_waitcnt for 124 reads in the ISA): the kernel ran at LDS LATENCY,
      // 5x below its LDS-bandwidth bound (SQ: 66 % of the wave cycles parked).  Queries outside the image have a zero f0
      // row, so they need no guard either.  The only branch left is wave-uniform (r < P).
      uint4 q[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) q[i] = *reinterpret_cast<const uint4*>(f0slot + 16 * i);
#pragma unroll
      for (int ri = 0; ri < NR; ++ri) {
        const int r = wave + NWV * ri;  // wave-uniform
        if (r < P) {
          const int yy = my_y + r;
          const bool rowok = yy >= 0 && yy < bh;
          const int rowbase = rowok ? yy * bwp : 0;
#pragma unroll
          for (int j = 0; j < P; ++j) {
            const int xx = my_x + j;
            const bool ok = rowok && xx >= 0 && xx < bw;
            const float t = LcDot<T>::dot(q, lds + (rowbase + (ok ? xx : 0)) * LC_PITCH, acc[ri][j]);
            acc[ri][j] = ok ? t : acc[ri][j];
          }
        }
      }
    } else if (qvalid) {
      // f32 operands / r = 7: the branch-free form needs more registers than there are (11-334 spilled); per-position guards
      uint4 q[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) q[i] = *reinterpret_cast<const uint4*>(f0slot + 16 * i);
#pragma unroll
      for (int ri = 0; ri < NR; ++ri) {
        const int r = wave + NWV * ri;
        const int yy = my_y + r;
        if (r < P && yy >= 0 && yy < bh) {
#pragma unroll
          for (int j = 0; j < P; ++j) {
            const int xx = my_x + j;
            if (xx >= 0 && xx < bw) acc[ri][j] = LcDot<T>::dot(q, lds + (yy * bwp + xx) * LC_PITCH, acc[ri][j]);
          }
        }
      }
    }
    __syncthreads();
  }
  // ---- exchange the integer-patch dots through LDS: D[q][r][j] (f32), then the bilinear combination per output tap
  float* Dl = reinterpret_cast<float*>(lds);  // [64][P*P]   (64 * 256 * 4 B = 64 KiB at r = 7)
  if constexpr (MFMA) {
    // window entries outside the (clipped) rectangle are zero: clear, then every lane scatters the entries of its query's
    // window out of its accumulators.  Register e of block nb holds slot 32 nb + 8 (e >> 2) + 4 hh + (e & 3) for query
    // 32 (wave & 1) + l31 (the C / D layout of the 32 x 32 MFMA); slot -> rectangle (row, column) incrementally.
    for (int i = tid; i < LC_TQ * LC_TQ * P * P / 4; i += NTHR) reinterpret_cast<f32x4*>(Dl)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const int l31 = lane & 31, hh = lane >> 5;
    const int q = 32 * (wave & 1) + l31;
    const int wx0 = qx0[q] - R - bx0, wy0 = qy0[q] - R - by0;  // window origin inside the rectangle
    float* dq = Dl + q * (P * P);
#pragma unroll
    for (int i = 0; i < NBW; ++i) {
      const int s0 = (NBG * i + (wave >> 1)) * 32 + 4 * hh;
      if ((NBG * i + (wave >> 1)) * 32 < (int)npx) {
        int py = s0 / bwp, px = s0 - py * bwp;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int rr = py - wy0, jj = px - wx0;
          if (py < bh && px < bw && rr >= 0 && rr < P && jj >= 0 && jj < P) dq[rr * P + jj] = macc[i][e];
          px += (e & 3) == 3 ? 5 : 1;  // next slot of this lane: +1 inside a run of 4, +5 to the next run (8 apart)
          while (px >= bwp) {
            px -= bwp;
            ++py;
          }
        }
      }
    }
  } else {
#pragma unroll
    for (int ri = 0; ri < NR; ++ri) {
      const int r = wave + NWV * ri;
      if (r < P) {
#pragma unroll
        for (int j = 0; j < P; ++j) Dl[lane * (P * P) + r * P + j] = acc[ri][j];
      }
    }
  }
  __syncthreads();
  for (int o = tid; o < LC_TQ * LC_TQ * K; o += NTHR) {
    const int q = o / K, k = o - q * K;
    const int qp = qpx[q];
    if (qp < 0) continue;
    const int j = k / KW, i = k - j * KW;
    const float wfx = qfx[q], wfy = qfy[q];
    const float* d = Dl + q * (P * P) + j * P + i;
    const float c = (1.f - wfy) * (1.f - wfx) * d[0] + (1.f - wfy) * wfx * d[1] + wfy * (1.f - wfx) * d[P] + wfy * wfx * d[P + 1];
    ElemIO<TOUT>::st(outp + ((long)b * HW + qp) * a.ldo + k, c * a.scale);
  }
  }  // tile loop
}

// Tile classifier: one THREAD per 8 x 8 query tile computes the bounding rectangle of the tile's integer patches (clipped
// to the image); the tile goes to the coherent list (rectangle fits the LDS stage) or to the gather list.
// ws: [0] gather count, [1] coherent count, [2] / [3] unused, then the two lists (tiles ints each).
// Appends are aggregated per wave (one atomicAdd per list and wave, positions by ballot rank), for two reasons measured
// with one atomic per tile: (a) 11 664 atomics on one word took 100 us (a counter word sustains ~90 updates / us), 20x the
// classification itself; (b) the lists came out in completion order, i.e. with the tiles of all images interleaved -
// the gather kernel then walked 16 feature maps at once (80 MB, MALL resident) instead of one or two (L2 resident) and ran
// 1.2-1.9x slower than the per-pixel kernel on the same queries.  Now a list is in tile order inside each group of 64.
template <int R>
__global__ __launch_bounds__(64) void local_corr_classify_kernel(const LocalCorrArgs a) {
  const int lane = threadIdx.x & 63;  // one wave per workgroup: 64 tiles, spread over as many CUs as there are groups
  const int tiles_x = (a.W + LC_TQ - 1) / LC_TQ, tiles_y = (a.H + LC_TQ - 1) / LC_TQ;
  const int tpi = tiles_x * tiles_y, tiles = a.B * tpi;
  const int tile = blockIdx.x * 64 + threadIdx.x;
  const bool valid = tile < tiles;
  bool coherent = false;
  if (valid) {
    const int b = tile / tpi;
    const int trem = tile - b * tpi;
    const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
    int xlo = 0x3fffffff, xhi = -0x3fffffff, ylo = 0x3fffffff, yhi = -0x3fffffff;
#pragma unroll
    for (int r = 0; r < LC_TQ; ++r) {
      const int qy = ty * LC_TQ + r;
      const float2* wrow = reinterpret_cast<const float2*>(a.warp) + (long)b * a.H * a.W + (long)min(qy, a.H - 1) * a.W;
#pragma unroll
      for (int c = 0; c < LC_TQ; ++c) {
        const int qx = tx * LC_TQ + c;
        const float2 wv = wrow[min(qx, a.W - 1)];  // clamped address, masked below: the 64 loads carry no branches
        int x0, y0;
        float fx, fy;
        unnormalize_floor(wv.x, a.W, x0, fx);
        unnormalize_floor(wv.y, a.H, y0, fy);
        if (qy < a.H && qx < a.W) {
          xlo = min(xlo, x0); xhi = max(xhi, x0);
          ylo = min(ylo, y0); yhi = max(yhi, y0);
        }
      }
    }
    const long bw = max(min(xhi + R + 1, a.W - 1) - max(xlo - R, 0) + 1, 0);
    const long bh = max(min(yhi + R + 1, a.H - 1) - max(ylo - R, 0) + 1, 0);
    coherent = bw * bh <= a.pxmax && !a.force_gather;
  }
  const unsigned long long mc = __ballot(valid && coherent), mg = __ballot(valid && !coherent);
  int basec = 0, baseg = 0;
  if (lane == 0) {
    if (mc) basec = atomicAdd(a.ws + 1, __popcll(mc));
    if (mg) baseg = atomicAdd(a.ws + 0, __popcll(mg));
  }
  basec = __shfl(basec, 0);
  baseg = __shfl(baseg, 0);
  const unsigned long long below = (1ull << lane) - 1ull;
  if (valid) {
    if (coherent) a.ws[4 + tiles + basec + __popcll(mc & below)] = tile;
    else a.ws[4 + baseg + __popcll(mg & below)] = tile;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Round 6: counting sort of the incoherent tiles' queries by the TS x TS bin of f1 their window starts in.
//   window origin ox = x0 - R, clamped to [-P, W]; bin = ((ox + P) / TS, (oy + P) / TS) of the query's image: the windows of a
//   bin lie in a rectangle of at most (TS + P - 1)^2 pixels (windows wholly outside the image contribute zeros wherever
//   they are put), which the launcher sizes to fit the tile kernel's stage.
// ws + ws_bins: count [nbins] | first item [nbins] | cursor [nbins];  ws[2] = number of items (groups of <= 64 queries of
// one bin);  ws + ws_qlist: items x 64 global pixel indices, -1 = padding (memset by the launcher).
// The order of the queries inside a bin depends on the atomics' arrival order; the RESULT of a query does not (its dot
// products are evaluated per (slot, query) with a fixed channel order whatever the item looks like).
template <int R> __device__ __forceinline__ int lc_bin_of(const LocalCorrArgs& a, long pix, int b) {
  constexpr int P = 2 * R + 2;
  int x0, y0;
  float fx, fy;
  unnormalize_floor(a.warp[pix * 2 + 0], a.W, x0, fx);
  unnormalize_floor(a.warp[pix * 2 + 1], a.H, y0, fy);
  const int ox = min(max(x0 - R, -P), a.W), oy = min(max(y0 - R, -P), a.H);
  return (b * a.bin_ny + (oy + P) / a.bin_ts) * a.bin_nx + (ox + P) / a.bin_ts;
}

// grid: one 64-thread workgroup per gather-list tile (the launch covers every tile of the call; surplus workgroups exit)
template <int R, bool SCATTER>
__global__ __launch_bounds__(64) void local_corr_bin_kernel(const LocalCorrArgs a) {
  const int li = blockIdx.x;
  const int nlist = a.ws[0];
  const int tile = a.ws[4 + li];
  if (li >= nlist) return;
  const int lane = threadIdx.x;
  const int tiles_x = (a.W + LC_TQ - 1) / LC_TQ, tiles_y = (a.H + LC_TQ - 1) / LC_TQ;
  const int tpi = tiles_x * tiles_y;
  const int b = tile / tpi;
  const int trem = tile - b * tpi;
  const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
  const int gy = ty * LC_TQ + (lane >> 3), gx = tx * LC_TQ + (lane & 7);
  if (gy >= a.H || gx >= a.W) return;
  const long pix = (long)b * a.H * a.W + (long)gy * a.W + gx;
  const int bin = lc_bin_of<R>(a, pix, b);
  const int nbins = a.B * a.bin_nx * a.bin_ny;
  int* cnt = a.ws + a.ws_bins;
  if constexpr (!SCATTER) {
    atomicAdd(cnt + bin, 1);
  } else {
    const int slot = atomicAdd(cnt + 2 * nbins + bin, 1);
    a.ws[a.ws_qlist + (long)cnt[nbins + bin] * 64 + slot] = (int)pix;
  }
}

// one workgroup: items per bin = ceil(count / 64), exclusive scan -> first item of every bin, total -> ws[2]
__global__ __launch_bounds__(1024) void local_corr_bin_scan_kernel(const LocalCorrArgs a) {
  __shared__ int part[1024];
  const int nbins = a.B * a.bin_nx * a.bin_ny;
  int* cnt = a.ws + a.ws_bins;
  const int tid = threadIdx.x;
  const int per = (nbins + 1023) / 1024;
  const int i0 = tid * per, i1 = min(i0 + per, nbins);
  int sum = 0;
  for (int i = i0; i < i1; ++i) sum += (cnt[i] + 63) >> 6;
  part[tid] = sum;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan of the 1024 partial sums
    const int v = tid >= off ? part[tid - off] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  int run = part[tid] - sum;
  for (int i = i0; i < i1; ++i) {
    cnt[nbins + i] = run;
    run += (cnt[i] + 63) >> 6;
  }
  if (tid == 1023) a.ws[2] = part[1023];
}

// Per-query gathers for the pixels of the tiles on the gather list: block = (list entry, round of 4 queries), one query per
// wave.  The launch covers every tile of the call; blocks beyond the list exit at once.  (Persistent workgroups pulling
// work with atomics were measured 1.6x slower: a single counter word saturates at ~90 dequeues / us, and pulling whole
// tiles instead serialises 16 latency-bound rounds per workgroup.)
template <int R, typename T, typename TOUT, int NW>
__global__ __launch_bounds__(64 * NW) void local_corr_list_kernel(const LocalCorrArgs a) {
  extern __shared__ __attribute__((aligned(16))) float f0s[];  // [NW waves][C]
  constexpr int RPT = 64 / NW;  // workgroups ("rounds") per tile: NW = 4 -> 16, NW = 8 -> 8
  const int li = blockIdx.x / RPT, rnd = blockIdx.x % RPT;
  // both scalar loads are issued before either is waited for (the list has one slot per tile of the call, so the second
  // address is valid whatever the count is): one L2 round trip at the head of every workgroup instead of two dependent ones
  const int nlist = a.ws[0];
  const int tile = a.ws[4 + li];
  if (li >= nlist) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tiles_x = (a.W + LC_TQ - 1) / LC_TQ, tiles_y = (a.H + LC_TQ - 1) / LC_TQ;
  const int tpi = tiles_x * tiles_y;
  const int b = tile / tpi;
  const int trem = tile - b * tpi;
  const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
  const int q = rnd * NW + wave;
  const int gy = ty * LC_TQ + (q >> 3), gx = tx * LC_TQ + (q & 7);
  const bool act = gy < a.H && gx < a.W;
  const long pix = (long)b * a.H * a.W + (long)gy * a.W + gx;
  float* myf0 = f0s + wave * a.C;
  lc_gather_stage_f0<R, T, TOUT>(a, pix, act, lane, myf0);
  __syncthreads();
  lc_gather_pixel<R, T, TOUT>(a, pix, act, lane, myf0);
}

// General per-tap form: warp[B,HW,K,2] arbitrary coordinates (plugin signature).
template <typename T, typename TOUT>
__global__ __launch_bounds__(256) void local_corr_general_kernel(const LocalCorrArgs a) {
  constexpr int CE = LcIO<T>::CE;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long HW = (long)a.H * a.W;
  const long total = (long)a.B * HW;
  const long pix = (long)blockIdx.x * 4 + wave;
  if (pix >= total) return;
  const int b = (int)(pix / HW);
  const long p = pix - (long)b * HW;
  const T* f0p = reinterpret_cast<const T*>(a.f0) + ((long)b * HW + p) * a.ld0;
  const int simg = (b + a.f1_shift) % a.nimg;
  const T* f1p = reinterpret_cast<const T*>(a.f1) + (long)simg * HW * a.ld1;
  TOUT* o = reinterpret_cast<TOUT*>(a.out) + pix * a.ldo;
  for (int k = 0; k < a.K; ++k) {
    int x0, y0;
    float fx, fy;
    unnormalize_floor(a.warp[(pix * a.K + k) * 2 + 0], a.W, x0, fx);
    unnormalize_floor(a.warp[(pix * a.K + k) * 2 + 1], a.H, y0, fy);
    if (a.nearest) {  // grid_sample mode="nearest": nearbyint of the un-normalised coordinate (ties to even), zero padding
      x0 = (int)rintf(unnormalize(a.warp[(pix * a.K + k) * 2 + 0], a.W));
      y0 = (int)rintf(unnormalize(a.warp[(pix * a.K + k) * 2 + 1], a.H));
      fx = fy = 0.f;  // wgt = {1, 0, 0, 0}: the single tap (y0, x0)
    }
    const float wgt[4] = {(1.f - fy) * (1.f - fx), (1.f - fy) * fx, fy * (1.f - fx), fy * fx};
    float sum = 0.f;
    for (int c = lane * CE; c < a.C; c += 64 * CE) {
      float q[CE];
      LcIO<T>::ld(f0p + c, q);
#pragma unroll
      for (int t = 0; t < (a.nearest ? 1 : 4); ++t) {
        const int yy = y0 + (t >> 1), xx = x0 + (t & 1);
        if (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) {
          float v[CE];
          LcIO<T>::ld(f1p + ((long)yy * a.W + xx) * a.ld1 + c, v);
          float d = 0.f;
#pragma unroll
          for (int j = 0; j < CE; ++j) d = fmaf(v[j], q[j], d);
          sum = fmaf(wgt[t], d, sum);
        }
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
    if (lane == 0) ElemIO<TOUT>::st(o + k, sum * a.scale);
  }
}

static int check_common(const LocalCorrArgs& a, int ce) {
  ROMA_REQUIRE(a.B > 0 && a.H > 0 && a.W > 0 && a.C > 0, "local_corr: empty problem");
  ROMA_REQUIRE(a.ld0 % ce == 0 && a.ld1 % ce == 0, "local_corr: feature strides must keep 16-byte alignment");
  ROMA_REQUIRE((reinterpret_cast<uintptr_t>(a.f0) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.f1) & 15) == 0,
               "local_corr: features must be 16-byte aligned");
  ROMA_REQUIRE(a.nimg > 0, "local_corr: nimg must be positive");
  return 0;
}

int g_lc_mode = -1;  // roma_tuning("lc_mode"): -1 / 0 = tiled (MFMA all-pairs for 16-bit features) + work list (default), 1 = every tile to the gather list, 2 = per-pixel launch (the form every other radius uses)

int g_lc_bin = -1;

// bin edge: the largest TS with (TS + P - 1)^2 <= the smallest stage any build of the tile kernel has for this radius (the
// 16-bit MFMA form's 320 slots for r <= 3) - one geometry for every precision keeps local_corr_ws_ints independent of it
static int lc_bin_ts(int radius) {
  const int P = 2 * radius + 2, pxmax = radius <= 3 ? 320 : 704;
  int ts = 1;
  while ((ts + P) * (ts + P) <= pxmax) ++ts;  // (ts + 1 + P - 1)^2
  return ts;
}
long local_corr_ws_ints(int B, int H, int W, int radius) {
  if (radius != 2 && radius != 3 && radius != 7) return 0;
  const long tiles = (long)B * ((H + LC_TQ - 1) / LC_TQ) * ((W + LC_TQ - 1) / LC_TQ);
  const int ts = lc_bin_ts(radius), P = 2 * radius + 2;
  const long nbins = (long)B * ((W + P) / ts + 1) * ((H + P) / ts + 1);
  return 4 + 2 * tiles + 3 * nbins + 64 * (tiles + nbins);
}

template <int R, typename T, typename TOUT, bool MFMA>
static int launch_tiled(const LocalCorrArgs& a0, hipStream_t stream) {
  LocalCorrArgs a = a0;
  const int tiles = a.B * ((a.H + LC_TQ - 1) / LC_TQ) * ((a.W + LC_TQ - 1) / LC_TQ);
  constexpr int P = 2 * R + 2;
  a.bin_ts = lc_bin_ts(R);
  a.bin_nx = (a.W + P) / a.bin_ts + 1;
  a.bin_ny = (a.H + P) / a.bin_ts + 1;
  const long nbins = (long)a.B * a.bin_nx * a.bin_ny;
  a.ws_bins = 4 + 2 * tiles;
  a.ws_qlist = a.ws_bins + 3 * (int)nbins;
  static const int bin_env = getenv("ROMA_LC_BIN") ? atoi(getenv("ROMA_LC_BIN")) : 1;
  const bool binned = (g_lc_bin >= 0 ? g_lc_bin : bin_env) != 0 && (long)a.B * a.H * a.W < (1l << 31) &&
                      (a.bin_ts + P - 1) * (a.bin_ts + P - 1) <= LcGeom<R, MFMA>::PXMAX;
  const size_t need = (size_t)local_corr_ws_ints(a.B, a.H, a.W, R) * sizeof(int);
  bool own_ws = false;
  if (!a.ws) {  // operator entry points: stream-ordered scratch (the model passes a slice of its arena)
    ROMA_CHECK_HIP(hipMallocAsync(reinterpret_cast<void**>(&a.ws), need, stream));
    own_ws = true;
  } else {
    ROMA_REQUIRE((size_t)a.ws_bytes >= need, "local_corr(window): work-list scratch too small");
  }
  a.force_gather = g_lc_mode == 1 ? 1 : 0;
  ROMA_CHECK_HIP(hipMemsetAsync(a.ws, 0, 4 * sizeof(int), stream));
  const size_t lds_tile = (size_t)LcGeom<R, MFMA>::STAGE + 5 * 64 * 4 + 32;
  a.pxmax = LcGeom<R, MFMA>::PXMAX;  // what the classifier calls a coherent tile: its rectangle fits this kernel's stage
  static bool attr_set[64] = {false};
  int dev = 0;
  ROMA_CHECK_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    ROMA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&local_corr_tile_kernel<R, T, TOUT, MFMA>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_tile));
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  hipLaunchKernelGGL((local_corr_classify_kernel<R>), dim3((unsigned)((tiles + 63) / 64)), dim3(64), 0, stream, a);
  ROMA_LAUNCH_CHECK();
  // (capping this launch at 1024 workgroups - the kernel strides over the list - changes nothing either way: measured)
  hipLaunchKernelGGL((local_corr_tile_kernel<R, T, TOUT, MFMA>), dim3((unsigned)tiles), dim3(LcGeom<R, MFMA>::NTHR), lds_tile, stream, a);
  ROMA_LAUNCH_CHECK();
  // gather list: one query per wave, four per workgroup.  (Eight waves per workgroup - to share the two dependent scalar
  // loads at the head of every workgroup - were measured SLOWER on the benchmark model's incoherent warps: 1.86 vs 1.60 ms
  // at r = 2, 0.96 vs 0.91 at r = 3, profiles/r03_final_visit.log; the variant was removed in round 4.)
  if (binned) {
    // incoherent tiles: sort their queries by target bin, then the LIST form of the tile kernel (one item = <= 64 queries of a bin)
    ROMA_CHECK_HIP(hipMemsetAsync(a.ws + a.ws_bins, 0, (size_t)3 * nbins * sizeof(int), stream));
    ROMA_CHECK_HIP(hipMemsetAsync(a.ws + a.ws_qlist, 0xff, (size_t)64 * (tiles + nbins) * sizeof(int), stream));
    hipLaunchKernelGGL((local_corr_bin_kernel<R, false>), dim3((unsigned)tiles), dim3(64), 0, stream, a);
    ROMA_LAUNCH_CHECK();
    hipLaunchKernelGGL(local_corr_bin_scan_kernel, dim3(1), dim3(1024), 0, stream, a);
    ROMA_LAUNCH_CHECK();
    hipLaunchKernelGGL((local_corr_bin_kernel<R, true>), dim3((unsigned)tiles), dim3(64), 0, stream, a);
    ROMA_LAUNCH_CHECK();
    static bool attr_set_l[64] = {false};
    if (dev < 0 || dev >= 64 || !attr_set_l[dev]) {
      ROMA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&local_corr_tile_kernel<R, T, TOUT, MFMA, true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_tile));
      if (dev >= 0 && dev < 64) attr_set_l[dev] = true;
    }
    // the number of items is only known on the device: at most one per tile's worth of queries plus one partial item per bin.
    // The kernel strides over the item list, so a capped grid is enough; surplus workgroups exit at once.
    const long max_items = (long)tiles + nbins;
    hipLaunchKernelGGL((local_corr_tile_kernel<R, T, TOUT, MFMA, true>), dim3((unsigned)std::min<long>(max_items, 8192)), dim3(LcGeom<R, MFMA>::NTHR),
                       lds_tile, stream, a);
    ROMA_LAUNCH_CHECK();
  } else {
    hipLaunchKernelGGL((local_corr_list_kernel<R, T, TOUT, 4>), dim3((unsigned)tiles * 16u), dim3(256), (size_t)4 * a.C * sizeof(float),
                       stream, a);
    ROMA_LAUNCH_CHECK();
  }
  if (own_ws) ROMA_CHECK_HIP(hipFreeAsync(a.ws, stream));
  return 0;
}

template <int R>
static int launch_window_r(const LocalCorrArgs& a, hipStream_t stream) {
  const long total = (long)a.B * a.H * a.W;
  dim3 grid((unsigned)((total + 3) / 4));
  size_t lds = (size_t)4 * a.C * sizeof(float);
  // algorithmic bytes: f0 + f1 read once, centre warp, K outputs
  const double es_in = a.in_dt == DT_F32 ? 4.0 : 2.0, es_out = a.out_dt == DT_F32 ? 4.0 : 2.0;
  char pname[64];
  snprintf(pname, sizeof pname, "local_corr_window_kernel<%d,%s>", R, a.in_dt == DT_F32 ? "f32" : ROMA_H16_NAME);
  ProfScope ps(pname, (double)total * (2.0 * a.C * es_in + 8.0 + (2.0 * R + 1) * (2.0 * R + 1) * es_out), "byte", stream);
  static const int env_mode = getenv("ROMA_LC_MODE") ? atoi(getenv("ROMA_LC_MODE")) : 0;
  const int mode = g_lc_mode >= 0 ? g_lc_mode : env_mode;
  const int cc = a.in_dt == DT_F32 ? 32 : 64;  // channels per 128-byte chunk of the tiled form
  // the tiled form is instantiated for the radii RoMa uses (roma_models.py:103-139: 7, 3, 2); other radii keep the
  // per-pixel kernel
  if constexpr (R == 2 || R == 3 || R == 7) {
    if (mode != 2 && a.C % cc == 0) {
      if (a.in_dt == DT_F32 && a.out_dt == DT_F32) return launch_tiled<R, float, float, false>(a, stream);
      if (a.in_dt == DT_F32) return launch_tiled<R, float, bf16_t, false>(a, stream);
      if (a.out_dt == DT_F32) return launch_tiled<R, bf16_t, float, true>(a, stream);
      return launch_tiled<R, bf16_t, bf16_t, true>(a, stream);
    }
  }
#define ROMA_LC(T, TOUT) hipLaunchKernelGGL((local_corr_window_kernel<R, T, TOUT>), grid, dim3(256), lds, stream, a)
  if (a.in_dt == DT_F32 && a.out_dt == DT_F32) ROMA_LC(float, float);
  else if (a.in_dt == DT_F32) ROMA_LC(float, bf16_t);
  else if (a.out_dt == DT_F32) ROMA_LC(bf16_t, float);
  else ROMA_LC(bf16_t, bf16_t);
#undef ROMA_LC
  ROMA_LAUNCH_CHECK();
  return 0;
}

int local_corr_window_launch(const LocalCorrArgs& a, hipStream_t stream) {
  const int ce = a.in_dt == DT_F32 ? 4 : 8;
  if (int rc = check_common(a, ce)) return rc;
  const int P = 2 * a.radius + 2;
  const int S = 64 / (P <= 8 ? 8 : 16);
  ROMA_REQUIRE(a.C % (ce * S) == 0, "local_corr(window): C must be a multiple of 16B-chunk x lanes-per-column");
  ROMA_REQUIRE((size_t)4 * a.C * sizeof(float) <= 64 * 1024, "local_corr(window): C too large for the LDS f0 stage");
  switch (a.radius) {
    case 1: return launch_window_r<1>(a, stream);
    case 2: return launch_window_r<2>(a, stream);
    case 3: return launch_window_r<3>(a, stream);
    case 4: return launch_window_r<4>(a, stream);
    case 5: return launch_window_r<5>(a, stream);
    case 6: return launch_window_r<6>(a, stream);
    case 7: return launch_window_r<7>(a, stream);
    default: set_error("local_corr(window): radius must be in 1..7"); return -1;
  }
}

int local_corr_general_launch(const LocalCorrArgs& a, hipStream_t stream) {
  const int ce = a.in_dt == DT_F32 ? 4 : 8;
  if (int rc = check_common(a, ce)) return rc;
  ROMA_REQUIRE(a.C % ce == 0 && a.K > 0, "local_corr(general): bad C or K");
  const long total = (long)a.B * a.H * a.W;
  dim3 grid((unsigned)((total + 3) / 4));
#define ROMA_LC(T, TOUT) hipLaunchKernelGGL((local_corr_general_kernel<T, TOUT>), grid, dim3(256), 0, stream, a)
  if (a.in_dt == DT_F32 && a.out_dt == DT_F32) ROMA_LC(float, float);
  else if (a.in_dt == DT_F32) ROMA_LC(float, bf16_t);
  else if (a.out_dt == DT_F32) ROMA_LC(bf16_t, float);
  else ROMA_LC(bf16_t, bf16_t);
#undef ROMA_LC
  ROMA_LAUNCH_CHECK();
  return 0;
}

}  // namespace roma

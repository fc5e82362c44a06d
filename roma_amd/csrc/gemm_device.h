// Device-side pieces of the GEMM kernel (gemm.hip): LDS row geometry, the LDS-DMA helper, the
// zero page and the compile-time specialised staged epilogues.
#pragma once
#include "gemm.h"

namespace roma {

constexpr int ROWB = 128;  // bytes per LDS row = 8 chunks of 16 B

template <typename T> struct InTraits;
template <> struct InTraits<float> { static constexpr int CE = 4; };
template <> struct InTraits<bf16_t> { static constexpr int CE = 8; };

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;

static __device__ __attribute__((aligned(256))) unsigned int g_zero_page[64];  // source of every out-of-range DMA chunk (one copy per TU)

__device__ inline float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }
// Same function for 16-bit outputs, built for the VALU: 128 values per lane per 256 x 256 tile make the activation a visible
// part of the fc1 GEMM (libm's branchy erff ~45 instructions: 27 % of it; the round-1..3 form - Abramowitz-Stegun 7.1.26, one
// v_rcp + one v_exp + 13 full-rate operations - still ~84 issue cycles per value, ~9 us per tile next to a ~30 us K loop).
//     gelu(x) = max(x, 0) - |x| / 2 * erfc(|x| / sqrt 2),      erfc(z / sqrt 2) = 2^-(z * Q(z)),  Q of degree 4
// (log2 of erfc is smooth and nearly quadratic; the coefficients are a weighted minimax fit on [0, 6], tools/fit_gelu.py).
// One quarter-rate v_exp and 8 full-rate operations; |x| is a source modifier.  |error| <= 7.2e-7 absolute over all x
// (f32 evaluation, tested against erf in tests/test_cpu_oracle.py), relative <= 1.2e-5 for x >= -2: 1/300 of the 2^-9 rounding
// of the store that follows.
__device__ __forceinline__ float gelu_erf_fast(float x) {
  const float z = fabsf(x);
  float q = fmaf(4.881172499153763e-4f, z, -7.198805455118418e-3f);
  q = fmaf(q, z, 5.2146803587675095e-2f);
  q = fmaf(q, z, 4.595957100391388e-1f);
  q = fmaf(q, z, 1.1510006189346313f);
  const float e = __builtin_amdgcn_exp2f(-(q * z));
  return fmaf(-0.5f * z, e, fmaxf(x, 0.f));
}

// The same function on four values with packed-f32 arithmetic (v_pk_fma_f32 / v_pk_mul_f32: two values per issue slot) -
// only in translation units compiled WITH packed-f32 instructions (ROMA_EPI_PK: gemm8p.hip, the DINOv2 fc1 GEMM).  Per
// element the same IEEE operations in the same order as gelu_erf_fast: bit-identical.  Round 6: the activation was 9 + 4
// (quarter-rate v_exp) issue slots per value = 1 660 of the ~1 900 VALU slots of the fc1 epilogue per wave and tile.
#if defined(ROMA_EPI_PK)
typedef float epi_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ epi_f32x2 gelu_erf_fast2(epi_f32x2 x) {
  const epi_f32x2 z = {fabsf(x[0]), fabsf(x[1])};
  epi_f32x2 q = __builtin_elementwise_fma(epi_f32x2{4.881172499153763e-4f, 4.881172499153763e-4f}, z,
                                          epi_f32x2{-7.198805455118418e-3f, -7.198805455118418e-3f});
  q = __builtin_elementwise_fma(q, z, epi_f32x2{5.2146803587675095e-2f, 5.2146803587675095e-2f});
  q = __builtin_elementwise_fma(q, z, epi_f32x2{4.595957100391388e-1f, 4.595957100391388e-1f});
  q = __builtin_elementwise_fma(q, z, epi_f32x2{1.1510006189346313f, 1.1510006189346313f});
  const epi_f32x2 t = q * z;
  const epi_f32x2 e = {__builtin_amdgcn_exp2f(-t[0]), __builtin_amdgcn_exp2f(-t[1])};
  const epi_f32x2 mz = z * epi_f32x2{-0.5f, -0.5f};
  const epi_f32x2 mx = {fmaxf(x[0], 0.f), fmaxf(x[1], 0.f)};
  return __builtin_elementwise_fma(mz, e, mx);
}
#endif

template <typename TOUT> __device__ inline void store4(TOUT* p, f32x4 v, bool vec, int nvalid) {
  if (vec && nvalid >= 4) {
    ElemIO<TOUT>::st4(p, v);
  } else {
    for (int j = 0; j < 4; ++j)
      if (j < nvalid) ElemIO<TOUT>::st(p + j, v[j]);
  }
}

__device__ __forceinline__ void glds16(const char* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// ---------------------------------------------------------------------------------------------------------------
// Plain (EPI_STD) epilogues, specialised at compile time on the activation and on "tile completely inside M x N".
// One generic epilogue with every mode / activation / tail case inlined 32x per tile was ~50k instructions: its
// straight-line path no longer fitted the instruction cache and cost ~8 us per 256x256 tile (as much as the K loop
// at K = 1024).  Each variant below is a few hundred instructions.
// XOR swizzle of the 16-byte chunk index inside a staged row of CPR chunks.  Power-of-two rows: row & (CPR-1).
// CPR = 12 (256x192 tiles, 192-byte rows = 48 dwords): 48*row mod 64 only takes 4 values, so the chunk is rotated
// inside its group of 4 by (row >> 2) - rows r, r+4, r+8, r+12 then land in different banks (the unswizzled layout
// measured SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.38 for this tile shape, 0.06 for 256x256).
template <int CPR> __device__ __forceinline__ int epi_swz(int row) {
  if constexpr ((CPR & (CPR - 1)) == 0) return row & (CPR - 1);
  else if constexpr (CPR % 4 == 0) return (row >> 2) & 3;
  else return 0;
}

// "this value is read here" for the compiler's wait-count bookkeeping (no instruction is emitted)
__device__ __forceinline__ void epi_consume(f32x4 v) { asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3])); }
__device__ __forceinline__ void epi_consume(uint4 v) { asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w)); }

// Per-column epilogue vectors (bias, LayerScale) of the wave's TN x 32 columns.  Each lane needs the same 4 consecutive
// columns of every 8-column group for ALL its rows, so they are loaded ONCE per tile, back to back, before the
// accumulators are touched: one vmcnt wait per tile.  (Loading them where they are used - inside the (tm, tn, rg) loops -
// made hipcc emit `global_load ; s_waitcnt vmcnt(0) ; use` 32..64 times per tile, each a serialised L2 round trip that
// also drains the LDS-DMA queue: ~9 us of the ~12 us epilogue of a 256 x 256 tile, profiles/r02_v11_gemm_overhead.log.)
template <int TN, bool FULL, bool HAS_S>
struct EpiCols {
  f32x4 b[TN][4], s[HAS_S ? TN : 1][4];
  bool has_b;
  // columns nw0 + tn * 32 + 8 * rg + 4 * h + [0, 4), tn in [0, TN)
  __device__ __forceinline__ void load(const GemmArgs& a, int nw0, int h) {
    has_b = a.bias != nullptr;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int n = nw0 + tn * 32 + 8 * rg + 4 * h;
        // no bias: -0.0, the exact additive identity (x + -0.0 == x for every x including -0.0), so that the epilogues add
        // unconditionally - with `if (has_b)` hipcc computed BOTH alpha * acc and fma(alpha, acc, bias) for every value and
        // selected (12 VALU instructions per 4 values instead of 4: round 6 ISA count)
        f32x4 bv = {-0.f, -0.f, -0.f, -0.f}, sv = {1.f, 1.f, 1.f, 1.f};
        if (FULL || n + 3 < a.N) {
          if (has_b) bv = *reinterpret_cast<const f32x4*>(a.bias + n);
          if (HAS_S && a.scale) sv = *reinterpret_cast<const f32x4*>(a.scale + n);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (n + j < a.N) {
              if (has_b) bv[j] = a.bias[n + j];
              if (HAS_S && a.scale) sv[j] = a.scale[n + j];
            }
        }
        b[tn][rg] = bv;
        if constexpr (HAS_S) s[tn][rg] = sv;
      }
    // Consume every vector HERE (an empty asm that reads it): the compiler then places its one vmcnt wait right behind the
    // loads.  Without it, a path that never uses a vector (padding row blocks, empty tails) leaves "load pending" on those
    // registers in hipcc's bookkeeping, and the K loop - which reuses them for fragments - gets an s_waitcnt vmcnt(0) at
    // its top that drains the LDS-DMA pipeline every K tile.
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        epi_consume(b[tn][rg]);
        if constexpr (HAS_S) epi_consume(s[tn][rg]);
      }
  }
};

// alpha * acc + bias as ONE fma per value (what hipcc's contraction made of `alpha * acc` followed by `+= bias` all along;
// written out so that it stays one instruction - and one rounding - in the packed-f32 translation units too).  bias is -0.0
// where the caller passed none (EpiCols::load): fma(alpha, acc, -0.0) == alpha * acc exactly.
__device__ __forceinline__ f32x4 epi_axpb(float alpha, f32x4 accv, f32x4 bv) {
#if defined(ROMA_EPI_PK)
  const epi_f32x2 al = {alpha, alpha};
  const epi_f32x2 r0 = __builtin_elementwise_fma(al, epi_f32x2{accv[0], accv[1]}, epi_f32x2{bv[0], bv[1]});
  const epi_f32x2 r1 = __builtin_elementwise_fma(al, epi_f32x2{accv[2], accv[3]}, epi_f32x2{bv[2], bv[3]});
  return f32x4{r0[0], r0[1], r1[0], r1[1]};
#else
  f32x4 v;
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = fmaf(alpha, accv[j], bv[j]);
  return v;
#endif
}

// activation and per-column scale of values that already carry their bias
template <int ACT, bool BF16_OUT, bool HAS_S>
__device__ __forceinline__ f32x4 epi_act(f32x4 v, f32x4 sv) {
  if (ACT == ACT_RELU) {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
  } else if (ACT == ACT_GELU) {
#if defined(ROMA_EPI_PK)
    if constexpr (BF16_OUT) {
      const epi_f32x2 g0 = gelu_erf_fast2(epi_f32x2{v[0], v[1]}), g1 = gelu_erf_fast2(epi_f32x2{v[2], v[3]});
      v = f32x4{g0[0], g0[1], g1[0], g1[1]};
    } else
#endif
    {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = BF16_OUT ? gelu_erf_fast(v[j]) : gelu_erf(v[j]);
    }
  }
  if constexpr (HAS_S) v *= sv;  // (1.0 where the caller passed no scale)
  return v;
}
template <int ACT, bool BF16_OUT, bool HAS_S>
__device__ __forceinline__ f32x4 epi_apply(f32x4 v, f32x4 bv, f32x4 sv) {
  v += bv;  // (-0.0 where the caller passed no bias: EpiCols::load)
  return epi_act<ACT, BF16_OUT, HAS_S>(v, sv);
}

// bf16 output: the MFMA layout gives each lane 4 consecutive n of ONE row, i.e. a wave store would scatter 8-byte
// pieces over 32 rows (measured 0.56 TB/s).  Stage the wave's tile through its private LDS slice instead and write
// whole rows: 16 B per lane, 128..384 contiguous bytes per row.  (One wave's LDS operations complete in order.)
// RES: add a bf16 residual (a.res_bf16, may alias C) to the rows as they leave LDS - the same lane reads and then
// writes each 16-byte piece, so the in-place update is race free.
template <int TM, int TN, int ACT, bool FULL, bool RES = false, bool HAS_S = false>
__device__ __forceinline__ void epi_staged_bf16(const f32x16 (&acc)[TN][TM], const GemmArgs& a, bf16_t* Cb, char* ws,
                                                long mw0, int nw0, int lane, const bf16_t* Rb = nullptr) {
  constexpr int RB = TN * 64;    // staged row: TN*32 bf16
  constexpr int CPR = TN * 4;    // 16-byte chunks per row
  constexpr int NPC = (32 * CPR) / 64;  // 16-byte pieces per lane per 32-row block
  const int l31 = lane & 31, h = lane >> 5;
  // HAS_S: a per-column scale (another 32 registers next to the 128 accumulators: the compiler spills).  The model does
  // not need it - LayerScale is folded into the weights in bf16 mode (model.hip pack_weights) - it is kept for callers of
  // the operator entry point, only together with the bf16 residual; gemm_launch routes other uses to the generic epilogue.
  EpiCols<TN, FULL, HAS_S> cols;
  cols.load(a, nw0, h);
  // Geometry of the 16-byte pieces this lane moves from the staged rows to C: the same for every 32-row block, so the LDS
  // offset and the per-lane byte offset into C / the residual are formed ONCE; the row block only moves the wave-uniform
  // base.  (Round 6: per piece and block hipcc re-derived row / chunk / swizzle and a 64-bit m * ldc product - ~22 VALU
  // instructions, three of them quarter-rate multiplies - and waited for each ds_read right behind it; now the NPC reads
  // are issued back to back.)
  int rdo[NPC], prow[NPC], pn[NPC];
  unsigned doff[NPC], roff[RES ? NPC : 1];
#pragma unroll
  for (int i = 0; i < NPC; ++i) {
    const int c = lane + 64 * i;
    const int row = c / CPR, ch = c - row * CPR;
    rdo[i] = row * RB + ((ch ^ epi_swz<CPR>(row)) << 4);
    prow[i] = row;
    pn[i] = nw0 + ch * 8;
    doff[i] = (unsigned)(row * (int)a.ldc + ch * 8) * 2u;  // (row < 32: fits 32 bits for every ldc < 2^25)
    if constexpr (RES) roff[i] = (unsigned)(row * (int)a.ldr + ch * 8) * 2u;
  }
  char* cbase = reinterpret_cast<char*>(Cb + mw0 * a.ldc + nw0);
  const char* rbase = RES ? reinterpret_cast<const char*>(Rb + mw0 * a.ldr + nw0) : nullptr;
  const long dstep = 64 * a.ldc, rstep = 64 * a.ldr;  // 32 rows, in bytes
  // bf16 residual rows in a rolling buffer: piece i of block tm + 1 is requested the moment piece i of block tm has been
  // consumed, so a request has the rest of the row pass plus the next staging pass to come back from L2 / HBM (a second
  // buffer would cost 16-24 registers next to the accumulators: it spilled).  The lane that reads a 16-byte piece is the
  // lane that later writes it, and blocks are disjoint rows: in-place is race free.
  uint4 rres[RES ? NPC : 1];
  auto piece_ok = [&](int tm, int i) { return FULL || (mw0 + tm * 32 + prow[i] < a.M && pn[i] + 8 <= a.N); };
  auto load_res = [&](int tm, int i) {
    rres[i] = make_uint4(0, 0, 0, 0);
    if (piece_ok(tm, i)) rres[i] = *reinterpret_cast<const uint4*>(rbase + (long)tm * rstep + roff[i]);
  };
  if constexpr (RES) {
#pragma unroll
    for (int i = 0; i < NPC; ++i) load_res(0, i);
  }
  const bool nt = (a.dbg & 1024) != 0;  // streaming (non-temporal) stores, the default (gemm_launch): acknowledged sooner in the vmcnt queue
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const f32x16& ab = acc[tn][tm];
        f32x4 v = epi_axpb(a.alpha, f32x4{ab[4 * rg], ab[4 * rg + 1], ab[4 * rg + 2], ab[4 * rg + 3]}, cols.b[tn][rg]);
        v = epi_act<ACT, true, HAS_S>(v, cols.s[HAS_S ? tn : 0][rg]);
        uint2 pk;
        pk.x = pack_bf16x2(v[0], v[1]);
        pk.y = pack_bf16x2(v[2], v[3]);
        const int ch = tn * 4 + rg;
        *reinterpret_cast<uint2*>(ws + l31 * RB + ((ch ^ epi_swz<CPR>(l31)) << 4) + 8 * h) = pk;
      }
    uint4 pv[NPC];
#pragma unroll
    for (int i = 0; i < NPC; ++i) pv[i] = *reinterpret_cast<const uint4*>(ws + rdo[i]);
    if constexpr (RES) {
#pragma unroll
      for (int i = 0; i < NPC; ++i) {
        uint4 v = pv[i];
        epi_consume(rres[i]);  // consumed on every path (see EpiCols::load)
        if (piece_ok(tm, i)) {
          const uint4 r = rres[i];
          const unsigned* vp = reinterpret_cast<const unsigned*>(&v);
          const unsigned* rp = reinterpret_cast<const unsigned*>(&r);
          unsigned o[4];
#pragma unroll
          for (int j = 0; j < 4; ++j)
            o[j] = pack_bf16x2(h16_lo(vp[j]) + h16_lo(rp[j]),
                               h16_hi(vp[j]) + h16_hi(rp[j]));
          v = make_uint4(o[0], o[1], o[2], o[3]);
        }
        if (tm + 1 < TM) load_res(tm + 1, i);  // the register is free again: request the same piece of the next block
        pv[i] = v;
      }
    }
    if (a.dbg & 1) continue;  // ablation (tools/bench_gemm_overhead.py): no stores
    char* cb = cbase + (long)tm * dstep;
    if (FULL) {
      if (nt) {
        typedef unsigned u32x4n __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int i = 0; i < NPC; ++i)
          __builtin_nontemporal_store(u32x4n{pv[i].x, pv[i].y, pv[i].z, pv[i].w}, reinterpret_cast<u32x4n*>(cb + doff[i]));
      } else {
#pragma unroll
        for (int i = 0; i < NPC; ++i) *reinterpret_cast<uint4*>(cb + doff[i]) = pv[i];
      }
    } else {
#pragma unroll
      for (int i = 0; i < NPC; ++i) {
        const long m = mw0 + tm * 32 + prow[i];
        const int n = pn[i];
        if (m >= a.M || n >= a.N) continue;
        bf16_t* d = reinterpret_cast<bf16_t*>(cb + doff[i]);
        if (n + 8 <= a.N) {
          *reinterpret_cast<uint4*>(d) = pv[i];
        } else {
          const bf16_t* e = reinterpret_cast<const bf16_t*>(&pv[i]);
          for (int j = 0; j < 8; ++j)
            if (n + j < a.N) d[j] = e[j];
        }
      }
    }
  }
}

// EPI_QKV with bf16 outputs.  The GEMM rows are padded to npad tokens per image (gemm_launch), so every 32-row MFMA
// block lies inside one image at a 32-aligned token: q / k rows are written like the plain staged epilogue (one head
// = 2*hd contiguous bytes per token), and V is staged TRANSPOSED ([d][token]) so that V^T leaves as 16-byte pieces of
// 8 consecutive tokens instead of single bf16 elements.
template <int TM, int TN>
__device__ __forceinline__ void epi_staged_qkv(const f32x16 (&acc)[TN][TM], const GemmArgs& a, char* ws, long mw0, int nw0,
                                               int lane) {
  constexpr int RB = TN * 64;
  constexpr int CPR = TN * 4;
  const int l31 = lane & 31, h = lane >> 5;
  const int D = a.heads * a.hd;
  const int which = nw0 / D;         // wave-uniform: D % (TN*32) == 0
  const int nrel = nw0 - which * D;  // first column of this wave inside q / k / v
  const float sc = which == 0 ? a.qscale : 1.0f;
  EpiCols<TN, true, false> cols;  // N = 3 * heads * hd is a multiple of the wave's 64 columns: always a full vector
  cols.load(a, nw0, h);
  // the 16-byte pieces this lane moves out of the staged block: LDS offset and element offset inside the image's q / k
  // ([head][token][d]) or V^T ([head][d][token]) are the same for every 32-row block (round 6: were re-derived, with a
  // division by hd, per piece and block)
  constexpr int NPC = (32 * CPR) / 64;
  int rdo[NPC], poff[NPC], prow[NPC];
#pragma unroll
  for (int i = 0; i < NPC; ++i) {
    const int c = lane + 64 * i;
    if (which < 2) {
      const int row = c / CPR, ch = c - row * CPR;
      const int nr = nrel + ch * 8;
      const int head = nr / a.hd, d = nr - head * a.hd;
      rdo[i] = row * RB + ((ch ^ epi_swz<CPR>(row)) << 4);
      poff[i] = (head * a.npad + row) * a.hd + d;
      prow[i] = row;
    } else {
      const int dl = c >> 2, tg = c & 3;
      rdo[i] = dl * 64 + tg * 16;
      poff[i] = (nrel + dl) * a.npad + tg * 8;  // (head * hd + d == the column inside V)
      prow[i] = 0;
    }
  }
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const long mw = mw0 + tm * 32;
    if (mw >= a.M) continue;
    const int qb = (int)(mw / a.npad);
    const int qt0 = (int)(mw - (long)qb * a.npad);
    if (qt0 >= a.ntok) continue;  // block of padding rows only
    if (which < 2) {
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int n = nw0 + tn * 32 + 8 * rg + 4 * h;
          f32x4 v;
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = acc[tn][tm][4 * rg + j];
          v += cols.b[tn][rg];
          v *= sc;
          uint2 pk;
          pk.x = pack_bf16x2(v[0], v[1]);
          pk.y = pack_bf16x2(v[2], v[3]);
          const int ch = tn * 4 + rg;
          *reinterpret_cast<uint2*>(ws + l31 * RB + ((ch ^ epi_swz<CPR>(l31)) << 4) + 8 * h) = pk;
        }
      bf16_t* dstb = reinterpret_cast<bf16_t*>(which == 0 ? a.q : a.k) + ((long)qb * a.heads * a.npad + qt0) * a.hd;
      uint4 pv[NPC];
#pragma unroll
      for (int i = 0; i < NPC; ++i) pv[i] = *reinterpret_cast<const uint4*>(ws + rdo[i]);
      if (a.dbg & 1) continue;
#pragma unroll
      for (int i = 0; i < NPC; ++i)
        if (qt0 + prow[i] < a.ntok) *reinterpret_cast<uint4*>(dstb + poff[i]) = pv[i];  // padding rows stay zero
    } else {
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int nl = tn * 32 + 8 * rg + 4 * h;
          f32x4 v;
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = acc[tn][tm][4 * rg + j];
          v += cols.b[tn][rg];
          const uint32_t p0 = pack_bf16x2(v[0], v[1]), p1 = pack_bf16x2(v[2], v[3]);
          unsigned short* col = reinterpret_cast<unsigned short*>(ws + nl * 64 + l31 * 2);  // [d][token]
          col[0] = (unsigned short)(p0 & 0xffffu);
          col[32] = (unsigned short)(p0 >> 16);
          col[64] = (unsigned short)(p1 & 0xffffu);
          col[96] = (unsigned short)(p1 >> 16);
        }
      static_assert(NPC == TN * 2, "V^T pieces");
      bf16_t* vt = reinterpret_cast<bf16_t*>(a.vt) + (long)qb * a.heads * a.hd * a.npad + qt0;
      uint4 pv[NPC];
#pragma unroll
      for (int i = 0; i < NPC; ++i) pv[i] = *reinterpret_cast<const uint4*>(ws + rdo[i]);
      if (a.dbg & 1) continue;
#pragma unroll
      for (int i = 0; i < NPC; ++i) *reinterpret_cast<uint4*>(vt + poff[i]) = pv[i];
    }
  }
}

// f32 output (+ optional f32 residual, which may alias C): same idea, one 32 x 32 MFMA block (4 KiB) at a time.
// Direct stores from the MFMA layout touch 32 rows x 32 B per instruction; staged, a wave instruction covers
// 8 rows x 128 contiguous bytes for both the residual read and the store.  Requires N % 4 == 0.
template <int TM, int TN, int ACT, bool FULL>
__device__ __forceinline__ void epi_staged_f32(const f32x16 (&acc)[TN][TM], const GemmArgs& a, float* Cb, const float* Rb,
                                               char* ws, long mw0, int nw0, int lane) {
  const int l31 = lane & 31, h = lane >> 5;
  // one 32-column block at a time (tn outer): its bias / scale vectors are 32 registers, loaded once for all TM row blocks
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    EpiCols<1, FULL, true> cols;
    cols.load(a, nw0 + tn * 32, h);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const long mw = mw0 + tm * 32;
      const int nw = nw0 + tn * 32;
      // the f32 residual pieces of this 32 x 32 block: issued before the staging pass, added after it (the same lane reads
      // and then writes each piece, so the in-place update C = res + ... stays race free)
      f32x4 rres[4];
      if (Rb) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int c = lane + 64 * i;
          const int row = c >> 3, ch = c & 7;
          const long m = mw + row;
          const int n = nw + ch * 4;
          rres[i] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (FULL || (m < a.M && n < a.N)) rres[i] = *reinterpret_cast<const f32x4*>(Rb + m * a.ldr + n);
        }
      }
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const f32x16& ab = acc[tn][tm];
        f32x4 v = epi_axpb(a.alpha, f32x4{ab[4 * rg], ab[4 * rg + 1], ab[4 * rg + 2], ab[4 * rg + 3]}, cols.b[0][rg]);
        v = epi_act<ACT, false, true>(v, cols.s[0][rg]);
        *reinterpret_cast<f32x4*>(ws + l31 * 128 + (((2 * rg + h) ^ (l31 & 7)) << 4)) = v;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i;
        const int row = c >> 3, ch = c & 7;
        f32x4 v = *reinterpret_cast<const f32x4*>(ws + row * 128 + ((ch ^ (row & 7)) << 4));
        const long m = mw + row;
        const int n = nw + ch * 4;
        if (Rb) {
          epi_consume(rres[i]);  // consumed on every path (see EpiCols::load)
          v += rres[i];
        }
        if (!FULL && (m >= a.M || n >= a.N)) continue;
        if (a.dbg & 1) continue;
        *reinterpret_cast<f32x4*>(Cb + m * a.ldc + n) = v;
      }
    }
  }
}

// EPI_COSK with f32 output (the GP's Gram matrices, matcher.py:191-200: exp((x.y / (|x||y| + 1e-6) - 1) / T) + diag), staged like
// epi_staged_f32.  Round 4: the generic epilogue wrote these 1600 x 1600 x 16 matrices straight from the MFMA layout - 32 rows x
// 32 B per wave store, ~0.8 TB/s - which made a 42-GFLOP launch take 200 us.  Same arithmetic as the generic path (true
// division, libm expf: the Gram matrix feeds a Cholesky with condition ~1e3, so no fast exp here): bit-identical results.
// nxb / nyb: the batch item's row / column norms.  Tails are bounds-checked (n = 1600 is not a multiple of the tile).
template <int TM, int TN>
__device__ __forceinline__ void epi_staged_cosk(const f32x16 (&acc)[TN][TM], const GemmArgs& a, float* Cb, const float* nxb,
                                                const float* nyb, char* ws, long mw0, int nw0, int lane) {
  const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int nw = nw0 + tn * 32;
    f32x4 nyv[4];
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int n = nw + 8 * rg + 4 * h;
      nyv[rg] = f32x4{1.f, 1.f, 1.f, 1.f};
      if (n + 3 < a.N) {
        nyv[rg] = *reinterpret_cast<const f32x4*>(nyb + n);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (n + j < a.N) nyv[rg][j] = nyb[n + j];
      }
      epi_consume(nyv[rg]);
    }
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const long mw = mw0 + tm * 32;
      const long mrow = mw + l31;
      const float nxm = mrow < a.M ? nxb[mrow] : 1.f;
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int n = nw + 8 * rg + 4 * h;
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float c = (a.alpha * acc[tn][tm][4 * rg + j]) / (nxm * nyv[rg][j] + 1e-6f);
          float kk = expf((c - 1.0f) * a.inv_t);
          if (a.diag_add != 0.f && mrow == n + j) kk += a.diag_add;
          v[j] = kk;
        }
        *reinterpret_cast<f32x4*>(ws + l31 * 128 + (((2 * rg + h) ^ (l31 & 7)) << 4)) = v;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i;
        const int row = c >> 3, ch = c & 7;
        const f32x4 v = *reinterpret_cast<const f32x4*>(ws + row * 128 + ((ch ^ (row & 7)) << 4));
        const long m = mw + row;
        const int n = nw + ch * 4;
        if (m >= a.M || n >= a.N) continue;
        if (a.dbg & 1) continue;
        *reinterpret_cast<f32x4*>(Cb + m * a.ldc + n) = v;  // N % 4 == 0: whole pieces only
      }
    }
  }
}

}  // namespace roma

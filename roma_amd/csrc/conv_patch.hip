// 3x3 convolution (pad 1) + bias + ReLU for the VGG layers with Cout >= 256 (conv3_1 .. conv4_4, encoders.py:17-27), 16-bit NHWC:
// the implicit GEMM of gemm8p.hip with the activation operand RESIDENT in LDS across the nine taps.
//
// Why: gemm8p's conv form stages, for every 64-deep K tile, 256 pixels x 128 B of
// activations AND 256 couts x 128 B of weights - 64 KB per K tile and CU through the L2 -> LDS path, which is what bounds the
// loop (0.39 of the MFMA peak in round 5; moving the re-reads from the fabric into L2 changed nothing: r05_v15_conv_k_order_ab.log).
// But the nine taps of one 64-channel slab read the SAME pixels, one pixel apart.  Here a workgroup owns a 2-D patch of TY x TX
// <= 256 output pixels; the patch with its one-pixel halo ((TY + 2)(TX + 2) <= 384 rows of 128 B) is staged ONCE per channel slab
// and serves all nine taps - a tap is an offset on the fragment-read addresses - while only the weight tiles stream:
// 48 + 9 x 32 KB per nine K tiles instead of 9 x 64.
//
// Everything else is gemm8p's loop, restated: 256 x 256 x 64 tiles, 8 waves = 2 wave groups one barrier apart, the k-half
// phase schedule (8 / 6 / 6 / 4 fragment reads, 8 MFMAs per phase under s_setprio), 128-byte LDS rows with the XOR chunk
// swizzle chunk ^ ((row >> 1) & 7) applied on the DMA source and on the read side, weights as the FIRST MFMA operand (a lane
// owns 4 consecutive couts of one pixel), W half-tiles permuted so that a wave's 64 columns are two 32-row blocks, inline-asm
// fragment reads whose waits name their destination registers.  K order = slab major, k = ((ci / 64) * 9 + tap) * 64 + ci % 64
// (GemmArgs::conv_korder = 1): the same products in the same order as gemm8p's conv form with that packing - bit-identical
// results (tests/test_gpu_ops.py::test_conv3x3_patch_*).
//
// LDS (all 160 KB): patch buffers 2 x 48 KB (slab s + 1 arrives during taps 0-2 of slab s), weight buffers 2 x 32 KB.
// DMA per K tile (slab s, tap t): P1 W half 1 of tile kt + 1, P2 / P3 one patch piece each (t < 3: 6 x 8 = 48 pieces of
// 1 KB), P4 W half 0 of tile kt + 2; ONE counted wait per K tile, in front of P4's first barrier: everything but the pieces
// issued in P2 .. P4 of this tile must have landed (the weights of kt + 1; a patch piece is never waited for on its own - it
// is older than the next tile's weights).  No prefetch across output tiles: the epilogue stages through the idle patch
// buffer and a workgroup barrier separates it from the next tile's prologue (2-4 % of a tile's K loop).
#include "conv_patch.h"

#include <stdio.h>
#include <stdlib.h>

#include <algorithm>

#include "gemm_device.h"

namespace roma {

static __device__ __attribute__((aligned(256))) unsigned int g_cp_zero[64];  // source of every out-of-image DMA chunk

int g_conv_patch = -1;  // roma_tuning("conv_patch", v): 1 = this kernel for slab-major VGG layers (default), 0 = gemm8p, -1 = env ROMA_CONV_PATCH

struct ConvPatchArgs {
  const bf16_t* in;    // [B, H, W, Cin]
  const bf16_t* w;     // [Cout][9 * Cin] slab major
  const float* bias;   // [Cout]
  bf16_t* out;         // [B, H, W, Cout]
  int B, H, W, Cin, Cout;
  int TY, TX;          // output patch (TY * TX <= 256), PW = TX + 2, PR = (TY + 2) * PW <= 384
  int nty, ntx, ntiles;
  int inv_tx, inv_pw;  // 65536 / d + 1: (r * inv) >> 16 == r / d for r < 512
};

#define CP_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
#define CP_DS_READ(REG, ADDR) asm volatile("ds_read_b128 %0, %1" : "=v"(REG) : "v"(ADDR))
#define CP_DS_READ_O(REG, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(REG) : "v"(ADDR), "n"(OFF))

__global__ __launch_bounds__(512, 2) void conv3x3_patch_kernel(const ConvPatchArgs a) {
  constexpr int BN = 256, BK = 64;
  constexpr int PATCH = 48 * 1024, WBUF = 32 * 1024, WOFF = 2 * PATCH;  // LDS: [patch 0][patch 1][W 0][W 1]
  constexpr int TM = 4, TN = 2;
  extern __shared__ __attribute__((aligned(1024))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int l31 = lane & 31, h = lane >> 5;
  const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem);
  const char* zsrc = reinterpret_cast<const char*>(g_cp_zero);
  const int NT = a.Cout / BN;
  const int nslab = a.Cin / BK, nkt = 9 * nslab;
  const int PW = a.TX + 2, PR = (a.TY + 2) * PW, npix = a.TY * a.TX;

  // ---- W fragment read addresses (as gemm8p): LDS row 32 wc + l31 of a half-tile, 16-byte slot (2 g + h) ^ ((l31 >> 1) & 7)
  unsigned wrd[4];
  {
    const int sw = (l31 >> 1) & 7;
#pragma unroll
    for (int g = 0; g < 4; ++g) wrd[g] = (unsigned)((32 * wc + l31) * ROWB + (((2 * g + h) ^ sw) << 4));
  }
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 fa[2][2], fw0[2], fw1[2], fw0n[2], fw1n[2];

  // waits: every register the covered reads write is a read-write operand, so no consumer can move above the wait
#define CPK_WAIT_A() \
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[1][0]), "+v"(fa[1][1])::"memory")
#define CPK_WAIT_AW(Wv)                                                                                        \
  asm volatile("s_waitcnt lgkmcnt(0)"                                                                          \
               : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(Wv[0]), "+v"(Wv[1])::"memory")
#define CPK_WAIT_AWW(Wv, Vv)                                                                                   \
  asm volatile("s_waitcnt lgkmcnt(0)"                                                                          \
               : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(Wv[0]), "+v"(Wv[1]), "+v"(Vv[0]), \
                 "+v"(Vv[1])::"memory")
  // 8 MFMAs over the four accumulators of row half MH; k order per accumulator = gemm8p's
#define CPK_MFMA(MH, W0, W1)                                                                                   \
  _Pragma("unroll") for (int gl = 0; gl < 2; ++gl) {                                                           \
    _Pragma("unroll") for (int mt = 0; mt < 2; ++mt)                                                           \
        acc[0][(MH) * 2 + mt] = mfma_h16_32x32x16(W0[gl], fa[mt][gl], acc[0][(MH) * 2 + mt]);                 \
    _Pragma("unroll") for (int mt = 0; mt < 2; ++mt)                                                           \
        acc[1][(MH) * 2 + mt] = mfma_h16_32x32x16(W1[gl], fa[mt][gl], acc[1][(MH) * 2 + mt]);                 \
  }
#define CPK_PHASE(WAIT, MF)                      \
  __builtin_amdgcn_sched_barrier(0);             \
  __builtin_amdgcn_s_barrier();                  \
  WAIT;                                          \
  __builtin_amdgcn_sched_barrier(0);             \
  __builtin_amdgcn_s_setprio(1);                 \
  MF;                                            \
  __builtin_amdgcn_s_setprio(0);                 \
  __builtin_amdgcn_sched_barrier(0);             \
  __builtin_amdgcn_s_barrier();                  \
  __builtin_amdgcn_sched_barrier(0);
  // W fragments of half NH, k-pair GP from the W buffer at LDS address WB
#define CPK_READ_W(DST, NH, GP, WB) \
  _Pragma("unroll") for (int gl = 0; gl < 2; ++gl) CP_DS_READ_O(DST[gl], (WB) + wrd[2 * (GP) + gl], (NH) * 128 * ROWB);
  // A fragments of the two 32-pixel blocks whose k-group-0 addresses are A0 / A1, k-pair GP: chunk (2 g + h) ^ sw = position of
  // k-group 0 with bits 5-6 flipped by g
#define CPK_READ_A(A0, A1, GP)                                          \
  _Pragma("unroll") for (int gl = 0; gl < 2; ++gl) {                    \
    CP_DS_READ(fa[0][gl], (A0) ^ (unsigned)((2 * (GP) + gl) << 5));     \
    CP_DS_READ(fa[1][gl], (A1) ^ (unsigned)((2 * (GP) + gl) << 5));     \
  }

  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.w), 0, (int)((long)a.Cout * 9 * a.Cin * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.in), 0, (int)((long)a.B * a.H * a.W * a.Cin * 2), 0x00020000);
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    // ---- tile -> (image, patch row, patch column, cout tile)
    const int ni = tile % NT;
    int pt = tile / NT;
    const int txi = pt % a.ntx;
    pt /= a.ntx;
    const int tyi = pt % a.nty;
    const int b = pt / a.nty;
    const int ty0 = tyi * a.TY, tx0 = txi * a.TX, n0 = ni * BN;

    // ---- DMA descriptors.  Weights: wave w stages pieces 2 w, 2 w + 1 of every 128-row half-tile (LDS row i of half hf is
    // cout n0 + (i >> 5) * 64 + hf * 32 + (i & 31)); lane -> (row r8 of the piece, 16-byte slot), slot holds chunk slot ^ ((i >> 1) & 7)
    // Round 6 (gemm8p.hip): both operands through buffer descriptors - SGPR base + 32-bit lane offset + SGPR slab / K offset;
    // halo pixels outside the image carry an offset beyond num_records and read zeros in hardware
    unsigned w_off[2][2];
    unsigned poff[6];  // patch piece wave + 8 j: byte offset of this lane's chunk from a.in (slab 0), 0x80000000 = zeros
    unsigned prow[2][2];  // patch row (at the centre tap) of this lane's pixel in block (mh, mt), as a byte offset
    {
      int ln_ = lane;  // opaque: derived values are computed here, once per tile
      asm volatile("" : "+v"(ln_));
      const int r8 = ln_ >> 3, slot = ln_ & 7;
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int i = 16 * wave + 8 * j + r8;
          const int chunk = slot ^ ((i >> 1) & 7);
          const int gn = n0 + (i >> 5) * 64 + hf * 32 + (i & 31);
          w_off[hf][j] = (unsigned)(((long)gn * (9 * a.Cin) + chunk * 8) * 2);
        }
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const int pr = (wave + 8 * j) * 8 + r8;
        const int py = (pr * a.inv_pw) >> 16, px = pr - py * PW;
        const int iy = ty0 - 1 + py, ix = tx0 - 1 + px;
        const int chunk = slot ^ ((pr >> 1) & 7);
        const bool ok = pr < PR && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        poff[j] = ok ? (unsigned)(((((long)b * a.H + iy) * a.W + ix) * a.Cin) * 2 + chunk * 16) : 0x80000000u;
      }
      const int l31_ = ln_ & 31;
#pragma unroll
      for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const int r = min(128 * wr + 64 * mh + 32 * mt + l31_, npix - 1);  // rows beyond the patch: any valid row, never stored
          const int py = (r * a.inv_tx) >> 16, px = r - py * a.TX;
          prow[mh][mt] = (unsigned)(((py + 1) * PW + px + 1) * ROWB);
        }
    }
#define CP_BL16(RS, VOFF, SOFF, DST) \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(RS, (__attribute__((address_space(3))) void*)(DST), 16, (int)(VOFF), (int)(SOFF), 0, 0)
#define CP_ISSUE_W(HF, KT, BSEL)                                                                              \
  {                                                                                                           \
    char* dst_ = smem + WOFF + (BSEL) * WBUF + (HF) * 128 * ROWB + (2 * wave) * 1024;                         \
    const int soff_ = (KT) * (BK * 2);                                                                        \
    CP_BL16(rs_w, w_off[HF][0], soff_, dst_);                                                                 \
    CP_BL16(rs_w, w_off[HF][1], soff_, dst_ + 1024);                                                          \
  }
  // patch piece J (wave + 8 J) of channel slab S into patch buffer PSEL
#define CP_ISSUE_P(J, S, PSEL) CP_BL16(rs_in, poff[J], (S) * (BK * 2), smem + (PSEL) * PATCH + (wave + 8 * (J)) * 1024);

    // ---- prologue: patch of slab 0, K tile 0 complete, W half 0 of K tile 1 under way
#pragma unroll
    for (int j = 0; j < 6; ++j) CP_ISSUE_P(j, 0, 0)
    CP_ISSUE_W(0, 0, 0) CP_ISSUE_W(1, 0, 0)
    CP_ISSUE_W(0, 1, 1)
    CP_WAIT_VM(2);
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();  // group 1 runs one barrier behind group 0 inside the K loop

    f32x16 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Round 6: the K loop by slab - the nine taps of every slab but the last as nine STEADY copies of the body with the tap a
    // constant, the last slab on the general copy (conv_patch_ktile.inc).  The single loop carried ~25 branches per K tile.
    int kt = 0, slab = 0;
    unsigned ca[4];  // fragment addresses of the current tap's blocks (conv_patch_ktile.inc); first: tap 0 of slab 0 (patch buffer 0)
    {
      const int toff0 = (-PW - 1) * ROWB;
#pragma unroll
      for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const unsigned rb_ = prow[mh][mt] + (unsigned)toff0;
          ca[2 * mh + mt] = lds0 + rb_ + ((((rb_ >> 8) & 7u) ^ (unsigned)h) << 4);
        }
    }
#define CP_STEADY 1
    for (; slab + 1 < nslab; ++slab) {
      {
#define CP_TAP 0
#include "conv_patch_ktile.inc"
#undef CP_TAP
      }
      ++kt;
      {
#define CP_TAP 1
#include "conv_patch_ktile.inc"
#undef CP_TAP
      }
      ++kt;
      {
#define CP_TAP 2
#include "conv_patch_ktile.inc"
#undef CP_TAP
      }
      ++kt;
      {
#define CP_TAP 3
#include "conv_patch_ktile.inc"
#undef CP_TAP
      }
      ++kt;
      {
#define CP_TAP 4
#include "conv_patch_ktile.inc"
#undef CP_TAP
      }
      ++kt;
      {
#define CP_TAP 5
#include "conv_patch_ktile.inc"
#undef CP_TAP
      }
      ++kt;
      {
#define CP_TAP 6
#include "conv_patch_ktile.inc"
#undef CP_TAP
      }
      ++kt;
      {
#define CP_TAP 7
#include "conv_patch_ktile.inc"
#undef CP_TAP
      }
      ++kt;
      {
#define CP_TAP 8
#include "conv_patch_ktile.inc"
#undef CP_TAP
      }
      ++kt;
    }
#undef CP_STEADY
#define CP_STEADY 0
    for (int tap = 0; kt < nkt; ++kt, ++tap) {
#define CP_TAP tap
#include "conv_patch_ktile.inc"
#undef CP_TAP
    }
#undef CP_STEADY
    if (wr == 0) __builtin_amdgcn_s_barrier();  // group 0 meets group 1's extra barrier: both groups are past every LDS read

    // ---- epilogue: bias + ReLU, staged through the patch buffer the last slab did NOT read (nobody touches it), whole 16-byte
    // pieces of a pixel's 64 couts per lane; pixel (ty0 + r / TX, tx0 + r % TX) of tile-local row r
    {
      const int lastpb = (nslab - 1) & 1;
      char* ws = smem + (lastpb ^ 1) * PATCH + wave * 4096;
      int lane_e = lane;
      asm volatile("" : "+v"(lane_e));
      const int el31 = lane_e & 31, eh = lane_e >> 5;
      constexpr int RB = TN * 64, CPR = TN * 4;
      const int nw0 = n0 + wc * 64;
      GemmArgs ga;  // (EpiCols reads bias / N only)
      ga.bias = a.bias;
      ga.N = a.Cout;
      EpiCols<TN, true, false> cols;
      cols.load(ga, nw0, eh);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = acc[tn][tm][4 * rg + j];
            v = epi_apply<ACT_RELU, true, false>(v, cols.b[tn][rg], cols.s[0][rg]);
            uint2 pk;
            pk.x = pack_bf16x2(v[0], v[1]);
            pk.y = pack_bf16x2(v[2], v[3]);
            const int ch = tn * 4 + rg;
            *reinterpret_cast<uint2*>(ws + el31 * RB + ((ch ^ epi_swz<CPR>(el31)) << 4) + 8 * eh) = pk;
          }
#pragma unroll
        for (int i = 0; i < (32 * CPR) / 64; ++i) {
          const int c = lane_e + 64 * i;
          const int row = c / CPR, ch = c - row * CPR;
          const uint4 v = *reinterpret_cast<const uint4*>(ws + row * RB + ((ch ^ epi_swz<CPR>(row)) << 4));
          const int r = 128 * wr + 32 * tm + row;
          const int py = (r * a.inv_tx) >> 16, px = r - py * a.TX;
          const int y = ty0 + py, x = tx0 + px;
          if (r < npix && y < a.H && x < a.W) {
            bf16_t* dst = a.out + (((long)b * a.H + y) * a.W + x) * a.Cout + nw0 + ch * 8;
            typedef unsigned u32x4n __attribute__((ext_vector_type(4)));
            __builtin_nontemporal_store(u32x4n{v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4n*>(dst));
          }
        }
      }
    }
    // every wave is done with its staging slice and the stores are on their way before the next tile's prologue refills LDS
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
#undef CP_ADDR
#undef CP_ISSUE_P
#undef CP_ISSUE_W
#undef CPK_READ_A
#undef CPK_READ_W
#undef CPK_PHASE
#undef CPK_MFMA
#undef CPK_WAIT_AWW
#undef CPK_WAIT_AW
#undef CPK_WAIT_A
}

// patch shape: TY x TX <= 256 pixels, (TY + 2)(TX + 2) <= 384 rows, least padding over the image (ties: the squarer patch)
static void choose_patch(int H, int W, int* ty, int* tx) {
  long best = -1;
  for (int t_y = 4; t_y <= 64; ++t_y)
    for (int t_x = 4; t_x <= 64; ++t_x) {
      if (t_y * t_x > 256 || (t_y + 2) * (t_x + 2) > 384) continue;
      const long covered = (long)((H + t_y - 1) / t_y) * ((W + t_x - 1) / t_x) * 256;  // MFMA rows spent (a tile is 256 rows whatever it holds)
      const long score = covered * 64 + (t_y > t_x ? t_y - t_x : t_x - t_y);
      if (best < 0 || score < best) {
        best = score;
        *ty = t_y;
        *tx = t_x;
      }
    }
}

bool conv_patch_supported(const GemmArgs& a) {
  return a.conv_c > 0 && a.conv_korder == 1 && a.conv_c % 64 == 0 && a.conv_c >= 128 && a.conv_c <= 512 && a.N % 256 == 0 &&
         a.in_dt == DT_BF16 && a.out_dt == DT_BF16 && a.act == ACT_RELU && a.bias && a.mode == EPI_STD && !a.scale && !a.res &&
         !a.res_bf16 && a.alpha == 1.0f && a.batch == 1 && a.batch2 == 1 && a.ldc == a.N && a.ldw == 9l * a.conv_c &&
         (reinterpret_cast<uintptr_t>(a.C) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.A) & 15) == 0;
}

// 0 = launched, 1 = not this kernel's problem, < 0 = error
int conv_patch_try_launch(const GemmArgs& g, hipStream_t stream) {
  static const int use_env = getenv("ROMA_CONV_PATCH") ? atoi(getenv("ROMA_CONV_PATCH")) : 1;
  if (!(g_conv_patch >= 0 ? g_conv_patch : use_env)) return 1;
  if (!conv_patch_supported(g)) return 1;
  const int H = g.conv_h, W = g.conv_w;
  const long hw = (long)H * W;
  if (hw <= 0 || g.M % hw != 0 || H < 4 || W < 4) return 1;
  const int B = (int)(g.M / hw);
  if ((long)g.M * g.conv_c * 2 >= (1l << 31) - (1 << 20)) return 1;  // 32-bit source offsets below 2^31 (0x80000000 = the out-of-image marker of the buffer descriptor)
  ConvPatchArgs a;
  a.in = reinterpret_cast<const bf16_t*>(g.A);
  a.w = reinterpret_cast<const bf16_t*>(g.W);
  a.bias = g.bias;
  a.out = reinterpret_cast<bf16_t*>(g.C);
  a.B = B; a.H = H; a.W = W; a.Cin = g.conv_c; a.Cout = g.N;
  choose_patch(H, W, &a.TY, &a.TX);
  a.nty = (H + a.TY - 1) / a.TY;
  a.ntx = (W + a.TX - 1) / a.TX;
  const long nt = (long)B * a.nty * a.ntx * (g.N / 256);
  if (nt >= (1l << 30)) return 1;
  a.ntiles = (int)nt;
  a.inv_tx = 65536 / a.TX + 1;
  a.inv_pw = 65536 / (a.TX + 2) + 1;
  const size_t lds = 160 * 1024;
  char pname[96];
  snprintf(pname, sizeof pname, "conv3x3_patch_kernel<" ROMA_H16_NAME ",relu>");
  ProfScope ps(pname, 2.0 * (double)g.M * g.N * g.K, "flop", stream);
  static bool attr_set[64] = {false};
  int dev = 0;
  ROMA_CHECK_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    ROMA_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_patch_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  const long gx = std::min<long>(nt, 256);
  hipLaunchKernelGGL(conv3x3_patch_kernel, dim3((unsigned)gx), dim3(512), lds, stream, a);
  ROMA_LAUNCH_CHECK();
  return 0;
}

}  // namespace roma

// Depthwise 5x5 (+folded BN, ReLU) for the WIDE ConvRefiner scales (C = 576 / 1152 / 1408; 16-bit storage) - the
// "wave-private ring" form of dwconv5x5_kernel (elementwise.hip).  romatch/models/matcher.py:106-122.
//
// What bounds the register-prefetch kernel (elementwise.hip) is not bytes and not FMAs: at 254 VGPRs it runs two waves
// per SIMD, each with ONE input row (8 x 8-byte loads per lane) in flight, i.e. 32 KiB per CU - against an HBM latency of
// 2-3 us under load that is ~4 TB/s by Little's law - and its waves issue only 42 % of their cycles
// (profiles/r02_pmc_sq_summary.json).  The LDS-DMA ring of round 1 fixed the depth but bought a workgroup barrier per
// row (the four waves shared the ring) and lost.  Here every WAVE owns its own ring:
//
//   * a wave = 64 channels (one 128-byte line per pixel) x 16 output columns x a strip of rows; lane = (4 channels,
//     4 columns).  Per input row it needs 20 pixels x 128 B = 2 560 B: three `global_load_lds_dwordx4` (16 B per lane,
//     whole cache lines, the halo columns fetched once per wave instead of once per lane);
//   * the ring holds NR = 6 rows per wave (18 KiB; 72 KiB per workgroup, two workgroups per CU) and the DMA runs five
//     rows ahead: ~100 KiB in flight per CU.  Only the issuing wave reads its slots, so the one thing that orders a read
//     behind its DMA is that wave's own counted `s_waitcnt vmcnt` - NO barrier anywhere, the waves drift freely;
//   * pixels sit in the ring at slot s(x) = x + x / 4 (every fifth 128-byte slot stays empty): the two 16-lane halves of
//     a `ds_read_b64` group read pixels 4 apart, i.e. 5 slots = an odd number of 128-byte rows apart - opposite halves of
//     the 64 banks, conflict free;
//   * the 25 x 4 tap weights of a lane stay in registers for the whole strip, the taps run column-major so that only four
//     converted columns are live: 100 weight + 80 accumulator registers + the row pipeline = ~215 VGPRs, two waves per SIMD.
//     The five rolling accumulator rows are rotated with moves.  Renaming them instead (the row loop unrolled by five) was
//     built first: hipcc's allocation then needs ~330 registers - at two waves per SIMD it spilled 100-325 of them (every
//     scratch reload is a VMEM operation that drains the counted DMA queue), and at one wave per SIMD the kernel ran
//     0.54 ms against the 0.45 ms of dwconv5x5_kernel at 16 x 216 x 216 x 576 (profiles/r03_v5_dwconv_ring.log);
//   * arithmetic and its order per accumulator are those of dwconv5x5_kernel: results are bit-identical (tests).
#include <stdlib.h>

#include <algorithm>

#include "elementwise.h"
#include "gemm.h"  // DT_*

namespace roma {

typedef __attribute__((ext_vector_type(2))) float f32x2;
#define ROMA_LDS __attribute__((address_space(3)))
typedef ROMA_LDS unsigned char lds_u8;

__device__ __attribute__((aligned(256))) unsigned int g_dwr_zero_page[64];  // source of every out-of-image / empty-slot piece

constexpr int DWR_ROWB = 3072;                   // bytes per ring row: 24 slots x 128 B (20 pixels + 4 empty slots)
constexpr int DWR_PXW = 16;                      // output columns per wave
constexpr int DWR_NR = 6;                        // ring rows per wave
constexpr int DWR_RING = 4 * DWR_NR * DWR_ROWB;  // 72 KiB per workgroup
static_assert(3 * (DWR_NR - 1) + 4 * (DWR_NR - 1) <= 63, "vmcnt is a 6-bit counter");
static_assert(2 * DWR_RING <= 160 * 1024, "two workgroups per CU");

#define ROMA_DWR_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")

__device__ __forceinline__ void dwr_glds16(const char* src, lds_u8* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (ROMA_LDS void*)lds_wave_base, 16, 0, 0);
}

// Round 6: the ring reads of input row t + 1 are issued BEFORE the 200 FMAs of row t and waited for behind them (`cr` is
// written asynchronously by the inline-asm reads: nothing may touch it between dwr_read and dwr_wait - tools/audit_asm_reads.py).
// Until then every row opened with 8 reads + s_waitcnt lgkmcnt(0): an LDS round trip per row that only one other wave of the
// SIMD could cover.
__device__ __forceinline__ void dwr_read(unsigned long long (&cr)[8], unsigned rd) {
  asm volatile(
      "ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:128\n\tds_read_b64 %2, %8 offset:256\n\t"
      "ds_read_b64 %3, %8 offset:384\n\tds_read_b64 %4, %8 offset:640\n\tds_read_b64 %5, %8 offset:768\n\t"
      "ds_read_b64 %6, %8 offset:896\n\tds_read_b64 %7, %8 offset:1024"
      : "=&v"(cr[0]), "=&v"(cr[1]), "=&v"(cr[2]), "=&v"(cr[3]), "=&v"(cr[4]), "=&v"(cr[5]), "=&v"(cr[6]), "=&v"(cr[7])
      : "v"(rd)
      : "memory");
}
__device__ __forceinline__ void dwr_wait(unsigned long long (&cr)[8]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(cr[0]), "+v"(cr[1]), "+v"(cr[2]), "+v"(cr[3]), "+v"(cr[4]), "+v"(cr[5]), "+v"(cr[6]), "+v"(cr[7])::"memory");
}

// one input row t (already in `cr`): convert, start the reads of row t + 1 (ring address rd_next) into `cr`, feed the five
// rolling output rows (acc[k] = output row t - 4 + k, tap row ky = 4 - k), finish and store row t - 4 at `orow`, rotate
__device__ __forceinline__ void dwr_row(f32x2 (&acc)[5][4][2], const f32x4 (&wreg)[25], f32x2 bias0, f32x2 bias1,
                                        unsigned long long (&cr)[8], unsigned rd_next, bool store, bf16_t* orow, long Cp, int npx) {
  f32x2 v[8][2];
#define ROMA_DWR_CVT(J)                                                  \
  {                                                                      \
    const uint32_t lo_ = (uint32_t)cr[J], hi_ = (uint32_t)(cr[J] >> 32); \
    v[J][0] = f32x2{h16_lo(lo_), h16_hi(lo_)};                           \
    v[J][1] = f32x2{h16_lo(hi_), h16_hi(hi_)};                           \
  }
  ROMA_DWR_CVT(0) ROMA_DWR_CVT(1) ROMA_DWR_CVT(2) ROMA_DWR_CVT(3) ROMA_DWR_CVT(4) ROMA_DWR_CVT(5) ROMA_DWR_CVT(6) ROMA_DWR_CVT(7)
#undef ROMA_DWR_CVT
#pragma unroll
  for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(v[j][0]), "+v"(v[j][1]));  // the conversions are done HERE: cr is free
  dwr_read(cr, rd_next);
#pragma unroll
  for (int kx = 0; kx < 5; ++kx) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const f32x4 wx = wreg[(4 - k) * 5 + kx];
      const f32x2 w0 = f32x2{wx[0], wx[1]}, w1 = f32x2{wx[2], wx[3]};
#pragma unroll
      for (int px = 0; px < 4; ++px) {
        acc[k][px][0] = v[px + kx][0] * w0 + (k == 4 && kx == 0 ? bias0 : acc[k][px][0]);  // (the new row starts from the bias:
        acc[k][px][1] = v[px + kx][1] * w1 + (k == 4 && kx == 0 ? bias1 : acc[k][px][1]);  //  no 16 v_mov per row to re-seed acc[4])
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (store) {
#pragma unroll
    for (int px = 0; px < 4; ++px) {
      if (px < npx) {  // (exec-masked: the instruction is issued for the wave as long as one lane owns column px)
        uint2 u;
        u.x = pack_relu_h16x2(acc[0][px][0][0], acc[0][px][0][1]);
        u.y = pack_relu_h16x2(acc[0][px][1][0], acc[0][px][1][1]);
        *reinterpret_cast<uint2*>(orow + (long)px * Cp) = u;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int px = 0; px < 4; ++px) {
      acc[k][px][0] = acc[k + 1][px][0];
      acc[k][px][1] = acc[k + 1][px][1];
    }
  dwr_wait(cr);
}

__global__ __launch_bounds__(256, 2) void dwconv5x5_ring_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out,
                                                                const float* __restrict__ w, const float* __restrict__ bias,
                                                                int B, int H, int W, int Cp, int SY, int nchunk, int nxg,
                                                                long ntasks) {
  constexpr int NR = DWR_NR;
  __shared__ __attribute__((aligned(1024))) unsigned char ring[DWR_RING];  // the DMA target: read with inline asm only
  // a wave's task = (image, strip, 64-channel chunk, 16-column tile), tiles fastest: the four waves of a workgroup take
  // four neighbouring column tiles (together they read whole rows of 64 + 4 pixels), and no wave is wasted when the tile
  // count is not a multiple of four (16 x 70 x 70 x 1152: 5 tiles - a workgroup of "four tiles of one chunk" idled 3 of 8).
  // Each XCD owns a contiguous band of workgroups (the row halos hit its own L2).
  const long nwg = (ntasks + 3) / 4, wg_per_xcd = (nwg + 7) / 8;
  const long lw = (long)(blockIdx.x % 8) * wg_per_xcd + blockIdx.x / 8;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long task = lw * 4 + wv;
  if (lw >= nwg || task >= ntasks) return;  // no barriers: a wave may leave on its own
  const int xg = (int)(task % nxg);
  long r = task / nxg;
  const int chunk = (int)(r % nchunk);
  r /= nchunk;
  const int yt = (H + SY - 1) / SY;
  const int ys = (int)(r % yt) * SY;
  const int b = (int)(r / yt);
  const int sy = min(SY, H - ys);
  const int T = sy + 4;  // input rows ys - 2 .. ys + sy + 1

  const int cg = lane & 15, xq = lane >> 4;
  const int c = chunk * 64 + cg * 4;

  // ---- tap weights and bias of this lane's 4 channels: registers for the whole strip
  f32x4 wreg[25];
#pragma unroll
  for (int t = 0; t < 25; ++t) wreg[t] = *reinterpret_cast<const f32x4*>(w + (long)t * Cp + c);
  const f32x4 bx = *reinterpret_cast<const f32x4*>(bias + c);
#pragma unroll
  for (int t = 0; t < 25; ++t) asm volatile("" : "+v"(wreg[t]));  // retire these loads here, keep them in registers
  const f32x2 bias0 = f32x2{bx[0], bx[1]}, bias1 = f32x2{bx[2], bx[3]};
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing of this wave in flight before the counted DMA stream starts

  const int xb = xg * DWR_PXW + xq * 4;   // first output column of this lane
  const int x0 = xg * DWR_PXW - 2;        // image column of ring pixel 0
  const int npx = min(4, W - xb);         // valid output columns of this lane (<= 0: the lane only helps with the DMA)

  // ---- DMA descriptors: piece p = 64 i + lane of a row -> ring slot p >> 3 (slot s holds pixel s - s / 5, s % 5 == 4
  // stays empty), 16-byte part p & 7 of the pixel's 128-byte channel line
  const char* zsrc = reinterpret_cast<const char*>(g_dwr_zero_page);
  const char* inb = reinterpret_cast<const char*>(in + ((long)b * H * W) * Cp) + (long)chunk * 128;
  // Round 6: the DMA goes through a buffer descriptor over this wave's slice of the image (base = image b, channel chunk): a
  // 32-bit lane offset fixed for the strip (pixel column + 16-byte part; 0x80000000 for empty slots and columns outside the image:
  // beyond num_records, the hardware returns zeros) plus the row offset in an SGPR.  No VALU in the issue, no per-row selects,
  // no 64-bit lane pointers (first version of this round: pointers carried and advanced per row; before: rebuilt per row).
  // Rows above / below the image take the marker offset for every lane behind a wave-uniform branch.
  const long pitch = (long)W * Cp * 2;  // (< 2^31: checked by the launcher)
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(inb), 0, (int)((long)H * pitch - (long)chunk * 128), 0x00020000);
  unsigned voff[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int p = 64 * i + lane, s = p >> 3, part = p & 7;
    const int x = x0 + s - s / 5;
    const bool ok = (s % 5 != 4) && x >= 0 && x < W;
    voff[i] = ok ? (unsigned)(x * Cp * 2 + part * 16) : 0x80000000u;
  }
  unsigned vout = 0x80000000u;  // every lane outside: the rows above / below the image
  asm volatile("" : "+v"(vout));
  lds_u8* const myring = (lds_u8*)ring + wv * (NR * DWR_ROWB);
#define ROMA_DWR_BL16(VOFF, SOFF, DST) \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (ROMA_LDS void*)(DST), 16, (int)(VOFF), (int)(SOFF), 0, 0)
#define ROMA_DWR_ISSUE(RROW, SLOT)                                                         \
  {                                                                                        \
    const int yy_ = ys - 2 + (RROW);                                                       \
    if ((RROW) < T && yy_ >= 0 && yy_ < H) { /* wave-uniform */                            \
      const int so_ = yy_ * (int)pitch;                                                    \
      _Pragma("unroll") for (int i = 0; i < 3; ++i) ROMA_DWR_BL16(voff[i], so_, myring + (SLOT) * DWR_ROWB + i * 1024); \
    } else {                                                                               \
      _Pragma("unroll") for (int i = 0; i < 3; ++i) ROMA_DWR_BL16(vout, 0, myring + (SLOT) * DWR_ROWB + i * 1024); \
    }                                                                                      \
  }

  f32x2 acc[5][4][2];
#pragma unroll
  for (int s5 = 0; s5 < 5; ++s5)
#pragma unroll
    for (int px = 0; px < 4; ++px) {
      acc[s5][px][0] = bias0;
      acc[s5][px][1] = bias1;
    }
  const unsigned ring_lds = (unsigned)(size_t)myring;
  const unsigned rd0 = ring_lds + (unsigned)(xq * 5 * 128 + cg * 8);
  bf16_t* const obase = out + ((long)b * H * W) * Cp + c;

  // ---- prologue: rows 0 .. NR - 2 in flight, row 0 in registers
#pragma unroll
  for (int rr = 0; rr < NR - 1; ++rr) ROMA_DWR_ISSUE(rr, rr);
  unsigned long long cr[8];
  ROMA_DWR_WAIT_VM(3 * (NR - 2));  // row 0 has landed: only rows 1 .. NR - 2 may be outstanding
  dwr_read(cr, rd0);
  dwr_wait(cr);

  int slot1 = 1;      // ring slot of input row t + 1
  int fill = NR - 1;  // ring slot the next DMA goes to (= slot of row t - 1)
  // Row t + 1 is read from the ring during row t's FMAs, so it must have landed here: at most the DMA issued AFTER it may be
  // outstanding - rows t + 2 .. t + NR - 1, 3 pieces each.  The output stores of the last iterations are younger than that DMA
  // too, but they may NOT be added to the allowance: vmcnt counts loads and stores in one counter, loads retire in order among
  // themselves and so do stores, but a store can retire before an OLDER load - with an allowance of 3 (NR - 2) + 4 k a wave
  // whose young stores were acknowledged early would pass the wait with row t + 1 still in flight and read the slot's previous
  // row (seen as 1-in-1000 one-ulp patches in the two-stream stress, profiles/r03_v20_determinism_stress.log).  With the
  // allowance equal to the number of younger LOADS the wait is exact: if row t + 1 were outstanding, so would be all
  // 3 (NR - 2) younger pieces.  (Past the strip's last row the read returns a stale slot that nobody uses.)
  // The output row pointer advances by one image row per stored row (round 6: it was rebuilt from 64-bit products per row).
  bf16_t* orow = obase + ((long)ys * W + xb) * Cp;
  const long orow_step = (long)W * Cp;
#pragma nounroll
  for (int t = 0; t < T; ++t) {
    ROMA_DWR_ISSUE(t + NR - 1, fill);
    ROMA_DWR_WAIT_VM(3 * (NR - 2));
    const bool st = t >= 4;
    dwr_row(acc, wreg, bias0, bias1, cr, rd0 + (unsigned)slot1 * DWR_ROWB, st && npx > 0, orow, (long)Cp, npx);
    if (st) orow += orow_step;
    fill = fill + 1 == NR ? 0 : fill + 1;
    slot1 = slot1 + 1 == NR ? 0 : slot1 + 1;
  }
#undef ROMA_DWR_ISSUE
  ROMA_DWR_WAIT_VM(0);  // trailing zero-page DMAs must not outlive the workgroup's LDS allocation
}

int g_dw_ring = -1;  // roma_tuning("dw_ring", v): 1 = this kernel for the large launches it is faster on (default), 2 = for every shape it takes, 0 = dwconv5x5_kernel, -1 = env ROMA_DW_RING

// 0 = launched, 1 = not this kernel's problem, < 0 = error
int dwconv5x5_ring_try_launch(const void* in, void* out, const float* w, const float* bias, int B, int H, int W, int Cp, int dt,
                              hipStream_t s) {
  static const int env = getenv("ROMA_DW_RING") ? atoi(getenv("ROMA_DW_RING")) : 1;
  if (!(g_dw_ring >= 0 ? g_dw_ring : env)) return 1;
  if (dt != DT_BF16 || Cp % 64 != 0 || Cp < 256 || H < 1 || W < 1) return 1;
  if ((long)H * W * Cp * 2 >= (1l << 31)) return 1;  // 32-bit byte offsets inside an image (buffer descriptor per image; 0x80000000 = outside)
  if ((reinterpret_cast<uintptr_t>(in) & 15) != 0 || (reinterpret_cast<uintptr_t>(out) & 7) != 0) return 1;
  if ((reinterpret_cast<uintptr_t>(w) & 15) != 0 || (reinterpret_cast<uintptr_t>(bias) & 15) != 0) return 1;
  const int nchunk = Cp / 64;
  const int nxg = (W + DWR_PXW - 1) / DWR_PXW;
  // strip height: a strip of SY rows reads SY + 4 input rows and pays ~3 rows of pipeline fill; 2048 waves are resident
  // (8 per CU) and take the tasks in rounds - pick the split of H that minimises rounds x (SY + 7)
  int SY = H;
  {
    long best = -1;
    const long per_strip = (long)B * nxg * nchunk;
    // (Rounds 3-6 capped a strip at 48 rows; without the cap the five wide shapes are 2.5 % faster in sum - 16 x 108^2 x 1152
    // 231 -> 216 us, 8 x 216^2 x 576 235 -> 224 - and none slower beyond noise: profiles/r06_v41_dwconv_strip_height.log.
    // ROMA_DWR_MAXSY: A/B.)
    static const int max_sy = getenv("ROMA_DWR_MAXSY") ? std::max(6, atoi(getenv("ROMA_DWR_MAXSY"))) : (1 << 20);
    for (int ns = (H + max_sy - 1) / max_sy; ns <= std::max(1, H / 6); ++ns) {
      const int sy = (H + ns - 1) / ns;
      const long nt = per_strip * ((H + sy - 1) / sy);
      const long cost = ((nt + 2047) / 2048) * (sy + 7);
      if (best < 0 || cost < best) {
        best = cost;
        SY = sy;
      }
    }
  }
  const long ntasks = (long)B * ((H + SY - 1) / SY) * nxg * nchunk;
  ROMA_REQUIRE(ntasks < (1l << 31), "dwconv5x5: grid too large");
  // Coarse tasks (a wave = 64 channels x 16 columns x a strip) need a problem that fills the 2048 resident waves: measured
  // per launch, ring / register-prefetch kernel (profiles/r03_v14_dwconv_ring_flat_tasks.log), B = 16 and B = 8 (one
  // sub-batch stream): 40^2 x 1408: 53.9 / 48.6 and 34.9 / 32.6 us; 70^2 x 1152: 118 / 142 and 62.7 / 59.0; 140^2 x 576:
  // 205 / 224 and 86.8 / 109; 108^2 x 1152: 220 / 258 and 112 / 137; 216^2 x 576: 407 / 463 and 204 / 240.  The crossover
  // sits between 45 M and 90 M elements.  ROMA_DW_RING=2 forces this kernel (tests, A/B).
  static const long min_elems = getenv("ROMA_DW_RING_MINELEMS") ? atol(getenv("ROMA_DW_RING_MINELEMS")) : (64l << 20);
  if ((g_dw_ring >= 0 ? g_dw_ring : env) != 2 && (long)B * H * W * Cp < min_elems) return 1;
  ProfScope ps("dwconv5x5_kernel<" ROMA_H16_NAME ">", 2.0 * (double)B * H * W * Cp * 2.0, "byte", s);
  const long nwg = (ntasks + 3) / 4;
  hipLaunchKernelGGL(dwconv5x5_ring_kernel, dim3((unsigned)(((nwg + 7) / 8) * 8)), dim3(256), 0, s, (const bf16_t*)in,
                     (bf16_t*)out, w, bias, B, H, W, Cp, SY, nchunk, nxg, ntasks);
  ROMA_LAUNCH_CHECK();
  return 0;
}

}  // namespace roma

// Go / no-go microbenchmark: the depthwise 5x5 stencil of the ConvRefiner blocks
// (matcher.py:106-122) on the matrix core instead of v_pk_fma_f32.
//
// Depthwise = one filter per channel, so the only MFMA whose blocks can be ONE channel each is the 16-block
// v_mfma_f32_4x4x4_16b_{bf16,f16}: per block D[4][4] = A[4][4] . B[4][4].  Lanes 4b .. 4b + 3 belong to block b, so
// every lane must hold 4 x-consecutive values of ONE channel.  With channels-last activations [y][x][c] that is a
// transpose; ds_read_b64_tr_b16 cannot supply it (its 16-lane group returns lane i = COLUMN i of a [4][16] block,
// i.e. 16 different channels in 16 consecutive lanes - the MFMA wants the same channel in 4 consecutive lanes, and the
// 8-byte pieces the transpose gathers are channel-contiguous because the LDS-DMA that filled the image wrote 16-byte,
// channel-contiguous chunks).  The layout that needs no transpose is "quad-interleaved":
//
//        act[y][q][c][4]   (q = storage quad = 4 consecutive pixels, c = channel; 8 bytes per (q, c))
//
// in which lane (b = channel, j) reads its operand with ONE aligned 8-byte load, straight from global memory (no LDS):
//   * K = 8 window: outputs x0 .. x0 + 3 need inputs x0 - 2 .. x0 + 5 = two quads when the input frame is shifted by 2
//     pixels against the output frame ("moving frame": a layer reads quads q, q + 1 of a row stored with offset a and
//     writes quad q of a row stored with offset a - 2; nine chained layers start at a = 18).
//   * per tap row dy two MFMAs (k halves h = 0, 1) with the Toeplitz operands A_h[i][k] = w[dy][4h + k - i] (0 <= . <= 4):
//     10 MFMAs x 8 cycles per 16 channels x 16 pixels against ~50 v_pk_fma_f32 x 5 cycles.
//   * a wave = 16 channels x 16 pixels (4 quads) x a strip of rows; the 5 live input rows are 20 VGPRs, the 10 Toeplitz
//     operands 20, the accumulator 4: ~70 VGPRs, 6-7 waves per SIMD, no LDS, no barrier, no DMA ring.
//
// Build and run (GPU box):  hipcc --offload-arch=gfx950 -O3 -o dwmfma dwmfma.hip && ./dwmfma
// Prints the operand-layout probe, max |error| against a host reference, and the time of the kernel and of a plain
// copy of the same bytes on B x H x W x C = 16 x 216 x 216 x 576 (the largest dwconv5x5 launch: 395-407 us on the
// v_pk_fma_f32 ring kernel, profiles/r03_v14_dwconv_ring_flat_tasks.log, r03_v22_exact_waits.log).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#define CK(x)                                                                     \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

static inline u16 f2bf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (u16)(u >> 16);
}
static inline float bf2f(u16 h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
__device__ inline u16 d_f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (u16)(u >> 16);
}

// ---------------------------------------------------------------- layout probe: one MFMA, every operand element tagged
__global__ void probe_kernel(float* out) {
  const int lane = threadIdx.x;
  // A element (lane, e) = lane + e / 8 exactly representable in bf16?  use small integers: lane*4+e < 256 fits 8 bits
  s16x4 a, b;
  for (int e = 0; e < 4; ++e) {
    a[e] = (short)d_f2bf((float)(lane * 4 + e));  // 0 .. 255: exact in bf16
    b[e] = (short)d_f2bf(e == 0 ? 1.f : 0.f);     // B = "pick k = 0" for every column
  }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  f32x4 d = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[lane * 4 + r] = d[r];
  // second probe: A = pick, B tagged
  for (int e = 0; e < 4; ++e) {
    b[e] = (short)d_f2bf((float)(lane * 4 + e));
    a[e] = (short)d_f2bf(e == 0 ? 1.f : 0.f);
  }
  d = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[256 + lane * 4 + r] = d[r];
}

// ---------------------------------------------------------------- the stencil
// in : [B][H + 4][WQ][C][4] bf16, rows shifted by 2 (row r holds y = r - 2; rows 0, 1, H + 2, H + 3 are zero), pixels
//      shifted by 2 (position p holds x = p - 2; positions outside the image are zero)
// out: [B][H][WQ][C][4] bf16, position p holds x = p (offset 0)
// w  : [C][25] f32 taps, scale / shift: [C] f32 (BatchNorm folded), ReLU
struct Args {
  const u16* in;
  u16* out;
  const float* w;
  const float* scale;
  const float* shift;
  int B, H, W, C, WQ, strip;
};

template <int MODE>  // 0 = full, 1 = no MFMA (loads + stores only), 2 = no stores
__global__ __launch_bounds__(256) void dwmfma_kernel(const Args a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cb = lane >> 2, j = lane & 3;  // block = channel within the group of 16; j = quad within the wave's 4
  const int ncg = a.C / 16, nxg = (a.W + 15) / 16, nst = (a.H + a.strip - 1) / a.strip;
  long task = (long)blockIdx.x * 4 + wave;
  const long ntask = (long)a.B * nst * nxg * ncg;
  if (task >= ntask) return;
  const int cg = (int)(task % ncg);
  task /= ncg;
  const int xg = (int)(task % nxg);
  task /= nxg;
  const int st = (int)(task % nst);
  const int b = (int)(task / nst);
  const int c = cg * 16 + cb;
  const int q = xg * 4 + j;                // output quad = first input quad
  const int y0 = st * a.strip, y1 = min(a.H, y0 + a.strip);
  const bool q_ok = q * 4 < a.W;           // quad holds at least one pixel of the image

  // Toeplitz operands: lane (channel, i) holds A_h[i][k], k = 0 .. 3 (here i = j: the lane's index inside its block)
  s16x4 A[5][2];
  {
    const float* wc = a.w + (long)c * 25;
    const int i = j;
#pragma unroll
    for (int dy = 0; dy < 5; ++dy)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int t = 4 * h + k - i;
          A[dy][h][k] = (short)d_f2bf((t >= 0 && t <= 4) ? wc[dy * 5 + t] : 0.f);
        }
  }
  const float sc = a.scale[c], sh = a.shift[c];

  const long in_row = (long)a.WQ * a.C * 4;  // elements per stored row
  const u16* ip = a.in + ((long)b * (a.H + 4)) * in_row + ((long)q * a.C + c) * 4;
  u16* op = a.out + ((long)b * a.H) * in_row + ((long)q * a.C + c) * 4;
  const long qstep = (long)a.C * 4;

  s16x4 r[5][2];  // input rows y - 2 .. y + 2 (stored rows y .. y + 4), two quads each
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int h = 0; h < 2; ++h) r[d][h] = *reinterpret_cast<const s16x4*>(ip + (long)(y0 + d) * in_row + h * qstep);

  for (int y = y0; y < y1; y += 5) {
#pragma unroll
    for (int u = 0; u < 5; ++u) {
      const int yy = y + u;
      if (yy < y1) {
        // newest row (stored row yy + 4) into slot (u + 4) % 5
#pragma unroll
        for (int h = 0; h < 2; ++h)
          r[(u + 4) % 5][h] = *reinterpret_cast<const s16x4*>(ip + (long)(yy + 4) * in_row + h * qstep);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (MODE != 1) {
#pragma unroll
          for (int dy = 0; dy < 5; ++dy)
#pragma unroll
            for (int h = 0; h < 2; ++h) acc = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(A[dy][h], r[(u + dy) % 5][h], acc, 0, 0, 0);
        } else {
#pragma unroll
          for (int dy = 0; dy < 5; ++dy) acc[0] += (float)r[(u + dy) % 5][0][0] + (float)r[(u + dy) % 5][1][1];
        }
        // BatchNorm + ReLU, zero outside the image (the next layer's padding), 8-byte store
        s16x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float v = fmaxf(fmaf(acc[i], sc, sh), 0.f);
          if (q * 4 + i >= a.W) v = 0.f;
          o[i] = (short)d_f2bf(v);
        }
        if (MODE != 2 && q_ok) *reinterpret_cast<s16x4*>(op + (long)yy * in_row) = o;
        if (MODE == 2) asm volatile("" ::"v"(o));
      }
    }
  }
}

__global__ void copy_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = in[i];
}

int main() {
  // ---- probe
  float* dprobe;
  CK(hipMalloc(&dprobe, 512 * 4));
  probe_kernel<<<1, 64>>>(dprobe);
  std::vector<float> hp(512);
  CK(hipMemcpy(hp.data(), dprobe, 512 * 4, hipMemcpyDeviceToHost));
  printf("probe 1 (A tagged lane*4+e, B picks k=0): D[lane][reg] for lanes 0..7\n");
  for (int l = 0; l < 8; ++l) printf("  lane %d: %g %g %g %g\n", l, hp[l * 4], hp[l * 4 + 1], hp[l * 4 + 2], hp[l * 4 + 3]);
  printf("probe 2 (B tagged, A picks k=0): lanes 0..7\n");
  for (int l = 0; l < 8; ++l) printf("  lane %d: %g %g %g %g\n", l, hp[256 + l * 4], hp[256 + l * 4 + 1], hp[256 + l * 4 + 2], hp[256 + l * 4 + 3]);

  // ---- correctness on a small problem, then timing on the large one
  for (int pass = 0; pass < 2; ++pass) {
    const int B = pass ? 16 : 2, H = pass ? 216 : 23, W = pass ? 216 : 38, C = pass ? 576 : 32;
    const int WQ = (W + 2 + 3) / 4 + 2;  // storage quads per row: 2-pixel left shift + right halo
    const long row = (long)WQ * C * 4;
    const long n_in = (long)B * (H + 4) * row, n_out = (long)B * H * row;
    std::vector<u16> hin(n_in, 0), hout(n_out, 0);
    std::vector<float> hw((long)C * 25), hs(C), hb(C);
    srand(1);
    auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    for (auto& v : hw) v = bf2f(f2bf(rnd() * 0.3f));
    for (int c = 0; c < C; ++c) {
      hs[c] = 0.5f + 0.5f * fabsf(rnd());
      hb[c] = 0.2f * rnd();
    }
    std::vector<float> img((long)B * H * W * C);
    for (auto& v : img) v = bf2f(f2bf(rnd()));
    for (int b = 0; b < B; ++b)
      for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x)
          for (int c = 0; c < C; ++c) {
            const int p = x + 2;
            hin[((long)b * (H + 4) + y + 2) * row + ((long)(p / 4) * C + c) * 4 + p % 4] = f2bf(img[(((long)b * H + y) * W + x) * C + c]);
          }
    u16 *din, *dout;
    float *dw, *ds, *db;
    CK(hipMalloc(&din, n_in * 2));
    CK(hipMalloc(&dout, n_out * 2));
    CK(hipMalloc(&dw, hw.size() * 4));
    CK(hipMalloc(&ds, C * 4));
    CK(hipMalloc(&db, C * 4));
    CK(hipMemcpy(din, hin.data(), n_in * 2, hipMemcpyHostToDevice));
    CK(hipMemset(dout, 0, n_out * 2));
    CK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(ds, hs.data(), C * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), C * 4, hipMemcpyHostToDevice));
    for (int strip : {27, 54, 108}) {
      Args a{din, dout, dw, ds, db, B, H, W, C, WQ, pass ? strip : 10};
      const long ntask = (long)B * ((H + a.strip - 1) / a.strip) * ((W + 15) / 16) * (C / 16);
      const unsigned grid = (unsigned)((ntask + 3) / 4);
      dwmfma_kernel<0><<<grid, 256>>>(a);
      CK(hipDeviceSynchronize());
      if (!pass) {
        CK(hipMemcpy(hout.data(), dout, n_out * 2, hipMemcpyDeviceToHost));
        double maxerr = 0, maxref = 0;
        for (int b = 0; b < B; ++b)
          for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x)
              for (int c = 0; c < C; ++c) {
                double s = 0;
                for (int dy = 0; dy < 5; ++dy)
                  for (int dx = 0; dx < 5; ++dx) {
                    const int yy = y + dy - 2, xx = x + dx - 2;
                    if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                    s += (double)hw[(long)c * 25 + dy * 5 + dx] * img[(((long)b * H + yy) * W + xx) * C + c];
                  }
                double ref = fmax(s * hs[c] + hb[c], 0.0);
                double got = bf2f(hout[((long)b * H + y) * row + ((long)(x / 4) * C + c) * 4 + x % 4]);
                maxerr = fmax(maxerr, fabs(got - ref));
                maxref = fmax(maxref, fabs(ref));
              }
        printf("small problem %dx%dx%dx%d: max |err| %.4g (max |ref| %.3g; bf16 output rounding ~ %.3g)\n", B, H, W, C, maxerr, maxref,
               maxref / 256);
        break;
      }
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0));
      CK(hipEventCreate(&e1));
      auto time_it = [&](auto fn) {
        fn();
        CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
          CK(hipEventRecord(e0));
          for (int it = 0; it < 10; ++it) fn();
          CK(hipEventRecord(e1));
          CK(hipEventSynchronize(e1));
          float ms;
          CK(hipEventElapsedTime(&ms, e0, e1));
          best = fminf(best, ms / 10);
        }
        return best * 1e3f;
      };
      const double bytes = 2.0 * (double)B * H * W * C * 2;  // algorithmic: read + write, 2 B per element
      const float t0 = time_it([&] { dwmfma_kernel<0><<<grid, 256>>>(a); });
      const float t1 = time_it([&] { dwmfma_kernel<1><<<grid, 256>>>(a); });
      const float t2 = time_it([&] { dwmfma_kernel<2><<<grid, 256>>>(a); });
      printf("strip %3d: full %.1f us (%.2f TB/s algorithmic), no MFMA %.1f us, no stores %.1f us; %ld wave tasks\n", strip, t0,
             bytes / t0 * 1e-6, t1, t2, ntask);
    }
    if (pass) {
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0));
      CK(hipEventCreate(&e1));
      const long n16 = n_out * 2 / 16;
      copy_kernel<<<2048, 256>>>((const uint4*)din, (uint4*)dout, n16);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      for (int it = 0; it < 10; ++it) copy_kernel<<<2048, 256>>>((const uint4*)din, (uint4*)dout, n16);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("copy of the same tensor: %.1f us (%.2f TB/s)\n", ms * 100, 2.0 * n16 * 16 / (ms / 10 * 1e-3) * 1e-12);
    }
    CK(hipFree(din));
    CK(hipFree(dout));
    CK(hipFree(dw));
    CK(hipFree(ds));
    CK(hipFree(db));
  }
  return 0;
}

// Weighted sampling WITHOUT replacement on the device - the two torch.multinomial(..., replacement=False) draws of
// RegressionMatcher.sample (romatch/models/matcher.py:615-627; TinyRoMa.sample, tiny.py:259-273).
//
// Exponential race (equivalent to sequential draws proportional to the remaining weights, and what ATen itself does for
// this case: q ~ Exp(1), top-k of w / q): key_i = E_i / w_i with E_i = -log(u_i); the k smallest keys are the sample.
// Zero weights get key = +inf and are never chosen.  u_i comes from a counter-based generator (two rounds of a 64-bit
// mixer on (seed, i)), so a draw is reproducible from its seed and needs no state.
//
// Selection of the k smallest of n keys without sorting: positive floats order like their bit patterns, so a 3-pass radix
// select (11 + 11 + 10 bits; LDS histograms, one small scan kernel per pass) finds the k-th key exactly; a final pass
// compacts every index with key < T and as many with key == T as are still missing.  n = 1.5 M, k = 40 000: ~25 us.
// The sample is then put in DRAW order (ascending race key = the order sequential draws would have produced them, which is
// what torch.multinomial returns): callers that truncate or stride the result (`matches[:N]`) get a random subset, not the
// spatially ordered one the compaction leaves.  k <= 40 000 in match(): an all-pairs rank (k^2 compares from LDS tiles) is
// ~30 us and needs no sort; above 65 536 (sample(num = 100 000) asks for k = 400 000) the same order comes from a bitonic
// network over the packed (key, index) words - see ORDER_ALLPAIRS_MAX.
#include "sampling.h"

#include <stdint.h>

namespace roma {

struct SelState {  // lives at the head of the workspace
  unsigned prefix;      // bits of the k-th key fixed so far
  unsigned remaining;   // rank of the k-th key inside the current prefix bucket (1-based)
  unsigned out_count;   // compaction cursor
  unsigned ties_left;   // how many keys == T still go out
  unsigned n_positive;  // number of finite keys (w > 0)
  unsigned pad[3];
};

__device__ __forceinline__ uint64_t mix64(uint64_t z) {  // splitmix64 finaliser
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void race_keys_kernel(const float* __restrict__ w, long n, uint64_t seed, float* __restrict__ keys,
                                                        SelState* st) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float wi = w[i];
  float key = __int_as_float(0x7f800000);  // +inf
  if (wi > 0.f) {
    const uint64_t r = mix64(mix64(seed + 0x9e3779b97f4a7c15ull * (uint64_t)(i + 1)) ^ seed);
    // (0, 1) exclusive: 23 random bits + 0.5 is exact in f32 (24 significant bits), so u is never rounded up to 1
    const float u = ((float)(r >> 41) + 0.5f) * (1.0f / 8388608.0f);
    key = -__logf(u) / wi;
    key = fminf(key, 3.0e38f);  // keep finite keys below +inf
    atomicAdd(&st->n_positive, 1u);
  }
  keys[i] = key;
}

// histogram of the `bits` bits at `shift` over the keys whose higher bits equal st->prefix (all keys in pass 0)
__global__ __launch_bounds__(256) void race_hist_kernel(const float* __restrict__ keys, long n, int shift, int bits, int pass,
                                                        const SelState* st, unsigned* __restrict__ hist) {
  __shared__ unsigned lh[2048];
  const int nb = 1 << bits;
  for (int i = threadIdx.x; i < nb; i += 256) lh[i] = 0;
  __syncthreads();
  const unsigned prefix = st->prefix;
  const int hi_shift = shift + bits;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const unsigned k = __float_as_uint(keys[i]);
    if (pass == 0 || (k >> hi_shift) == (prefix >> hi_shift)) atomicAdd(&lh[(k >> shift) & (nb - 1)], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nb; i += 256)
    if (lh[i]) atomicAdd(&hist[i], lh[i]);
}

// one workgroup: find the bucket that holds the st->remaining-th key, fix its bits, clear the histogram for the next pass
__global__ __launch_bounds__(256) void race_scan_kernel(unsigned* __restrict__ hist, int shift, int bits, int last, SelState* st) {
  __shared__ unsigned part[256];
  const int nb = 1 << bits, per = nb / 256;
  unsigned s = 0;
  for (int j = 0; j < per; ++j) s += hist[threadIdx.x * per + j];
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned rem = st->remaining, acc = 0;
    int t = 0;
    while (t < 255 && acc + part[t] < rem) acc += part[t++];
    int b = t * per;
    while (b < (t + 1) * per - 1 && acc + hist[b] < rem) acc += hist[b++];
    st->prefix |= (unsigned)b << shift;
    st->remaining = rem - acc;  // rank inside bucket b
    if (last) st->ties_left = rem - acc;  // keys equal to T that still belong to the sample
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nb; i += 256) hist[i] = 0;
}

__global__ __launch_bounds__(256) void race_compact_kernel(const float* __restrict__ keys, long n, SelState* st,
                                                           long long* __restrict__ out, long k) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const unsigned key = __float_as_uint(keys[i]), T = st->prefix;
  bool take = key < T;
  if (key == T) {  // ties at the threshold: the first `ties_left` that arrive
    const unsigned old = atomicSub(&st->ties_left, 1u);
    take = old >= 1u && old <= 0x7fffffffu;
  }
  if (take) {
    const unsigned pos = atomicAdd(&st->out_count, 1u);
    if ((long)pos < k) out[pos] = i;
  }
}

// rank of every selected element among the selected (key, index) pairs; out[rank] = index
__global__ __launch_bounds__(256) void race_order_kernel(const float* __restrict__ keys, const long long* __restrict__ sel, long k,
                                                         long long* __restrict__ out) {
  __shared__ float tk[256];
  __shared__ long long ti[256];
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long long mi = i < k ? sel[i] : 0;
  const float ki = i < k ? keys[mi] : 0.f;
  long rank = 0;
  for (long j0 = 0; j0 < k; j0 += 256) {
    const long j = j0 + threadIdx.x;
    __syncthreads();
    ti[threadIdx.x] = j < k ? sel[j] : -1;
    tk[threadIdx.x] = j < k ? keys[sel[j]] : 0.f;
    __syncthreads();
    const int m = (int)((k - j0) < 256 ? (k - j0) : 256);
    for (int t = 0; t < m; ++t) rank += (tk[t] < ki || (tk[t] == ki && ti[t] < mi)) ? 1 : 0;
  }
  if (i < k) out[rank] = mi;
}

// k > ORDER_ALLPAIRS_MAX (sample(num) beyond ~16 000 matches: k = 4 * num): k^2 compares would take seconds at k = 400 000.
// The selected (key bits, index) pairs - one 64-bit word each, ordered exactly like the all-pairs rule (key, then index) - go
// through a global-memory bitonic network instead: log2(P) (log2(P) + 1) / 2 passes over P = 2^ceil(log2 k) words (190 launches
// of ~5 us at k = 400 000; the stages that fit one workgroup's 2048 words run fused in LDS).  Same result as the all-pairs
// rank for every k; the switch is on size only.
constexpr long ORDER_ALLPAIRS_MAX = 65536;

static long order_pow2(long k) {
  long p = 2048;
  while (p < k) p <<= 1;
  return p;
}

__global__ __launch_bounds__(256) void race_pack_kernel(const float* __restrict__ keys, const long long* __restrict__ sel, long k, long P,
                                                        unsigned long long* __restrict__ comp) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  unsigned long long v = ~0ull;  // padding sorts last
  if (i < k) {
    const long long m = sel[i];
    v = ((unsigned long long)__float_as_uint(keys[m]) << 32) | (unsigned long long)(unsigned)m;  // n < 2^31
  }
  comp[i] = v;
}

// one compare-exchange pass of the network at distance j inside sorted runs of length kk (global memory)
__global__ __launch_bounds__(256) void bitonic_pass_kernel(unsigned long long* __restrict__ d, long j, long kk) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;  // one thread per pair
  const long i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
  const unsigned long long a = d[i], b = d[l];
  const bool asc = (i & kk) == 0;
  if ((a > b) == asc) {
    d[i] = b;
    d[l] = a;
  }
}

// every pass with distance j <= 1024 of the runs kk0 .. kk1 (kk0 <= kk1): 2048 consecutive words per workgroup in LDS.
// kk0 = 2, kk1 = 2048 sorts each 2048-word block; kk0 = kk1 = kk > 2048 finishes the passes j = 1024 .. 1 of run length kk.
__global__ __launch_bounds__(256) void bitonic_lds_kernel(unsigned long long* __restrict__ d, long kk0, long kk1) {
  __shared__ unsigned long long v[2048];
  const long base = (long)blockIdx.x * 2048;
  for (int t = threadIdx.x; t < 2048; t += 256) v[t] = d[base + t];
  __syncthreads();
  for (long kk = kk0; kk <= kk1; kk <<= 1) {
    for (int j = (int)(kk > 2048 ? 1024 : kk >> 1); j >= 1; j >>= 1) {
      for (int t = threadIdx.x; t < 1024; t += 256) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
        const unsigned long long a = v[i], b = v[l];
        const bool asc = ((base + i) & kk) == 0;
        if ((a > b) == asc) {
          v[i] = b;
          v[l] = a;
        }
      }
      __syncthreads();
    }
  }
  for (int t = threadIdx.x; t < 2048; t += 256) d[base + t] = v[t];
}

__global__ __launch_bounds__(256) void race_unpack_kernel(const unsigned long long* __restrict__ comp, long k, long long* __restrict__ out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < k) out[i] = (long long)(comp[i] & 0xffffffffull);
}

size_t multinomial_workspace_bytes(long n, long k) {
  size_t b = sizeof(SelState) + 2048 * sizeof(unsigned) + (size_t)n * sizeof(float) + (size_t)(k > 0 ? k : 0) * sizeof(long long) + 16;
  if (k > ORDER_ALLPAIRS_MAX) b += (size_t)order_pow2(k) * sizeof(unsigned long long) + 16;
  return b;
}

int multinomial_launch(const float* weights, long n, long k, unsigned long long seed, long long* out, void* ws, size_t ws_bytes,
                       hipStream_t s) {
  ROMA_REQUIRE(weights && out && ws && n > 0 && k > 0 && k <= n, "multinomial: bad arguments (need 0 < k <= n)");
  ROMA_REQUIRE(n < (1l << 31), "multinomial: n too large");
  ROMA_REQUIRE(ws_bytes >= multinomial_workspace_bytes(n, k), "multinomial: workspace too small (roma_op_multinomial_workspace)");
  SelState* st = reinterpret_cast<SelState*>(ws);
  unsigned* hist = reinterpret_cast<unsigned*>(st + 1);
  float* keys = reinterpret_cast<float*>(hist + 2048);
  long long* sel = reinterpret_cast<long long*>((reinterpret_cast<uintptr_t>(keys + n) + 15) & ~(uintptr_t)15);  // compaction order
  ROMA_CHECK_HIP(hipMemsetAsync(ws, 0, sizeof(SelState) + 2048 * sizeof(unsigned), s));
  const unsigned gn = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(race_keys_kernel, dim3(gn), dim3(256), 0, s, weights, n, (uint64_t)seed, keys, st);
  ROMA_LAUNCH_CHECK();
  // remaining = k (set on the device side of the stream: a 4-byte copy from a pinned-free immediate via memset is not
  // possible for arbitrary values, so a tiny kernel-less trick: hipMemcpyAsync from host stack would race with the host;
  // use hipMemsetD32Async)
  ROMA_CHECK_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(&st->remaining), (int)k, 1, s));
  const unsigned gh = (unsigned)std::min<long>((n + 255) / 256, 1024);
  const int shifts[3] = {21, 10, 0}, bits[3] = {11, 11, 10};
  for (int p = 0; p < 3; ++p) {
    hipLaunchKernelGGL(race_hist_kernel, dim3(gh), dim3(256), 0, s, keys, n, shifts[p], bits[p], p, st, hist);
    ROMA_LAUNCH_CHECK();
    hipLaunchKernelGGL(race_scan_kernel, dim3(1), dim3(256), 0, s, hist, shifts[p], bits[p], p == 2 ? 1 : 0, st);
    ROMA_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(race_compact_kernel, dim3(gn), dim3(256), 0, s, keys, n, st, sel, k);
  ROMA_LAUNCH_CHECK();
  if (k <= ORDER_ALLPAIRS_MAX) {
    hipLaunchKernelGGL(race_order_kernel, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, s, keys, sel, k, out);
    ROMA_LAUNCH_CHECK();
    return 0;
  }
  const long P = order_pow2(k);
  unsigned long long* comp = reinterpret_cast<unsigned long long*>((reinterpret_cast<uintptr_t>(sel + k) + 15) & ~(uintptr_t)15);
  hipLaunchKernelGGL(race_pack_kernel, dim3((unsigned)(P / 256)), dim3(256), 0, s, keys, sel, k, P, comp);
  ROMA_LAUNCH_CHECK();
  hipLaunchKernelGGL(bitonic_lds_kernel, dim3((unsigned)(P / 2048)), dim3(256), 0, s, comp, 2l, 2048l);
  ROMA_LAUNCH_CHECK();
  for (long kk = 4096; kk <= P; kk <<= 1) {
    for (long j = kk >> 1; j >= 2048; j >>= 1) {
      hipLaunchKernelGGL(bitonic_pass_kernel, dim3((unsigned)(P / 512)), dim3(256), 0, s, comp, j, kk);
      ROMA_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(bitonic_lds_kernel, dim3((unsigned)(P / 2048)), dim3(256), 0, s, comp, kk, kk);
    ROMA_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(race_unpack_kernel, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, s, comp, k, out);
  ROMA_LAUNCH_CHECK();
  return 0;
}

}  // namespace roma

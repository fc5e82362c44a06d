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
// The output order is arbitrary (the reference only indexes with it).
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
    const float u = ((float)(r >> 40) + 0.5f) * (1.0f / 16777216.0f);  // (0, 1): 24 random bits, never 0 or 1
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

size_t multinomial_workspace_bytes(long n) { return sizeof(SelState) + 2048 * sizeof(unsigned) + (size_t)n * sizeof(float); }

int multinomial_launch(const float* weights, long n, long k, unsigned long long seed, long long* out, void* ws, size_t ws_bytes,
                       hipStream_t s) {
  ROMA_REQUIRE(weights && out && ws && n > 0 && k > 0 && k <= n, "multinomial: bad arguments (need 0 < k <= n)");
  ROMA_REQUIRE(n < (1l << 31), "multinomial: n too large");
  ROMA_REQUIRE(ws_bytes >= multinomial_workspace_bytes(n), "multinomial: workspace too small (roma_op_multinomial_workspace)");
  SelState* st = reinterpret_cast<SelState*>(ws);
  unsigned* hist = reinterpret_cast<unsigned*>(st + 1);
  float* keys = reinterpret_cast<float*>(hist + 2048);
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
  hipLaunchKernelGGL(race_compact_kernel, dim3(gn), dim3(256), 0, s, keys, n, st, out, k);
  ROMA_LAUNCH_CHECK();
  return 0;
}

}  // namespace roma

// torch.multinomial(weights, k, replacement=False) on the device (matcher.py:615-627): see sampling.hip.
#pragma once
#include <algorithm>

#include "common.h"

namespace roma {
size_t multinomial_workspace_bytes(long n, long k);
// out: k distinct indices (int64), in draw order, drawn without replacement with probability proportional to weights
// (>= 0).  Like torch on a GPU, nobody checks that k weights are positive: if fewer are, zero-weight entries complete
// the sample (they come last).  ws: device workspace.
int multinomial_launch(const float* weights, long n, long k, unsigned long long seed, long long* out, void* ws, size_t ws_bytes,
                       hipStream_t s);
}  // namespace roma

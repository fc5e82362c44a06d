// Gaussian kernel density of the sampled matches (reference: romatch/utils/kde.py:4-12, used by
// RegressionMatcher.sample, romatch/models/matcher.py:598-629):
//     density[i] = sum_j exp(-||x_i - y_j||^2 / (2 std^2)),   y = x[::down]
// x: [n, 4] f32 (A-coordinates, B-coordinates of a match).  The reference evaluates this with a 40 000 x 40 000 fp16
// cdist (3.2 GB intermediate); here it is one all-pairs kernel with the reference points staged through LDS.
// half_inputs != 0 rounds the coordinates to fp16 first (the reference's `x.half()`); distances, exp and the row sum
// are always f32 (at least the reference's precision).
#pragma once
#include "common.h"

namespace roma {
int kde_launch(const float* x, long n, int down, float std, int half_inputs, float* density, hipStream_t s);
}  // namespace roma

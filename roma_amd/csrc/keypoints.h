// RegressionMatcher.match_keypoints (romatch/models/matcher.py:732-773) on the device:
//   1. sample_warp_at: x_A_to_B = grid_sample(warp[..., 2:4], x_A), cert_A = grid_sample(certainty, x_A)
//      (bilinear, zeros padding, align_corners=False - the same tap rule as the refiner warp);
//   2. mutual_nn: mutual nearest neighbours between x_A_to_B and x_B within max_dist, certainty above cert_th.
#pragma once
#include "common.h"

namespace roma {
int sample_warp_at_launch(const float* warp, const float* cert, int H, int W, const float* xa, long n, float* xa_to_b,
                          float* cert_a, hipStream_t s);
// match_b[i] = j (index into b) or -1.  ws_a / ws_b: 8-byte workspaces of na / nb entries.
int mutual_nn_launch(const float* a, long na, const float* b, long nb, const float* cert_a, float cert_th, float max_dist,
                     int* match_b, unsigned long long* ws_a, unsigned long long* ws_b, hipStream_t s);
// Tie-complete form = torch.nonzero of the reference's mask (matcher.py:756-762), two calls around one host read:
//   count: runs both nearest-neighbour passes, then offs[0 .. na] (int64) = exclusive prefix sums of the per-row match
//          counts, offs[na] = number of pairs;  fill: pairs[offs[na]][2] (int64: index into a, index into b), row-major.
int mutual_nn_count_launch(const float* a, long na, const float* b, long nb, const float* cert_a, float cert_th, float max_dist,
                           unsigned long long* ws_a, unsigned long long* ws_b, long long* offs, hipStream_t s);
int mutual_nn_fill_launch(const float* a, long na, const float* b, long nb, const float* cert_a, float cert_th, float max_dist,
                          const unsigned long long* ws_a, const unsigned long long* ws_b, long long* offs, long long* pairs,
                          hipStream_t s);
// conf_from_fb_consistency (matcher.py:672-699): flows [B,H,W,2] f32 -> in_th [B,H,W] f32 (0 / 1)
// visualize_warp (matcher.py:936-986): warp [H,W2,4], certainty [H,W2], images [3,im_h,im_w] f32 -> out [3,H,W2]
int visualize_warp_launch(const float* warp, const float* cert, const float* im_a, const float* im_b, int H, int W,
                          int symmetric, int im_h, int im_w, float* out, hipStream_t s);
int fb_consistency_launch(const float* flow_fwd, const float* flow_bwd, int B, int H, int W, float th_n, float* out,
                          hipStream_t s);
}  // namespace roma

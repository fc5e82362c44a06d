// Fused local-window correlation - the reference's one native operator
// (`local_corr.local_corr`, call site romatch/utils/local_correlation.py:22-35; semantics pinned
// to the in-repo torch fallback :39-74):
//   corr[b,p,k] = scale * sum_c f0[b,p,c] * bilinear_zeropad(f1[b'], coord[b,p,k])[c]
#pragma once
#include "common.h"

namespace roma {

struct LocalCorrArgs {
  const void* f0 = nullptr;    // [nimg, HW, C]  channels-last, row stride ld0
  const void* f1 = nullptr;    // [nimg, H, W, C] channels-last, pixel stride ld1
  const float* warp = nullptr; // window form: [B, HW, 2] centre (x,y) normalised;  general form: [B, HW, K, 2]
  void* out = nullptr;         // [B, HW, K] with row stride ldo (lets the op write into the refiner concat buffer)
  int B = 0, H = 0, W = 0, C = 0;
  int radius = 0;              // window form: K = (2r+1)^2
  int K = 0;                   // general form
  long ld0 = 0, ld1 = 0, ldo = 0;
  int nimg = 0, f1_shift = 0;  // image of f0 = b, image of f1 = (b + f1_shift) % nimg
  float scale = 1.f;           // 1/sqrt(C) when f0 is not pre-scaled
  int in_dt = 0, out_dt = 0;
  int nearest = 0;             // general form only: mode="nearest" of the plugin (local_correlation.py:19,30): one tap, weight 1
  // window form: device scratch for the tile work list, (2 * tiles + 4) ints with tiles = B * ceil(H/8) * ceil(W/8); nullptr =
  // allocate stream-ordered scratch for the call.  force_gather is set by the launcher (tuning switch).
  // Round 6: the queries of incoherent tiles are sorted into bins of f1 (local_corr.hip, LIST form of the tile kernel); the scratch
  // then also holds the bin counters and the sorted query list: local_corr_ws_ints(B, H, W, radius) ints in all.
  int* ws = nullptr;
  long ws_bytes = 0;
  int force_gather = 0;
  int pxmax = 0;               // set by the launcher: largest rectangle (pixels) the tile kernel's LDS stage holds
  // set by the launcher: bin geometry and the offsets (in ints) of the bin tables / the sorted query list inside ws
  int bin_ts = 0, bin_nx = 0, bin_ny = 0, ws_bins = 0, ws_qlist = 0;
};
extern int g_lc_mode;  // roma_tuning("lc_mode")
extern int g_lc_bin;   // roma_tuning("lc_bin"): 1 = incoherent tiles through the bin-sorted LIST form (default), 0 = per-query gathers

// ints of device scratch local_corr_window_launch needs for this problem (window form, tiled radii 2 / 3 / 7; 0 otherwise)
long local_corr_ws_ints(int B, int H, int W, int radius);

// Window form (integer-patch identity: all K taps share one fractional offset).
int local_corr_window_launch(const LocalCorrArgs& a, hipStream_t stream);
// General per-tap form (drop-in for the plugin signature).
int local_corr_general_launch(const LocalCorrArgs& a, hipStream_t stream);

}  // namespace roma

// Transformer blocks of the match() path as a stand-alone unit: one pre-norm block (DINOv2 and the coordinate decoder
// share it) and the whole DINOv2 ViT-L/14 forward (romatch/models/transformer/dinov2.py:192-237, encoders.py:55-66).
// The forward is also the C-ABI entry roma_vit_forward (include/roma_hip.h): the reference's timing script runs DINOv2
// in bfloat16 and everything else in binary16 (roma_models.py:183-188 vs matcher.py:46,341, encoders.py:7), and the
// 16-bit format is a property of the library BUILD - so the binary16 library hands its DINOv2 to the bfloat16 library
// through this entry (ROMA_MIXED, model.hip).
#pragma once
#include "../../include/roma_hip.h"
#include "common.h"

namespace roma {

struct VitScratch {
  void *ln = nullptr, *ao = nullptr, *hid = nullptr;     // [rows, 1024] x 2, [rows, 4096] (activation dtype)
  void *q = nullptr, *k = nullptr, *vt = nullptr;        // persistent zero-padded attention operands
};

// x (residual stream, x_dt = DT_F32 or DT_BF16) is updated in place.  rows = Bn * N tokens.
int vit_block_run(const roma_vit_block_t& w, void* x, int x_dt, long rows, int Bn, int N, int npad, int heads, int hd,
                  float eps, int act_dt, const VitScratch& s, hipStream_t st);

int vit_forward(const roma_vit_args_t& a, hipStream_t st);

}  // namespace roma

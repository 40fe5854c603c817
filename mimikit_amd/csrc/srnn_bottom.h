// Arguments of the fused SampleRNN bottom-tier kernel (see srnn_bottom.hip).
#pragma once
#include "mmk_common.h"

namespace mmk {

struct SrnnBottomArgs {
  int32_t B, H, Hm, Q, n_out, learn_temp;   // clips, hidden, MLP hidden, classes, classes + temperature column
  float min_temp, class_size;
  int32_t fs, up_slots;                     // bottom frame size (<= 16), slots of the tier above (frame_sizes[-2])
  int32_t n_steps;                          // consecutive steps of this launch (no tier above fires inside)
  const int64_t* tau_ptr; int64_t tau_off;  // first step t = *tau_ptr + tau_off
  int64_t* idx; int64_t idx_rs;             // (B, T) int64 classes, written in place
  const float* wb; const float* bb;         // framed conv weight (H, fs) row-major, bias (H)
  const float* upper;                       // (B, up_slots, H) output of the tier above
  const float* fc0_wp; const float* fc0_bias; const float* fc2_wp; const float* fc2_bias;   // packed (linear.hip)
  const float* fc0_raw; const float* fc2_raw;   // the same matrices row-major, (Hm, H) and (n_out, Hm), as bound (may be null)
  const float* a_comp; const float* b_comp; // composed mode of the one-clip kernel: (fs, Hm) = W0 wb_i and (Hm) = W0 bb + b0, or null
  const float* temperature; const float* uniforms; int64_t uni_ld, uni_off;
  float* logits_out; int64_t logits_ld;     // logits of the launch's last step
  unsigned long long* stamps;               // diagnostic: phase totals (100 MHz ticks) + launch count, or nullptr
};

bool srnn_bottom_supported(int H, int Hm, int n_out, int fs);
int launch_srnn_bottom(const SrnnBottomArgs& a, hipStream_t stream);

}  // namespace mmk

// Arguments of the fused LSTM time-step kernel (see lstm_step.hip).
#pragma once
#include "mmk_common.h"

namespace mmk {

struct LstmStepDir {
  const float* whh_wp;          // packed (linear.hip) W_hh: 4H rows (i, f, g, o), K = H
  const float* gadd;            // (M, ., 4H) precomputed W_ih x + b for this time step, row stride gadd_ld
  const float* h_in;            // (M, H) previous hidden state
  float* h_out;                 // (M, H) new hidden state (must not alias h_in: other workgroups still read it)
  float* c;                     // (M, H) cell state, updated in place
  float* y;                     // optional copy of h_out with row stride y_ld
};

struct LstmStepArgs {
  int32_t M, H, n_dir;          // rows, hidden size, directions processed by this launch (grid.z)
  int64_t gadd_ld, y_ld;
  int32_t zero_state;           // 1: h_{t-1} = c_{t-1} = 0 (first step of `lstm(x)` without a state): no recurrent product, nothing read
  LstmStepDir dir[2];
};

bool lstm_step_supported(int H);
int launch_lstm_step(const LstmStepArgs& a, hipStream_t stream);

}  // namespace mmk

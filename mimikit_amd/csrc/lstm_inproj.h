// Arguments of the input-projection kernel of a bidirectional LSTM layer (see lstm_inproj.hip).
#pragma once
#include "mmk_common.h"

namespace mmk {

struct LstmInProjDir {
  const float* wih_wp;          // packed (linear.hip) W_ih: 4H rows (i, f, g, o), K columns in k_chunks chunks of 16 (zero padded)
  const float* bias;            // 4H sums b_ih + b_hh, or null
  float* out;                   // (rows, 4H) with row stride out_ld
};

struct LstmInProjArgs {
  const float* x;               // (rows, K) with row stride x_ld; or, x_group > 0, row r at x + (r / x_group) x_group_stride + (r % x_group) x_ld:
  int64_t x_ld, out_ld;         //   the frames of a caller's (clip, frame, bin) tensor read where they lie (x_group frames per clip)
  int32_t x_group;
  int64_t x_group_stride;
  int64_t x_floats;             // floats from x that may be read (reads past them return zeros)
  int32_t rows, K, k_chunks, H;
  int32_t k_valid;              // > 0: a row holds only k_valid inputs (<= K); what lies behind them in memory is NOT the row's and is multiplied as 0
  LstmInProjDir dir[2];         // [forward, reverse]
};

bool lstm_inproj_supported(int rows, int K, int k_chunks, int H);
int launch_lstm_inproj(const LstmInProjArgs& a, int n_cu, hipStream_t stream);

}  // namespace mmk

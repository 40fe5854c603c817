// Arguments of the fused SampleRNN GRU-tier kernel (see srnn_gru.hip).
#pragma once
#include "mmk_common.h"

namespace mmk {

struct SrnnGruArgs {
  int32_t B, H, fs;                         // clips, hidden, frame size of this tier
  int32_t up_mod, div;                      // slots of the tier above per own step (0: top tier), own frame size
  float class_size;
  const int64_t* tau_ptr; int64_t tau_off;  // t = *tau_ptr + tau_off
  const int64_t* idx; int64_t idx_rs; int64_t shift;   // window idx[:, t + shift - fs : t + shift]
  const float* win_wp; const float* win_bias;   // input Linear (H x fs), packed (linear.hip), bias in row order
  const float* v_comp;                      // (G H, 16) pre-multiplied W_ih W_in, zero padded columns (fs <= 16), or nullptr
  const float* upper;                       // (B, up_mod, H) output of the tier above, or nullptr
  const float* wih_wp; const float* wih_bias; const float* whh_wp; const float* whh_bias;   // packed (linear.hip)
  int32_t w_tile_chunks;                    // K-chunks per packed tile (H/16 for separate matrices, 2 H/16 for [x | h])
  int32_t lstm;                             // 0: GRU (3 gates, separate biases) ; 1: LSTM (4 gates, summed bias in wih_bias)
  float* c;                                 // LSTM cell state (B, H), in place
  float* h_ring; int64_t h_slot_stride;     // [2][B][H]: slot (cnt & 1) is read, slot ((cnt + 1) & 1) written
  unsigned long long* h_gran;               // [B][H] granules {update number, new state}: with the up-sampler phase (ups_wp)
  int64_t* cnt; unsigned* done;             // update counter of the tier, finish ticket
  unsigned long long* stamps;               // diagnostic: phase totals (100 MHz ticks) + launch count, or nullptr
  // optional second phase: the tier's up-sampler, out[b][n] = W_up[n] . h_new[b] + bias[n]  (n < 16 ups_n_tiles),
  // behind a grid-wide barrier (the launcher only sets it when every workgroup of the grid is resident at once)
  const float* ups_wp; const float* ups_bias; int32_t ups_n_tiles; int32_t ups_n;
  float* ups_out; int64_t ups_out_ld;
  int* err;                                 // sticky error word (a barrier that timed out), or nullptr
};

bool srnn_gru_supported(int H, int fs, bool lstm);
bool srnn_gru_grid_resident(int H, int B);
int launch_srnn_gru(const SrnnGruArgs& a, hipStream_t stream);

}  // namespace mmk

// Arguments of the resident bi-LSTM kernel: all time steps of one bidirectional layer in ONE launch (see lstm_seq.hip).
#pragma once
#include "mmk_common.h"

namespace mmk {

struct LstmSeqDir {
  const float* whh_wp;          // packed (linear.hip) W_hh: 4H rows (i, f, g, o), K = H
  const float* gadd;            // precomputed W_ih x + b of ALL frames: row (clip, frame), strides gadd_ld / gadd_ts
  float* h;                     // (M, H) hidden state: read at the first step (unless zero_state), final state on return
  float* c;                     // (M, H) cell state, the same way
  float* y;                     // hidden state of every frame: row (clip, frame), strides y_ld / y_ts; null: not wanted
};

struct LstmSeqArgs {
  int32_t M, H, n_steps;        // rows (clips), hidden size, frames of the sequence (>= 2)
  int32_t zero_state;           // 1: h = c = 0 before the first frame
  int32_t rows_pad;             // rows of one state image in `xch` (>= M, the same for every launch on these buffers)
  int64_t gadd_ld, gadd_ts, y_ld, y_ts;   // strides between clips / between frames
  LstmSeqDir dir[2];            // [forward, reverse]: the reverse direction walks the frames from the last to the first
  float* xch;                   // (n_steps - 1, 2, rows_pad, H) state images the workgroups exchange, all words poisoned (0xFFFFFFFF)
  float* xch_next;              // the set the NEXT launch will use: this launch poisons it
  uint32_t* err;                // set when a workgroup gave up waiting (the outputs are then undefined)
  // The layer's output as the Seq2Seq encoder / decoder use it (s2s_lstm_v2.py:52-56, :100, :174): adjacent channels of [forward | reverse]
  // added up - column c of the H folded ones is h[2 c] + h[2 c + 1] of ONE direction - plus `res` (same rows, H columns) if given:
  float* fold;                  // (clip, frame) rows with strides y_ld / y_ts, H columns; or null
  const float* res;             // residual input, indexed like `fold`; or null
  float* pool;                  // (clip, H): the folded rows of a clip pooled over its frames (encoder, :105-113); or null
  int32_t pool_mode;            // 0 edge_sum: first + last   1 edge_mean   2 sum   3 mean
  unsigned long long* stamps;   // diagnostic build only: (16 phases, 8 waves, 8) words of workgroup (stamp_wg, 0, 0), else null
  int32_t stamp_wg;
};

bool lstm_seq_supported(int H, int M, int n_steps, int n_cu);
size_t lstm_seq_xch_floats(int H, int rows_pad, int n_steps);     // floats of ONE set
int launch_lstm_seq(const LstmSeqArgs& a, hipStream_t stream);

}  // namespace mmk

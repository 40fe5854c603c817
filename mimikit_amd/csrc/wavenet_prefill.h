// Kernels of the WaveNet warm-up-as-prefill path (see wavenet_prefill.hip).
#pragma once
#include "mmk_common.h"

namespace mmk {

struct WnPrefillArgs {
  int32_t M, N, n_tiles, k_chunks;          // rows (positions), real columns, column tiles, K-chunks of the packed matrix
  int32_t nseg;                             // 1..3 K segments, each a multiple of 16 wide
  int32_t seg_k[3];
  const float* seg[3]; int64_t seg_ld[3]; int64_t seg_batch[3];   // row 0 of clip 0, row stride, clip stride (floats)
  const float* wp; const float* bias;       // packed weights (linear.hip), bias in packed row order or nullptr
  float* out; int64_t out_ld, out_batch;
  const float* res_in; int64_t res_ld, res_batch;                 // residual epilogue only
};

// epilogue 0: gate -> out[:, N/2] ; 1: out = res_in + product + bias
int launch_wn_prefill(const WnPrefillArgs& a, int epilogue, int batch, hipStream_t stream);
int launch_wn_prefill_embed(const int64_t* idx, int64_t idx_rs, int64_t t_begin, const float* emb, int q_levels, int C, int n_pos,
                            float* out, int64_t out_batch, int batch, hipStream_t stream);
int launch_wn_prefill_scatter(const float* h, int64_t h_batch, int64_t t_begin, int64_t t_lo, int n_pos, int C, int B, int Mg, int Gc,
                              int Gn, float* rings, int64_t ring_floats_per_wg, int64_t ring_offset, int ring_mask,
                              hipStream_t stream);

}  // namespace mmk

// Arguments of the one-hand-off-per-layer persistent WaveNet kernel (see wavenet_chain.hip).
#pragma once
#include "mmk_common.h"

namespace mmk {

// One entry per ITERATION i = 0 .. L of a step: iteration i multiplies layer i's composed gate matrix (i < L) and
// layer i-1's [res ; skip] matrix (i >= 1).
struct WnChainIter {
  const float* A_wp;      // packed [2C/16 tiles][3 C/16 chunks][64][4]: K segments [tap 0 | tap 1 | tap 1 . W_res of layer i-1]
  const float* A_bias;    // packed order: b_dil + b_1x1 + tap 1 . b_res of layer i-1
  const float* B_wp;      // packed [(res + skip)/16 tiles][C/16][64][4] of layer i-1
  const float* B_bias;
  int64_t ring_offset;    // float offset of layer i's input-history ring inside a workgroup's block
  int32_t dil;            // of layer i
  int32_t ring_mask;
  int32_t prev_has_res;   // layer i-1 has residual rows
  int32_t pad_;
};

struct WnChainArgs {
  int32_t B, Gc, Gn, Mg;          // clips, clip groups, tile owners per group, clips per group (<= 4)
  int32_t L, C, C1;               // layers, channels (= skip = residual channels), conditioning channels (0 = none)
  int32_t q_levels, H1, n_classes, n_logits_pad, learn_temp;
  float min_temp;
  int32_t xcd_local;
  int64_t t0, n_steps;            // positions t0 .. t0+n_steps-1 are produced
  const WnChainIter* iters;       // L + 1 entries, device memory
  int64_t ring_floats_per_wg;
  const float* emb;               // (q_levels, C)
  int64_t* idx; int64_t idx_rs;   // (B, T) int64, written in place
  const float* condall;           // (B, cond_steps, L, 2C), see wavenet_persist.h
  int64_t cond_steps;
  const float* zeros;
  const float* fc0_wp; const float* fc0_bias; const float* fc2_wp; const float* fc2_bias;
  const float* temperature; const float* uniforms; int64_t uni_ld;
  float* logits_out; int64_t logits_ld;
  // exchange state (zeroed before every launch): gran_y and gran_h hold TWO generations each (iteration parity)
  unsigned long long *gran_h, *gran_y, *gran_skip, *gran_hid, *gran_logit, *gran_idx;
  float* h_rings;
  int32_t* err_flag;
  unsigned* xcd_count;
  unsigned long long* stamps;     // diagnostic build (MMK_WN_STAMPS=1)
};

size_t wn_chain_lds_bytes(const WnChainArgs& a);
bool wn_chain_supported(int C, int Mg, int L);
int launch_wavenet_chain(const WnChainArgs& a, hipStream_t stream);

}  // namespace mmk

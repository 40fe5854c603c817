// Arguments of the XCD-pipelined, weight-stationary persistent WaveNet kernel (see wavenet_pipe.hip).
#pragma once
#include "wavenet_chain.h"

namespace mmk {

struct WnPipeArgs {
  int32_t B, Gc, Gn, Mg;          // clips, clip groups (<= 8), tile owners per stage, clips per group (<= 4)
  int32_t L, C, C1;               // layers, channels, conditioning channels (0 = none)
  int32_t n_it;                   // iterations per stage: iteration i of a step runs on stage i / n_it (one stage per XCD)
  int32_t q_levels, H1, n_classes, n_logits_pad, learn_temp;
  float min_temp;
  int64_t t0, n_steps;
  const WnChainIter* iters;       // L + 1 entries; ring_offset = float offset inside the OWNING stage's workgroup block
  int64_t ring_floats_per_wg;     // ring block of one workgroup: its stage's layers x slots x (Gc Mg clips) x C
  const float* emb;
  int64_t* idx; int64_t idx_rs;
  const float* condall; int64_t cond_steps;
  const float* zeros;
  const float* fc0_wp; const float* fc0_bias; const float* fc2_wp; const float* fc2_bias;
  const float* temperature; const float* uniforms; int64_t uni_ld;
  float* logits_out; int64_t logits_ld;
  // Exchange state (zeroed before every launch).  Buffers written with plain stores (XCD-local hand-offs: the line lives
  // dirty in the producer XCD's L2) are NEVER shared between stages - a late write-back from one XCD must not clobber
  // what another XCD published at the same address - and buffers that cross XCDs are only written write-through (sc1).
  unsigned long long *gran_yl, *gran_hl;      // [8 stages][Gc][2 generations][16 C]: inside a stage
  unsigned long long *gran_hown;              // [8 stages][Gc][16 C]: the last layer input a stage produced, for its own history ring
  unsigned long long *gran_yx, *gran_hx;      // [Gc][2 (stage parity)][16 C]: from a stage to the next one
  unsigned long long *gran_skipfwd;           // [Gc][2][16 C]: running skip sums handed from stage to stage
  unsigned long long *gran_skip, *gran_hid, *gran_logit;   // head stage only
  unsigned long long *gran_idx;               // [Gc][16]: sampled classes, head stage -> stage 0
  float* h_rings;                 // [8 stages][Gn][ring_floats_per_wg]
  int32_t* err_flag;
  unsigned* xcd_count;
  unsigned long long* stamps;
  int32_t stamp_stage, stamp_owner, stamp_wave;   // diagnostic build: which workgroup's / wave's lane 0 records the stamps
};

size_t wn_pipe_lds_bytes(const WnPipeArgs& a);
bool wn_pipe_supported(int C, int Mg, int Gc, int L);
int wn_pipe_iters_per_stage(int L);
int launch_wavenet_pipe(const WnPipeArgs& a, hipStream_t stream);

}  // namespace mmk

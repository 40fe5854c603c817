// Arguments of the persistent WaveNet kernel (see wavenet_persist.hip).
#pragma once
#include "mmk_common.h"

namespace mmk {

struct WnLayerTab {
  int32_t dil;
  int32_t has_res;
  int32_t ring_mask;      // history ring size - 1 (power of two >= dil + 1)
  int32_t pad_;
  int64_t ring_offset;    // float offset of this layer's input-history ring inside a workgroup's block
  const float* A_wp;      // packed [2C/16 tiles][kcA][64][4]
  const float* A_bias;    // packed order or nullptr
  const float* B_wp;      // packed [(res + skip)/16 tiles][C/16][64][4]
  const float* B_bias;
};

struct WnPersistArgs {
  // geometry
  int32_t B, Gc, Gn, Mg;          // clips, clip groups, tile owners per group, clips per group (<= 16)
  int32_t L, C, S, C1;            // layers, channels, skip channels (== C), conditioning channels (0 = none)
  int32_t kcA;                    // K-chunks of one packed A row block: 2*C/16 + C1/16 (the kernel uses the first 2*C/16)
  int32_t q_levels, H1, n_classes, n_logits_pad, learn_temp;
  float min_temp;
  int32_t teacher_forced;         // warm-up: inputs come from idx[], no head
  int32_t force_tiles;            // 16-row MFMA tiles also for groups of at most 4 clips (the plan's MMK_WN_SMALL=0 switch: a parity-test mode)
  int64_t tf_end;                 // warm-up: first position that generation will consume (layers above the need line are skipped)
  int32_t xcd_local;              // 1: one clip group per XCD, hand-offs through that XCD's L2 (see .hip)
  int64_t t0, n_steps;            // positions t0 .. t0+n_steps-1 are produced (newest input = t0-1+s)
  // layers (table in device memory: too large for the 4 KiB kernel-argument segment)
  const WnLayerTab* layers;
  int64_t ring_floats_per_wg;
  // io
  const float* emb;               // (q_levels, C)
  int64_t* idx;                   // (B, T) int64, row stride idx_rs; written in place
  int64_t idx_rs;
  const float* condall;           // (B, cond_steps, L, 2C): conv_1x1_l(c[tau0 + s]) in packed gate order, no bias; C1 == 0: unused
  int64_t cond_steps;
  const float* zeros;             // a few words of zeros (address of every load that has nothing to fetch)
  // head
  const float* fc0_wp; const float* fc0_bias; const float* fc2_wp; const float* fc2_bias;
  const float* temperature; const float* uniforms; int64_t uni_ld;
  float* logits_out; int64_t logits_ld;
  // exchange state (zeroed before every launch) and private rings
  unsigned long long *gran_h, *gran_y, *gran_skip, *gran_hid, *gran_logit, *gran_idx;
  float* h_rings;                 // [Gc*Gn][ring_floats_per_wg]: each workgroup's copy of past layer inputs
  int32_t* err_flag;              // 1: hand-off timeout, 2: workgroups were not spread 8 x Gn over the XCDs
  unsigned* xcd_count;            // [8] arrivals per XCD + [1] total, zeroed before every launch
  unsigned long long* stamps;     // diagnostic build (MMK_WN_STAMPS=1): 16 phase totals of workgroup 1, 100 MHz ticks
};

size_t wn_persist_lds_bytes(const WnPersistArgs& a);
int launch_wavenet_persist(const WnPersistArgs& a, hipStream_t stream);

}  // namespace mmk

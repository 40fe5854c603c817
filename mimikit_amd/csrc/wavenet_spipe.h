// Arguments of the stage-pipeline WaveNet kernel (see wavenet_spipe.hip): one LAYER per stage, a stage = C / 32 CUs that keep the
// layer's matrices in registers for the whole launch, the clips stream through the stages ONE AT A TIME.
#pragma once
#include "mmk_common.h"

namespace mmk {

constexpr int kSpMaxLayers = 31;        // + the head stage = 32 stages = 8 XCDs x 4 stage slots
constexpr int kSpSlots = 4;             // message generations per (stage, clip): step s writes s & 3 and poisons (s + 2) & 3
constexpr unsigned kSpPoison = 0xFFFFFFFFu;   // a NaN no layer produces: "this word of the message has not arrived yet"

// the raw state_dict tensors of one layer (device pointers; any of the biases may be null)
struct WnSpRaw {
  const float* wd;      // conv_dil weight (2C, C, 2): rows [f ; g], tap 0 = delayed sample
  const float* bd;      // (2C)
  const float* w1;      // conv_1x1 weight (2C, C1) of the conditioning input, or null
  const float* b1;      // its bias (2C), or null
  const float* wr;      // conv_res weight (C, C) or null (layer without residual)
  const float* br;      // (C)
  const float* ws;      // conv_skip weight (C, C)
  const float* bs;      // (C)
};

struct WnSpipeArgs {
  int32_t B, L, C, C1;                // clips (<= kSpMaxClips), layers (<= 31), channels (256), conditioning channels (0 = none)
  int32_t learn_temp;
  float min_temp;
  int32_t Bmax;
  int64_t t0, n_steps;                // positions t0 .. t0 + n_steps - 1 are produced
  // per-stage register images (built at commit by wn_spipe_build_image)
  const float* img_chain;             // [L][C / 8 waves][44][64][4]
  const float* img_helper;            // [L][C / 8 waves][32][64][4]
  const float* cst_chain;             // [L][C / 8][64]   residual bias of the layer below
  const float* cst_helper;            // [L][C / 8][64]   gate bias (dilated + conditioning conv + tap 1 . b_res below)
  const float* head_w0;               // (128, C): fc0 . W_skip of the LAST layer, row-major
  const float* head_b0;               // (128): fc0 bias + sum over all layers of fc0 . b_skip
  const float* fc2_w;                 // (256 [+ 1], 128) row-major (the state_dict tensor)
  const float* fc2_b;
  // the launch path's history rings: [ring slots][Bmax][C], slot = position & (ring - 1)
  float* hist[kSpMaxLayers];
  int32_t ring[kSpMaxLayers];
  int32_t dil[kSpMaxLayers];
  const float* emb;                   // (256, C)
  int64_t* idx; int64_t idx_rs;
  const float* cproj; int64_t cond_steps;     // (Bmax, cond_steps, C1): the conditioning input after its LinearIO, for the block's positions
  const float* temperature; const float* uniforms; int64_t uni_ld;
  float* logits_out; int64_t logits_ld;
  // exchange state: every word 0xFFFFFFFF before every launch
  unsigned* msg;                      // [L + 1][Bmax][4][2 C]: what stage s receives: per producing wave 8 x | 8 y
  unsigned* hidmsg;                   // [L + 1][Bmax][4][128]: running hidden pre-activations of the head's first Linear, handed on INSIDE an XCD
  unsigned* hidgrp;                   // [8][Bmax][4][128]: the sum an XCD's stages have accumulated, written by its last stage, added up by the head
  unsigned* xcd_count;                // [8] arrivals per XCD (zeroed before every launch)
  int32_t* err_flag;
  unsigned long long* stamps;         // diagnostic build only
  int32_t stamp_stage;
  int32_t pair;                       // two clips per visit (wavenet_spipe_pair.inc): 1 ask for it, 0 refuse it, < 0 by the clip count
  int32_t dbg;                        // diagnostic build, timing experiments (results wrong): 1 no conditioning reads, 2 delayed rows from L2
};

constexpr int kSpMaxClips = 128;        // clips in the ring (the LDS image of the prepared gate terms is 512 B per clip)
constexpr int kSpPairMinClips = 54;     // two clips per visit from this many clips on (wavenet_spipe_pair.inc; measured on cfg 4, us per step, one / two clips per visit:
                                        // 48 clips 53.0 / 58.6, 56 clips 60.3 / 58.7, 64 clips 68 / 58.7, 96 clips 100 / 77.4, 128 clips 133 / 98.8)
// which form a launch of B clips takes: pair > 0 asks for two clips per visit, 0 refuses it, < 0 leaves it to the clip count (an even number of clips, at least 24)
inline bool wn_spipe_pair_form(int B, int pair) { return B % 2 == 0 && B >= 24 && B <= kSpMaxClips && (pair > 0 || (pair < 0 && B >= kSpPairMinClips)); }

bool wn_spipe_supported(int C, int S, int H1, int n_classes, int L, int n_cond, int cond_dim, int batch);
int64_t wn_spipe_img_chain_floats(int L, int C);
int64_t wn_spipe_img_helper_floats(int L, int C);
int64_t wn_spipe_cst_floats(int L, int C);
int64_t wn_spipe_msg_words(int L, int C, int Bmax);
int64_t wn_spipe_hidmsg_words(int L, int Bmax);
int64_t wn_spipe_hidgrp_words(int Bmax);
// commit: raw (device array of L entries), C1 = conditioning channels (0: none, <= C), f0 = the head's first Linear (128, C) and its bias
int wn_spipe_build_image(const WnSpRaw* raw_dev, int L, int C, int C1, const float* f0, const float* fb0, float* img_chain, float* img_helper,
                         float* cst_chain, float* cst_helper, float* head_w0, float* head_b0, hipStream_t stream);
int launch_wavenet_spipe(const WnSpipeArgs& a, hipStream_t stream);

}  // namespace mmk

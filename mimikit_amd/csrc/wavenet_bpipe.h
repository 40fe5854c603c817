// Arguments of the batched stage-pipeline WaveNet kernel (see wavenet_bpipe.hip): the stage pipeline's geometry (one LAYER per stage of
// 8 CUs, the layer's matrices in registers for the whole launch), but the clips travel in GROUPS OF 16 and a visit is a set of
// v_mfma_f32_16x16x4_f32 products: the large-batch regime of BASELINE config 4 (more clips per GPU than the one-clip ring serves at its beat).
#pragma once
#include "mmk_common.h"
#include "wavenet_spipe.h"

namespace mmk {

constexpr int kBpGroup = 16;                       // clips per visit: the N of the 16x16x4 product
constexpr int kBpMaxClips = 512;                   // 32 groups: as many as there are stages
constexpr int kBpMsgWords = 2 * 4096 + 2048;       // what a stage receives per group and step: x (256 x 16) | y (256 x 16) | running hidden pre-activations (128 x 16)
// A-operand images per CU of a layer stage (floats): gate tiles x [x_{s-2} | y_{s-2} | y_{s-1}] (4 chain waves x 192 registers) | gate tiles x [delayed x | c]
// (4 helper waves x 128) | residual tiles x a K half of y_{s-1} (4 x 32) | hidden-unit tile x a K quarter of y_{s-1} (4 x 16), each register 64 lanes
constexpr int kBpChainRegs = 192;
constexpr int kBpCuFloats = (4 * kBpChainRegs + 4 * 128 + 4 * 32 + 4 * 16) * 64;
constexpr int kBpCstFloats = (4 + 2) * 256;        // per CU: gate bias of the 4 tiles | residual bias of the 2 tiles, in the products' output layout

struct WnBpipeArgs {
  int32_t B, L, C1, learn_temp;       // clips (<= kBpMaxClips), layers (<= 31), conditioning channels (0 = none, <= 256), temperature column
  float min_temp;
  int32_t Bmax;
  int64_t t0, n_steps;                // positions t0 .. t0 + n_steps - 1 are produced
  const float* img;                   // [L][8][kBpCuFloats]
  const float* cst;                   // [L][8][kBpCstFloats]
  const float* head_w0;               // (128, 256): fc0 . W_skip of the LAST layer, row-major (wavenet_spipe.hip's image kernel makes it)
  const float* head_b0;               // (128): fc0 bias + sum over all layers of fc0 . b_skip
  const float* fc2_w;                 // (257, 128) row-major, padded by the plan (row 256: the temperature)
  const float* fc2_b;                 // (257 ..): -inf for classes that do not exist
  float* hist[kSpMaxLayers];          // the launch path's history rings: [ring slots][Bmax][256], slot = position & (ring - 1)
  int32_t ring[kSpMaxLayers];
  int32_t dil[kSpMaxLayers];
  const float* emb;                   // (256, 256)
  int64_t* idx; int64_t idx_rs;
  const float* cproj; int64_t cond_steps;     // (Bmax, cond_steps, C1): the conditioning input after its LinearIO, for the block's positions
  const float* temperature; const float* uniforms; int64_t uni_ld;
  float* logits_out; int64_t logits_ld;
  unsigned* msg;                      // [L + 1][groups][4][kBpMsgWords]: every word 0xFFFFFFFF before every launch
  unsigned* xcd_count;                // [8] arrivals per XCD (zeroed before every launch)
  int32_t* err_flag;
  unsigned long long* stamps;         // diagnostic build only: phase totals of stage stamp_stage's CU 0 (10 ns ticks)
  int32_t stamp_stage;
};

bool wn_bpipe_supported(int C, int S, int H1, int n_classes, int L, int n_cond, int cond_dim, int batch);
int64_t wn_bpipe_img_floats(int L);
int64_t wn_bpipe_cst_floats(int L);
int64_t wn_bpipe_msg_words(int L, int Bmax);
// commit: raw (device array of L entries, as for the stage pipeline), C1 = conditioning channels, f0 = the head's first Linear (128, 256)
int wn_bpipe_build_image(const WnSpRaw* raw_dev, int L, int C1, const float* f0, float* img, float* cst, hipStream_t stream);
int launch_wavenet_bpipe(const WnBpipeArgs& a, hipStream_t stream);

}  // namespace mmk

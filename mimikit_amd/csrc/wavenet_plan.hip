// WaveNet generate plan (host side): dilation-queue state in HBM, per-step launch
// sequence built from the fused linear kernels, hipGraph replay of step blocks.
//
// Reference algorithm: WaveNet.generate_step == WaveNet.forward over the
// rf-long window (wavenet_v2.py:276-293, :447-452) -- every step recomputes all
// rf positions.  Here each layer keeps a queue of its own past inputs
// (ring of (k-1)*d+1 slots), so a step touches every layer once:
//   z   = sum_j W[:,:,j] . h_l[tau-(k-1-j)d] + sum_c W1x1_c . cond_c[tau] + b   (:141-150)
//   y   = tanh(z_f) * sigmoid(z_g)                                               (:151)
//   skp = Wskip . y + b (+ skp) ; h_{l+1}[tau] = h_l[tau] + Wres . y + b         (:165-176)
// which is the same arithmetic on the same operands as the window form.
#include <stdlib.h>

#include "plan_util.h"
#include "wavenet_chain.h"
#include "wavenet_lpipe.h"
#include "wavenet_persist.h"
#include "wavenet_prefill.h"
#include "wavenet_spipe.h"
#include "wavenet_bpipe.h"

using namespace mmk;

// clips from which the stage pipeline's networks run in groups of 16 on the matrix pipe (wavenet_bpipe.hip).  Measured on cfg 4, us per step: a group's trip
// is 107 whatever the batch up to ~10 groups (128 clips 107 against the one-clip ring's 136, 256 clips 140 against 272 as two passes); the ring's beat is
// 1.06 per clip: 104 clips are where the two meet
constexpr int kBpipeMinClips = 105;
// Round 6: the ring with TWO clips per visit (wavenet_spipe_pair.inc) takes 128 clips in 100 us per step and 112 in 88, so up to the ring's 128 clips the groups of 16
// are left with the clip counts the pair form does not take (odd ones)
constexpr int kBpipeAlwaysClips = kSpMaxClips + 1;
static bool bpipe_by_default(int B) { return B >= kBpipeAlwaysClips || (B >= kBpipeMinClips && B % 2 != 0); }

struct WnCall {
  int M = 0;
  const void* in0 = nullptr;
  int64_t in0_rs = 0;
  const float* cond[MMK_MAX_COND] = {nullptr, nullptr, nullptr, nullptr};      // (int64 class indices where cond_q_levels[j] > 0)
  int64_t cond_rs[MMK_MAX_COND] = {0, 0, 0, 0};
  const float* temperature = nullptr;
  const float* uniforms = nullptr;
  int64_t uni_ld = 0;
  int64_t uni_off = 0;
};

// an output module beyond the first (one per target, wavenet_v2.py:240-243, :293): an MLPIO + sampler of its own geometry on the same hidden vector
struct WnHead {
  std::vector<PackedLinear> mlp;
  int q = 0, hidden = 0, learn_temp = 0;
  float min_temp = 0.f;
  float* logits = nullptr;
  int logits_ld = 0;
};

struct mmk_wavenet_plan {
  Tuning tune;                  // the config's execution switches (plan_util.h): never the environment in the product library
  mmk_wavenet_config cfg;
  Binder binder;
  bool committed = false;
  int L = 0, C = 0, S = 0, Bmax = 0, n_cond = 0;
  std::vector<int> ksz, dil, ring;
  std::vector<PackedLinear> Aff;                 // with_affine_residuals: the layers' ParametrizedLinear (3 C x C)
  float* affbuf = nullptr;                       // its (Bmax, 3 C) outputs
  std::vector<char> has_res;
  int64_t rf = 1;
  int head_in = 0;

  const float* emb = nullptr;
  const float* cond_emb[MMK_MAX_COND] = {nullptr, nullptr, nullptr, nullptr};   // class conditioning inputs: their EmbeddingIO tables as bound
  int n_tgt = 1;
  std::vector<WnHead> xheads;                    // targets 1 .. (written to cond[k - 1]); launch path
  PackedLinear in0_lin;
  std::vector<PackedLinear> cond_lin, A, Bm, mlp;

  std::vector<float*> hist;
  std::vector<float*> cbuf;
  float* ybuf = nullptr;
  float* skipbuf = nullptr;
  float* hid[2] = {nullptr, nullptr};
  float* logits = nullptr;
  int logits_ld = 0;
  int64_t* tau = nullptr;

  hipStream_t cap_stream = nullptr;
  GraphCache gc;

  // persistent-kernel mode (wavenet_persist.hip): chosen at create time when the geometry allows it
  bool persistent = false;
  int Gc = 0, Gn = 0, Mg = 0, C1 = 0, n_logits_pad = 0;
  static constexpr int kCondBlock = 1024;       // positions whose conditioning is projected per launch
  std::vector<int64_t> ring_offset;  // per layer: float offset of its history ring inside a workgroup's block
  std::vector<int> ring_mask;
  int64_t ring_floats_per_wg = 0;
  PackedLinear cond_all;             // rows = L x 2C gate rows of every layer's conditioning 1x1 conv, K = C1
  WnLayerTab* layer_tab = nullptr;
  unsigned long long *gran_h = nullptr, *gran_y = nullptr, *gran_skip = nullptr, *gran_hid = nullptr,
                     *gran_logit = nullptr, *gran_idx = nullptr;
  int64_t gran_words = 0;       // contiguous block [gran_h .. gran_idx] + err word, zeroed before every launch
  float* h_rings = nullptr;     // per workgroup: past inputs of every layer (the delayed tap reads them)
  float* cproj = nullptr;       // (Bmax, kCondBlock, C1) conditioning after its LinearIO
  float* cpad = nullptr;        // the block's (or the prompt's) conditioning rows of all clips, compact, every row padded to whole 16-float chunks:
  int64_t cpad_rows = 0;        // what the tiled GEMM multiplies (513 bins per row as they arrive cannot be read 16 bytes at a time)
  float* condall = nullptr;     // (Bmax, kCondBlock, L, 2C) every layer's conditioning product, packed gate order
  float* zero_pad = nullptr;    // 64 floats that stay zero
  // warm-up as a prefill (wavenet_prefill.hip): two ping-pong layer inputs, the gated output, the projected
  // conditioning, each (Bmax, pf_P, .) for a prompt window of up to pf_P positions
  int64_t pf_P = 0;
  float *pf_h[2] = {nullptr, nullptr}, *pf_y = nullptr, *pf_c = nullptr;
  int32_t* err_flag = nullptr;
  unsigned* xcd_count = nullptr;
  bool xcd_local = false;       // one clip group per XCD, hand-offs through the XCD's L2 (verified in-kernel)
  // one hand-off per layer (wavenet_chain.hip): gate matrices with the K segments [tap 0 | tap 1 | tap 1 . W_res of the
  // layer below], pre-multiplied at commit
  bool chain = false;
  // four workgroups per clip that own whole layers (wavenet_lpipe.hip): small networks (C = S = 64, H1 = 128, 256 classes)
  bool lpipe = false;
  unsigned long long *lp_xg = nullptr, *lp_cg = nullptr;
  int64_t lp_gran_words = 0;
  std::vector<PackedLinear> Ac;
  WnChainIter* iter_tab = nullptr;
  float* compose_scratch = nullptr;   // (2C, C) product + 2C bias terms of one layer
  // one layer per stage of 8 CUs, clips streamed through one at a time (wavenet_spipe.hip): C = 256, <= 31 layers, <= 32 clips
  bool spipe = false;
  int last_pair = 0;           // the last stage-pipeline launch took two clips per visit (wavenet_spipe_pair.inc)
  // the same stages, the clips in groups of 16 on the matrix pipe (wavenet_bpipe.hip): the stage pipeline's large batches, <= 512 clips.  A
  // sub-mode of `spipe` (prefill into the launch path's rings, padded head, redo path are shared)
  bool bpipe = false;
  float *bp_img = nullptr, *bp_cst = nullptr;
  unsigned* bp_msg = nullptr;
  float *sp_img_chain = nullptr, *sp_img_helper = nullptr, *sp_cst_chain = nullptr, *sp_cst_helper = nullptr, *sp_head_w0 = nullptr, *sp_head_b0 = nullptr;
  unsigned *sp_msg = nullptr, *sp_hidmsg = nullptr, *sp_hidgrp = nullptr;
  WnSpRaw* sp_raw = nullptr;
  const float *sp_fc2_w = nullptr, *sp_fc2_b = nullptr;
  // the stage pipeline's head is 128 hidden units x 256 classes and its helpers multiply ONE conditioning row: a narrower head is padded
  // (zero rows / columns, -inf bias for the classes that do not exist), two conditioning inputs' projections and 1x1 matrices laid side by side
  float *sp_f0p = nullptr, *sp_fb0p = nullptr, *sp_fc2p = nullptr, *sp_fc2bp = nullptr, *sp_w1cat = nullptr, *sp_b1cat = nullptr, *sp_logits = nullptr;
  static constexpr int kSpH1 = 128, kSpQ = 256, kSpLogitsLd = 260;
  PackedLinear lp_mlp0, lp_mlp1;      // the layer pipeline's copy of the same padded head, in the packed layout it reads

  void layout_persistent(Carver& c) {
    layer_tab = c.take<WnLayerTab>(L);
    const int gq = spipe ? 0 : Gc;   // (the stage pipeline keeps its messages in blocks of its own; only the XCD counters + error word here)
    const int64_t n_h = (int64_t)gq * 2 * 16 * C, n_y = n_h, n_s = (int64_t)gq * 16 * S,   // y, h: two generations each
                  n_hid = (int64_t)gq * 16 * cfg.mlp_hidden, n_l = (int64_t)gq * 16 * n_logits_pad, n_i = (int64_t)gq * 16;
    gran_words = n_h + n_y + n_s + n_hid + n_l + n_i + 8 + 2;   // + XCD registration counters + sticky error word
    unsigned long long* base = c.take<unsigned long long>(gran_words);
    gran_h = base;
    gran_y = gran_h + (base ? n_h : 0);
    gran_skip = gran_y + (base ? n_y : 0);
    gran_hid = gran_skip + (base ? n_s : 0);
    gran_logit = gran_hid + (base ? n_hid : 0);
    gran_idx = gran_logit + (base ? n_l : 0);
    xcd_count = reinterpret_cast<unsigned*>(gran_idx + (base ? n_i : 0));
    err_flag = reinterpret_cast<int32_t*>(gran_idx + (base ? n_i + 8 : 0));
    if (lpipe) {
      sp_f0p = c.take<float>((int64_t)kSpH1 * C);
      sp_fb0p = c.take<float>(kSpH1);
      sp_fc2p = c.take<float>((int64_t)(kSpQ + 1) * kSpH1);
      sp_fc2bp = c.take<float>(kSpLogitsLd);
      sp_logits = c.take<float>((int64_t)Bmax * kSpLogitsLd);
      lp_mlp0.set_geometry(kSpH1, {C});
      lp_mlp1.set_geometry(kSpQ + 1, {kSpH1});
      lp_mlp0.carve(c, true);
      lp_mlp1.carve(c, true);
      lp_gran_words = (int64_t)(kLpStages + 1) * Bmax * 128 + (int64_t)Bmax * 16 + (int64_t)Bmax * 2;   // x | skip, class (a line per clip), XCC ids
      lp_xg = c.take<unsigned long long>(lp_gran_words);
      lp_cg = lp_xg + (lp_xg ? (int64_t)(kLpStages + 1) * Bmax * 128 : 0);
    }
    if (spipe) {
      sp_img_chain = c.take<float>(wn_spipe_img_chain_floats(L, C));
      sp_img_helper = c.take<float>(wn_spipe_img_helper_floats(L, C));
      sp_cst_chain = c.take<float>(wn_spipe_cst_floats(L, C));
      sp_cst_helper = c.take<float>(wn_spipe_cst_floats(L, C));
      sp_head_w0 = c.take<float>((int64_t)kSpH1 * C);
      sp_head_b0 = c.take<float>(kSpH1);
      sp_f0p = c.take<float>((int64_t)kSpH1 * C);
      sp_fb0p = c.take<float>(kSpH1);
      sp_fc2p = c.take<float>((int64_t)(kSpQ + 1) * kSpH1);
      sp_fc2bp = c.take<float>(kSpLogitsLd);
      sp_logits = c.take<float>((int64_t)Bmax * kSpLogitsLd);
      if (n_cond == 2) {
        sp_w1cat = c.take<float>((int64_t)L * 2 * C * C1);
        sp_b1cat = c.take<float>((int64_t)L * 2 * C);
      }
      if (bpipe) {
        bp_img = c.take<float>(wn_bpipe_img_floats(L));
        bp_cst = c.take<float>(wn_bpipe_cst_floats(L));
        bp_msg = c.take<unsigned>(wn_bpipe_msg_words(L, Bmax));
      } else {
        sp_msg = c.take<unsigned>(wn_spipe_msg_words(L, C, Bmax) + wn_spipe_hidmsg_words(L, Bmax) + wn_spipe_hidgrp_words(Bmax));   // one block: poisoned by one memset
        sp_hidmsg = sp_msg + (sp_msg ? wn_spipe_msg_words(L, C, Bmax) : 0);
        sp_hidgrp = sp_hidmsg + (sp_msg ? wn_spipe_hidmsg_words(L, Bmax) : 0);
      }
      sp_raw = c.take<WnSpRaw>(L);
    }
    h_rings = spipe ? nullptr : c.take<float>((int64_t)Gc * Gn * ring_floats_per_wg);
    cproj = C1 > 0 ? c.take<float>((int64_t)Bmax * kCondBlock * C1) : nullptr;
    if (C1 > 0) {
      int kmax = 16;
      for (int j = 0; j < n_cond; ++j) kmax = (int)round_up(cfg.cond_in_dim[j], 16) > kmax ? (int)round_up(cfg.cond_in_dim[j], 16) : kmax;
      const int64_t prompt_rows = round_up(rf, 32);      // (= pf_P, set further down)
      cpad_rows = (int64_t)Bmax * (prompt_rows > kCondBlock ? prompt_rows : kCondBlock);
      cpad = c.take<float>(cpad_rows * kmax);
    }
    // (the stage pipeline multiplies the layers' conditioning products itself, from cproj)
    condall = (C1 > 0 && !spipe) ? c.take<float>((int64_t)Bmax * kCondBlock * L * 2 * C) : nullptr;
    if (C1 > 0 && !spipe) cond_all.carve(c, false);
    if (chain) {
      for (auto& pl : Ac) pl.carve(c, true);
      iter_tab = c.take<WnChainIter>(L + 1);
      compose_scratch = c.take<float>((int64_t)2 * C * C + 2 * C);
    }
    zero_pad = c.take<float>(64);
    pf_P = round_up(rf, 32);
    pf_h[0] = c.take<float>((int64_t)Bmax * pf_P * C);
    pf_h[1] = c.take<float>((int64_t)Bmax * pf_P * C);
    pf_y = c.take<float>((int64_t)Bmax * pf_P * C);
    pf_c = C1 > 0 ? c.take<float>((int64_t)Bmax * pf_P * C1) : nullptr;
  }

  void layout(Carver& c) {
    const bool bias = cfg.bias != 0;
    if (cfg.q_levels == 0) in0_lin.carve(c, true);
    for (auto& p : cond_lin) p.carve(c, true);
    for (auto& p : A) p.carve(c, bias);
    for (auto& p : Aff) p.carve(c, bias);
    affbuf = Aff.empty() ? nullptr : c.take<float>((int64_t)Bmax * 3 * C);
    for (auto& p : Bm)
      if (p.n_tiles > 0) p.carve(c, bias);
    for (auto& p : mlp) p.carve(c, true);
    int hmax = cfg.mlp_hidden > 0 ? cfg.mlp_hidden : 1;
    for (auto& h : xheads) {
      for (auto& m : h.mlp) m.carve(c, true);
      h.logits_ld = (int)round_up(h.q + (h.learn_temp ? 1 : 0), 4);
      h.logits = c.take<float>((int64_t)Bmax * h.logits_ld);
      hmax = h.hidden > hmax ? h.hidden : hmax;
    }
    hist.resize(L + 1);
    for (int l = 0; l < L; ++l) hist[l] = c.take<float>((int64_t)ring[l] * Bmax * C);
    hist[L] = c.take<float>((int64_t)Bmax * C);  // sink for the last layer's dilated output
    cbuf.resize(n_cond);
    for (int j = 0; j < n_cond; ++j) cbuf[j] = c.take<float>((int64_t)Bmax * cfg.cond_dim[j]);
    ybuf = c.take<float>((int64_t)Bmax * C);
    skipbuf = c.take<float>((int64_t)Bmax * (S > 0 ? S : 1));
    hid[0] = c.take<float>((int64_t)Bmax * hmax);
    hid[1] = c.take<float>((int64_t)Bmax * hmax);
    logits_ld = (int)round_up(cfg.out_dim + (cfg.learn_temp ? 1 : 0), 4);
    logits = c.take<float>((int64_t)Bmax * logits_ld);
    tau = c.take<int64_t>(8 + 256 + 2048 + 256 + 8);   // [0]: the launch path's position; [8 ..]: stamps of the diagnostic builds
    if (persistent) layout_persistent(c);
  }

  Addr hist_slot(int l, int offset) const {
    if (l >= L) return addr_static(hist[L]);
    return addr_ring(hist[l], (int64_t)Bmax * C, offset, ring[l]);
  }
  // where layer l's gated output y lives
  Addr y_addr(int l) const {
    if (has_res[l] || l + 1 >= L) return addr_static(ybuf);
    return hist_slot(l + 1, 0);
  }
};

// out[n][k] = sum_c A[n][c] R[c][k] and out_bias[n] = sum_c A[n][c] r[c] for n < N, with A addressed as a[n a_rs + c a_cs]
// and R (C x C), r (C) a 1x1 convolution of the layer below.  Accumulated in fp64 and rounded once: the pre-multiplied
// matrix is as close to the exact product as fp32 allows.  Used for
//   * tap 1 of a k = 2 dilated convolution (2C rows, a = wd + 1, a_rs = 2C, a_cs = 2) times the residual convolution, and
//   * the head's first Linear (H1 rows) times a layer's skip convolution (the stage pipeline).
__global__ __launch_bounds__(256) void compose_kernel(const float* __restrict__ a, int64_t a_rs, int64_t a_cs, int N,
                                                     const float* __restrict__ wr, const float* __restrict__ br,
                                                     float* __restrict__ out, float* __restrict__ out_bias, int C) {
  const int64_t total = (int64_t)N * C;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total + N; idx += (int64_t)gridDim.x * blockDim.x) {
    if (idx < total) {
      const int n = (int)(idx / C), k = (int)(idx % C);
      double acc = 0.0;
      for (int c = 0; c < C; ++c) acc += (double)a[n * a_rs + c * a_cs] * (double)wr[(int64_t)c * C + k];
      out[idx] = (float)acc;
    } else {
      const int n = (int)(idx - total);
      double acc = 0.0;
      if (br)
        for (int c = 0; c < C; ++c) acc += (double)a[n * a_rs + c * a_cs] * (double)br[c];
      out_bias[n] = (float)acc;
    }
  }
}

static int derive(mmk_wavenet_plan* p) {
  const mmk_wavenet_config& c = p->cfg;
  if (c.n_layers < 1 || c.n_layers > MMK_MAX_LAYERS) return fail(MMK_ERR_INVALID, "wavenet: n_layers=%d out of range", c.n_layers);
  if (c.dim_dilated < 1 || c.max_batch < 1) return fail(MMK_ERR_INVALID, "wavenet: dim_dilated / max_batch must be positive");
  if (c.n_cond < 0 || c.n_cond > MMK_MAX_COND) return fail(MMK_ERR_INVALID, "wavenet: n_cond=%d out of range", c.n_cond);
  if (c.q_levels == 0 && c.in_dim < 1) return fail(MMK_ERR_INVALID, "wavenet: in_dim required when q_levels == 0");
  if (c.head_kind == 0 && (c.mlp_hidden < 1 || c.mlp_n_hidden < 0 || c.mlp_n_hidden > MMK_MAX_MLP_HIDDEN))
    return fail(MMK_ERR_INVALID, "wavenet: bad MLP head geometry");
  if (c.head_kind < 0 || c.head_kind > 2) return fail(MMK_ERR_INVALID, "wavenet: head_kind %d unknown", c.head_kind);
  if (c.exec_mode < 0 || c.exec_mode > 1) return fail(MMK_ERR_INVALID, "wavenet: exec_mode %d unknown", c.exec_mode);
  if (c.head_kind != 0 && c.q_levels != 0) return fail(MMK_ERR_UNSUPPORTED, "wavenet: linear head needs a continuous input 0");
  if (c.head_kind != 0 && c.out_dim != c.in_dim) return fail(MMK_ERR_UNSUPPORTED, "wavenet: linear head out_dim must equal in_dim");
  p->L = c.n_layers;
  p->C = c.dim_dilated;
  p->S = c.skips_dim;
  p->Bmax = c.max_batch;
  p->n_cond = c.n_cond;
  p->ksz.assign(c.kernel_size, c.kernel_size + p->L);
  p->dil.assign(c.dilation, c.dilation + p->L);
  p->ring.resize(p->L);
  p->has_res.resize(p->L);
  p->rf = 1;
  for (int l = 0; l < p->L; ++l) {
    if (p->ksz[l] < 1 || p->dil[l] < 1) return fail(MMK_ERR_INVALID, "wavenet: layer %d has kernel %d dilation %d", l, p->ksz[l], p->dil[l]);
    if (p->ksz[l] + p->n_cond > kMaxSeg)
      return fail(MMK_ERR_UNSUPPORTED, "wavenet: kernel_size + n_cond = %d exceeds %d K-segments", p->ksz[l] + p->n_cond, kMaxSeg);
    const int cause = (p->ksz[l] - 1) * p->dil[l];  // WNLayer.cause, wavenet_v2.py:74
    int ring = 1;
    while (ring < cause + 1) ring <<= 1;  // power-of-two queue: slot = position & (ring - 1)
    p->ring[l] = ring;
    p->rf += cause;
    // has_residuals (:78) with input_dim == dims_dilated[0]; last layer built with residuals_dim=None (:216)
    p->has_res[l] = c.residuals_dim != 0 && c.residuals_dim == p->C && (c.res_explicit ? c.layer_has_res[l] != 0 : l != p->L - 1);
  }
  p->head_in = p->S > 0 ? p->S : p->C;

  // geometry of the packed matrices
  if (c.q_levels == 0) p->in0_lin.set_geometry(p->C, {c.in_dim});
  p->cond_lin.resize(p->n_cond);
  for (int j = 0; j < p->n_cond; ++j) p->cond_lin[j].set_geometry(c.cond_dim[j], {c.cond_q_levels[j] > 0 ? 1 : c.cond_in_dim[j]});   // (a class input has its table instead)
  p->n_tgt = c.n_targets > 1 ? c.n_targets : 1;
  if (p->n_tgt > MMK_MAX_STREAMS || p->n_tgt > 1 + p->n_cond)
    return fail(MMK_ERR_UNSUPPORTED, "wavenet: %d targets for %d inputs (the loop writes output k into input k)", p->n_tgt, 1 + p->n_cond);
  bool multi = p->n_tgt > 1;
  for (int j = 0; j < p->n_cond; ++j) multi = multi || c.cond_q_levels[j] > 0;
  p->xheads.clear();
  for (int k = 1; k < p->n_tgt; ++k) {
    WnHead h;
    h.q = c.x_out_dim[k]; h.hidden = c.x_mlp_hidden[k]; h.learn_temp = c.x_learn_temp[k]; h.min_temp = c.x_min_temp[k];
    const int nh = c.x_mlp_n_hidden[k];
    if (h.q < 2 || h.hidden < 1 || nh < 0 || nh > MMK_MAX_MLP_HIDDEN) return fail(MMK_ERR_INVALID, "wavenet: bad MLP head geometry of target %d", k);
    if (c.cond_q_levels[k - 1] < h.q)
      return fail(MMK_ERR_INVALID, "wavenet: target %d draws from %d classes, but input %d is %s", k, h.q, k,
                  c.cond_q_levels[k - 1] > 0 ? "a stream of fewer classes" : "not a class stream");
    PackedLinear f0;
    f0.set_geometry(h.hidden, {p->S > 0 ? p->S : p->C});
    h.mlp.push_back(f0);
    for (int i = 0; i < nh; ++i) {
      PackedLinear m;
      m.set_geometry(h.hidden, {h.hidden});
      h.mlp.push_back(m);
    }
    PackedLinear out;
    out.set_geometry(h.q + (h.learn_temp ? 1 : 0), {h.hidden});
    h.mlp.push_back(out);
    p->xheads.push_back(h);
  }
  p->A.resize(p->L);
  p->Bm.resize(p->L);
  p->Aff.clear();
  if (c.with_affine_residuals) {
    p->Aff.resize(p->L);
    for (auto& a : p->Aff) a.set_geometry(3 * p->C, {p->C});
  }
  for (int l = 0; l < p->L; ++l) {
    std::vector<int> ks;
    for (int j = 0; j < p->ksz[l]; ++j) ks.push_back(p->C);
    for (int j = 0; j < p->n_cond; ++j) ks.push_back(c.cond_dim[j]);
    p->A[l].set_geometry(c.gated ? 2 * p->C : p->C, ks);
    const int n_res_pad = p->has_res[l] ? (int)round_up(p->C, 16) : 0;
    if (p->has_res[l] || p->S > 0)
      p->Bm[l].set_geometry(n_res_pad + p->S, {p->C});
    else
      p->Bm[l] = PackedLinear();
  }
  p->mlp.clear();
  if (c.head_kind == 0) {
    PackedLinear first;
    first.set_geometry(c.mlp_hidden, {p->head_in});
    p->mlp.push_back(first);
    for (int i = 0; i < c.mlp_n_hidden; ++i) {
      PackedLinear h;
      h.set_geometry(c.mlp_hidden, {c.mlp_hidden});
      p->mlp.push_back(h);
    }
    PackedLinear last;
    last.set_geometry(c.out_dim + (c.learn_temp ? 1 : 0), {c.mlp_hidden});
    p->mlp.push_back(last);
  } else {
    PackedLinear only;
    only.set_geometry(c.out_dim, {p->head_in});
    p->mlp.push_back(only);
  }

  // ---- persistent-kernel mode: one launch for all steps (wavenet_persist.hip) -------------------
  // Geometry it covers: gated k=2 layers, embedding input, MLP head without extra hidden layers,
  // C = skips = residuals in {32..256 step 32}, at most one conditioning input of a multiple of 16
  // channels.  Anything else stays on the per-layer launch path.
  const char* env = p->tune.get("MMK_WN_PERSISTENT");
  bool ok = !(env && env[0] == '0') && c.exec_mode != 1;     // (exec_mode 1: the caller asks for the per-layer launch path)
  ok = ok && c.gated && c.act_f == ACT_TANH && c.act_g == ACT_SIGMOID && c.mlp_act == ACT_MISH && c.q_levels > 0 && c.head_kind == 0 && c.mlp_n_hidden == 0 && c.n_cond <= 2;      // (the default gate is built into the persistent kernels and the prefill)
  ok = ok && !multi;            // class conditioning streams and further targets: the launch path (a step's conditioning row depends on the step before)
  ok = ok && p->C % 32 == 0 && p->C <= 256 && p->S == p->C && c.residuals_dim == p->C && !c.layerwise_inputs && !c.with_affine_residuals;
  for (int l = 0; l < p->L; ++l) ok = ok && (p->has_res[l] != 0) == (l != p->L - 1);   // (reverse_layer_order: launch path)
  for (int j = 0; j < c.n_cond; ++j) ok = ok && c.cond_dim[j] % 16 == 0;
  for (int l = 0; l < p->L; ++l) ok = ok && p->ksz[l] == 2;
  // (the stage pipeline takes any head of up to 128 hidden units and two conditioning inputs; the other persistent kernels whole tiles of 16 and one)
  const bool ok_sp = ok && c.mlp_hidden >= 1;
  // (two conditioning inputs: the layer pipeline only - their projections side by side in a row, the 1x1 matrices side by side in the GEMM's K; decided below)
  ok = ok && c.mlp_hidden % 16 == 0 && c.mlp_hidden >= 16 && (c.n_cond <= 1 || (c.n_cond == 2 && p->C == 64));
  p->persistent = false;
  // The persistent kernels need every workgroup of their grid resident at once (one per CU: their LDS carve does not leave room
  // for a second), and the XCD-local / pipelined placements one stage or clip group per XCD of an 8-XCD device: ask the device
  // (a partitioned or smaller GPU falls back to agent-scope hand-offs or to the launch path instead of timing out)
  int n_cu = 256, n_xcc = 8;
  {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess) {
      if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n_cu = v;
      if (hipDeviceGetAttribute(&v, hipDeviceAttributeNumberOfXccs, dev) == hipSuccess && v > 0) n_xcc = v;
    }
  }
  if (ok) {
    p->Gn = 2 * p->C / 16;
    int gc_max = n_cu / p->Gn;
    if (gc_max < 1) gc_max = 1;
    // clips per group: one MFMA row tile (16) at most, and one epilogue element per thread
    // (16 columns x Mg clips <= 64 * C/32 threads)
    const int mg_cap = 16;                           // one MFMA row tile; one epilogue element per I/O thread
    int gc = (p->Bmax + 7) / 8;                      // aim at 8 clips per group
    // the one-hand-off kernel (wavenet_chain.hip) takes groups of at most 4 clips: prefer those when the grid allows it
    // ... and it pays where the XCD's L2 (4 MiB) holds its weights: the pre-multiplied matrices are a third more bytes,
    // and a layer set that has to come over the fabric every step is bound by that stream (measured, DESIGN.md 5.2:
    // cfg 4 streams 62 MB per XCD and step at ~0.9 TB/s = 70 us, more than the two-hand-off kernel takes)
    // (round 4's sweep, DESIGN 5.6: up to 128 channels the one-hand-off kernel is the fastest wherever it runs - 44 against 57 us per step at
    //  64 channels x 30 layers, 20 against 40 at 128 x 10, 48 against 62 - 64 at 128 x 30 -, whether its pre-multiplied matrices fit an L2 or
    //  not; at 256 channels it loses to the other kernels)
    const char* chenv = p->tune.get("MMK_WN_CHAIN");
    const bool chain_default = p->C <= 128;
    bool chain_wanted = p->L >= 2 && (chenv ? chenv[0] != '0' : chain_default);
    for (int l = 0; l + 1 < p->L; ++l) chain_wanted = chain_wanted && p->has_res[l];
    if (chain_wanted && (p->Bmax + 3) / 4 <= gc_max) gc = (p->Bmax + 3) / 4;
    const char* genv = p->tune.get("MMK_WN_GROUPS");
    if (genv && atoi(genv) > 0) gc = atoi(genv);
    if (gc < (p->Bmax + mg_cap - 1) / mg_cap) gc = (p->Bmax + mg_cap - 1) / mg_cap;
    if (gc > gc_max) gc = gc_max;
    if (gc > p->Bmax) gc = p->Bmax;
    p->Mg = (p->Bmax + gc - 1) / gc;
    p->Gc = (p->Bmax + p->Mg - 1) / p->Mg;
    // XCD-local mode: always 8 groups (one per XCD, some possibly without clips), 8 * Gn workgroups
    const char* xenv = p->tune.get("MMK_WN_XCD_LOCAL");
    p->xcd_local = !(xenv && xenv[0] == '0') && n_xcc == 8 && 8 * p->Gn <= n_cu && (p->Bmax + 7) / 8 <= mg_cap;
    if (p->xcd_local) {
      p->Gc = 8;
      p->Mg = (p->Bmax + 7) / 8;
    }
    p->C1 = 0;
    for (int j = 0; j < c.n_cond; ++j) p->C1 += c.cond_dim[j];
    p->n_logits_pad = (int)round_up(c.out_dim + (c.learn_temp ? 1 : 0), 16);
    if (p->Mg <= mg_cap && p->Gc * p->Gn <= n_cu) {
      p->persistent = true;
      p->ring_offset.assign(p->L, 0);
      p->ring_mask.assign(p->L, 0);
      int64_t off = 0;
      for (int l = 0; l < p->L; ++l) {
        int ring = 1;
        while (ring < p->dil[l] + 1) ring <<= 1;
        p->ring_offset[l] = off;
        p->ring_mask[l] = ring - 1;
        off += (int64_t)ring * p->Mg * p->C;      // one slot = the group's clips x C
      }
      p->ring_floats_per_wg = off;
      if (off * 4 >= ((int64_t)1 << 32)) p->persistent = false;   // ring offsets are 32-bit byte offsets in the kernel
      {
        WnPersistArgs probe = {};   // the kernel's LDS carve must fit (wide MLP heads may not)
        probe.C = p->C; probe.S = p->S; probe.H1 = c.mlp_hidden; probe.n_logits_pad = p->n_logits_pad; probe.L = p->L;
        probe.Gn = p->Gn;
        if (wn_persist_lds_bytes(probe) > 160 * 1024) p->persistent = false;
      }
      if (p->C1 > 0) p->cond_all.set_geometry(p->L * 2 * p->C, {p->C1});
    }
  }
  // ---- stage pipeline (wavenet_spipe.hip): one layer per stage of 8 CUs, 4 stages per XCD, clips streamed through one at a time.
  // 256 channels, <= 31 layers, <= 32 clips, the same layer structure as the other persistent kernels; needs the whole 8 x 32 CU
  // chip and the warm-up as a prefill into the launch path's rings.  MMK_WN_SPIPE=0, or forcing another kernel (MMK_WN_PIPE=1 /
  // MMK_WN_CHAIN=1), turns it off.
  p->spipe = false;
  if (ok_sp) {
    const char* senv = p->tune.get("MMK_WN_SPIPE");
    const char* fenv = p->tune.get("MMK_WN_PREFILL");
    const char* penv = p->tune.get("MMK_WN_PIPE");
    const char* cenv = p->tune.get("MMK_WN_CHAIN");
    bool ok5 = !(senv && senv[0] == '0') && !(fenv && fenv[0] == '0') && !(penv && penv[0] == '1') && !(cenv && cenv[0] == '1');
    int cond_total = 0;
    for (int j = 0; j < c.n_cond; ++j) cond_total += c.cond_dim[j];
    // (the classes the network is fed are the ones it draws, and the first one - the prompt's last sample - is clamped to the head's 256)
    ok5 = ok5 && n_xcc == 8 && n_cu == 256 && c.q_levels <= 256 && c.out_dim <= c.q_levels;
    // Groups of 16 clips on the matrix pipe (wavenet_bpipe.hip) where the one-clip ring is beat-bound: a group's step is a trip of L + 1 visits of
    // ~3.5 us whatever the batch (up to ~10 groups; beyond, a stage's ~9.5 us per visit is the beat), the ring's is ~1.1 us per clip.  MMK_WN_BPIPE=0 turns it off, =1 takes it for any batch.
    const char* benv = p->tune.get("MMK_WN_BPIPE");
    bool ok6 = ok5 && !(benv && benv[0] == '0') && wn_bpipe_supported(p->C, p->S, c.mlp_hidden, c.out_dim, p->L, c.n_cond, cond_total, p->Bmax);
    ok6 = ok6 && ((benv && benv[0] == '1') || (bpipe_by_default(p->Bmax) && p->L >= 16));
    ok5 = ok5 && (ok6 || wn_spipe_supported(p->C, p->S, c.mlp_hidden, c.out_dim, p->L, c.n_cond, cond_total, p->Bmax));
    p->bpipe = false;
    // A ring of few stages is beat-bound early (one clip's trip: ~1.3 us per stage; ~1.25 us per clip once the clips queue up): 10 layers x 32 clips
    // 38 us per step against 31 on the two-hand-off kernel, x 64 clips 74 against 43 (round 4's sweep).  Asked for by name
    // (MMK_WN_SPIPE=1) it is taken all the same.
    if (!(senv && senv[0] == '1') && !ok6 && p->L <= 15 && 1.25 * p->Bmax > 3.0 * p->L + 8.0) ok5 = false;
    if (ok5) {
      p->spipe = true;
      p->bpipe = ok6;
      p->persistent = true;
      p->xcd_local = false;
      p->chain = false;
      p->C1 = cond_total;
      p->n_logits_pad = (int)round_up(c.out_dim + (c.learn_temp ? 1 : 0), 16);
      p->ring_offset.assign(p->L, 0);
      p->ring_mask.assign(p->L, 0);
      p->ring_floats_per_wg = 0;
    }
  }
  // one hand-off per layer: every layer but the last needs its residual 1x1 (it is folded into the next layer's tap-1
  // product), groups of at most 4 clips (4x4 MFMA blocks)
  p->chain = false;
  p->Ac.clear();
  if (p->persistent && !p->spipe) {
    const char* cenv = p->tune.get("MMK_WN_CHAIN");
    const char* pforce = p->tune.get("MMK_WN_PIPE");
    bool ok2 = (cenv ? cenv[0] != '0' : p->C <= 128) && wn_chain_supported(p->C, p->Mg, p->L) && !(pforce && pforce[0] == '1');
    for (int l = 0; l + 1 < p->L; ++l) ok2 = ok2 && p->has_res[l];
    if (ok2) {
      WnChainArgs probe = {};
      probe.C = p->C; probe.H1 = c.mlp_hidden; probe.n_logits_pad = p->n_logits_pad; probe.L = p->L; probe.Gn = p->Gn;
      ok2 = wn_chain_lds_bytes(probe) <= 160 * 1024;
    }
    if (ok2) {
      p->chain = true;
      p->Ac.resize(p->L);
      for (auto& pl : p->Ac) pl.set_geometry(2 * p->C, {p->C, p->C, p->C});
    }
  }
  // ---- layer pipeline (wavenet_lpipe.hip): small networks whose layers fit a fraction of a CU's registers; 32 workgroups per 8 clips,
  // all resident, four per clip on one XCD; the warm-up is the prefill, scattered into the launch path's rings.  MMK_WN_LPIPE=0: off.
  p->lpipe = false;
  if (p->persistent && !p->spipe) {
    const char* lenv = p->tune.get("MMK_WN_LPIPE");
    const char* fenv = p->tune.get("MMK_WN_PREFILL");
    bool ok4 = !(lenv && lenv[0] == '0') && !(fenv && fenv[0] == '0') && n_xcc == 8 && 32 * ((p->Bmax + 7) / 8) <= n_cu && c.q_levels <= 256 && c.out_dim <= c.q_levels;
    ok4 = ok4 && wn_lpipe_supported(p->C, p->S, c.mlp_hidden, c.out_dim, p->L, c.n_cond, p->Bmax);
    for (int l = 0; l + 1 < p->L; ++l) ok4 = ok4 && p->has_res[l];
    p->lpipe = ok4;
  }
  if (p->persistent && !p->spipe && !p->lpipe && c.n_cond > 1) {      // two inputs without the layer pipeline: the launch path
    p->persistent = false;
    p->chain = false;
    p->Ac.clear();
  }
  return MMK_OK;
}

extern "C" int mmk_wavenet_plan_create(const mmk_wavenet_config* cfg, mmk_wavenet_plan** out) {
  if (!cfg || !out) return fail(MMK_ERR_INVALID, "wavenet_plan_create: null argument");
  if (cfg->act_f < 0 || cfg->act_f > ACT_COS || cfg->act_g < 0 || cfg->act_g > ACT_COS || cfg->mlp_act < 0 || cfg->mlp_act > ACT_COS)
    return fail(MMK_ERR_INVALID, "wavenet_plan_create: act_f / act_g / mlp_act outside MMK_ACT_*");
  mmk_wavenet_plan* p = new mmk_wavenet_plan();
  p->cfg = *cfg;
  p->tune.parse(cfg->tuning, sizeof(cfg->tuning));
  int rc = derive(p);
  if (rc != MMK_OK) {
    delete p;
    return rc;
  }
  *out = p;
  return MMK_OK;
}

extern "C" void mmk_wavenet_plan_destroy(mmk_wavenet_plan* p) {
  if (!p) return;
  p->gc.reset();
  if (p->cap_stream) (void)hipStreamDestroy(p->cap_stream);
  delete p;
}

extern "C" int mmk_wavenet_plan_bind(mmk_wavenet_plan* p, const char* key, const float* dev_ptr, int64_t numel) {
  if (!p || !key || !dev_ptr) return fail(MMK_ERR_INVALID, "wavenet_plan_bind: null argument");
  p->binder.bind(key, dev_ptr, numel);
  p->committed = false;
  return MMK_OK;
}

extern "C" int64_t mmk_wavenet_receptive_field(const mmk_wavenet_plan* p) { return p ? p->rf : 0; }

extern "C" size_t mmk_wavenet_workspace_bytes(const mmk_wavenet_plan* p) {
  if (!p) return 0;
  mmk_wavenet_plan tmp = *p;  // layout() only writes pointers; run it on a copy
  tmp.gc = GraphCache();
  tmp.cap_stream = nullptr;
  Carver c(nullptr);
  tmp.layout(c);
  return c.used();
}

__global__ void vec_add_kernel(const float* __restrict__ a, const float* __restrict__ b, int n, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = a[i] + b[i];
}

extern "C" int mmk_wavenet_commit(mmk_wavenet_plan* p, void* workspace, size_t workspace_bytes, mmk_stream_t stream) {
  if (!p || !workspace) return fail(MMK_ERR_INVALID, "wavenet_commit: null argument");
  if ((reinterpret_cast<uintptr_t>(workspace) & 255) != 0) return fail(MMK_ERR_WORKSPACE, "wavenet_commit: workspace must be 256-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const mmk_wavenet_config& c = p->cfg;
  Carver carve(workspace);
  p->layout(carve);
  if (carve.used() > workspace_bytes)
    return fail(MMK_ERR_WORKSPACE, "wavenet_commit: workspace of %zu bytes, %zu needed", workspace_bytes, carve.used());
  MMK_HIP(hipStreamSynchronize(st));   // replays of the cached graph may still be queued: wait before destroying it
  p->gc.reset();  // pointers inside a cached graph are stale now
  MMK_HIP(hipMemsetAsync(workspace, 0, carve.used(), st));

  Binder& b = p->binder;
  b.clear_missing();
  const int C = p->C, S = p->S, L = p->L;
  const bool bias = c.bias != 0;
  auto key = [](const std::string& s) { return s; };

  // input 0
  if (c.q_levels > 0) {
    p->emb = b.need(key("input_modules.0.0.weight"), (int64_t)c.q_levels * C);
  } else {
    const float* w = b.need("input_modules.0.0.weight", (int64_t)C * c.in_dim);
    const float* bb = b.need("input_modules.0.0.bias", C);
    if (w && bb) {
      MMK_TRY(pack_rect(p->in0_lin.Wp, p->in0_lin.k_chunks, 0, 1, C, 0, c.in_dim, w, c.in_dim, 1, st));
      MMK_TRY(pack_bias(p->in0_lin.bias, 0, 1, C, bb, 0, st));
    }
  }
  // conditioning inputs (LinearIO, modules/io.py:115-122)
  for (int j = 0; j < p->n_cond; ++j) {
    const std::string base = "input_modules." + std::to_string(j + 1) + ".0.";
    if (c.cond_q_levels[j] > 0) {           // EmbeddingIO (modules/io.py:132-136): the table itself, no bias
      p->cond_emb[j] = b.need(base + "weight", (int64_t)c.cond_q_levels[j] * c.cond_dim[j]);
      continue;
    }
    const float* w = b.need(base + "weight", (int64_t)c.cond_dim[j] * c.cond_in_dim[j]);
    const float* bb = b.need(base + "bias", c.cond_dim[j]);
    if (w && bb) {
      MMK_TRY(pack_rect(p->cond_lin[j].Wp, p->cond_lin[j].k_chunks, 0, 1, c.cond_dim[j], 0, c.cond_in_dim[j], w,
                        c.cond_in_dim[j], 1, st));
      MMK_TRY(pack_bias(p->cond_lin[j].bias, 0, 1, c.cond_dim[j], bb, 0, st));
    }
  }
  // layers
  for (int l = 0; l < L; ++l) {
    const std::string ly = "layers." + std::to_string(l) + ".";
    const int k = p->ksz[l];
    const int rows_src = c.gated ? 2 * C : C;
    const std::string dil_base = ly + (c.gated ? "conv_dil.0.0." : "conv_dil.0.");
    const float* wd = b.need(dil_base + "weight", (int64_t)rows_src * C * k);
    const float* bd = bias ? b.need(dil_base + "bias", rows_src) : nullptr;
    if (c.with_affine_residuals) {
      PackedLinear& F = p->Aff[l];
      const float* wa = b.need(ly + "aff_res.params.weight", (int64_t)3 * C * C);
      const float* ba = bias ? b.need(ly + "aff_res.params.bias", 3 * C) : nullptr;
      if (wa) MMK_TRY(pack_rect(F.Wp, F.k_chunks, 0, 1, 3 * C, 0, C, wa, C, 1, st));
      if (ba) MMK_TRY(pack_bias(F.bias, 0, 1, 3 * C, ba, 0, st));
    }
    PackedLinear& A = p->A[l];
    if (wd) {
      for (int j = 0; j < k; ++j) {
        // tap j multiplies x[tau - (k-1-j)*d]  (cross-correlation, tap 0 = most delayed)
        if (c.gated) {
          MMK_TRY(pack_rect(A.Wp, A.k_chunks, 0, 2, C, A.seg_chunk0[j], C, wd + j, (int64_t)C * k, k, st));
          MMK_TRY(pack_rect(A.Wp, A.k_chunks, 1, 2, C, A.seg_chunk0[j], C, wd + (int64_t)C * C * k + j, (int64_t)C * k, k, st));
        } else {
          MMK_TRY(pack_rect(A.Wp, A.k_chunks, 0, 1, C, A.seg_chunk0[j], C, wd + j, (int64_t)C * k, k, st));
        }
      }
    }
    if (bd) {
      if (c.gated) {
        MMK_TRY(pack_bias(A.bias, 0, 2, C, bd, 0, st));
        MMK_TRY(pack_bias(A.bias, 1, 2, C, bd + C, 0, st));
      } else {
        MMK_TRY(pack_bias(A.bias, 0, 1, C, bd, 0, st));
      }
    }
    for (int j = 0; j < p->n_cond; ++j) {
      const std::string cb = ly + "conv_1x1." + std::to_string(j) + (c.gated ? ".0." : ".");
      const int cd = c.cond_dim[j];
      const float* w1 = b.need(cb + "weight", (int64_t)rows_src * cd);
      const float* b1 = bias ? b.need(cb + "bias", rows_src) : nullptr;
      if (w1) {
        if (c.gated) {
          MMK_TRY(pack_rect(A.Wp, A.k_chunks, 0, 2, C, A.seg_chunk0[k + j], cd, w1, cd, 1, st));
          MMK_TRY(pack_rect(A.Wp, A.k_chunks, 1, 2, C, A.seg_chunk0[k + j], cd, w1 + (int64_t)C * cd, cd, 1, st));
          if (p->persistent && p->C1 > 0 && !p->spipe) {   // same gate-interleaved rows, all layers stacked (bias stays in A.bias); input j's columns behind input j - 1's
            int kc0 = 0;
            for (int jj = 0; jj < j; ++jj) kc0 += c.cond_dim[jj] / 16;
            MMK_TRY(pack_rect(p->cond_all.Wp, p->cond_all.k_chunks, l * 2 * C, 2, C, kc0, cd, w1, cd, 1, st));
            MMK_TRY(pack_rect(p->cond_all.Wp, p->cond_all.k_chunks, l * 2 * C + 1, 2, C, kc0, cd, w1 + (int64_t)C * cd, cd, 1, st));
          }
        } else {
          MMK_TRY(pack_rect(A.Wp, A.k_chunks, 0, 1, C, A.seg_chunk0[k + j], cd, w1, cd, 1, st));
        }
      }
      if (b1) {
        if (c.gated) {
          MMK_TRY(pack_bias(A.bias, 0, 2, C, b1, 1, st));
          MMK_TRY(pack_bias(A.bias, 1, 2, C, b1 + C, 1, st));
        } else {
          MMK_TRY(pack_bias(A.bias, 0, 1, C, b1, 1, st));
        }
      }
    }
    PackedLinear& Bm = p->Bm[l];
    if (Bm.n_tiles > 0) {
      const int n_res_pad = p->has_res[l] ? (int)round_up(C, 16) : 0;
      if (p->has_res[l]) {
        const float* wr = b.need(ly + "conv_res.weight", (int64_t)C * C);
        const float* br = bias ? b.need(ly + "conv_res.bias", C) : nullptr;
        if (wr) MMK_TRY(pack_rect(Bm.Wp, Bm.k_chunks, 0, 1, C, 0, C, wr, C, 1, st));
        if (br) MMK_TRY(pack_bias(Bm.bias, 0, 1, C, br, 0, st));
      }
      if (S > 0) {
        const float* ws = b.need(ly + "conv_skip.weight", (int64_t)S * C);
        const float* bs = bias ? b.need(ly + "conv_skip.bias", S) : nullptr;
        if (ws) MMK_TRY(pack_rect(Bm.Wp, Bm.k_chunks, n_res_pad, 1, S, 0, C, ws, C, 1, st));
        if (bs) MMK_TRY(pack_bias(Bm.bias, n_res_pad, 1, S, bs, 0, st));
      }
    }
  }
  // one-hand-off-per-layer mode: gate matrices [tap 0 | tap 1 | tap 1 . W_res(l-1)], bias + tap 1 . b_res(l-1)
  if (p->chain) {
    for (int l = 0; l < L; ++l) {
      const std::string ly = "layers." + std::to_string(l) + ".";
      const float* wd = b.need(ly + "conv_dil.0.0.weight", (int64_t)2 * C * C * 2);
      const float* bd = bias ? b.need(ly + "conv_dil.0.0.bias", 2 * C) : nullptr;
      PackedLinear& A = p->Ac[l];
      if (!wd) continue;
      for (int jt = 0; jt < 2; ++jt) {
        MMK_TRY(pack_rect(A.Wp, A.k_chunks, 0, 2, C, A.seg_chunk0[jt], C, wd + jt, (int64_t)C * 2, 2, st));
        MMK_TRY(pack_rect(A.Wp, A.k_chunks, 1, 2, C, A.seg_chunk0[jt], C, wd + (int64_t)C * C * 2 + jt, (int64_t)C * 2, 2, st));
      }
      if (bd) {
        MMK_TRY(pack_bias(A.bias, 0, 2, C, bd, 0, st));
        MMK_TRY(pack_bias(A.bias, 1, 2, C, bd + C, 0, st));
      }
      if (p->n_cond == 1 && bias) {
        const float* b1 = b.need(ly + "conv_1x1.0.0.bias", 2 * C);
        if (b1) {
          MMK_TRY(pack_bias(A.bias, 0, 2, C, b1, 1, st));
          MMK_TRY(pack_bias(A.bias, 1, 2, C, b1 + C, 1, st));
        }
      }
      if (l >= 1) {
        const std::string lp = "layers." + std::to_string(l - 1) + ".";
        const float* wr = b.need(lp + "conv_res.weight", (int64_t)C * C);
        const float* br = bias ? b.need(lp + "conv_res.bias", C) : nullptr;
        if (wr) {
          float* prod = p->compose_scratch;
          float* pbias = prod + (int64_t)2 * C * C;
          hipLaunchKernelGGL(compose_kernel, dim3(512), dim3(256), 0, st, wd + 1, (int64_t)2 * C, (int64_t)2, 2 * C, wr, br, prod, pbias, C);
          MMK_HIP(hipGetLastError());
          MMK_TRY(pack_rect(A.Wp, A.k_chunks, 0, 2, C, A.seg_chunk0[2], C, prod, C, 1, st));
          MMK_TRY(pack_rect(A.Wp, A.k_chunks, 1, 2, C, A.seg_chunk0[2], C, prod + (int64_t)C * C, C, 1, st));
          MMK_TRY(pack_bias(A.bias, 0, 2, C, pbias, 1, st));
          MMK_TRY(pack_bias(A.bias, 1, 2, C, pbias + C, 1, st));
        }
      }
    }
  }
  // stage pipeline: the per-lane register images of every stage, from the raw tensors (W1 R and fc0 W_skip composed in fp64)
  if (p->spipe) {
    std::vector<WnSpRaw> raw(L);
    for (int l = 0; l < L; ++l) {
      const std::string ly = "layers." + std::to_string(l) + ".";
      WnSpRaw& r = raw[l];
      r.wd = b.need(ly + "conv_dil.0.0.weight", (int64_t)2 * C * C * 2);
      r.bd = bias ? b.need(ly + "conv_dil.0.0.bias", 2 * C) : nullptr;
      r.w1 = p->n_cond == 1 ? b.need(ly + "conv_1x1.0.0.weight", (int64_t)2 * C * p->C1) : nullptr;
      r.b1 = (bias && p->n_cond == 1) ? b.need(ly + "conv_1x1.0.0.bias", 2 * C) : nullptr;
      if (p->n_cond == 2) {      // sum_j conv_1x1_j(c_j) (wavenet_v2.py:141-147) = [W_0 | W_1] [c_0 ; c_1]: the two matrices side by side, the biases added up
        float* wcat = p->sp_w1cat + (int64_t)l * 2 * C * p->C1;
        float* bcat = p->sp_b1cat + (int64_t)l * 2 * C;
        int col = 0;
        for (int j = 0; j < 2; ++j) {
          const int dj = c.cond_dim[j];
          const float* wj = b.need(ly + "conv_1x1." + std::to_string(j) + ".0.weight", (int64_t)2 * C * dj);
          if (wj) MMK_HIP(hipMemcpy2DAsync(wcat + col, (size_t)p->C1 * sizeof(float), wj, (size_t)dj * sizeof(float), (size_t)dj * sizeof(float), 2 * C,
                                           hipMemcpyDeviceToDevice, st));
          col += dj;
        }
        r.w1 = wcat;
        if (bias) {
          const float* b0 = b.need(ly + "conv_1x1.0.0.bias", 2 * C);
          const float* b1 = b.need(ly + "conv_1x1.1.0.bias", 2 * C);
          if (b0 && b1) {
            hipLaunchKernelGGL(vec_add_kernel, dim3((2 * C + 255) / 256), dim3(256), 0, st, b0, b1, 2 * C, bcat);
            MMK_HIP(hipGetLastError());
          }
          r.b1 = bcat;
        }
      }
      r.wr = p->has_res[l] ? b.need(ly + "conv_res.weight", (int64_t)C * C) : nullptr;
      r.br = (bias && p->has_res[l]) ? b.need(ly + "conv_res.bias", C) : nullptr;
      r.ws = b.need(ly + "conv_skip.weight", (int64_t)C * C);
      r.bs = bias ? b.need(ly + "conv_skip.bias", C) : nullptr;
    }
    const int H1 = c.mlp_hidden;
    const float* f0 = b.need("output_modules.0.estimator.0.fc.0.weight", (int64_t)H1 * C);
    const float* fb0 = b.need("output_modules.0.estimator.0.fc.0.bias", H1);
    p->sp_fc2_w = b.need("output_modules.0.estimator.0.fc.2.weight", (int64_t)(c.out_dim + (c.learn_temp ? 1 : 0)) * H1);
    p->sp_fc2_b = b.need("output_modules.0.estimator.0.fc.2.bias", c.out_dim + (c.learn_temp ? 1 : 0));
    if (b.missing().empty()) {
      // the head as the kernel knows it: 128 hidden units x 256 classes (+ the temperature row at index 256).  Hidden units that do not exist
      // have zero rows in fc0 and zero columns in fc2 (Mish(0) = 0 anyway), classes that do not exist a zero row and a bias of -inf: never the
      // maximum, probability 0 in a draw (the workspace was cleared at the start of this commit)
      constexpr int PH = mmk_wavenet_plan::kSpH1, PQ = mmk_wavenet_plan::kSpQ;
      const int Q = c.out_dim;
      MMK_HIP(hipMemcpyAsync(p->sp_f0p, f0, (size_t)H1 * C * sizeof(float), hipMemcpyDeviceToDevice, st));
      MMK_HIP(hipMemcpyAsync(p->sp_fb0p, fb0, (size_t)H1 * sizeof(float), hipMemcpyDeviceToDevice, st));
      MMK_HIP(hipMemcpy2DAsync(p->sp_fc2p, (size_t)PH * sizeof(float), p->sp_fc2_w, (size_t)H1 * sizeof(float), (size_t)H1 * sizeof(float), Q,
                               hipMemcpyDeviceToDevice, st));
      MMK_TRY(launch_fill(p->sp_fc2bp, -INFINITY, mmk_wavenet_plan::kSpLogitsLd, st));
      MMK_HIP(hipMemcpyAsync(p->sp_fc2bp, p->sp_fc2_b, (size_t)Q * sizeof(float), hipMemcpyDeviceToDevice, st));
      if (c.learn_temp) {
        MMK_HIP(hipMemcpyAsync(p->sp_fc2p + (int64_t)PQ * PH, p->sp_fc2_w + (int64_t)Q * H1, (size_t)H1 * sizeof(float), hipMemcpyDeviceToDevice, st));
        MMK_HIP(hipMemcpyAsync(p->sp_fc2bp + PQ, p->sp_fc2_b + Q, sizeof(float), hipMemcpyDeviceToDevice, st));
      }
      MMK_HIP(hipMemcpyAsync(p->sp_raw, raw.data(), sizeof(WnSpRaw) * L, hipMemcpyHostToDevice, st));
      MMK_TRY(wn_spipe_build_image(p->sp_raw, L, C, p->C1, p->sp_f0p, p->sp_fb0p, p->sp_img_chain, p->sp_img_helper, p->sp_cst_chain, p->sp_cst_helper,
                                   p->sp_head_w0, p->sp_head_b0, st));
      if (p->bpipe) MMK_TRY(wn_bpipe_build_image(p->sp_raw, L, p->C1, p->sp_f0p, p->bp_img, p->bp_cst, st));
      MMK_HIP(hipStreamSynchronize(st));   // `raw` is host-local
    }
  }
  // head
  if (c.head_kind == 0) {
    const std::string hb = "output_modules.0.estimator.0.fc.";
    for (size_t i = 0; i < p->mlp.size(); ++i) {
      PackedLinear& m = p->mlp[i];
      const std::string kb = hb + std::to_string(2 * i) + ".";
      const float* w = b.need(kb + "weight", (int64_t)m.N * m.segK[0]);
      const float* bb = b.need(kb + "bias", m.N);
      if (w) MMK_TRY(pack_rect(m.Wp, m.k_chunks, 0, 1, m.N, 0, m.segK[0], w, m.segK[0], 1, st));
      if (bb) MMK_TRY(pack_bias(m.bias, 0, 1, m.N, bb, 0, st));
    }
  } else {
    PackedLinear& m = p->mlp[0];
    const float* w = b.need("output_modules.0.0.weight", (int64_t)m.N * m.segK[0]);
    const float* bb = b.need("output_modules.0.0.bias", m.N);
    if (w) MMK_TRY(pack_rect(m.Wp, m.k_chunks, 0, 1, m.N, 0, m.segK[0], w, m.segK[0], 1, st));
    if (bb) MMK_TRY(pack_bias(m.bias, 0, 1, m.N, bb, 0, st));
  }
  if (p->lpipe) {      // the layer pipeline's head is 128 hidden units x 256 classes too: the padded head (as for the stage pipeline), packed
    constexpr int PH = mmk_wavenet_plan::kSpH1, PQ = mmk_wavenet_plan::kSpQ;
    const int H1 = c.mlp_hidden, Q = c.out_dim;
    const float* f0 = b.need("output_modules.0.estimator.0.fc.0.weight", (int64_t)H1 * C);
    const float* fb0 = b.need("output_modules.0.estimator.0.fc.0.bias", H1);
    const float* f2 = b.need("output_modules.0.estimator.0.fc.2.weight", (int64_t)(Q + (c.learn_temp ? 1 : 0)) * H1);
    const float* fb2 = b.need("output_modules.0.estimator.0.fc.2.bias", Q + (c.learn_temp ? 1 : 0));
    if (f0 && fb0 && f2 && fb2) {
      MMK_HIP(hipMemcpyAsync(p->sp_f0p, f0, (size_t)H1 * C * sizeof(float), hipMemcpyDeviceToDevice, st));
      MMK_HIP(hipMemcpyAsync(p->sp_fb0p, fb0, (size_t)H1 * sizeof(float), hipMemcpyDeviceToDevice, st));
      MMK_HIP(hipMemcpy2DAsync(p->sp_fc2p, (size_t)PH * sizeof(float), f2, (size_t)H1 * sizeof(float), (size_t)H1 * sizeof(float), Q, hipMemcpyDeviceToDevice, st));
      MMK_TRY(launch_fill(p->sp_fc2bp, -INFINITY, mmk_wavenet_plan::kSpLogitsLd, st));
      MMK_HIP(hipMemcpyAsync(p->sp_fc2bp, fb2, (size_t)Q * sizeof(float), hipMemcpyDeviceToDevice, st));
      if (c.learn_temp) {
        MMK_HIP(hipMemcpyAsync(p->sp_fc2p + (int64_t)PQ * PH, f2 + (int64_t)Q * H1, (size_t)H1 * sizeof(float), hipMemcpyDeviceToDevice, st));
        MMK_HIP(hipMemcpyAsync(p->sp_fc2bp + PQ, fb2 + Q, sizeof(float), hipMemcpyDeviceToDevice, st));
      } else {
        MMK_TRY(launch_fill(p->sp_fc2bp + PQ, 0.f, 1, st));
      }
      MMK_TRY(pack_rect(p->lp_mlp0.Wp, p->lp_mlp0.k_chunks, 0, 1, PH, 0, C, p->sp_f0p, C, 1, st));
      MMK_TRY(pack_bias(p->lp_mlp0.bias, 0, 1, PH, p->sp_fb0p, 0, st));
      MMK_TRY(pack_rect(p->lp_mlp1.Wp, p->lp_mlp1.k_chunks, 0, 1, PQ + 1, 0, PH, p->sp_fc2p, PH, 1, st));
      MMK_TRY(pack_bias(p->lp_mlp1.bias, 0, 1, PQ + 1, p->sp_fc2bp, 0, st));
    }
  }
  for (size_t k = 0; k < p->xheads.size(); ++k) {
    WnHead& h = p->xheads[k];
    for (size_t i = 0; i < h.mlp.size(); ++i) {
      PackedLinear& m = h.mlp[i];
      const std::string kb = "output_modules." + std::to_string(k + 1) + ".estimator.0.fc." + std::to_string(2 * i) + ".";
      const float* w = b.need(kb + "weight", (int64_t)m.N * m.segK[0]);
      const float* bb = b.need(kb + "bias", m.N);
      if (w) MMK_TRY(pack_rect(m.Wp, m.k_chunks, 0, 1, m.N, 0, m.segK[0], w, m.segK[0], 1, st));
      if (bb) MMK_TRY(pack_bias(m.bias, 0, 1, m.N, bb, 0, st));
    }
  }
  if (!b.missing().empty()) return fail(MMK_ERR_KEY, "wavenet_commit: state_dict tensor %s", b.missing().c_str());
  if (p->persistent && !p->spipe) {
    std::vector<WnLayerTab> tab(L);
    for (int l = 0; l < L; ++l) {
      tab[l].dil = p->dil[l];
      tab[l].has_res = p->has_res[l] ? 1 : 0;
      tab[l].ring_mask = p->ring_mask[l];
      tab[l].pad_ = 0;
      tab[l].ring_offset = p->ring_offset[l];
      tab[l].A_wp = p->A[l].Wp;
      tab[l].A_bias = p->A[l].bias;
      tab[l].B_wp = p->Bm[l].Wp;
      tab[l].B_bias = p->Bm[l].bias;
    }
#ifdef MMK_DIAG
    if (const char* x = p->tune.get("MMK_WN_EXPERIMENT_SAME_WEIGHTS"); x && x[0] == '1') {
      // timing experiment only (results are wrong; diagnostic build): every layer reads layer 0's weights, which then stay in L2
      for (int l = 1; l < L; ++l) { tab[l].A_wp = tab[0].A_wp; tab[l].B_wp = tab[0].B_wp; }
    }
#endif
    MMK_HIP(hipMemcpyAsync(p->layer_tab, tab.data(), sizeof(WnLayerTab) * L, hipMemcpyHostToDevice, st));
    std::vector<WnChainIter> it(L + 1);
    if (p->chain) {
      for (int i = 0; i <= L; ++i) {
        const int la = i < L ? i : 0;
        it[i].A_wp = p->Ac[la].Wp;
        it[i].A_bias = p->Ac[la].bias;
        const PackedLinear* Bsrc = i >= 1 ? &p->Bm[i - 1] : nullptr;
        it[i].B_wp = Bsrc ? Bsrc->Wp : nullptr;
        it[i].B_bias = Bsrc ? Bsrc->bias : nullptr;
        it[i].ring_offset = p->ring_offset[la];
        it[i].dil = p->dil[la];
        it[i].ring_mask = p->ring_mask[la];
        it[i].prev_has_res = (i >= 1 && p->has_res[i - 1]) ? 1 : 0;
        it[i].pad_ = 0;
      }
      MMK_HIP(hipMemcpyAsync(p->iter_tab, it.data(), sizeof(WnChainIter) * (L + 1), hipMemcpyHostToDevice, st));
    }
    MMK_HIP(hipStreamSynchronize(st));   // `tab` is host-local
  }
  if (!p->cap_stream) MMK_HIP(hipStreamCreateWithFlags(&p->cap_stream, hipStreamNonBlocking));
  p->committed = true;
  return MMK_OK;
}

// enqueue one step at position tau = *plan->tau + tau_off
static int emit_step(mmk_wavenet_plan* p, const WnCall& call, int64_t tau_off, bool with_head, hipStream_t st) {
  const mmk_wavenet_config& c = p->cfg;
  const int C = p->C, S = p->S, L = p->L, M = call.M;
  g_prof_tag = 2;
  // input module 0
  if (c.q_levels > 0) {
    MMK_TRY(launch_embed((const int64_t*)call.in0, call.in0_rs, 0, p->emb, C, c.q_levels, p->hist_slot(0, 0), C, M,
                         p->tau, tau_off, st));
  } else {
    LinearArgs a = {};
    p->in0_lin.fill(a);
    a.seg[0].x = addr_time(call.in0, c.in_dim, 0, 1, 0);
    a.seg[0].ld = call.in0_rs;
    a.M = M; a.tau_ptr = p->tau; a.tau_off = tau_off;
    a.epilogue = EPI_STORE; a.act = ACT_NONE;
    a.out = p->hist_slot(0, 0); a.out_ld = C;
    MMK_TRY(launch_linear(a, st));
  }
  for (int j = 0; j < p->n_cond; ++j) {
    if (c.cond_q_levels[j] > 0) {      // a class stream: the row of its table for position tau (written by the step before when a target feeds it)
      MMK_TRY(launch_embed(reinterpret_cast<const int64_t*>(call.cond[j]), call.cond_rs[j], 0, p->cond_emb[j], c.cond_dim[j], c.cond_q_levels[j],
                           addr_static(p->cbuf[j]), c.cond_dim[j], M, p->tau, tau_off, st));
      continue;
    }
    LinearArgs a = {};
    p->cond_lin[j].fill(a);
    a.seg[0].x = addr_time(call.cond[j], c.cond_in_dim[j], 0, 1, 0);
    a.seg[0].ld = call.cond_rs[j];
    a.M = M; a.tau_ptr = p->tau; a.tau_off = tau_off;
    a.epilogue = EPI_STORE; a.act = ACT_NONE;
    a.out = addr_static(p->cbuf[j]); a.out_ld = c.cond_dim[j];
    MMK_TRY(launch_linear(a, st));
  }
  for (int l = 0; l < L; ++l) {
    const int k = p->ksz[l], d = p->dil[l];
    if (c.with_affine_residuals) {   // the layer's newest input becomes x_hat * a + b, in place: both taps and the residual sum see it
      LinearArgs a = {};
      p->Aff[l].fill(a);
      a.seg[0].x = p->hist_slot(l, 0); a.seg[0].ld = C;
      a.M = M; a.tau_ptr = p->tau; a.tau_off = tau_off;
      a.epilogue = EPI_STORE; a.act = ACT_NONE;
      a.out = addr_static(p->affbuf); a.out_ld = 3 * C;
      g_prof_tag = 2;
      MMK_TRY(launch_linear(a, st));
      MMK_TRY(launch_affine_rows(p->affbuf, 3 * C, p->hist_slot(l, 0), C, M, C, p->tau, tau_off, st));
    }
    {
      LinearArgs a = {};
      p->A[l].fill(a);
      for (int j = 0; j < k; ++j) {
        a.seg[j].x = p->hist_slot(l, -(k - 1 - j) * d);
        a.seg[j].ld = C;
      }
      for (int j = 0; j < p->n_cond; ++j) {
        a.seg[k + j].x = addr_static(p->cbuf[j]);
        a.seg[k + j].ld = c.cond_dim[j];
      }
      a.M = M; a.tau_ptr = p->tau; a.tau_off = tau_off;
      a.epilogue = c.gated ? EPI_GATE : EPI_STORE;
      a.act = c.act_f;                      // act_f(z_f) act_g(z_g) / act_f(z) (wavenet_v2.py:151, :163)
      a.act2 = c.act_g;
      a.out = p->y_addr(l); a.out_ld = C;
      g_prof_tag = 0;
      MMK_TRY(launch_linear(a, st));
    }
    if (p->Bm[l].n_tiles > 0) {
      LinearArgs a = {};
      p->Bm[l].fill(a);
      a.seg[0].x = p->y_addr(l);
      a.seg[0].ld = C;
      a.M = M; a.tau_ptr = p->tau; a.tau_off = tau_off;
      a.epilogue = EPI_RES_SKIP;
      a.n_res = p->has_res[l] ? C : 0;
      a.n_res_pad = p->has_res[l] ? (int)round_up(C, 16) : 0;
      a.n_skip = S;
      a.skip_first = (l == 0);
      a.res_in = p->hist_slot(l, 0); a.res_in_ld = C;
      a.res_out = p->hist_slot(l + 1, 0); a.res_out_ld = C;
      a.skip = addr_static(p->skipbuf); a.skip_ld = S > 0 ? S : 1;
      g_prof_tag = 1;
      MMK_TRY(launch_linear(a, st));
    }
    if (c.layerwise_inputs) {   // dilated = dilated + inputs[0]   (:285-286): the layer's output at tau + the embedded input at tau
      g_prof_tag = 2;
      const Addr dst = l + 1 < L ? p->hist_slot(l + 1, 0) : (p->has_res[l] ? p->hist_slot(L, 0) : addr_static(p->ybuf));
      MMK_TRY(launch_add_rows(p->hist_slot(0, 0), C, dst, C, M, C, p->tau, tau_off, st));
    }
  }
  g_prof_tag = 2;
  if (!with_head) return MMK_OK;
  // without skips the head reads the last layer's output: its gated units, or (reverse_layer_order) the residual sum
  const float* x = S > 0 ? p->skipbuf : (p->has_res[L - 1] ? p->hist[L] : p->ybuf);
  int x_ld = p->head_in;
  if (c.head_kind == 0) {
    for (size_t i = 0; i < p->mlp.size(); ++i) {
      const bool last = (i + 1 == p->mlp.size());
      LinearArgs a = {};
      p->mlp[i].fill(a);
      a.seg[0].x = addr_static(x);
      a.seg[0].ld = x_ld;
      a.M = M; a.tau_ptr = p->tau; a.tau_off = tau_off;
      a.epilogue = EPI_STORE;
      a.act = last ? (int)ACT_NONE : c.mlp_act;  // MLPIO.activation (modules/io.py:205: Mish unless the spec says otherwise)
      float* o = last ? p->logits : p->hid[i & 1];
      a.out = addr_static(o);
      a.out_ld = last ? p->logits_ld : c.mlp_hidden;
      MMK_TRY(launch_linear(a, st));
      x = o;
      x_ld = (int)a.out_ld;
    }
    SampleArgs s = {};
    s.logits = p->logits; s.ld = p->logits_ld; s.rows = M; s.n_classes = c.out_dim; s.has_temp_col = c.learn_temp;
    s.min_temp = c.min_temp; s.temperature = call.temperature; s.uniforms = call.uniforms;
    s.uniform_ld = call.uni_ld; s.uni_off = call.uni_off;
    s.out = (int64_t*)call.in0; s.out_row_stride = call.in0_rs; s.out_tau_off = 1;
    s.tau_ptr = p->tau; s.tau_off = tau_off;
    MMK_TRY(launch_sample(s, st));
    const float* x_head = S > 0 ? p->skipbuf : (p->has_res[L - 1] ? p->hist[L] : p->ybuf);
    for (size_t k = 0; k < p->xheads.size(); ++k) {     // one output module per target on the same vector (:293); output k + 1 goes into input k + 1
      const WnHead& h = p->xheads[k];
      const float* xk = x_head;
      int xk_ld = p->head_in;
      for (size_t i = 0; i < h.mlp.size(); ++i) {
        const bool last = (i + 1 == h.mlp.size());
        LinearArgs a = {};
        h.mlp[i].fill(a);
        a.seg[0].x = addr_static(xk);
        a.seg[0].ld = xk_ld;
        a.M = M; a.tau_ptr = p->tau; a.tau_off = tau_off;
        a.epilogue = EPI_STORE;
        a.act = last ? (int)ACT_NONE : c.mlp_act;
        float* o = last ? h.logits : p->hid[i & 1];
        a.out = addr_static(o);
        a.out_ld = last ? h.logits_ld : h.hidden;
        MMK_TRY(launch_linear(a, st));
        xk = o;
        xk_ld = (int)a.out_ld;
      }
      SampleArgs sk = {};
      sk.logits = h.logits; sk.ld = h.logits_ld; sk.rows = M; sk.n_classes = h.q; sk.has_temp_col = h.learn_temp;
      sk.min_temp = h.min_temp; sk.temperature = call.temperature;
      sk.uniforms = call.uniforms ? call.uniforms + (int64_t)(k + 1) * M * call.uni_ld : nullptr;     // (n_targets, batch, n_steps)
      sk.uniform_ld = call.uni_ld; sk.uni_off = call.uni_off;
      sk.out = reinterpret_cast<int64_t*>(const_cast<float*>(call.cond[k])); sk.out_row_stride = call.cond_rs[k]; sk.out_tau_off = 1;
      sk.tau_ptr = p->tau; sk.tau_off = tau_off;
      MMK_TRY(launch_sample(sk, st));
    }
  } else {
    LinearArgs a = {};
    p->mlp[0].fill(a);
    a.seg[0].x = addr_static(x);
    a.seg[0].ld = x_ld;
    a.M = M; a.tau_ptr = p->tau; a.tau_off = tau_off;
    a.epilogue = EPI_STORE; a.act = c.head_kind == 1 ? ACT_ABS : ACT_NONE;
    a.out = addr_time(call.in0, c.in_dim, 1, 1, 0);
    a.out_ld = call.in0_rs;
    MMK_TRY(launch_linear(a, st));
  }
  return MMK_OK;
}

static constexpr int kGraphSteps = 8;

// rows (clip b, position t0 + i), i < n, of a (batch, T, K) tensor -> compact rows b n + i of Kp = K rounded up to 16 floats, zero padded
__global__ void cond_compact_kernel(const float* __restrict__ src, int64_t clip_stride, int64_t t0, int K, int Kp, int n, int B, float* __restrict__ dst) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)B * n * Kp) return;
  const int k = (int)(e % Kp);
  const int64_t row = e / Kp;
  const int b = (int)(row / n), i = (int)(row - (int64_t)b * n);
  dst[e] = k < K ? src[(int64_t)b * clip_stride + (t0 + i) * K + k] : 0.f;
}

// c[b, t0 + i, :] = LinearIO_j(cond_j[b, t0 + i, :]) (modules/io.py:115-122) for i < n and all clips -> out + (b out_clip_stride + i) C1 + col:
// the rows made compact and 16-float aligned, then ONE tiled GEMM over them (the row-tile kernel on the rows as they lie - 513 floats apart,
// readable 4 bytes at a time - ran the cfg-4 block's 32 768 x 513 x 256 product in 528 us, 20 TFLOP/s; MMK_WN_COND_GEMM=0: that form)
static int project_cond(mmk_wavenet_plan* p, const WnCall& call, int j, int col, int64_t t0, int64_t n, float* out, int64_t out_clip_stride, hipStream_t st) {
  const mmk_wavenet_config& c = p->cfg;
  const int K = c.cond_in_dim[j], Kp = (int)round_up(K, 16), B = call.M;
  const char* genv = p->tune.get("MMK_WN_COND_GEMM");
  const int64_t M = (int64_t)B * n;
  if (!(genv && genv[0] == '0') && p->cpad && M <= p->cpad_rows && M < ((int64_t)1 << 31) && gemm_bias_act_supported(p->cpad, Kp, (int)M, K)) {
    const int64_t total = M * Kp;
    hipLaunchKernelGGL(cond_compact_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, call.cond[j], call.cond_rs[j], t0, K, Kp, (int)n, B, p->cpad);
    MMK_HIP(hipGetLastError());
    GemmRowMap rm;
    rm.group = (int)n; rm.kept = (int)n; rm.group_stride = out_clip_stride * p->C1; rm.row_stride = p->C1;
    return launch_gemm_bias_act(p->cpad, Kp, p->cond_lin[j].Wp, p->cond_lin[j].bias, p->cond_lin[j].n_tiles, p->cond_lin[j].k_chunks, c.cond_dim[j], K,
                                out + col, p->C1, (int)M, ACT_NONE, st, rm);
  }
  LinearArgs a = {};      // all clips in one launch: row m = position m % n of clip m / n
  p->cond_lin[j].fill(a);
  a.seg[0].x = addr_static(call.cond[j] + t0 * K);
  a.seg[0].ld = K;
  a.M = (int)M; a.tau_ptr = nullptr; a.tau_off = 0;
  a.epilogue = EPI_STORE; a.act = ACT_NONE;
  a.out = addr_static(out + col);
  a.out_ld = p->C1;
  a.row_group = (int)n; a.x_group_stride = call.cond_rs[j]; a.out_group_stride = out_clip_stride * p->C1;
  return launch_linear(a, st);
}

// persistent mode: blocks of up to kCondBlock positions, each = (conditioning projection GEMMs) +
// (clear of the hand-off words) + ONE kernel that runs every step of the block
static int run_persistent(mmk_wavenet_plan* p, const WnCall& call, int64_t tau0, int64_t n, bool with_head, hipStream_t st) {
  const mmk_wavenet_config& c = p->cfg;
  for (int64_t done = 0; done < n;) {
    // (the layer pipeline without conditioning has no block to prepare: one launch reads its weights once for up to 2^20 steps)
    const int64_t block = (p->lpipe && p->C1 == 0) ? ((int64_t)1 << 20) : (int64_t)p->kCondBlock;
    const int64_t nb = (n - done) < block ? (n - done) : block;
    const int64_t tau_b = tau0 + done;
    if (p->C1 > 0) {
      // c[b, tau, :] = LinearIO(cond[b, tau, :]) for the block's positions (modules/io.py:115-122)
      g_prof_tag = 2;
      for (int j = 0, col = 0; j < p->n_cond; col += c.cond_dim[j], ++j)      // (two inputs - stage pipeline only: side by side in a row of C1)
        MMK_TRY(project_cond(p, call, j, col, tau_b, nb, p->cproj, p->kCondBlock, st));
      // every layer's conv_1x1(c) for the same positions (wavenet_v2.py:140-150): off the per-sample chain; one
      // GEMM over all clips (M = positions, N = L x 2C, K = C1)
      if (!p->spipe) MMK_TRY(launch_gemm_f32(p->cproj, p->C1, (int64_t)p->kCondBlock * p->C1, p->cond_all.Wp, p->cond_all.n_tiles,
                              p->cond_all.k_chunks, p->cond_all.N, p->C1, p->condall, (int64_t)p->L * 2 * p->C,
                              (int64_t)p->kCondBlock * p->L * 2 * p->C, (int)nb, call.M, st));
    }
    // hand-off words are zeroed before EVERY launch (epochs restart at 1); the error word after them is sticky
    MMK_HIP(hipMemsetAsync(p->gran_h, 0, (size_t)(p->gran_words - 2) * sizeof(unsigned long long), st));
    const char* stamp_env = diag_only("MMK_WN_STAMPS");
    if (p->lpipe) {
      if (!with_head) return fail(MMK_ERR_STATE, "wavenet: the layer-pipeline kernel has no teacher-forced mode (warm-up is a prefill)");
      MMK_HIP(hipMemsetAsync(p->lp_xg, 0, (size_t)p->lp_gran_words * sizeof(unsigned long long), st));
      WnLpipeArgs k = {};
      k.B = call.M; k.L = p->L;
      wn_lpipe_split(p->L, k.first);
      k.learn_temp = c.learn_temp; k.min_temp = c.min_temp;
      k.t0 = tau_b + 1; k.n_steps = nb;
      k.layers = p->layer_tab;
      for (int l = 0; l < p->L; ++l) { k.hist[l] = p->hist[l]; k.ring[l] = p->ring[l]; }
      k.Bmax = p->Bmax;
      k.condall = p->C1 > 0 ? p->condall : nullptr; k.cond_steps = p->kCondBlock;
      k.kcA = 2 * (p->C / 16) + (int)round_up(p->C1, 16) / 16;
      k.emb = p->emb; k.idx = (int64_t*)call.in0; k.idx_rs = call.in0_rs;
      k.fc0_wp = p->lp_mlp0.Wp; k.fc0_bias = p->lp_mlp0.bias; k.fc2_wp = p->lp_mlp1.Wp; k.fc2_bias = p->lp_mlp1.bias;
      k.temperature = call.temperature;
      k.uniforms = call.uniforms ? call.uniforms + done : nullptr;
      k.uni_ld = call.uni_ld;
      k.logits_out = p->sp_logits; k.logits_ld = mmk_wavenet_plan::kSpLogitsLd;
      k.xg = p->lp_xg; k.cg = p->lp_cg; k.err_flag = p->err_flag;
      k.xcc_ids = reinterpret_cast<unsigned*>(p->lp_cg + (int64_t)p->Bmax * 16);
      {
        const char* xe = p->tune.get("MMK_WN_XCD_LOCAL");
        k.xcd_local = !(xe && xe[0] == '0');
      }
      MMK_TRY(launch_wavenet_lpipe(k, st));
      done += nb;
      continue;
    }
    if (p->bpipe) {
      if (!with_head) return fail(MMK_ERR_STATE, "wavenet: the stage-pipeline kernels have no teacher-forced mode (warm-up is a prefill)");
      MMK_HIP(hipMemsetAsync(p->bp_msg, 0xFF, (size_t)wn_bpipe_msg_words(p->L, call.M) * sizeof(unsigned), st));      // every word of THIS call's groups "not arrived" (the kernel lays the blocks out for them)
      WnBpipeArgs k = {};
      k.B = call.M; k.L = p->L; k.C1 = p->C1;
      k.learn_temp = c.learn_temp; k.min_temp = c.min_temp; k.Bmax = p->Bmax;
      k.t0 = tau_b + 1; k.n_steps = nb;
      k.img = p->bp_img; k.cst = p->bp_cst;
      k.head_w0 = p->sp_head_w0; k.head_b0 = p->sp_head_b0; k.fc2_w = p->sp_fc2p; k.fc2_b = p->sp_fc2bp;
      for (int l = 0; l < p->L; ++l) { k.hist[l] = p->hist[l]; k.ring[l] = p->ring[l]; k.dil[l] = p->dil[l]; }
      k.emb = p->emb; k.idx = (int64_t*)call.in0; k.idx_rs = call.in0_rs;
      k.cproj = p->cproj; k.cond_steps = p->kCondBlock;
      k.temperature = call.temperature;
      k.uniforms = call.uniforms ? call.uniforms + done : nullptr;
      k.uni_ld = call.uni_ld;
      k.logits_out = p->sp_logits; k.logits_ld = mmk_wavenet_plan::kSpLogitsLd;
      k.msg = p->bp_msg; k.xcd_count = p->xcd_count; k.err_flag = p->err_flag;
      k.stamps = (stamp_env && stamp_env[0] == '1') ? reinterpret_cast<unsigned long long*>(p->tau + 8) : nullptr;
      k.stamp_stage = diag_only("MMK_WN_STAMP_STAGE") ? atoi(diag_only("MMK_WN_STAMP_STAGE")) : 1;
      MMK_TRY(launch_wavenet_bpipe(k, st));
      done += nb;
      continue;
    }
    if (p->spipe) {
      if (!with_head) return fail(MMK_ERR_STATE, "wavenet: the stage-pipeline kernel has no teacher-forced mode (warm-up is a prefill)");
      // every message word starts as poison (0xFFFFFFFF): "not arrived"
      MMK_HIP(hipMemsetAsync(p->sp_msg, 0xFF, (size_t)(wn_spipe_msg_words(p->L, p->C, p->Bmax) + wn_spipe_hidmsg_words(p->L, p->Bmax) + wn_spipe_hidgrp_words(p->Bmax)) * sizeof(unsigned), st));
      WnSpipeArgs k = {};
      k.B = call.M; k.L = p->L; k.C = p->C; k.C1 = p->C1;
      k.learn_temp = c.learn_temp; k.min_temp = c.min_temp; k.Bmax = p->Bmax;
      k.t0 = tau_b + 1; k.n_steps = nb;
      k.img_chain = p->sp_img_chain; k.img_helper = p->sp_img_helper; k.cst_chain = p->sp_cst_chain; k.cst_helper = p->sp_cst_helper;
      k.head_w0 = p->sp_head_w0; k.head_b0 = p->sp_head_b0; k.fc2_w = p->sp_fc2p; k.fc2_b = p->sp_fc2bp;
      for (int l = 0; l < p->L; ++l) { k.hist[l] = p->hist[l]; k.ring[l] = p->ring[l]; k.dil[l] = p->dil[l]; }
      k.emb = p->emb; k.idx = (int64_t*)call.in0; k.idx_rs = call.in0_rs;
      k.cproj = p->cproj; k.cond_steps = p->kCondBlock;
      k.temperature = call.temperature;
      k.uniforms = call.uniforms ? call.uniforms + done : nullptr;
      k.uni_ld = call.uni_ld;
      k.logits_out = p->sp_logits; k.logits_ld = mmk_wavenet_plan::kSpLogitsLd;      // (256 classes + the temperature: mmk_wavenet_last_logits picks the network's own columns)
      k.msg = p->sp_msg; k.hidmsg = p->sp_hidmsg; k.hidgrp = p->sp_hidgrp; k.xcd_count = p->xcd_count; k.err_flag = p->err_flag;
      k.stamps = (stamp_env && stamp_env[0] == '1') ? reinterpret_cast<unsigned long long*>(p->tau + 8) : nullptr;
      k.stamp_stage = diag_only("MMK_WN_STAMP_STAGE") ? atoi(diag_only("MMK_WN_STAMP_STAGE")) : 1;
      k.dbg = (k.stamps && diag_only("MMK_WN_SPIPE_DBG")) ? atoi(diag_only("MMK_WN_SPIPE_DBG")) : 0;
      { const char* pe = p->tune.get("MMK_WN_SPIPE_PAIR"); k.pair = pe ? (pe[0] == '1' ? 1 : 0) : -1; }
      p->last_pair = wn_spipe_pair_form(k.B, k.pair) ? 1 : 0;
      MMK_TRY(launch_wavenet_spipe(k, st));
      done += nb;
      continue;
    }
    if (p->chain && with_head) {
      WnChainArgs k = {};
      k.B = call.M; k.Gc = p->Gc; k.Gn = p->Gn; k.Mg = p->Mg;
      k.L = p->L; k.C = p->C; k.C1 = p->C1;
      k.q_levels = c.q_levels; k.H1 = c.mlp_hidden; k.n_classes = c.out_dim; k.n_logits_pad = p->n_logits_pad;
      k.learn_temp = c.learn_temp; k.min_temp = c.min_temp;
      k.xcd_local = p->xcd_local ? 1 : 0;
      k.t0 = tau_b + 1; k.n_steps = nb;
      k.iters = p->iter_tab; k.ring_floats_per_wg = p->ring_floats_per_wg;
      k.emb = p->emb; k.idx = (int64_t*)call.in0; k.idx_rs = call.in0_rs;
      k.condall = p->condall; k.cond_steps = p->kCondBlock; k.zeros = p->zero_pad;
      k.fc0_wp = p->mlp[0].Wp; k.fc0_bias = p->mlp[0].bias; k.fc2_wp = p->mlp[1].Wp; k.fc2_bias = p->mlp[1].bias;
      k.temperature = call.temperature;
      k.uniforms = call.uniforms ? call.uniforms + done : nullptr;
      k.uni_ld = call.uni_ld;
      k.logits_out = p->logits; k.logits_ld = p->logits_ld;
      k.gran_h = p->gran_h; k.gran_y = p->gran_y; k.gran_skip = p->gran_skip; k.gran_hid = p->gran_hid;
      k.gran_logit = p->gran_logit; k.gran_idx = p->gran_idx;
      k.h_rings = p->h_rings; k.err_flag = p->err_flag; k.xcd_count = p->xcd_count;
      k.stamps = (stamp_env && stamp_env[0] == '1') ? reinterpret_cast<unsigned long long*>(p->tau + 8) : nullptr;
      MMK_TRY(launch_wavenet_chain(k, st));
      done += nb;
      continue;
    }
    WnPersistArgs a = {};
    a.B = call.M; a.Gc = p->Gc; a.Gn = p->Gn; a.Mg = p->Mg;
    a.L = p->L; a.C = p->C; a.S = p->S; a.C1 = p->C1;
    a.kcA = 2 * (p->C / 16) + p->C1 / 16;
    a.q_levels = c.q_levels; a.H1 = c.mlp_hidden; a.n_classes = c.out_dim; a.n_logits_pad = p->n_logits_pad;
    a.learn_temp = c.learn_temp; a.min_temp = c.min_temp;
    a.teacher_forced = with_head ? 0 : 1;
    { const char* sm = p->tune.get("MMK_WN_SMALL"); a.force_tiles = (sm && sm[0] == '0') ? 1 : 0; }
    a.tf_end = tau0 + n;   // warm-up: generation starts consuming here
    a.xcd_local = p->xcd_local ? 1 : 0;
    a.xcd_count = p->xcd_count;
    a.t0 = tau_b + 1; a.n_steps = nb;
    a.layers = p->layer_tab; a.ring_floats_per_wg = p->ring_floats_per_wg;
    a.emb = p->emb; a.idx = (int64_t*)call.in0; a.idx_rs = call.in0_rs;
    a.condall = p->condall; a.cond_steps = p->kCondBlock; a.zeros = p->zero_pad;
    a.fc0_wp = p->mlp[0].Wp; a.fc0_bias = p->mlp[0].bias; a.fc2_wp = p->mlp[1].Wp; a.fc2_bias = p->mlp[1].bias;
    a.temperature = call.temperature;
    a.uniforms = call.uniforms ? call.uniforms + done : nullptr;   // column s of this block = done + s
    a.uni_ld = call.uni_ld;
    a.logits_out = p->logits; a.logits_ld = p->logits_ld;
    a.gran_h = p->gran_h; a.gran_y = p->gran_y; a.gran_skip = p->gran_skip; a.gran_hid = p->gran_hid;
    a.gran_logit = p->gran_logit; a.gran_idx = p->gran_idx;
    a.h_rings = p->h_rings; a.err_flag = p->err_flag;
    {
      const char* senv = diag_only("MMK_WN_STAMPS");
      a.stamps = (senv && senv[0] == '1') ? reinterpret_cast<unsigned long long*>(p->tau + 8) : nullptr;
    }
    MMK_TRY(launch_wavenet_persist(a, st));
    done += nb;
  }
  return MMK_OK;
}

// Warm-up of the persistent path as a prefill: all positions of [t_begin, t_end) go through a layer at once (GEMMs
// with the gate / residual epilogues), then the tail of every layer input is copied into the step kernel's rings.
// The same staircase as the step path: layer l only runs where its output is still needed at t_end.
static int prefill(mmk_wavenet_plan* p, const WnCall& call, int64_t t_begin, int64_t t_end, hipStream_t st) {
  const mmk_wavenet_config& c = p->cfg;
  const int C = p->C, L = p->L, C1 = p->C1, B = call.M;
  const int64_t n = t_end - t_begin, P = p->pf_P;
  MMK_TRY(launch_wn_prefill_embed((const int64_t*)call.in0, call.in0_rs, t_begin, p->emb, c.q_levels, C, (int)n, p->pf_h[0],
                                  P * C, B, st));
  if (C1 > 0) {
    for (int j = 0, col = 0; j < p->n_cond; col += c.cond_dim[j], ++j)      // c[b, t, :] = LinearIO(cond[b, t, :]); two inputs' rows side by side
      MMK_TRY(project_cond(p, call, j, col, t_begin, n, p->pf_c, P, st));
  }
  std::vector<int64_t> sfx(L, 0);   // sfx[l] = sum of the dilations above layer l
  for (int l = L - 2; l >= 0; --l) sfx[l] = sfx[l + 1] + p->dil[l + 1];
  int cur = 0;
  for (int l = 0; l < L; ++l) {
    const int d = p->dil[l];
    // the ring of layer l holds its input at the last d positions
    const int64_t t_lo = t_end - d > t_begin ? t_end - d : t_begin;
    if (p->lpipe || p->spipe)  // the launch path's rings: [slot][Bmax][C]
      MMK_TRY(launch_wn_lpipe_scatter(p->pf_h[cur], P * C, t_begin, t_lo, (int)(t_end - t_lo), C, B, p->Bmax, p->hist[l], p->ring[l], st));
    else
      MMK_TRY(launch_wn_prefill_scatter(p->pf_h[cur], P * C, t_begin, t_lo, (int)(t_end - t_lo), C, B, p->Mg, p->Gc, p->Gn, p->h_rings,
                                        p->ring_floats_per_wg, p->ring_offset[l], p->ring_mask[l], st));
    if (l == L - 1) break;
    int64_t r_lo = t_end - sfx[l];                       // first position whose output is still needed
    if (r_lo < t_begin + d) r_lo = t_begin + d;          // the delayed input must exist
    if (r_lo >= t_end) break;
    const int64_t p0 = r_lo - t_begin;
    const int M = (int)(t_end - r_lo);
    const float* h = p->pf_h[cur];
    {
      WnPrefillArgs a = {};
      a.M = M; a.N = 2 * C; a.n_tiles = p->A[l].n_tiles; a.k_chunks = p->A[l].k_chunks;
      a.nseg = C1 > 0 ? 3 : 2;
      a.seg_k[0] = C; a.seg[0] = h + (p0 - d) * C; a.seg_ld[0] = C; a.seg_batch[0] = P * C;
      a.seg_k[1] = C; a.seg[1] = h + p0 * C; a.seg_ld[1] = C; a.seg_batch[1] = P * C;
      if (C1 > 0) { a.seg_k[2] = C1; a.seg[2] = p->pf_c + p0 * C1; a.seg_ld[2] = C1; a.seg_batch[2] = P * C1; }
      a.wp = p->A[l].Wp; a.bias = p->A[l].bias;
      a.out = p->pf_y + p0 * C; a.out_ld = C; a.out_batch = P * C;
      MMK_TRY(launch_wn_prefill(a, 0, B, st));
    }
    {
      WnPrefillArgs a = {};
      a.M = M; a.N = C; a.n_tiles = C / 16; a.k_chunks = p->Bm[l].k_chunks;
      a.nseg = 1;
      a.seg_k[0] = C; a.seg[0] = p->pf_y + p0 * C; a.seg_ld[0] = C; a.seg_batch[0] = P * C;
      a.wp = p->Bm[l].Wp; a.bias = p->Bm[l].bias;     // rows [res ; skip]: the first C packed rows
      a.res_in = h + p0 * C; a.res_ld = C; a.res_batch = P * C;
      a.out = p->pf_h[cur ^ 1] + p0 * C; a.out_ld = C; a.out_batch = P * C;
      MMK_TRY(launch_wn_prefill(a, 1, B, st));
    }
    cur ^= 1;
  }
  return MMK_OK;
}

// run n steps starting at position tau0 (device counter is set here)
static int run_steps(mmk_wavenet_plan* p, const WnCall& call, int64_t tau0, int64_t n, bool with_head, hipStream_t st) {
  if (n <= 0) return MMK_OK;
  if (p->persistent) return run_persistent(p, call, tau0, n, with_head, st);
  MMK_TRY(launch_set_i64(p->tau, tau0, st));
  int64_t done = 0;
  if (n >= 2 * kGraphSteps) {
    std::vector<int64_t> key = {call.M, (int64_t)(uintptr_t)call.in0, call.in0_rs, with_head ? 1 : 0,
                                (int64_t)(uintptr_t)call.temperature, (int64_t)(uintptr_t)call.uniforms, call.uni_ld,
                                call.uni_off};
    for (int j = 0; j < p->n_cond; ++j) {
      key.push_back((int64_t)(uintptr_t)call.cond[j]);
      key.push_back(call.cond_rs[j]);
    }
    if (!p->gc.exec || p->gc.key != key) {
      MMK_HIP(hipStreamSynchronize(st));  // a cached graph may still be in flight
      p->gc.reset();
      MMK_HIP(hipStreamBeginCapture(p->cap_stream, hipStreamCaptureModeThreadLocal));
      int rc = MMK_OK;
      for (int s = 0; s < kGraphSteps && rc == MMK_OK; ++s) rc = emit_step(p, call, s, with_head, p->cap_stream);
      if (rc == MMK_OK) rc = launch_bump(p->tau, kGraphSteps, p->cap_stream);
      hipGraph_t g = nullptr;
      hipError_t e = hipStreamEndCapture(p->cap_stream, &g);
      if (rc != MMK_OK) {
        if (g) (void)hipGraphDestroy(g);
        return rc;
      }
      if (e != hipSuccess) return fail(MMK_ERR_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e));
      p->gc.graph = g;
      MMK_HIP(hipGraphInstantiate(&p->gc.exec, g, nullptr, nullptr, 0));
      p->gc.key = key;
      p->gc.steps = kGraphSteps;
    }
    const int64_t reps = n / kGraphSteps;
    for (int64_t r = 0; r < reps; ++r) MMK_HIP(hipGraphLaunch(p->gc.exec, st));
    done = reps * kGraphSteps;
  }
  for (int64_t s = done; s < n; ++s) MMK_TRY(emit_step(p, call, s - done, with_head, st));
  if (n > done) MMK_TRY(launch_bump(p->tau, n - done, st));
  return MMK_OK;
}

static int check_call(mmk_wavenet_plan* p, int32_t batch, const void* in0, const void* const* cond,
                      const int64_t* cond_rs, WnCall& call) {
  if (!p) return fail(MMK_ERR_INVALID, "wavenet: null plan");
  if (!p->committed) return fail(MMK_ERR_STATE, "wavenet: plan not committed (bind weights, then mmk_wavenet_commit)");
  if (batch < 1 || batch > p->Bmax) return fail(MMK_ERR_INVALID, "wavenet: batch %d outside [1, %d]", batch, p->Bmax);
  if (!in0) return fail(MMK_ERR_INVALID, "wavenet: null input");
  if (p->n_cond > 0 && (!cond || !cond_rs)) return fail(MMK_ERR_INVALID, "wavenet: %d conditioning inputs expected", p->n_cond);
  call.M = batch;
  call.in0 = in0;
  for (int j = 0; j < p->n_cond; ++j) {
    if (!cond[j]) return fail(MMK_ERR_INVALID, "wavenet: conditioning input %d is null", j);
    call.cond[j] = static_cast<const float*>(cond[j]);
    call.cond_rs[j] = cond_rs[j];
  }
  return MMK_OK;
}

extern "C" int mmk_wavenet_warmup(mmk_wavenet_plan* p, int32_t batch, const void* in0, int64_t in0_row_stride,
                                  const void* const* cond, const int64_t* cond_row_stride, int64_t t_begin,
                                  int64_t t_end, mmk_stream_t stream) {
  WnCall call;
  MMK_TRY(check_call(p, batch, in0, cond, cond_row_stride, call));
  call.in0_rs = in0_row_stride;
  if (t_begin < 0 || t_end < t_begin) return fail(MMK_ERR_INVALID, "wavenet_warmup: bad range [%lld, %lld)", (long long)t_begin, (long long)t_end);
  const char* penv = p->tune.get("MMK_WN_PREFILL");
  if (p->lpipe || p->spipe) {
    // the stage-owned rings are only filled by the prefill; a longer window than the receptive field adds nothing to the
    // ring entries generation reads (each is determined by the rf - 1 positions before t_end)
    if (t_end - t_begin > p->rf - 1) t_begin = t_end - (p->rf - 1);
    return t_end > t_begin ? prefill(p, call, t_begin, t_end, (hipStream_t)stream) : MMK_OK;
  }
  if (p->persistent && t_end > t_begin && t_end - t_begin <= p->pf_P && !(penv && penv[0] == '0'))
    return prefill(p, call, t_begin, t_end, (hipStream_t)stream);
  return run_steps(p, call, t_begin, t_end - t_begin, false, (hipStream_t)stream);
}

extern "C" int mmk_wavenet_generate(mmk_wavenet_plan* p, int32_t batch, void* in0, int64_t in0_row_stride,
                                    const void* const* cond, const int64_t* cond_row_stride, int64_t t0,
                                    int64_t n_steps, const float* temperature, const float* uniforms,
                                    mmk_stream_t stream) {
  WnCall call;
  MMK_TRY(check_call(p, batch, in0, cond, cond_row_stride, call));
  call.in0_rs = in0_row_stride;
  if (t0 < 1 || n_steps < 0) return fail(MMK_ERR_INVALID, "wavenet_generate: bad t0/n_steps");
  if (temperature && !uniforms) return fail(MMK_ERR_INVALID, "wavenet_generate: temperature given without uniforms");
  if (temperature && p->cfg.head_kind != 0) return fail(MMK_ERR_INVALID, "wavenet_generate: this head has no sampler");
  call.temperature = temperature;
  call.uniforms = uniforms;
  call.uni_ld = n_steps;
  call.uni_off = -(t0 - 1);
  // the step that writes position t consumes position tau = t-1 as its newest input
  return run_steps(p, call, t0 - 1, n_steps, true, (hipStream_t)stream);
}

extern "C" int mmk_wavenet_last_logits(mmk_wavenet_plan* p, int32_t batch, float* out, int64_t ld, mmk_stream_t stream) {
  if (!p || !out) return fail(MMK_ERR_INVALID, "wavenet_last_logits: null argument");
  if (!p->committed) return fail(MMK_ERR_STATE, "wavenet_last_logits: plan not committed");
  if (p->cfg.head_kind != 0) return fail(MMK_ERR_UNSUPPORTED, "wavenet_last_logits: only for the MLP head");
  const int n = p->cfg.out_dim + (p->cfg.learn_temp ? 1 : 0);
  if (p->spipe || p->lpipe) {      // these kernels' head writes 256 classes + the temperature at column 256: the network's classes, then its temperature
    constexpr int PQ = mmk_wavenet_plan::kSpQ, PLD = mmk_wavenet_plan::kSpLogitsLd;
    MMK_HIP(hipMemcpy2DAsync(out, ld * sizeof(float), p->sp_logits, PLD * sizeof(float), p->cfg.out_dim * sizeof(float), batch, hipMemcpyDeviceToDevice,
                             (hipStream_t)stream));
    if (p->cfg.learn_temp)
      MMK_HIP(hipMemcpy2DAsync(out + p->cfg.out_dim, ld * sizeof(float), p->sp_logits + PQ, PLD * sizeof(float), sizeof(float), batch,
                               hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return MMK_OK;
  }
  MMK_HIP(hipMemcpy2DAsync(out, ld * sizeof(float), p->logits, p->logits_ld * sizeof(float), n * sizeof(float), batch,
                           hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return MMK_OK;
}

extern "C" int mmk_wavenet_last_logits_of(mmk_wavenet_plan* p, int32_t target, int32_t batch, float* out, int64_t ld, mmk_stream_t stream) {
  if (target == 0) return mmk_wavenet_last_logits(p, batch, out, ld, stream);
  if (!p || !out) return fail(MMK_ERR_INVALID, "wavenet_last_logits_of: null argument");
  if (!p->committed) return fail(MMK_ERR_STATE, "wavenet_last_logits_of: plan not committed");
  if (target < 0 || target >= p->n_tgt) return fail(MMK_ERR_INVALID, "wavenet_last_logits_of: target %d of %d", target, p->n_tgt);
  const WnHead& h = p->xheads[target - 1];
  const int n = h.q + (h.learn_temp ? 1 : 0);
  MMK_HIP(hipMemcpy2DAsync(out, ld * sizeof(float), h.logits, h.logits_ld * sizeof(float), n * sizeof(float), batch,
                           hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return MMK_OK;
}

extern "C" int mmk_wavenet_profile_steps(mmk_wavenet_plan* p, int32_t batch, void* in0, int64_t in0_row_stride,
                                         const void* const* cond, const int64_t* cond_row_stride, int64_t t0,
                                         int64_t n_steps, double* ms_total, int64_t* launches, mmk_stream_t stream) {
  WnCall call;
  MMK_TRY(check_call(p, batch, in0, cond, cond_row_stride, call));
  if (!ms_total || !launches || t0 < 1 || n_steps < 1) return fail(MMK_ERR_INVALID, "wavenet_profile_steps: bad arguments");
  if (p->persistent) return fail(MMK_ERR_UNSUPPORTED, "wavenet_profile_steps: the plan runs one persistent kernel per call; time mmk_wavenet_generate instead");
  call.in0_rs = in0_row_stride;
  hipStream_t st = (hipStream_t)stream;
  std::vector<ProfRecord> records;
  records.reserve((size_t)n_steps * (2 * p->L + 8));
  MMK_TRY(launch_set_i64(p->tau, t0 - 1, st));
  g_prof = &records;
  int rc = MMK_OK;
  for (int64_t s = 0; s < n_steps && rc == MMK_OK; ++s) rc = emit_step(p, call, s, true, st);
  g_prof = nullptr;
  if (rc == MMK_OK) rc = launch_bump(p->tau, n_steps, st);
  hipError_t e = hipStreamSynchronize(st);
  for (int k = 0; k < 3; ++k) { ms_total[k] = 0.0; launches[k] = 0; }
  for (auto& r : records) {
    float ms = 0.f;
    if (e == hipSuccess && hipEventElapsedTime(&ms, r.start, r.stop) == hipSuccess && r.tag >= 0 && r.tag < 3) {
      ms_total[r.tag] += ms;
      launches[r.tag] += 1;
    }
    (void)hipEventDestroy(r.start);
    (void)hipEventDestroy(r.stop);
  }
  if (e != hipSuccess) return fail(MMK_ERR_HIP, "wavenet_profile_steps: %s", hipGetErrorString(e));
  return rc;
}

extern "C" int mmk_wavenet_mode(const mmk_wavenet_plan* p) {
  return (p && p->persistent) ? (p->bpipe ? 6 : p->spipe ? 5 : (p->lpipe ? 4 : (p->chain ? 2 : 1))) : 0;
}

extern "C" int mmk_wavenet_pair_visits(const mmk_wavenet_plan* p) { return p ? p->last_pair : 0; }

extern "C" int mmk_wavenet_inject_sync_error(mmk_wavenet_plan* p, mmk_stream_t stream) {
  if (!p) return fail(MMK_ERR_INVALID, "wavenet_inject_sync_error: null plan");
  if (!p->committed) return fail(MMK_ERR_STATE, "wavenet_inject_sync_error: plan not committed");
  if (!p->persistent) return MMK_OK;      // (the launch path has no hand-off that could time out)
  const int32_t one = 1;                  // the word a timed-out wait inside the kernel sets
  MMK_HIP(hipMemcpyAsync(p->err_flag, &one, sizeof(one), hipMemcpyHostToDevice, (hipStream_t)stream));
  MMK_HIP(hipStreamSynchronize((hipStream_t)stream));
  return MMK_OK;
}

extern "C" int mmk_wavenet_sync_status(mmk_wavenet_plan* p, mmk_stream_t stream) {
  if (!p) return fail(MMK_ERR_INVALID, "wavenet_sync_status: null plan");
  MMK_HIP(hipStreamSynchronize((hipStream_t)stream));
  if (!p->committed || !p->persistent) return MMK_OK;
  int32_t flag = 0;
  MMK_HIP(hipMemcpy(&flag, p->err_flag, sizeof(flag), hipMemcpyDeviceToHost));
  if (flag == 2)
    return fail(MMK_ERR_STATE, "wavenet: the persistent kernel's workgroups were not spread 8 x %d over the XCDs; rerun with MMK_WN_XCD_LOCAL=0 (agent-scope hand-offs)", p->Gn);
  if (flag != 0)
    return fail(MMK_ERR_STATE, "wavenet: a hand-off inside the persistent kernel timed out (error word 0x%x) - its workgroups were not all resident (another kernel "
                "holding CUs?); the samples of this call are invalid, rerun it (MMK_WN_PERSISTENT=0 selects the per-layer launch path)", (unsigned)flag);
  {
    const char* senv = diag_only("MMK_WN_STAMPS");
    if (senv && senv[0] == '1') {
      unsigned long long st[256];
      MMK_HIP(hipMemcpy(st, p->tau + 8, p->spipe ? sizeof(st) : 24 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      if (p->bpipe) {      // (wavenet_bpipe.hip: the marks of chain wave 0 and helper wave 4 of CU 0 of stage MMK_WN_STAMP_STAGE, totals of the last launch)
        fprintf(stderr, "[mmk stamps] batched stage pipeline, us per mark - chain wave 0:");
        for (int k = 0; k < 16; ++k) fprintf(stderr, " [%d]=%.0f", k, (double)st[k] * 0.01);
        fprintf(stderr, "\n[mmk stamps] helper wave 4:");
        for (int k = 0; k < 16; ++k) fprintf(stderr, " [%d]=%.0f", k, (double)st[16 + k] * 0.01);
        {
          static const char* role[4] = {"chain wave 0", "chain wave 2", "helper wave 4", "helper wave 6"};
          const unsigned long long t0 = st[160];
          for (int r = 0; r < 4; ++r) {
            fprintf(stderr, "\n[mmk trace] %s, three visits in the middle of the launch, marks 0 .. 7 in 10 ns ticks from chain wave 0's first:", role[r]);
            for (int v = 0; v < 3; ++v) {
              fprintf(stderr, " |");
              for (int k = 0; k < 8; ++k) fprintf(stderr, " %lld", (long long)(st[160 + 24 * r + 8 * v + k] - t0));
            }
          }
        }
        fprintf(stderr, "\n[mmk stamps] group 0, last step, per stage [y of the stage below stored -> y complete here | -> y stored], 10 ns ticks:");
        for (int l = 1; l < p->L; ++l) fprintf(stderr, " %d:[%lld|%lld]", l, (long long)(st[64 + 2 * l] - st[64 + 2 * (l - 1) + 1]), (long long)(st[64 + 2 * l + 1] - st[64 + 2 * l]));
        fprintf(stderr, "\n[mmk stamps] the same group: early products out minus y of the stage below stored (negative: the early half was waiting for y), 10 ns ticks:");
        for (int l = 1; l < p->L; ++l) fprintf(stderr, " %d:%lld", l, (long long)(st[128 + l] - st[64 + 2 * (l - 1) + 1]));
        fprintf(stderr, "\n");
      } else if (p->spipe) {
        fprintf(stderr, "[mmk stamps] last stage-pipeline launch, chain wave 0 of CU 0 of stage MMK_WN_STAMP_STAGE (default 1), shader cycles per visit: "
                        "wait for the message=%.0f; products + gate + publish=%.0f; poison + ring store=%.0f; visits=%llu\n",
                st[3] ? (double)st[0] / (double)st[3] : 0.0, st[3] ? (double)st[1] / (double)st[3] : 0.0, st[3] ? (double)st[2] / (double)st[3] : 0.0, st[3]);
        fprintf(stderr, "[mmk stamps] helper wave 0 of that CU, cycles per iteration: rows requested=%.0f; wait for the message=%.0f; hidden-unit products=%.0f; "
                        "their hand-over (rest of the poll + store)=%.0f; rows landed=%.0f; bias products=%.0f\n",
                st[3] ? (double)st[6] / (double)st[3] : 0.0, st[3] ? (double)st[7] / (double)st[3] : 0.0, st[3] ? (double)st[8] / (double)st[3] : 0.0,
                st[3] ? (double)st[9] / (double)st[3] : 0.0, st[3] ? (double)st[10] / (double)st[3] : 0.0, st[3] ? (double)st[11] / (double)st[3] : 0.0);
        if (const char* d = diag_only("MMK_WN_SPIPE_DBG"); d && (atoi(d) & 4)) {
          fprintf(stderr, "[mmk stamps] FREE RUN (nothing waits for a message; wrong results): ns per visit of stage s when its inbox is never empty:");
          for (int l = 0; l <= p->L; ++l) fprintf(stderr, " %d:%.0f", l, st[3] ? 10.0 * (double)(st[144 + l] - st[182 + l]) / (double)st[3] * (l == p->L ? 8.0 : 1.0) : 0.0);
          fprintf(stderr, " (the last entry: the head, one of its eight workgroups x 8)\n");
        }
        fprintf(stderr, "[mmk stamps] extra looks per visit=%.2f; clip 0, last step, publish time of stage s minus stage s - 1 in 10 ns ticks:",
                st[3] ? (double)st[4] / (double)st[3] : 0.0);
        for (int l = 1; l < p->L; ++l) fprintf(stderr, " %lld", (long long)(st[16 + l] - st[16 + l - 1]));
        fprintf(stderr, "\n[mmk stamps] per stage [publish of the stage below -> my quarter seen | -> all quarters + flags | -> my publish]:");
        for (int l = 1; l < p->L; ++l)
          fprintf(stderr, " %d:[%lld %lld %lld]", l, (long long)(st[64 + l] - st[16 + l - 1]), (long long)(st[112 + l] - st[64 + l]), (long long)(st[16 + l] - st[112 + l]));
        fprintf(stderr, " | head: last layer -> class of the step before the last: n/a; step before last's class -> last step's stage 0 publish=%lld; "
                        "last stage publish -> head done=%lld\n",
                (long long)(st[16] - st[16 + p->L + 1]), (long long)(st[16 + p->L + 2] - st[16 + p->L - 1]));
        {   // the last step of every clip: mean over the clips of [publish of the stage below -> seen | seen -> my publish], 10 ns ticks
          std::vector<unsigned long long> all(256 + 2048 + 256);
          MMK_HIP(hipMemcpy(all.data(), p->tau + 8, all.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
          fprintf(stderr, "[mmk stamps] the middle step of the launch, mean over the clips, per stage [publish of the stage below -> message seen | seen -> my publish]:");
          double tot = 0;
          for (int l = 1; l < p->L && l < 32; ++l) {
            double a = 0, b = 0;
            int n = 0;
            for (int cidx = 0; cidx < 32 && cidx < p->Bmax; ++cidx) {
              const unsigned long long pub0 = all[256 + (l - 1) * 32 + cidx], seen = all[256 + 1024 + l * 32 + cidx], pub1 = all[256 + l * 32 + cidx];
              if (!pub0 || !seen || !pub1) continue;
              a += (double)((long long)(seen - pub0)); b += (double)((long long)(pub1 - seen)); ++n;
            }
            if (n) { fprintf(stderr, " %d:[%.0f %.0f]", l, a / n, b / n); tot += (a + b) / n; }
          }
          fprintf(stderr, " | sum %.0f\n", tot);
          {   // head workgroup 0, thread 0: shader cycles per visit in the phases of the head
            const double nv = (double)((p->Bmax + 7) / 8) * (double)(st[3] / (unsigned long long)(p->Bmax > 0 ? p->Bmax : 1));
            if (nv > 0)
              fprintf(stderr, "[mmk stamps] head, cycles per visit of workgroup 0: XCD sums collected=%.0f; wait for y + staging + barrier=%.0f; products + last sum + Mish + barrier=%.0f; "
                              "logits + barrier=%.0f; argmax / draw=%.0f; embedding row + publish=%.0f\n",
                      (double)all[176] / nv, (double)all[177] / nv, (double)all[178] / nv, (double)all[179] / nv, (double)all[180] / nv, (double)all[181] / nv);
          }
          fprintf(stderr, "[mmk stamps] the middle step, clip 5: publish time of a stage's CUs 1 .. 7 minus its CU 0's, 10 ns ticks:");
          for (int l = 0; l < p->L && l < 31; ++l) {
            fprintf(stderr, " %d:[", l);
            for (int q = 1; q < 8; ++q) fprintf(stderr, "%lld%s", (long long)(all[256 + 2048 + l * 8 + q] - all[256 + 2048 + l * 8]), q < 7 ? " " : "]");
          }
          fprintf(stderr, "\n");
        }
        return MMK_OK;
      }
      if (p->chain) {
        fprintf(stderr, "[mmk stamps] last chain launch, I/O wave 0 of workgroup 1, totals in ms: wait products=%.3f; epilogues + publish=%.3f; "
                        "ring store=%.3f; wait y/h=%.3f; head=%.3f [skip wait + fc0=%.3f]; step start=%.3f | matrix wave 0: requests=%.3f; "
                        "operands + MFMA=%.3f; B1 + small operands=%.3f; B1b + second half=%.3f; wait B4=%.3f; shader clock=%.0f MHz\n",
                st[0] * 1e-5, st[1] * 1e-5, st[2] * 1e-5, st[3] * 1e-5, st[6] * 1e-5, st[16] * 1e-5, st[7] * 1e-5, st[8] * 1e-5, st[9] * 1e-5,
                st[10] * 1e-5, st[11] * 1e-5, st[12] * 1e-5, st[15] ? 100.0 * (double)st[14] / (double)st[15] : 0.0);
        return MMK_OK;
      }
      const char* names[7] = {"wait phase A", "epilogue A + publish", "wait y", "wait phase B", "epilogue B + publish", "wait h'", "head"};
      const int slot[7] = {0, 1, 2, 3, 4, 5, 6};
      fprintf(stderr, "[mmk stamps] last persistent launch, I/O wave 0 of workgroup 1, totals in ms:");
      for (int i = 0; i < 7; ++i) fprintf(stderr, " %s=%.3f;", names[i], st[slot[i]] * 1e-5);
      fprintf(stderr, " [step start=%.3f; head: skip wait + fc0=%.3f, rest=%.3f]", st[7] * 1e-5, st[16] * 1e-5, st[6] * 1e-5);
      const char* mnames[6] = {"requests", "phase A", "B1", "wait y", "phase B + small operands", "wait h'"};
      fprintf(stderr, " | matrix wave 0:");
      for (int i = 0; i < 6; ++i) fprintf(stderr, " %s=%.3f;", mnames[i], st[8 + i] * 1e-5);
      fprintf(stderr, " shader clock=%.0f MHz;", st[15] ? 100.0 * (double)st[14] / (double)st[15] : 0.0);
      fprintf(stderr, "\n");
    }
  }
  return MMK_OK;
}

// Seq2SeqLSTMNetwork generate plan (host side).
//
// Reference: Seq2SeqLSTMNetwork.forward/decode (s2s_lstm_v2.py:246-253),
// EncoderLSTM.forward (:93-113, edge_sum down-sampling), DecoderLSTM.forward
// (:155-179, linear_resample up-sampling).  One generate_step maps hop input
// frames to hop output frames through two bidirectional LSTMs:
//   - input projections of all hop frames are one GEMM (M = batch*hop),
//   - the 2*hop recurrent steps are skinny GEMMs + a fused LSTM cell,
//   - the "[fwd|bwd].view(.., D, 2).sum(-1)" of the reference pairs ADJACENT
//     channels of the concatenation (:100, :174); reproduced literally.
#include "plan_util.h"
#include "lstm_step.h"
#include "lstm_seq.h"
#include "lstm_inproj.h"

using namespace mmk;


namespace {

// gather a strided (batch, hop, dim) window into rows (b*hop + t) of a padded buffer
__global__ void gather_frames_kernel(const float* __restrict__ x, int64_t bs, int64_t fs, int hop, int dim,
                                     float* __restrict__ out, int out_ld) {
  const int row = blockIdx.x;  // b*hop + t
  const int b = row / hop, t = row % hop;
  const float* src = x + b * bs + t * fs;
  float* dst = out + (int64_t)row * out_ld;
  for (int c = threadIdx.x; c < out_ld; c += blockDim.x) dst[c] = c < dim ? src[c] : 0.f;
}

// class indices (batch, hop) -> rows (b*hop + t) of the embedding table: nn.Embedding under ZipReduceVariables, whose weight for
// a single input is 1 (s2s_lstm_v2.py:205-210, modules/io.py:299-313).  An index outside the table (torch raises for it) reads
// the nearest row.
__global__ void embed_rows_kernel(const int64_t* __restrict__ idx, int64_t bs, int64_t es, int hop, int n_classes, int dim,
                                  const float* __restrict__ table, float* __restrict__ out, int out_ld) {
  const int row = blockIdx.x;  // b*hop + t
  const int b = row / hop, t = row % hop;
  int64_t k = idx[b * bs + t * es];
  k = k < 0 ? 0 : (k >= n_classes ? n_classes - 1 : k);
  const float* src = table + k * dim;
  float* dst = out + (int64_t)row * out_ld;
  for (int c = threadIdx.x; c < out_ld; c += blockDim.x) dst[c] = c < dim ? src[c] * 1.f : 0.f;
}

// scatter rows (b*hop + t) to a strided (batch, n_out<=hop, dim) destination
__global__ void scatter_frames_kernel(const float* __restrict__ in, int in_ld, int hop, int n_out, int dim,
                                      float* __restrict__ y, int64_t bs, int64_t fs) {
  const int row = blockIdx.x;
  const int b = row / hop, t = row % hop;
  if (t >= n_out) return;
  const float* src = in + (int64_t)row * in_ld;
  float* dst = y + b * bs + t * fs;
  for (int c = threadIdx.x; c < dim; c += blockDim.x) dst[c] = src[c];
}

// y'[r][c] = cat(f, b)[r][2c] + cat(f, b)[r][2c+1]                (s2s_lstm_v2.py:100)
__device__ __forceinline__ float cat_at(const float* f, const float* b, int D, int i) { return i < D ? f[i] : b[i - D]; }

// y.view(..., D, 2).sum(-1) of the [forward | backward] concatenation (:100, :174), optionally x + y (:101-104, :175-178)
__global__ void pair_sum_kernel(const float* __restrict__ of, const float* __restrict__ ob, int D, int rows,
                                const float* __restrict__ res, float* __restrict__ out) {
  const int r = blockIdx.x;
  if (r >= rows) return;
  const float* f = of + (int64_t)r * D;
  const float* b = ob + (int64_t)r * D;
  for (int c = threadIdx.x; c < D; c += blockDim.x) {
    const float y = cat_at(f, b, D, 2 * c) + cat_at(f, b, D, 2 * c + 1);
    out[(int64_t)r * D + c] = res ? res[(int64_t)r * D + c] + y : y;
  }
}

// the encoder's pooling over the hop frames of its (folded) output  (:105-113):  unfold(1, hop, hop), then
//   mode 0 edge_sum: first + last      1 edge_mean: (first + last) / 2      2 sum: all frames      3 mean: sum / hop
__global__ void pool_frames_kernel(const float* __restrict__ x, int D, int hop, int mode, float* __restrict__ out) {
  const int bidx = blockIdx.x;
  const float* x0 = x + (int64_t)(bidx * hop) * D;
  for (int c = threadIdx.x; c < D; c += blockDim.x) {
    float v = 0.f;
    if (mode < 2) {
      v = x0[c] + x0[(int64_t)(hop - 1) * D + c];
      if (mode == 1) v *= 0.5f;
    } else {
      for (int t = 0; t < hop; ++t) v += x0[(int64_t)t * D + c];
      if (mode == 3) v /= (float)hop;
    }
    out[(int64_t)bidx * D + c] = v;
  }
}

// the last encoder layer's fold and the pooling in one pass (the folded frames of that layer are read by nothing else)
__global__ void pair_sum_pool_kernel(const float* __restrict__ of, const float* __restrict__ ob, int D, int hop, int mode,
                                     const float* __restrict__ res, float* __restrict__ out) {
  const int bidx = blockIdx.x;
  for (int c = threadIdx.x; c < D; c += blockDim.x) {
    auto folded = [&](int t) -> float {
      const int64_t r = (int64_t)bidx * hop + t;
      const float y = cat_at(of + r * D, ob + r * D, D, 2 * c) + cat_at(of + r * D, ob + r * D, D, 2 * c + 1);
      return res ? res[r * D + c] + y : y;
    };
    float v = 0.f;
    if (mode < 2) {
      v = folded(0) + folded(hop - 1);
      if (mode == 1) v *= 0.5f;
    } else {
      for (int t = 0; t < hop; ++t) v += folded(t);
      if (mode == 3) v /= (float)hop;
    }
    out[(int64_t)bidx * D + c] = v;
  }
}

// dec_upsampling="repeat": x.repeat_interleave(hop, 1)   (:161)
__global__ void repeat_rows_kernel(const float* __restrict__ x, int D, int hop, float* __restrict__ out) {
  const int row = blockIdx.x;                       // b * hop + t
  const float* src = x + (int64_t)(row / hop) * D;
  for (int c = threadIdx.x; c < D; c += blockDim.x) out[(int64_t)row * D + c] = src[c];
}

// dec_upsampling="interp" (:162-165): x.expand(-1, hop, -1) + F.interpolate(h_n.permute(1, 2, 0), (hop,)).permute(0, 2, 1) -
// nearest-neighbour interpolation of the encoder's two final states (forward, reverse) over the hop frames: frame t gets
// direction floor(2 t / hop)
__global__ void interp_rows_kernel(const float* __restrict__ x, const float* __restrict__ h_fwd, const float* __restrict__ h_rev, int D,
                                   int hop, float* __restrict__ out) {
  const int row = blockIdx.x;                       // b * hop + t
  const int b = row / hop, t = row % hop;
  const float* hsrc = ((2 * t) / hop == 0 ? h_fwd : h_rev) + (int64_t)b * D;
  const float* src = x + (int64_t)b * D;
  for (int c = threadIdx.x; c < D; c += blockDim.x) out[(int64_t)row * D + c] = src[c] + hsrc[c];
}

}  // namespace

struct BiLstm {
  PackedLinear ih[2], hh[2];  // [fwd, reverse]
};

struct mmk_s2s_plan {
  Tuning tune;                  // the config's execution switches (plan_util.h): never the environment in the product library
  mmk_s2s_config cfg;
  Binder binder;
  bool committed = false;
  int D = 0, hop = 0, Bmax = 0, in_pad = 0, out_pad = 0;
  std::vector<BiLstm> enc, dec;        // the bi-LSTM layers of each side
  PackedLinear fc_out, dec_fc, out_lin, enc_fc;     // enc_fc: LinearResampler(D, 1 / hop, 1) of enc_downsampling="linear_resample"
  float *xin = nullptr, *gi[2] = {nullptr, nullptr}, *gates = nullptr;
  float *h[2] = {nullptr, nullptr}, *c[2] = {nullptr, nullptr};
  float* h2[2] = {nullptr, nullptr};   // second state buffer of each direction (the fused step kernel ping-pongs)
  bool fused_lstm = false;
  // the resident sequence kernel (lstm_seq.hip): two sets of exchange images (a launch uses one and poisons the other), the word a
  // timed-out workgroup raises, launches since the last status check
  bool seq_lstm = false;
  float* xch[2] = {nullptr, nullptr};
  int xch_cur = 0;
  uint32_t* seq_err = nullptr;
  unsigned long long* seq_stamps = nullptr;      // diagnostic build: phase stamps of the last resident launch
  int n_cu = 0;
  int64_t seq_launches = 0;
  float *of = nullptr, *ob = nullptr, *es = nullptr, *coded = nullptr, *z = nullptr, *ysum = nullptr, *yout = nullptr;
  float* yalt = nullptr;                         // second folded-output buffer (layer n reads one, writes the other)
  float* compose_tmp = nullptr;                  // (hop D, D): dec.fc . enc.fc_out, row-major, before it is packed
  bool fc_composed = false;                      // dec_fc holds the pre-multiplied matrix: the coded frame is never materialised
  float *hs[2] = {nullptr, nullptr}, *cs[2] = {nullptr, nullptr};   // the encoder's final state: every decoder layer starts from it
  // discrete IO: the embedding table (a copy: in_classes x D), the MLP head's Linears, its hidden rows and raw outputs
  float* embed = nullptr;
  float* gemm_partial = nullptr;                 // split-K partial sums of the GEMM launches that would not fill the chip
  int gemm_ksplit = 0;                           // > 0: every GEMM launch splits K this many ways (tuning MMK_GEMM_KSPLIT: a parity-test mode)
  static constexpr int64_t kPartialFloats = 512 * 64 * 64;   // k_split x workgroups <= 512 tiles of 64 x 64
  std::vector<PackedLinear> mlp;
  float *hid[2] = {nullptr, nullptr}, *logits = nullptr;
  int logits_ld = 0;

  void layout(Carver& cv) {
    for (auto& l : enc) for (int d = 0; d < 2; ++d) { l.ih[d].carve(cv, true); l.hh[d].carve(cv, false); }
    for (auto& l : dec) for (int d = 0; d < 2; ++d) { l.ih[d].carve(cv, true); l.hh[d].carve(cv, false); }
    fc_out.carve(cv, false);
    if (cfg.enc_downsampling == 4) enc_fc.carve(cv, true);
    dec_fc.carve(cv, true);
    const int64_t rows = (int64_t)Bmax * hop;
    if (cfg.head_kind == 1) {
      for (auto& m : mlp) m.carve(cv, true);
      hid[0] = cv.take<float>(rows * cfg.mlp_hidden);
      hid[1] = cv.take<float>(rows * cfg.mlp_hidden);
      logits = cv.take<float>(rows * logits_ld);
    } else {
      out_lin.carve(cv, true);
    }
    if (cfg.in_classes > 0) embed = cv.take<float>((int64_t)cfg.in_classes * D);
    gemm_partial = cv.take<float>(kPartialFloats);
    xin = cv.take<float>(rows * in_pad);
    gi[0] = cv.take<float>(rows * 4 * D);
    gi[1] = cv.take<float>(rows * 4 * D);
    gates = cv.take<float>((int64_t)Bmax * 4 * D);
    for (int d = 0; d < 2; ++d) { h[d] = cv.take<float>((int64_t)Bmax * D); c[d] = cv.take<float>((int64_t)Bmax * D); }
    for (int d = 0; d < 2; ++d) h2[d] = cv.take<float>((int64_t)Bmax * D);
    if (seq_lstm) {
      for (int k = 0; k < 2; ++k) xch[k] = cv.take<float>((int64_t)lstm_seq_xch_floats(D, Bmax, hop));
      seq_err = reinterpret_cast<uint32_t*>(cv.take<float>(64));
      seq_stamps = reinterpret_cast<unsigned long long*>(cv.take<float>(2 * 16 * 8 * 8));
    }
    of = cv.take<float>(rows * D);
    ob = cv.take<float>(rows * D);
    es = cv.take<float>((int64_t)Bmax * D);
    coded = cv.take<float>((int64_t)Bmax * D);
    compose_tmp = cfg.dec_upsampling == 0 ? cv.take<float>((int64_t)hop * D * D) : nullptr;
    z = cv.take<float>(rows * D);
    ysum = cv.take<float>(rows * D);
    yalt = cv.take<float>(rows * D);
    for (int d = 0; d < 2; ++d) { hs[d] = cv.take<float>((int64_t)Bmax * D); cs[d] = cv.take<float>((int64_t)Bmax * D); }
    yout = cv.take<float>(rows * out_pad);
  }
};

static int derive(mmk_s2s_plan* p) {
  const mmk_s2s_config& c = p->cfg;
  if (c.in_dim < 1 || c.out_dim < 1 || c.model_dim < 1 || c.hop < 1 || c.max_batch < 1) return fail(MMK_ERR_INVALID, "s2s: bad dimensions");
  if (c.enc_n_lstm < 1 || c.enc_n_lstm > 8 || c.dec_n_lstm < 1 || c.dec_n_lstm > 8)
    return fail(MMK_ERR_UNSUPPORTED, "s2s: 1 .. 8 bi-LSTM layers per side, got %d + %d", c.enc_n_lstm, c.dec_n_lstm);
  if (c.enc_downsampling < 0 || c.enc_downsampling > 4 || c.dec_upsampling < 0 || c.dec_upsampling > 2)
    return fail(MMK_ERR_UNSUPPORTED, "s2s: enc_downsampling %d / dec_upsampling %d are not covered", c.enc_downsampling, c.dec_upsampling);
  if (c.enc_downsampling == 4 && c.model_dim % c.hop != 0)
    return fail(MMK_ERR_INVALID, "s2s: enc_downsampling='linear_resample' needs hop (%d) to divide model_dim (%d)", c.hop, c.model_dim);
  if (c.model_dim % 2 != 0) return fail(MMK_ERR_UNSUPPORTED, "s2s: model_dim must be even");
  if (c.in_classes < 0 || (c.in_classes > 0 && c.in_dim != c.model_dim))
    return fail(MMK_ERR_INVALID, "s2s: an embedding input has in_dim == model_dim (got %d, %d)", c.in_dim, c.model_dim);
  if (c.head_kind < 0 || c.head_kind > 1) return fail(MMK_ERR_INVALID, "s2s: head_kind %d unknown", c.head_kind);
  if (c.head_kind == 1 && (c.mlp_hidden < 1 || c.mlp_n_hidden < 0 || c.mlp_n_hidden > MMK_MAX_MLP_HIDDEN))
    return fail(MMK_ERR_INVALID, "s2s: bad MLP head geometry (hidden %d, %d extra blocks)", c.mlp_hidden, c.mlp_n_hidden);
  if ((c.head_kind == 1) != (c.in_classes > 0))
    return fail(MMK_ERR_UNSUPPORTED, "s2s: class indices in and out go together (in_classes %d, head_kind %d): the loop feeds outputs back", c.in_classes, c.head_kind);
  p->D = c.model_dim;
  p->hop = c.hop;
  p->Bmax = c.max_batch;
  p->in_pad = (int)round_up(c.in_dim, 4);
  p->out_pad = (int)round_up(c.out_dim, 4);
  p->enc.assign((size_t)c.enc_n_lstm, BiLstm());
  p->dec.assign((size_t)c.dec_n_lstm, BiLstm());
  for (int d = 0; d < 2; ++d) {
    for (size_t n = 0; n < p->enc.size(); ++n) {
      p->enc[n].ih[d].set_geometry(4 * p->D, {n == 0 ? p->in_pad : p->D});
      p->enc[n].hh[d].set_geometry(4 * p->D, {p->D});
    }
    for (auto& l : p->dec) {
      l.ih[d].set_geometry(4 * p->D, {p->D});
      l.hh[d].set_geometry(4 * p->D, {p->D});
    }
  }
  p->fc_out.set_geometry(p->D, {p->D});
  if (c.enc_downsampling == 4) p->enc_fc.set_geometry(p->D / p->hop, {p->D});
  p->dec_fc.set_geometry(p->hop * p->D, {p->D});
  p->out_lin.set_geometry(c.out_dim, {p->D});
  p->mlp.clear();
  if (c.head_kind == 1) {   // Linear, Mish, [Linear, Mish] * n, Linear   (networks/mlp.py:42-53)
    PackedLinear first, last;
    first.set_geometry(c.mlp_hidden, {p->D});
    p->mlp.push_back(first);
    for (int i = 0; i < c.mlp_n_hidden; ++i) {
      PackedLinear h;
      h.set_geometry(c.mlp_hidden, {c.mlp_hidden});
      p->mlp.push_back(h);
    }
    last.set_geometry(c.out_dim + (c.learn_temp ? 1 : 0), {c.mlp_hidden});
    p->mlp.push_back(last);
    p->logits_ld = (int)round_up(c.out_dim + (c.learn_temp ? 1 : 0), 4);
  }
  const char* fenv = p->tune.get("MMK_S2S_FUSED");
  p->fused_lstm = !(fenv && fenv[0] == '0') && lstm_step_supported(p->D);
  {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) p->n_cu = prop.multiProcessorCount;
  }
  // one launch per bi-LSTM layer where all its workgroups fit on the chip at once (exec_mode 1: the per-step kernel, which
  // needs no co-residency - what a caller asks for after mmk_s2s_sync_status reported a timed-out wait)
  const char* senv = p->tune.get("MMK_S2S_SEQ");
  p->seq_lstm = p->fused_lstm && c.exec_mode != 1 && !(senv && senv[0] == '0') && lstm_seq_supported(p->D, 1, p->hop, p->n_cu);
  return MMK_OK;
}

extern "C" int mmk_s2s_plan_create(const mmk_s2s_config* cfg, mmk_s2s_plan** out) {
  if (!cfg || !out) return fail(MMK_ERR_INVALID, "s2s_plan_create: null argument");
  mmk_s2s_plan* p = new mmk_s2s_plan();
  p->cfg = *cfg;
  p->tune.parse(cfg->tuning, sizeof(cfg->tuning));
  if (const char* ks = p->tune.get("MMK_GEMM_KSPLIT")) p->gemm_ksplit = atoi(ks);
  int rc = derive(p);
  if (rc != MMK_OK) {
    delete p;
    return rc;
  }
  *out = p;
  return MMK_OK;
}

extern "C" void mmk_s2s_plan_destroy(mmk_s2s_plan* p) { delete p; }

extern "C" int mmk_s2s_plan_bind(mmk_s2s_plan* p, const char* key, const float* dev_ptr, int64_t numel) {
  if (!p || !key || !dev_ptr) return fail(MMK_ERR_INVALID, "s2s_plan_bind: null argument");
  p->binder.bind(key, dev_ptr, numel);
  p->committed = false;
  return MMK_OK;
}

extern "C" size_t mmk_s2s_workspace_bytes(const mmk_s2s_plan* p) {
  if (!p) return 0;
  mmk_s2s_plan tmp = *p;
  Carver c(nullptr);
  tmp.layout(c);
  return c.used();
}

static int pack_lstm(mmk_s2s_plan* p, BiLstm& l, const std::string& base, int in_dim, hipStream_t st) {
  Binder& b = p->binder;
  const int D = p->D;
  const char* sfx[2] = {"", "_reverse"};
  for (int d = 0; d < 2; ++d) {
    const float* wih = b.need(base + "weight_ih_l0" + sfx[d], (int64_t)4 * D * in_dim);
    const float* whh = b.need(base + "weight_hh_l0" + sfx[d], (int64_t)4 * D * D);
    const float* bih = b.need(base + "bias_ih_l0" + sfx[d], 4 * D);
    const float* bhh = b.need(base + "bias_hh_l0" + sfx[d], 4 * D);
    if (wih) MMK_TRY(pack_rect(l.ih[d].Wp, l.ih[d].k_chunks, 0, 1, 4 * D, 0, in_dim, wih, in_dim, 1, st));
    if (whh) MMK_TRY(pack_rect(l.hh[d].Wp, l.hh[d].k_chunks, 0, 1, 4 * D, 0, D, whh, D, 1, st));
    if (bih) MMK_TRY(pack_bias(l.ih[d].bias, 0, 1, 4 * D, bih, 0, st));
    if (bhh) MMK_TRY(pack_bias(l.ih[d].bias, 0, 1, 4 * D, bhh, 1, st));
  }
  return MMK_OK;
}

// C[r][k] = sum_j A[r][j] B[j][k]   (A: (R, D) = dec.fc weight, B: (D, D) = enc.fc_out weight): fp64 accumulation, one rounding.
// The decoder's up-sampler then takes the pooled encoder frame directly: z = dec.fc(fc_out(e)) = (dec.fc . fc_out) e + b
// (s2s_lstm_v2.py:113, :158-159) - one launch and 4 MB of weights less per generate_step, same operands in another association.
__global__ void s2s_compose_kernel(const float* __restrict__ A, const float* __restrict__ B, int R, int D, float* __restrict__ C) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;      // consecutive threads: consecutive columns of B (coalesced)
  const int r = blockIdx.y;
  if (k >= D || r >= R) return;
  double acc = 0.0;
  for (int j = 0; j < D; ++j) acc += (double)A[(int64_t)r * D + j] * (double)B[(int64_t)j * D + k];
  C[(int64_t)r * D + k] = (float)acc;
}

extern "C" int mmk_s2s_commit(mmk_s2s_plan* p, void* workspace, size_t workspace_bytes, mmk_stream_t stream) {
  if (!p || !workspace) return fail(MMK_ERR_INVALID, "s2s_commit: null argument");
  if ((reinterpret_cast<uintptr_t>(workspace) & 255) != 0) return fail(MMK_ERR_WORKSPACE, "s2s_commit: workspace must be 256-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const mmk_s2s_config& c = p->cfg;
  Carver carve(workspace);
  p->layout(carve);
  if (carve.used() > workspace_bytes)
    return fail(MMK_ERR_WORKSPACE, "s2s_commit: workspace of %zu bytes, %zu needed", workspace_bytes, carve.used());
  MMK_HIP(hipMemsetAsync(workspace, 0, carve.used(), st));
  if (p->seq_lstm) {
    for (int k = 0; k < 2; ++k) MMK_HIP(hipMemsetAsync(p->xch[k], 0xFF, lstm_seq_xch_floats(p->D, p->Bmax, p->hop) * sizeof(float), st));
    p->xch_cur = 0;
  }
  Binder& b = p->binder;
  b.clear_missing();
  const int D = p->D;
  for (size_t n = 0; n < p->enc.size(); ++n)
    MMK_TRY(pack_lstm(p, p->enc[n], "enc.lstm." + std::to_string(n) + ".", n == 0 ? c.in_dim : D, st));
  for (size_t n = 0; n < p->dec.size(); ++n) MMK_TRY(pack_lstm(p, p->dec[n], "dec.lstm." + std::to_string(n) + ".", D, st));
  if (const float* w = b.need("enc.fc_out.weight", (int64_t)D * D))
    MMK_TRY(pack_rect(p->fc_out.Wp, p->fc_out.k_chunks, 0, 1, D, 0, D, w, D, 1, st));
  if (c.enc_downsampling == 4) {   // LinearResampler(D, 1 / hop, 1): Linear(D, D / hop) per frame, the hop results concatenated (:21-23)
    if (const float* w = b.need("enc.fc.fc.weight", (int64_t)(D / p->hop) * D))
      MMK_TRY(pack_rect(p->enc_fc.Wp, p->enc_fc.k_chunks, 0, 1, D / p->hop, 0, D, w, D, 1, st));
    if (const float* bb = b.need("enc.fc.fc.bias", D / p->hop)) MMK_TRY(pack_bias(p->enc_fc.bias, 0, 1, D / p->hop, bb, 0, st));
  }
  p->fc_composed = false;
  if (c.dec_upsampling == 0) {   // "repeat" / "interp" have no up-sampling weights
    const float* w = b.need("dec.fc.fc.weight", (int64_t)p->hop * D * D);
    const float* wo = b.need("enc.fc_out.weight", (int64_t)D * D);
    const char* cenv = p->tune.get("MMK_S2S_COMPOSED");
    if (w && wo && !(cenv && cenv[0] == '0')) {
      hipLaunchKernelGGL(s2s_compose_kernel, dim3((D + 255) / 256, p->hop * D), dim3(256), 0, st, w, wo, p->hop * D, D, p->compose_tmp);
      MMK_HIP(hipGetLastError());
      w = p->compose_tmp;
      p->fc_composed = true;
    }
    if (w)
      MMK_TRY(pack_rect(p->dec_fc.Wp, p->dec_fc.k_chunks, 0, 1, p->hop * D, 0, D, w, D, 1, st));
    if (const float* bb = b.need("dec.fc.fc.bias", (int64_t)p->hop * D)) MMK_TRY(pack_bias(p->dec_fc.bias, 0, 1, p->hop * D, bb, 0, st));
  }
  if (c.head_kind == 1) {
    const std::string hb = "output_module.heads.0.estimator.0.fc.";
    for (size_t i = 0; i < p->mlp.size(); ++i) {
      PackedLinear& m = p->mlp[i];
      const std::string kb = hb + std::to_string(2 * i) + ".";
      const float* w = b.need(kb + "weight", (int64_t)m.N * m.segK[0]);
      const float* bb = b.need(kb + "bias", m.N);
      if (w) MMK_TRY(pack_rect(m.Wp, m.k_chunks, 0, 1, m.N, 0, m.segK[0], w, m.segK[0], 1, st));
      if (bb) MMK_TRY(pack_bias(m.bias, 0, 1, m.N, bb, 0, st));
    }
  } else {
    if (const float* w = b.need("output_module.heads.0.0.weight", (int64_t)c.out_dim * D))
      MMK_TRY(pack_rect(p->out_lin.Wp, p->out_lin.k_chunks, 0, 1, c.out_dim, 0, D, w, D, 1, st));
    if (const float* bb = b.need("output_module.heads.0.0.bias", c.out_dim)) MMK_TRY(pack_bias(p->out_lin.bias, 0, 1, c.out_dim, bb, 0, st));
  }
  if (c.in_classes > 0)
    if (const float* w = b.need("input_module.heads.0.0.weight", (int64_t)c.in_classes * D))
      MMK_HIP(hipMemcpyAsync(p->embed, w, (size_t)c.in_classes * D * sizeof(float), hipMemcpyDeviceToDevice, st));
  if (!b.missing().empty()) return fail(MMK_ERR_KEY, "s2s_commit: state_dict tensor %s", b.missing().c_str());
  p->committed = true;
  return MMK_OK;
}

static int plain_linear(const PackedLinear& w, const float* x, int64_t ldx, int M, float* y, int64_t ldy, int act,
                        hipStream_t st, float* partial = nullptr, int forced_k_split = 0) {
  // GEMM-shaped calls (all hop frames of all clips at once) take the tiled kernel; MMK_S2S_GEMM=0 keeps the row-tile one
  static const bool tiled = [] { const char* e = diag_only("MMK_S2S_GEMM"); return !(e && e[0] == '0'); }();
  if (tiled && w.nseg == 1 && gemm_bias_act_supported(x, ldx, M, w.segK[0]))
    return launch_gemm_bias_act(x, ldx, w.Wp, w.bias, w.n_tiles, w.k_chunks, w.N, w.segK[0], y, ldy, M, act, st, GemmRowMap(), partial,
                                partial ? mmk_s2s_plan::kPartialFloats : 0, forced_k_split);
  // few rows against a large matrix (dec.fc, enc.fc_out): the weight-streaming kernel; MMK_S2S_SKINNY=0 keeps the row-tile one
  static const bool skinny = [] { const char* e = diag_only("MMK_S2S_SKINNY"); return !(e && e[0] == '0'); }();
  if (skinny && w.nseg == 1 && w.n_tiles >= 32 && skinny_linear_supported(x, ldx, M, w.segK[0], w.k_chunks))
    return launch_skinny_linear(x, ldx, w.Wp, w.bias, w.n_tiles, w.k_chunks, w.N, w.segK[0], y, ldy, M, act, st);
  LinearArgs a = {};
  w.fill(a);
  a.seg[0].x = addr_static(x);
  a.seg[0].ld = ldx;
  a.M = M;
  a.tau_ptr = nullptr;
  a.epilogue = EPI_STORE;
  a.act = act;
  a.out = addr_static(y);
  a.out_ld = ldy;
  return launch_linear(a, st);
}

// one bidirectional LSTM over hop frames; rows of x/of/ob are (b*hop + t)
// where the rows (clip, frame) of a layer's input lie when they are the caller's frames, read in place (group = frames per clip)
struct FrameMap {
  int group = 0, valid = 0;      // frames per clip; inputs per frame (what follows them in memory belongs to something else)
  int64_t clip_stride = 0, frame_stride = 0, floats = 0;
};

static bool inproj_enabled() {
  static const bool on = [] { const char* e = diag_only("MMK_S2S_INPROJ"); return !(e && e[0] == '0'); }();
  return on;
}

// what becomes of a layer's (forward | reverse) output: folded rows (+ residual) in `fold`, or those pooled per clip in `pool`
struct FoldSpec {
  float* fold = nullptr;
  const float* res = nullptr;
  float* pool = nullptr;
  int pool_mode = 0;
};

// the fold as its own launch, for the paths that leave the two directions' rows in of / ob
static int fold_rows(mmk_s2s_plan* p, int M, const FoldSpec& fs, hipStream_t st) {
  const int rows = M * p->hop;
  if (fs.pool) hipLaunchKernelGGL(pair_sum_pool_kernel, dim3(M), dim3(256), 0, st, p->of, p->ob, p->D, p->hop, fs.pool_mode, fs.res, fs.pool);
  else hipLaunchKernelGGL(pair_sum_kernel, dim3(rows), dim3(256), 0, st, p->of, p->ob, p->D, rows, fs.res, fs.fold);
  MMK_HIP(hipGetLastError());
  return MMK_OK;
}

static int run_bilstm(mmk_s2s_plan* p, BiLstm& l, const float* x, int x_ld, int M, bool zero_state, hipStream_t st, const FoldSpec& fs,
                      const FrameMap& fm = FrameMap()) {
  const int D = p->D, hop = p->hop;
  const int64_t rows = (int64_t)M * hop;
  // the input half of both directions: one launch with W_ih in registers where the chip holds it (lstm_inproj.hip), else a GEMM each
  const bool inproj = inproj_enabled() && p->seq_lstm && l.ih[0].nseg == 1 && lstm_inproj_supported((int)rows, l.ih[0].segK[0], l.ih[0].k_chunks, D);
  if (fm.group > 0 && !inproj) return fail(MMK_ERR_STATE, "s2s: frames in place need the input-projection kernel");
  if (inproj) {
    LstmInProjArgs ia = {};
    ia.x = x; ia.x_ld = x_ld; ia.out_ld = 4 * D; ia.rows = (int)rows; ia.K = l.ih[0].segK[0]; ia.k_chunks = l.ih[0].k_chunks; ia.H = D;
    ia.x_floats = rows * x_ld;
    if (fm.group > 0) { ia.x_group = fm.group; ia.x_group_stride = fm.clip_stride; ia.x_ld = fm.frame_stride; ia.x_floats = fm.floats; ia.k_valid = fm.valid; }
    for (int d = 0; d < 2; ++d) { ia.dir[d].wih_wp = l.ih[d].Wp; ia.dir[d].bias = l.ih[d].bias; ia.dir[d].out = p->gi[d]; }
    MMK_TRY(launch_lstm_inproj(ia, p->n_cu, st));
  }
  for (int d = 0; d < 2; ++d) {
    if (!inproj) MMK_TRY(plain_linear(l.ih[d], x, x_ld, (int)rows, p->gi[d], 4 * D, ACT_NONE, st, p->gemm_partial, p->gemm_ksplit));
    if (zero_state && !p->fused_lstm) {
      // `lstm(x,)` : fresh zero state on every call                    (s2s_lstm_v2.py:97); the fused step kernel takes a flag
      MMK_HIP(hipMemsetAsync(p->h[d], 0, (size_t)p->Bmax * D * sizeof(float), st));
      MMK_HIP(hipMemsetAsync(p->c[d], 0, (size_t)p->Bmax * D * sizeof(float), st));
    }
  }
  if (p->seq_lstm && lstm_seq_supported(D, M, hop, p->n_cu)) {
    LstmSeqArgs a = {};
    a.M = M; a.H = D; a.n_steps = hop; a.zero_state = zero_state ? 1 : 0; a.rows_pad = p->Bmax;
    a.gadd_ld = (int64_t)hop * 4 * D; a.gadd_ts = 4 * D; a.y_ld = (int64_t)hop * D; a.y_ts = D;
    for (int d = 0; d < 2; ++d) {
      a.dir[d].whh_wp = l.hh[d].Wp; a.dir[d].gadd = p->gi[d]; a.dir[d].h = p->h[d]; a.dir[d].c = p->c[d];
      a.dir[d].y = d == 0 ? p->of : p->ob;
    }
    a.xch = p->xch[p->xch_cur]; a.xch_next = p->xch[p->xch_cur ^ 1]; a.err = p->seq_err;
    // the fold (and the encoder's pooling) happen in the cell: nobody reads the two directions' rows themselves
    a.fold = fs.fold; a.res = fs.res; a.pool = fs.pool; a.pool_mode = fs.pool_mode;
    a.dir[0].y = a.dir[1].y = nullptr;
    if (const char* senv = diag_only("MMK_S2S_STAMPS"); senv && senv[0] == '1') {
      a.stamps = p->seq_stamps;
      a.stamp_wg = diag_only("MMK_S2S_STAMP_WG") ? atoi(diag_only("MMK_S2S_STAMP_WG")) : 0;
    }
    p->xch_cur ^= 1;
    p->seq_launches += 1;
    return launch_lstm_seq(a, st);
  }
  if (p->fused_lstm) {
    // both directions of a time step in one launch; the state ping-pongs between h and h2 (an even number of steps
    // leaves it in h, an odd one is copied back)
    float* cur[2] = {p->h[0], p->h[1]};
    float* nxt[2] = {p->h2[0], p->h2[1]};
    for (int s = 0; s < hop; ++s) {
      LstmStepArgs a = {};
      a.M = M; a.H = D; a.n_dir = 2; a.gadd_ld = (int64_t)hop * 4 * D; a.y_ld = (int64_t)hop * D;
      a.zero_state = (zero_state && s == 0) ? 1 : 0;
      for (int d = 0; d < 2; ++d) {
        const int t = d == 0 ? s : hop - 1 - s;
        a.dir[d].whh_wp = l.hh[d].Wp;
        a.dir[d].gadd = p->gi[d] + (int64_t)t * 4 * D;
        a.dir[d].h_in = cur[d]; a.dir[d].h_out = nxt[d]; a.dir[d].c = p->c[d];
        a.dir[d].y = (d == 0 ? p->of : p->ob) + (int64_t)t * D;
      }
      MMK_TRY(launch_lstm_step(a, st));
      for (int d = 0; d < 2; ++d) { float* tmp = cur[d]; cur[d] = nxt[d]; nxt[d] = tmp; }
    }
    if (cur[0] != p->h[0])
      for (int d = 0; d < 2; ++d)
        MMK_HIP(hipMemcpyAsync(p->h[d], cur[d], (size_t)M * D * sizeof(float), hipMemcpyDeviceToDevice, st));
    return fold_rows(p, M, fs, st);
  }
  for (int s = 0; s < hop; ++s) {
    for (int d = 0; d < 2; ++d) {
      const int t = d == 0 ? s : hop - 1 - s;
      MMK_TRY(plain_linear(l.hh[d], p->h[d], D, M, p->gates, 4 * D, ACT_NONE, st));
      float* y = (d == 0 ? p->of : p->ob) + (int64_t)t * D;
      MMK_TRY(launch_lstm_cell(p->gates, 4 * D, p->gi[d] + (int64_t)t * 4 * D, (int64_t)hop * 4 * D, p->h[d], D,
                               p->c[d], D, y, (int64_t)hop * D, M, D, st));
    }
  }
  return fold_rows(p, M, fs, st);
}

// one step's input and output: frames (x, y) or class indices (xi, yi), strides per clip and per frame / position
struct S2SIo {
  const float* x = nullptr; float* y = nullptr;
  const int64_t* xi = nullptr; int64_t* yi = nullptr;
  int64_t xbs = 0, xfs = 0, ybs = 0, yfs = 0;
};

static int s2s_step(mmk_s2s_plan* p, int M, const S2SIo& io, int n_out, hipStream_t st) {
  const mmk_s2s_config& c = p->cfg;
  const int D = p->D, hop = p->hop;
  const int rows = M * hop;
  float* y = io.y;
  const int64_t ybs = io.ybs, yfs = io.yfs;
  // continuous inputs: the first encoder layer's input projection reads the caller's frames where they lie (lstm_inproj.hip); else
  // they (or the embedded classes) are gathered into padded rows first
  FrameMap fm;
  if (c.in_classes <= 0 && inproj_enabled() && p->seq_lstm && lstm_seq_supported(D, M, hop, p->n_cu) && p->enc[0].ih[0].nseg == 1 &&
      lstm_inproj_supported(rows, p->enc[0].ih[0].segK[0], p->enc[0].ih[0].k_chunks, D) && io.xbs >= 0 && io.xfs >= 0) {
    fm.group = hop; fm.valid = c.in_dim; fm.clip_stride = io.xbs; fm.frame_stride = io.xfs;
    fm.floats = (int64_t)(M - 1) * io.xbs + (int64_t)(hop - 1) * io.xfs + c.in_dim;
    if (fm.floats * (int64_t)sizeof(float) >= ((int64_t)1 << 31)) fm = FrameMap();
  }
  if (c.in_classes > 0)
    hipLaunchKernelGGL(embed_rows_kernel, dim3(rows), dim3(256), 0, st, io.xi, io.xbs, io.xfs, hop, c.in_classes, D, p->embed, p->xin, p->in_pad);
  else if (fm.group == 0)
    hipLaunchKernelGGL(gather_frames_kernel, dim3(rows), dim3(256), 0, st, io.x, io.xbs, io.xfs, hop, c.in_dim, p->xin, p->in_pad);
  MMK_HIP(hipGetLastError());
  const size_t state_bytes = (size_t)M * D * sizeof(float);
  // encoder: every layer starts from a zero state; x = y, or x + y from the second layer on (:96-104)
  const float* xl = p->xin;
  int xl_ld = p->in_pad;
  float* fold = p->ysum;
  bool pooled = false;
  for (size_t n = 0; n < p->enc.size(); ++n) {
    FoldSpec fs;
    fs.res = (n > 0 && c.enc_apply_residuals) ? xl : nullptr;
    const bool pool_here = n + 1 == p->enc.size() && c.enc_downsampling != 4;     // last layer: fold + pool together
    if (pool_here) { fs.pool = p->es; fs.pool_mode = c.enc_downsampling; }
    else fs.fold = fold;
    if (n == 0 && fm.group > 0) MMK_TRY(run_bilstm(p, p->enc[n], io.x, xl_ld, M, true, st, fs, fm));
    else MMK_TRY(run_bilstm(p, p->enc[n], xl, xl_ld, M, true, st, fs));
    if (pool_here) {
      pooled = true;
      break;
    }
    xl = fold; xl_ld = D;
    fold = fold == p->ysum ? p->yalt : p->ysum;
  }
  if (c.enc_downsampling == 4) {
    // rows (b hop + t) -> D / hop features each: (M, hop, D / hop) contiguous IS the reshape to (M, 1, D)   (resamplers.py:21-23)
    MMK_TRY(plain_linear(p->enc_fc, xl, D, rows, p->es, D / hop, ACT_NONE, st));
  } else if (!pooled) {
    hipLaunchKernelGGL(pool_frames_kernel, dim3(M), dim3(256), 0, st, xl, D, hop, c.enc_downsampling, p->es);
    MMK_HIP(hipGetLastError());
  }
  const bool composed_up = c.dec_upsampling == 0 && p->fc_composed;
  if (!composed_up) MMK_TRY(plain_linear(p->fc_out, p->es, D, M, p->coded, D, ACT_NONE, st));
  // decoder: up-sampling to hop frames, every bi-LSTM seeded with the LAST encoder layer's (h_n, c_n)  (:158-171)
  if (c.dec_upsampling == 0) {
    MMK_TRY(plain_linear(p->dec_fc, composed_up ? p->es : p->coded, D, M, p->z, (int64_t)hop * D, ACT_NONE, st));
  } else if (c.dec_upsampling == 1) {
    hipLaunchKernelGGL(repeat_rows_kernel, dim3(rows), dim3(256), 0, st, p->coded, D, hop, p->z);
    MMK_HIP(hipGetLastError());
  } else {   // the encoder's final states are still in h[0] (forward) / h[1] (reverse)
    hipLaunchKernelGGL(interp_rows_kernel, dim3(rows), dim3(256), 0, st, p->coded, p->h[0], p->h[1], D, hop, p->z);
    MMK_HIP(hipGetLastError());
  }
  if (p->dec.size() > 1)
    for (int d = 0; d < 2; ++d) {
      MMK_HIP(hipMemcpyAsync(p->hs[d], p->h[d], state_bytes, hipMemcpyDeviceToDevice, st));
      MMK_HIP(hipMemcpyAsync(p->cs[d], p->c[d], state_bytes, hipMemcpyDeviceToDevice, st));
    }
  xl = p->z;
  fold = p->ysum;
  for (size_t n = 0; n < p->dec.size(); ++n) {
    if (n > 0)
      for (int d = 0; d < 2; ++d) {
        MMK_HIP(hipMemcpyAsync(p->h[d], p->hs[d], state_bytes, hipMemcpyDeviceToDevice, st));
        MMK_HIP(hipMemcpyAsync(p->c[d], p->cs[d], state_bytes, hipMemcpyDeviceToDevice, st));
      }
    FoldSpec fs;
    fs.fold = fold;
    fs.res = c.dec_apply_residuals ? xl : nullptr;
    MMK_TRY(run_bilstm(p, p->dec[n], xl, D, M, false, st, fs));
    xl = fold;
    fold = fold == p->ysum ? p->yalt : p->ysum;
  }
  if (c.head_kind == 1) {
    // MLP head over all hop positions of all clips, then one wave per row: learned-temperature division + argmax
    const float* hx = xl;
    int64_t hx_ld = D;
    for (size_t i = 0; i < p->mlp.size(); ++i) {
      const bool last = i + 1 == p->mlp.size();
      float* o = last ? p->logits : p->hid[i & 1];
      const int64_t o_ld = last ? p->logits_ld : c.mlp_hidden;
      MMK_TRY(plain_linear(p->mlp[i], hx, hx_ld, rows, o, o_ld, last ? (int)ACT_NONE : c.mlp_act, st, p->gemm_partial, p->gemm_ksplit));   // MLPIO.activation (modules/io.py:205: Mish by default)
      hx = o;
      hx_ld = o_ld;
    }
    SampleArgs sa = {};
    sa.logits = p->logits; sa.ld = p->logits_ld; sa.rows = rows; sa.n_classes = c.out_dim; sa.has_temp_col = c.learn_temp;
    sa.min_temp = c.min_temp;
    sa.out = io.yi; sa.out_row_stride = yfs; sa.group = hop; sa.kept = n_out; sa.group_stride = ybs;
    return launch_sample(sa, st);
  }
  {
    // the output projection writes the caller's (batch, frame, bin) rows itself when the tiled kernel takes it
    static const bool tiled = [] { const char* e = diag_only("MMK_S2S_GEMM"); return !(e && e[0] == '0'); }();
    const PackedLinear& w = p->out_lin;
    if (tiled && w.nseg == 1 && gemm_bias_act_supported(xl, D, rows, w.segK[0])) {
      GemmRowMap rm;
      rm.group = hop; rm.kept = n_out; rm.group_stride = ybs; rm.row_stride = yfs;
      return launch_gemm_bias_act(xl, D, w.Wp, w.bias, w.n_tiles, w.k_chunks, w.N, w.segK[0], y, 0, rows, c.out_abs ? ACT_ABS : ACT_NONE, st, rm,
                                  p->gemm_partial, mmk_s2s_plan::kPartialFloats, p->gemm_ksplit);
    }
  }
  MMK_TRY(plain_linear(p->out_lin, xl, D, rows, p->yout, p->out_pad, c.out_abs ? ACT_ABS : ACT_NONE, st));
  hipLaunchKernelGGL(scatter_frames_kernel, dim3(rows), dim3(256), 0, st, p->yout, p->out_pad, hop, n_out, c.out_dim, y,
                     ybs, yfs);
  MMK_HIP(hipGetLastError());
  return MMK_OK;
}

extern "C" int mmk_s2s_step(mmk_s2s_plan* p, int32_t batch, const float* x, int64_t x_batch_stride, int64_t x_frame_stride,
                            float* y, int64_t y_batch_stride, int64_t y_frame_stride, mmk_stream_t stream) {
  if (!p || !x || !y) return fail(MMK_ERR_INVALID, "s2s_step: null argument");
  if (!p->committed) return fail(MMK_ERR_STATE, "s2s_step: plan not committed");
  if (batch < 1 || batch > p->Bmax) return fail(MMK_ERR_INVALID, "s2s_step: batch %d outside [1, %d]", batch, p->Bmax);
  if (p->cfg.in_classes > 0) return fail(MMK_ERR_INVALID, "s2s_step: this plan takes class indices (mmk_s2s_step_classes)");
  S2SIo io;
  io.x = x; io.xbs = x_batch_stride; io.xfs = x_frame_stride; io.y = y; io.ybs = y_batch_stride; io.yfs = y_frame_stride;
  return s2s_step(p, batch, io, p->hop, (hipStream_t)stream);
}

extern "C" int mmk_s2s_step_classes(mmk_s2s_plan* p, int32_t batch, const int64_t* x, int64_t x_batch_stride, int64_t x_elem_stride,
                                    int64_t* y, int64_t y_batch_stride, int64_t y_elem_stride, mmk_stream_t stream) {
  if (!p || !x || !y) return fail(MMK_ERR_INVALID, "s2s_step_classes: null argument");
  if (!p->committed) return fail(MMK_ERR_STATE, "s2s_step_classes: plan not committed");
  if (batch < 1 || batch > p->Bmax) return fail(MMK_ERR_INVALID, "s2s_step_classes: batch %d outside [1, %d]", batch, p->Bmax);
  if (p->cfg.in_classes <= 0) return fail(MMK_ERR_INVALID, "s2s_step_classes: this plan takes frames (mmk_s2s_step)");
  S2SIo io;
  io.xi = x; io.xbs = x_batch_stride; io.xfs = x_elem_stride; io.yi = y; io.ybs = y_batch_stride; io.yfs = y_elem_stride;
  return s2s_step(p, batch, io, p->hop, (hipStream_t)stream);
}

extern "C" int mmk_s2s_last_logits(mmk_s2s_plan* p, int32_t batch, float* out, int64_t out_ld, mmk_stream_t stream) {
  if (!p || !out) return fail(MMK_ERR_INVALID, "s2s_last_logits: null argument");
  if (!p->committed || p->cfg.head_kind != 1) return fail(MMK_ERR_STATE, "s2s_last_logits: only for a committed plan with the MLP head");
  if (batch < 1 || batch > p->Bmax) return fail(MMK_ERR_INVALID, "s2s_last_logits: batch %d outside [1, %d]", batch, p->Bmax);
  const int n = p->cfg.out_dim + (p->cfg.learn_temp ? 1 : 0);
  if (out_ld < n) return fail(MMK_ERR_INVALID, "s2s_last_logits: out_ld %lld < %d", (long long)out_ld, n);
  MMK_HIP(hipMemcpy2DAsync(out, (size_t)out_ld * sizeof(float), p->logits, (size_t)p->logits_ld * sizeof(float), (size_t)n * sizeof(float),
                           (size_t)batch * p->hop, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return MMK_OK;
}

extern "C" int mmk_s2s_generate(mmk_s2s_plan* p, int32_t batch, float* frames, int64_t batch_stride, int64_t frame_stride,
                                int64_t t0, int64_t n_steps, int64_t t_total, mmk_stream_t stream) {
  if (!p || !frames) return fail(MMK_ERR_INVALID, "s2s_generate: null argument");
  if (!p->committed) return fail(MMK_ERR_STATE, "s2s_generate: plan not committed");
  if (batch < 1 || batch > p->Bmax) return fail(MMK_ERR_INVALID, "s2s_generate: batch %d outside [1, %d]", batch, p->Bmax);
  if (t0 < p->hop) return fail(MMK_ERR_INVALID, "s2s_generate: t0=%lld is shorter than hop=%d", (long long)t0, p->hop);
  if (p->cfg.in_classes > 0) return fail(MMK_ERR_INVALID, "s2s_generate: this plan takes class indices (mmk_s2s_generate_classes)");
  // loops/generate.py:207-219 : steps covered by the previous call's hop outputs are skipped
  for (int64_t t = t0; t < t0 + n_steps && t < t_total; t += p->hop) {
    const int64_t room = t_total - t;
    const int n_out = (int)(room < p->hop ? room : p->hop);
    S2SIo io;
    io.x = frames + (t - p->hop) * frame_stride; io.y = frames + t * frame_stride;
    io.xbs = io.ybs = batch_stride; io.xfs = io.yfs = frame_stride;
    MMK_TRY(s2s_step(p, batch, io, n_out, (hipStream_t)stream));
  }
  return MMK_OK;
}

extern "C" int mmk_s2s_generate_classes(mmk_s2s_plan* p, int32_t batch, int64_t* classes, int64_t batch_stride, int64_t elem_stride,
                                        int64_t t0, int64_t n_steps, int64_t t_total, mmk_stream_t stream) {
  if (!p || !classes) return fail(MMK_ERR_INVALID, "s2s_generate_classes: null argument");
  if (!p->committed) return fail(MMK_ERR_STATE, "s2s_generate_classes: plan not committed");
  if (batch < 1 || batch > p->Bmax) return fail(MMK_ERR_INVALID, "s2s_generate_classes: batch %d outside [1, %d]", batch, p->Bmax);
  if (t0 < p->hop) return fail(MMK_ERR_INVALID, "s2s_generate_classes: t0=%lld is shorter than hop=%d", (long long)t0, p->hop);
  if (p->cfg.in_classes <= 0) return fail(MMK_ERR_INVALID, "s2s_generate_classes: this plan takes frames (mmk_s2s_generate)");
  for (int64_t t = t0; t < t0 + n_steps && t < t_total; t += p->hop) {
    const int64_t room = t_total - t;
    const int n_out = (int)(room < p->hop ? room : p->hop);
    S2SIo io;
    io.xi = classes + (t - p->hop) * elem_stride; io.yi = classes + t * elem_stride;
    io.xbs = io.ybs = batch_stride; io.xfs = io.yfs = elem_stride;
    MMK_TRY(s2s_step(p, batch, io, n_out, (hipStream_t)stream));
  }
  return MMK_OK;
}

extern "C" int mmk_s2s_sync_status(mmk_s2s_plan* p, mmk_stream_t stream) {
  if (!p) return fail(MMK_ERR_INVALID, "s2s_sync_status: null argument");
  hipStream_t st = (hipStream_t)stream;
  if (!p->committed || !p->seq_lstm) {
    MMK_HIP(hipStreamSynchronize(st));
    return MMK_OK;
  }
  uint32_t word = 0;
  MMK_HIP(hipMemcpyAsync(&word, p->seq_err, sizeof(word), hipMemcpyDeviceToHost, st));
  MMK_HIP(hipStreamSynchronize(st));
  if (const char* senv = diag_only("MMK_S2S_STAMPS"); senv && senv[0] == '1') {
    std::vector<unsigned long long> h(16 * 8 * 8);
    MMK_HIP(hipMemcpy(h.data(), p->seq_stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    const unsigned long long t0 = h[0];
    fprintf(stderr, "[mmk stamps] last resident bi-LSTM launch, workgroup (MMK_S2S_STAMP_WG, 0, 0), per phase and wave, 10 ns ticks since the first phase of wave 0: "
                    "start / products done / barrier passed / cell done / fragments re-requested\n");
    for (int ph = 0; ph < 16; ++ph) {
      fprintf(stderr, "[mmk stamps] phase %2d:", ph);
      for (int wv = 0; wv < 8; wv += 1) {
        const unsigned long long* e = h.data() + (ph * 8 + wv) * 8;
        fprintf(stderr, "  w%d %lld/%lld/%lld/%lld/%llu", wv, (long long)(e[0] - t0), (long long)(e[1] - t0), (long long)(e[2] - t0), (long long)(e[3] - t0), e[4]);
      }
      fprintf(stderr, "\n");
    }
    const double ticks = (double)(h[(15 * 8) * 8] - h[8 * 8]), clocks = (double)(h[(15 * 8) * 8 + 5] - h[8 * 8 + 5]);
    if (ticks > 0) fprintf(stderr, "[mmk stamps] in-kernel clock over phases 1 .. 15 of wave 0: %.0f shader clocks in %.0f ticks of 10 ns = %.3f GHz\n", clocks, ticks, clocks / ticks * 0.1);
  }
  if (word == 0) return MMK_OK;
  // a workgroup of the resident bi-LSTM kernel gave up waiting: the exchange images are in an unknown state - poison both sets again
  MMK_HIP(hipMemsetAsync(p->seq_err, 0, sizeof(uint32_t), st));
  for (int k = 0; k < 2; ++k) MMK_HIP(hipMemsetAsync(p->xch[k], 0xFF, lstm_seq_xch_floats(p->D, p->Bmax, p->hop) * sizeof(float), st));
  MMK_HIP(hipStreamSynchronize(st));
  return fail(MMK_ERR_STATE, "s2s: a wait inside the resident bi-LSTM kernel timed out (code %u): the outputs since the last check are invalid", word);
}

extern "C" int mmk_s2s_inject_sync_error(mmk_s2s_plan* p, mmk_stream_t stream) {
  if (!p || !p->committed) return fail(MMK_ERR_STATE, "s2s_inject_sync_error: plan not committed");
  if (!p->seq_lstm) return fail(MMK_ERR_STATE, "s2s_inject_sync_error: this plan runs the per-step kernels, nothing can time out");
  const uint32_t word = 4;
  MMK_HIP(hipMemcpyAsync(p->seq_err, &word, sizeof(word), hipMemcpyHostToDevice, (hipStream_t)stream));
  MMK_HIP(hipStreamSynchronize((hipStream_t)stream));
  return MMK_OK;
}

extern "C" int64_t mmk_s2s_resident_launches(const mmk_s2s_plan* p) { return p ? p->seq_launches : 0; }

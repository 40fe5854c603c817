// SampleRNN generate plan (host side): tier schedule, recurrent state in HBM,
// per-step launch sequences replayed as hipGraphs of frame_sizes[0] steps.
//
// Reference: SampleRNN.generate_step (sample_rnn_v2.py:236-260), the warm-up of
// before_generate (:226-234) and SampleRNNTier.forward (:83-99).  The device
// counter holds the step index t (the position being predicted); data windows
// are idx[:, t+shift-fs : t+shift] with shift = prompt_len % rf during warm-up
// (the reference slides its window by that offset, :229-234) and 0 afterwards.
#include "plan_util.h"
#include "srnn_bottom.h"
#include "srnn_gru.h"
#include "srnn_resident.h"

using namespace mmk;



struct SrnnCall {
  int M = 0;
  const int64_t* idx = nullptr;                       // input 0 (the fused kernels know one input)
  int64_t idx_rs = 0;
  const int64_t* xidx[MMK_MAX_STREAMS] = {nullptr, nullptr, nullptr, nullptr};   // every input (entry 0 = idx); target k is written to input k
  int64_t xidx_rs[MMK_MAX_STREAMS] = {0, 0, 0, 0};
  int64_t shift = 0;
  const float* temperature = nullptr;
  const float* uniforms = nullptr;
  int64_t uni_ld = 0;
  int64_t uni_off = 0;
};

// a stacked recurrent layer above the first one of a tier (n_rnn > 1): its input is the layer below's new state
struct SrnnDeep {
  PackedLinear gates, gates_hh;
  float *h = nullptr, *c = nullptr;
};

// an output module beyond the first (one per target, sample_rnn_v2.py:181-182): an MLPIO + sampler of its own geometry on the same hidden vector
struct SrnnHead {
  std::vector<PackedLinear> mlp;
  int q = 0, hidden = 0, learn_temp = 0;
  float min_temp = 0.f;
  float* logits = nullptr;
  int logits_ld = 0;
};

struct SrnnTier {
  int fs = 0, up = 0;
  std::vector<SrnnDeep> deep;
  PackedLinear in_lin, gates, gates_hh, up_lin;
  float *h = nullptr, *c = nullptr, *out = nullptr;   // h: [2][Bmax][H] (the fused GRU kernel alternates the slots)
  float *win_raw = nullptr, *bin_raw = nullptr;       // input Linear in its state_dict layout
  int64_t* cnt = nullptr;                             // update counter (slot of the current state = cnt & 1)
  unsigned* done = nullptr;
  unsigned long long* h_gran = nullptr;               // [Bmax][H]: the new state as granules (fused up-sampler phase)
  float* v_comp = nullptr;                            // [G H][16]: W_ih W_in, zero padded (frame sizes <= 16; srnn_gru.hip composed mode)
  // resident mode (srnn_resident.hip): the input half's matrix and constants composed at commit, the granule arrays of the block
  float* v_full = nullptr;                            // [G H][fsp]: W_ih W_in for any frame size
  float* gconst = nullptr;                            // [2][G H]: the gates' input-half constant (top tier: W_ih b_in + b_ih; else with slot 0 of the link) | GRU: b_hh
  float* link_wp = nullptr;                           // tiers with a tier above: packed W_ih W_up,above, rows j G H + g H + u, K = H
  float* link_bias = nullptr;                         // [up,above][G H]: W_ih (b_in + b_up,above[slot]) + b_ih (LSTM: + b_hh)
  unsigned long long* rh_gran = nullptr;              // [2][Bmax][H]
  unsigned long long* rout_gran = nullptr;            // [Bmax][up][G H] gate rows for the tier below (last recurrent tier: [Bmax][S][Hm])
};

struct mmk_srnn_plan {
  Tuning tune;                  // the config's execution switches (plan_util.h): never the environment in the product library
  mmk_srnn_config cfg;
  Binder binder;
  bool committed = false;
  int n_rnn_tiers = 0, H = 0, G = 0, Bmax = 0;
  std::vector<SrnnTier> tiers;
  PackedLinear bottom;
  std::vector<PackedLinear> mlp;
  // more than one input / target (include/mmk.h): one launch per operation; the ZipReduceVariables weights are folded into the packed
  // input products at commit (zip_w: the weights of tier i at [4 i, 4 i + 4), zip_tmp: a scaled copy on its way into the packed matrix)
  int n_in = 1, n_tgt = 1, in_class[MMK_MAX_STREAMS] = {0, 0, 0, 0};
  bool multi = false;
  std::vector<SrnnHead> xheads;                 // targets 1 ..
  float *zip_w = nullptr, *zip_tmp = nullptr;
  float *xbuf = nullptr, *gi = nullptr, *gh = nullptr, *xbot = nullptr, *hid[2] = {nullptr, nullptr}, *logits = nullptr;
  int logits_ld = 0;
  int64_t* tau = nullptr;
  hipStream_t cap_stream = nullptr;
  GraphCache gc;
  // resident mode (srnn_resident.hip): ONE launch per generate block, every tier and the bottom tier resident with their weights in registers
  int64_t resident_blocks = 0;
  int64_t resident_warmups = 0;                 // warm-ups that ran as one teacher-forced resident launch
  unsigned long long* cls_gran = nullptr;       // [Bmax][256]
  unsigned long long* res_gran = nullptr;       // the tiers' granule arrays, one region (cleared at the start of every resident block)
  int64_t res_gran_count = 0;
  unsigned long long* res_prog = nullptr;       // [kResMaxTiers] progress words of a teacher-forced launch, inside res_gran (cleared with it)
  unsigned long long* res_stamps = nullptr;     // diagnostic build: phase totals per role
  float* cp_wp = nullptr;                       // last recurrent tier: W0 W_up[slot] for the slots 1 .. S - 1, packed tiles (rpb rows per unit block)
  float* cp0 = nullptr;                         // (Hm, H): W0 W_up[slot 0]
  float* bcs = nullptr;                         // (S, Hm): W0 (b_up[slot] + bb) + b0
  int cp_rpb = 0, cp_tiles = 0;
  bool resident_ready = false;                  // every composed operand of the mode was built at commit
  // fused bottom tier (srnn_bottom.hip): chosen at create time when the geometry allows it
  bool fused_bottom = false;
  bool fused_gru = false;                       // srnn_gru.hip: input linear + both gate products + cell in one launch
  float *wb_raw = nullptr, *bb_raw = nullptr;   // framed conv weight / bias in their state_dict layout
  const float* mlp_raw[2] = {nullptr, nullptr}; // fc0 / fc2 weights as bound (row-major; the caller keeps them alive with the plan)
  float *a_comp = nullptr, *b_comp = nullptr;   // bottom tier, composed mode: (fs, Hm) = W0 wb_i, (Hm) = W0 bb + b0 (srnn_bottom.hip)
  bool bottom_composed = false;

  void layout(Carver& c) {
    const bool bias = cfg.rnn_bias != 0;
    for (auto& t : tiers) {
      t.in_lin.carve(c, true);
      t.gates.carve(c, bias);
      if (cfg.rnn_kind == 1) t.gates_hh.carve(c, bias);
      t.up_lin.carve(c, true);
      t.h = c.take<float>((int64_t)2 * Bmax * H);
      t.c = c.take<float>((int64_t)Bmax * H);
      t.win_raw = c.take<float>((int64_t)H * t.fs);
      t.bin_raw = c.take<float>(H);
      t.cnt = c.take<int64_t>(4);
      t.done = c.take<unsigned>(4);
      t.h_gran = c.take<unsigned long long>((int64_t)Bmax * H);
      t.v_comp = c.take<float>((int64_t)G * H * 16);
      t.v_full = c.take<float>((int64_t)G * H * round_up(t.fs, 4));
      t.gconst = c.take<float>((int64_t)2 * G * H);
      {
        const size_t ti = (size_t)(&t - tiers.data());
        const int up_above = ti > 0 ? tiers[ti - 1].up : 0;
        t.link_wp = c.take<float>((int64_t)up_above * G * H * H);
        t.link_bias = c.take<float>((int64_t)up_above * G * H);
      }
      t.out = c.take<float>((int64_t)Bmax * t.up * H);
      for (auto& d : t.deep) {
        d.gates.carve(c, bias);
        if (cfg.rnn_kind == 1) d.gates_hh.carve(c, bias);
        d.h = c.take<float>((int64_t)Bmax * H);
        d.c = c.take<float>((int64_t)Bmax * H);
      }
    }
    bottom.carve(c, true);
    for (auto& m : mlp) m.carve(c, true);
    int hid_w = cfg.mlp_hidden, fs_max = 1;
    for (auto& h : xheads) {
      for (auto& m : h.mlp) m.carve(c, true);
      h.logits_ld = (int)round_up(h.q + (h.learn_temp ? 1 : 0), 4);
      h.logits = c.take<float>((int64_t)Bmax * h.logits_ld);
      hid_w = h.hidden > hid_w ? h.hidden : hid_w;
    }
    for (int i = 0; i < cfg.n_tiers; ++i) fs_max = cfg.frame_size[i] > fs_max ? cfg.frame_size[i] : fs_max;
    zip_w = c.take<float>(4 * MMK_MAX_TIERS);
    zip_tmp = c.take<float>((int64_t)H * fs_max + H);
    xbuf = c.take<float>((int64_t)Bmax * H);
    gi = c.take<float>((int64_t)Bmax * G * H);
    gh = c.take<float>((int64_t)Bmax * G * H);
    xbot = c.take<float>((int64_t)Bmax * H);
    hid[0] = c.take<float>((int64_t)Bmax * hid_w);
    hid[1] = c.take<float>((int64_t)Bmax * hid_w);
    logits_ld = (int)round_up(cfg.q_levels + (cfg.learn_temp ? 1 : 0), 4);
    logits = c.take<float>((int64_t)Bmax * logits_ld);
    tau = c.take<int64_t>(32);
    cls_gran = c.take<unsigned long long>((int64_t)Bmax * 256);
    {
      // the tiers' granule arrays of resident mode, one region: [2][Bmax][H] of new state per tier, then its rows for the tier below
      // ([Bmax][up][G H] gate rows; the last recurrent tier: [Bmax][S][mlp_hidden] rows composed with the head's first layer)
      res_gran_count = 0;
      for (size_t i = 0; i < tiers.size(); ++i) {
        const bool last = i + 1 == tiers.size();
        res_gran_count += (int64_t)2 * Bmax * H + (last ? (int64_t)Bmax * tiers[i].up * cfg.mlp_hidden : (int64_t)Bmax * tiers[i].up * G * H);
      }
      res_gran_count += kResMaxTiers;      // + a progress word per tier (teacher-forced launches)
      res_gran = c.take<unsigned long long>(res_gran_count);
      unsigned long long* at = res_gran;
      for (size_t i = 0; i < tiers.size(); ++i) {
        const bool last = i + 1 == tiers.size();
        tiers[i].rh_gran = at;
        if (at) at += (int64_t)2 * Bmax * H;
        tiers[i].rout_gran = at;
        if (at) at += last ? (int64_t)Bmax * tiers[i].up * cfg.mlp_hidden : (int64_t)Bmax * tiers[i].up * G * H;
      }
      res_prog = at;
    }
    res_stamps = c.take<unsigned long long>(8 * (1 + kResMaxTiers));
    {
      const int S = tiers.back().up, KC = (H + 15) / 16;
      cp_rpb = ((S - 1) * cfg.mlp_hidden + KC - 1) / KC;
      cp_tiles = (cp_rpb + 15) / 16;
      cp_wp = c.take<float>((int64_t)KC * cp_tiles * KC * 256);
      cp0 = c.take<float>((int64_t)cfg.mlp_hidden * H);
      bcs = c.take<float>((int64_t)S * cfg.mlp_hidden);
    }
    a_comp = c.take<float>((int64_t)cfg.frame_size[cfg.n_tiers - 1] * cfg.mlp_hidden);
    b_comp = c.take<float>(cfg.mlp_hidden);
    wb_raw = c.take<float>((int64_t)H * cfg.frame_size[cfg.n_tiers - 1]);
    bb_raw = c.take<float>(H);
  }
};

static int derive(mmk_srnn_plan* p) {
  const mmk_srnn_config& c = p->cfg;
  if (c.n_tiers < 2 || c.n_tiers > MMK_MAX_TIERS) return fail(MMK_ERR_INVALID, "srnn: n_tiers=%d outside [2, %d]", c.n_tiers, MMK_MAX_TIERS);
  if (c.hidden_dim < 1 || c.max_batch < 1 || c.q_levels < 2) return fail(MMK_ERR_INVALID, "srnn: bad hidden_dim / max_batch / q_levels");
  if (c.rnn_kind < 0 || c.rnn_kind > 2) return fail(MMK_ERR_INVALID, "srnn: rnn_kind %d unknown", c.rnn_kind);
  if (c.mlp_hidden < 1 || c.mlp_n_hidden < 0 || c.mlp_n_hidden > MMK_MAX_MLP_HIDDEN) return fail(MMK_ERR_INVALID, "srnn: bad MLP head geometry");
  p->H = c.hidden_dim;
  p->Bmax = c.max_batch;
  p->G = c.rnn_kind == 0 ? 4 : (c.rnn_kind == 1 ? 3 : 1);
  p->n_rnn_tiers = c.n_tiers - 1;
  p->tiers.resize(p->n_rnn_tiers);
  p->n_in = c.n_inputs > 1 ? c.n_inputs : 1;
  p->n_tgt = c.n_targets > 1 ? c.n_targets : 1;
  if (p->n_in > MMK_MAX_STREAMS) return fail(MMK_ERR_UNSUPPORTED, "srnn: %d inputs (at most %d)", p->n_in, MMK_MAX_STREAMS);
  if (p->n_tgt > p->n_in) return fail(MMK_ERR_UNSUPPORTED, "srnn: %d targets for %d inputs (the loop writes output k into input k)", p->n_tgt, p->n_in);
  if (c.inputs_mode < 0 || c.inputs_mode > 2) return fail(MMK_ERR_INVALID, "srnn: inputs_mode %d unknown", c.inputs_mode);
  for (int m = 0; m < p->n_in; ++m) {
    p->in_class[m] = c.in_class[m] > 0 ? c.in_class[m] : c.q_levels;
    if (p->in_class[m] < 2) return fail(MMK_ERR_INVALID, "srnn: input %d has %d classes", m, p->in_class[m]);
  }
  p->multi = p->n_in > 1 || p->n_tgt > 1 || p->in_class[0] != c.q_levels;

  for (int i = 0; i < p->n_rnn_tiers; ++i) {
    SrnnTier& t = p->tiers[i];
    t.fs = c.frame_size[i];
    const int next = (i < c.n_tiers - 2) ? c.frame_size[i + 1] : 1;  // from_config, sample_rnn_v2.py:155-158
    if (t.fs < 1 || next < 1 || t.fs % next != 0)
      return fail(MMK_ERR_INVALID, "srnn: frame_sizes[%d]=%d is not a multiple of the next tier's %d", i, t.fs, next);
    if (c.frame_size[0] % t.fs != 0)
      return fail(MMK_ERR_UNSUPPORTED, "srnn: frame_sizes[%d]=%d does not divide frame_sizes[0]=%d", i, t.fs, c.frame_size[0]);
    t.up = t.fs / next;
    t.in_lin.set_geometry(p->H, std::vector<int>(p->n_in, t.fs));      // K = [frame of input 0 | frame of input 1 | ...]
    if (c.rnn_kind == 1) {
      t.gates.set_geometry(3 * p->H, {p->H});
      t.gates_hh.set_geometry(3 * p->H, {p->H});
    } else {
      t.gates.set_geometry(p->G * p->H, {p->H, p->H});
    }
    t.up_lin.set_geometry(p->H * t.up, {p->H});
    const int n_rnn = c.n_rnn > 1 ? c.n_rnn : 1;
    t.deep.resize(n_rnn - 1);
    for (auto& d : t.deep) {
      if (c.rnn_kind == 1) {
        d.gates.set_geometry(3 * p->H, {p->H});
        d.gates_hh.set_geometry(3 * p->H, {p->H});
      } else {
        d.gates.set_geometry(p->G * p->H, {p->H, p->H});
      }
    }
  }
  if (c.frame_size[c.n_tiers - 1] < 1) return fail(MMK_ERR_INVALID, "srnn: bad bottom frame size");
  p->bottom.set_geometry(p->H, std::vector<int>(p->n_in, c.frame_size[c.n_tiers - 1]));
  p->mlp.clear();
  PackedLinear first;
  first.set_geometry(c.mlp_hidden, {p->H});
  p->mlp.push_back(first);
  for (int i = 0; i < c.mlp_n_hidden; ++i) {
    PackedLinear h;
    h.set_geometry(c.mlp_hidden, {c.mlp_hidden});
    p->mlp.push_back(h);
  }
  PackedLinear last;
  last.set_geometry(c.q_levels + (c.learn_temp ? 1 : 0), {c.mlp_hidden});
  p->mlp.push_back(last);
  p->xheads.clear();
  for (int k = 1; k < p->n_tgt; ++k) {
    SrnnHead h;
    h.q = c.x_q_levels[k]; h.hidden = c.x_mlp_hidden[k]; h.learn_temp = c.x_learn_temp[k]; h.min_temp = c.x_min_temp[k];
    const int nh = c.x_mlp_n_hidden[k];
    if (h.q < 2 || h.hidden < 1 || nh < 0 || nh > MMK_MAX_MLP_HIDDEN) return fail(MMK_ERR_INVALID, "srnn: bad MLP head geometry of target %d", k);
    if (h.q > p->in_class[k]) return fail(MMK_ERR_INVALID, "srnn: target %d draws from %d classes, input %d holds %d", k, h.q, k, p->in_class[k]);
    PackedLinear f0;
    f0.set_geometry(h.hidden, {p->H});
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
  const char* fenv = p->tune.get("MMK_SRNN_FUSED");
  p->fused_bottom = !(fenv && fenv[0] == '0') && c.mlp_n_hidden == 0 && c.mlp_act == ACT_MISH &&      // (the fused and resident kernels have Mish built in)
                    srnn_bottom_supported(p->H, c.mlp_hidden, c.q_levels + (c.learn_temp ? 1 : 0), c.frame_size[c.n_tiers - 1]);
  p->fused_gru = !(fenv && fenv[0] == '0') && (c.rnn_kind == 1 || c.rnn_kind == 0);   // GRU or LSTM tiers
  for (auto& t : p->tiers) p->fused_gru = p->fused_gru && srnn_gru_supported(p->H, t.fs, c.rnn_kind == 0);
  if (c.n_rnn > 1) p->fused_gru = false;    // stacked layers: one launch per op (the fused kernel's up-sampler reads layer 0)
  if (p->multi) p->fused_gru = p->fused_bottom = false;     // the fused kernels know one class stream, read and written
  if (c.n_rnn > 8) return fail(MMK_ERR_UNSUPPORTED, "srnn: n_rnn=%d", c.n_rnn);
  return MMK_OK;
}

extern "C" int mmk_srnn_plan_create(const mmk_srnn_config* cfg, mmk_srnn_plan** out) {
  if (!cfg || !out) return fail(MMK_ERR_INVALID, "srnn_plan_create: null argument");
  mmk_srnn_plan* p = new mmk_srnn_plan();
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

extern "C" void mmk_srnn_plan_destroy(mmk_srnn_plan* p) {
  if (!p) return;
  p->gc.reset();
  if (p->cap_stream) (void)hipStreamDestroy(p->cap_stream);
  delete p;
}

extern "C" int mmk_srnn_plan_bind(mmk_srnn_plan* p, const char* key, const float* dev_ptr, int64_t numel) {
  if (!p || !key || !dev_ptr) return fail(MMK_ERR_INVALID, "srnn_plan_bind: null argument");
  p->binder.bind(key, dev_ptr, numel);
  p->committed = false;
  return MMK_OK;
}

extern "C" size_t mmk_srnn_workspace_bytes(const mmk_srnn_plan* p) {
  if (!p) return 0;
  mmk_srnn_plan tmp = *p;
  tmp.gc = GraphCache();
  tmp.cap_stream = nullptr;
  Carver c(nullptr);
  tmp.layout(c);
  return c.used();
}

// ZipReduceVariables' weights (modules/io.py:296-302, :305-308): sum -> 1, mean -> 1 / M, static_mix -> softmax of the parameter
__global__ void srnn_zip_weights_kernel(const float* __restrict__ param, int M, int mode, float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (mode == 2 && param) {
    float mx = param[0], sum = 0.f, e[MMK_MAX_STREAMS];
    for (int m = 1; m < M; ++m) mx = fmaxf(mx, param[m]);
    for (int m = 0; m < M; ++m) { e[m] = expf(param[m] - mx); sum += e[m]; }
    for (int m = 0; m < M; ++m) out[m] = e[m] / sum;
  } else {
    for (int m = 0; m < M; ++m) out[m] = mode == 1 ? 1.f / (float)M : 1.f;
  }
}
// dst = src * w[0]: head m's matrix and bias with its weight folded in ((W x + b) w = (w W) x + w b)
__global__ void srnn_scale_kernel(const float* __restrict__ src, int64_t n, const float* __restrict__ w, float* __restrict__ dst) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n) dst[e] = src[e] * w[0];
}

// V[r][i] = sum_k W_ih[r][k] W_in[k][i]  (i < fs; the other columns zero): fp64 accumulation, rounded once
__global__ void srnn_compose_kernel(const float* __restrict__ wih, const float* __restrict__ win, int rows, int H, int fs, float* __restrict__ V) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)rows * 16) return;
  const int r = (int)(e >> 4), i = (int)(e & 15);
  double acc = 0.0;
  if (i < fs)
    for (int k = 0; k < H; ++k) acc += (double)wih[(int64_t)r * H + k] * (double)win[(int64_t)k * fs + i];
  V[e] = (float)acc;
}

// bottom tier, composed mode: A[i][u] = sum_k W0[u][k] wb[k][i] (i < fs), b'[u] = sum_k W0[u][k] bb[k] + b0[u]; fp64, rounded once
__global__ void srnn_compose_bottom_kernel(const float* __restrict__ w0, const float* __restrict__ b0, const float* __restrict__ wb,
                                           const float* __restrict__ bb, int Hm, int H, int fs, float* __restrict__ A, float* __restrict__ B) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (fs + 1) * Hm) return;
  const int i = e / Hm, u = e - i * Hm;
  double acc = 0.0;
  if (i < fs) {
    for (int k = 0; k < H; ++k) acc += (double)w0[(int64_t)u * H + k] * (double)wb[(int64_t)k * fs + i];
    A[e] = (float)acc;
  } else {
    for (int k = 0; k < H; ++k) acc += (double)w0[(int64_t)u * H + k] * (double)bb[k];
    B[u] = (float)(acc + (double)b0[u]);
  }
}

// ---- resident mode (srnn_resident.hip): operands composed at commit, fp64 accumulation, rounded once -----------------------------
// V[r][i] = sum_k W_ih[r][k] W_in[k][i] for any frame size (rows of fsp = fs rounded up to 4, zero padded)
__global__ void srnn_compose_v_kernel(const float* __restrict__ wih, const float* __restrict__ win, int rows, int H, int fs, int fsp, float* __restrict__ V) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)rows * fsp) return;
  const int r = (int)(e / fsp), i = (int)(e - (int64_t)r * fsp);
  double acc = 0.0;
  if (i < fs)
    for (int k = 0; k < H; ++k) acc += (double)wih[(int64_t)r * H + k] * (double)win[(int64_t)k * fs + i];
  V[e] = (float)acc;
}
// gc[j][r] = sum_k W_ih[r][k] (b_in[k] + b_up[j H + k]) + b_ih[r] (+ b_hh[r] when `sum_hh`: the LSTM cell adds both) for the slots j < n_slots of the
// tier above (b_up null, n_slots 1: the top tier); hh[r] = b_hh[r] otherwise (the GRU cell keeps the recurrent bias inside r (W_hn h + b_hn)), when
// `hh` is given; null biases count as zero
__global__ void srnn_compose_gconst_kernel(const float* __restrict__ wih, const float* __restrict__ bin, const float* __restrict__ bup,
                                           const float* __restrict__ bih, const float* __restrict__ bhh, int rows, int H, int n_slots, int sum_hh,
                                           float* __restrict__ gc, float* __restrict__ hh) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= rows * n_slots) return;
  const int j = e / rows, r = e - j * rows;
  double acc = 0.0;
  for (int k = 0; k < H; ++k) acc += (double)wih[(int64_t)r * H + k] * ((double)bin[k] + (bup ? (double)bup[(int64_t)j * H + k] : 0.0));
  if (bih) acc += (double)bih[r];
  if (sum_hh && bhh) acc += (double)bhh[r];
  gc[e] = (float)acc;
  if (hh && j == 0) hh[r] = (!sum_hh && bhh) ? bhh[r] : 0.f;
}
// The link of a tier to the tier above: L[j G H + r][k] = sum_m W_ih[r][m] W_up[j H + m][k] - the gates' input half as a function of the tier above's
// STATE, slot by slot -, as packed MFMA tiles (linear.hip's order), K = H
__global__ void srnn_compose_link_kernel(const float* __restrict__ wih, const float* __restrict__ wup, int GH, int H, int n_slots, float* __restrict__ link_wp) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)n_slots * GH * H) return;
  const int k = (int)(e % H);
  const int64_t R = e / H;
  const int r = (int)(R % GH), j = (int)(R / GH);
  double acc = 0.0;
  for (int m = 0; m < H; ++m) acc += (double)wih[(int64_t)r * H + m] * (double)wup[((int64_t)j * H + m) * H + k];
  const int KC = H / 16, k16 = k & 15;
  link_wp[((((R >> 4) * KC + (k >> 4)) * 64) + (k16 >> 2) * 16 + (R & 15)) * 4 + (k16 & 3)] = (float)acc;
}
// The head's first layer through the last recurrent tier's up-sampler: C[j][u][k] = sum_m W0[u][m] W_up[j H + m][k].  Slot 0 goes out row-major
// (Hm, H) - the clip's own workgroup multiplies it -, the slots 1 .. S - 1 as packed MFMA tiles (linear.hip's order) of `rpb` rows per unit
// block: row g = (j - 1) Hm + u sits in unit block g / rpb at row g % rpb of that block's `tiles` tiles.
__global__ void srnn_compose_cp_kernel(const float* __restrict__ w0, const float* __restrict__ wup, int Hm, int H, int S, int rpb, int tiles,
                                       float* __restrict__ cp0, float* __restrict__ cp_wp) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)S * Hm * H) return;
  const int k = (int)(e % H);
  const int64_t ju = e / H;
  const int u = (int)(ju % Hm), j = (int)(ju / Hm);
  double acc = 0.0;
  for (int m = 0; m < H; ++m) acc += (double)w0[(int64_t)u * H + m] * (double)wup[((int64_t)j * H + m) * H + k];
  if (j == 0) {
    cp0[(int64_t)u * H + k] = (float)acc;
    return;
  }
  const int g = (j - 1) * Hm + u, ub = g / rpb, r = g - ub * rpb;
  const int KC = H / 16, prow = ub * tiles * 16 + r, tile = prow >> 4, rin = prow & 15, c = k >> 4, k16 = k & 15;
  cp_wp[((((int64_t)tile * KC + c) * 64) + (k16 >> 2) * 16 + rin) * 4 + (k16 & 3)] = (float)acc;
}
// bcs[j][u] = sum_m W0[u][m] (b_up[j H + m] + bb[m]) + b0[u]
__global__ void srnn_compose_bcs_kernel(const float* __restrict__ w0, const float* __restrict__ b0, const float* __restrict__ bup,
                                        const float* __restrict__ bb, int Hm, int H, int S, float* __restrict__ bcs) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= S * Hm) return;
  const int j = e / Hm, u = e - j * Hm;
  double acc = 0.0;
  for (int m = 0; m < H; ++m) acc += (double)w0[(int64_t)u * H + m] * ((double)bup[(int64_t)j * H + m] + (double)bb[m]);
  bcs[e] = (float)(acc + (double)b0[u]);
}
// Start of a resident block at step t_begin: the class ring holds the 256 positions before it
__global__ void srnn_resident_init_kernel(unsigned long long* cls_gran, const int64_t* idx, int64_t idx_rs, int B, int64_t t_begin) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (int64_t)B * 256) {
    const int c = (int)(i >> 8), slot = (int)(i & 255);
    int64_t pos = ((t_begin - 1) & ~(int64_t)255) + slot;          // the position < t_begin with this residue
    if (pos >= t_begin) pos -= 256;
    cls_gran[i] = pos >= 0 ? (((unsigned long long)(unsigned)(pos + 1) << 32) | (unsigned)idx[(int64_t)c * idx_rs + pos]) : 0ull;
  }
}

extern "C" int mmk_srnn_reset(mmk_srnn_plan* p, mmk_stream_t stream) {
  if (!p || !p->committed) return fail(MMK_ERR_STATE, "srnn_reset: plan not committed");
  hipStream_t st = (hipStream_t)stream;
  {   // a grid barrier of the fused tier kernel that timed out during the previous generation (never seen: the launcher
      // only fuses when the grid is resident at once) would have left wrong samples behind - say so instead of hiding it
    int64_t err = 0;
    MMK_HIP(hipMemcpyAsync(&err, p->tau + 4, sizeof(err), hipMemcpyDeviceToHost, st));
    MMK_HIP(hipStreamSynchronize(st));
    if (err != 0) {
      MMK_HIP(hipMemsetAsync(p->tau + 4, 0, sizeof(int64_t), st));
      return fail(MMK_ERR_STATE, "srnn: a wait inside the tier / bottom kernels timed out during the previous generation (code %lld: 3 grid barrier of the tier kernel, "
                  "6 bottom role / 7 tier role of the resident kernel; MMK_SRNN_RESIDENT=0 runs the tiers and the bottom in turns)", (long long)err);
    }
  }
  for (auto& t : p->tiers) {
    // h0_init zeros / ones (SampleRNNTier._init_h0, sample_rnn_v2.py:118-119)
    MMK_TRY(launch_fill(t.h, p->cfg.h0_ones ? 1.f : 0.f, (int64_t)2 * p->Bmax * p->H, st));
    MMK_HIP(hipMemsetAsync(t.cnt, 0, 4 * sizeof(int64_t), st));
    MMK_HIP(hipMemsetAsync(t.done, 0, 4 * sizeof(unsigned), st));
    MMK_HIP(hipMemsetAsync(t.h_gran, 0, (size_t)p->Bmax * p->H * sizeof(unsigned long long), st));   // update numbers restart at 1
    MMK_TRY(launch_fill(t.c, p->cfg.h0_ones ? 1.f : 0.f, (int64_t)p->Bmax * p->H, st));
    MMK_HIP(hipMemsetAsync(t.out, 0, (size_t)p->Bmax * t.up * p->H * sizeof(float), st));
    for (auto& d : t.deep) {
      MMK_TRY(launch_fill(d.h, p->cfg.h0_ones ? 1.f : 0.f, (int64_t)p->Bmax * p->H, st));
      MMK_TRY(launch_fill(d.c, p->cfg.h0_ones ? 1.f : 0.f, (int64_t)p->Bmax * p->H, st));
    }
  }
  return MMK_OK;
}

// the input module of a tier over several inputs: head m's framed linear, times its ZipReduceVariables weight, into K segment m of `lin`
// (`mod` = "tiers.i.input_module.", `leaf` = the key between "heads.m." and "weight": "2." for a tier, "2.2.cv." for the bottom tier's convolution)
static int pack_zipped(mmk_srnn_plan* p, int tier, const std::string& mod, const std::string& leaf, PackedLinear& lin, int fs, hipStream_t st) {
  Binder& b = p->binder;
  const int H = p->H, M = p->n_in;
  float* wz = p->zip_w + 4 * tier;
  const float* param = p->cfg.inputs_mode == 2 ? b.need(mod + "weights", M) : nullptr;
  if (p->cfg.inputs_mode == 2 && !param) return MMK_OK;          // (reported with the other missing keys)
  hipLaunchKernelGGL(srnn_zip_weights_kernel, dim3(1), dim3(64), 0, st, param, M, (int)p->cfg.inputs_mode, wz);
  MMK_HIP(hipGetLastError());
  for (int m = 0; m < M; ++m) {
    const std::string hb = mod + "heads." + std::to_string(m) + "." + leaf;
    const float* w = b.need(hb + "weight", (int64_t)H * fs);
    const float* bb = b.need(hb + "bias", H);
    if (!w || !bb) continue;
    const int64_t nw = (int64_t)H * fs;
    hipLaunchKernelGGL(srnn_scale_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, st, w, nw, wz + m, p->zip_tmp);
    hipLaunchKernelGGL(srnn_scale_kernel, dim3((unsigned)((H + 255) / 256)), dim3(256), 0, st, bb, (int64_t)H, wz + m, p->zip_tmp + nw);
    MMK_HIP(hipGetLastError());
    MMK_TRY(pack_rect(lin.Wp, lin.k_chunks, 0, 1, H, lin.seg_chunk0[m], fs, p->zip_tmp, fs, 1, st));
    MMK_TRY(pack_bias(lin.bias, 0, 1, H, p->zip_tmp + nw, m > 0 ? 1 : 0, st));
  }
  return MMK_OK;
}

extern "C" int mmk_srnn_commit(mmk_srnn_plan* p, void* workspace, size_t workspace_bytes, mmk_stream_t stream) {
  if (!p || !workspace) return fail(MMK_ERR_INVALID, "srnn_commit: null argument");
  if ((reinterpret_cast<uintptr_t>(workspace) & 255) != 0) return fail(MMK_ERR_WORKSPACE, "srnn_commit: workspace must be 256-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const mmk_srnn_config& c = p->cfg;
  Carver carve(workspace);
  p->layout(carve);
  if (carve.used() > workspace_bytes)
    return fail(MMK_ERR_WORKSPACE, "srnn_commit: workspace of %zu bytes, %zu needed", workspace_bytes, carve.used());
  MMK_HIP(hipStreamSynchronize(st));   // replays of the cached graph may still be queued: wait before destroying it
  p->gc.reset();
  MMK_HIP(hipMemsetAsync(workspace, 0, carve.used(), st));
  Binder& b = p->binder;
  b.clear_missing();
  const int H = p->H, G = p->G;
  const bool bias = c.rnn_bias != 0;
  const float *last_wu = nullptr, *last_bu = nullptr;      // the last recurrent tier's up-sampler as bound (composed with the head below)
  const float *prev_wu = nullptr, *prev_bu = nullptr;      // the up-sampler of the tier above the one being committed (composed with its gates)
  int links_built = 0;
  p->resident_ready = false;
  for (int i = 0; i < p->n_rnn_tiers; ++i) {
    SrnnTier& t = p->tiers[i];
    const std::string tb = "tiers." + std::to_string(i) + ".";
    // input_module = ZipReduceVariables([Sequential(Linearizer, Unfold, Linear)])  -> heads.0.2
    const float* w = b.need(tb + "input_module.heads.0.2.weight", (int64_t)H * t.fs);
    const float* bb = b.need(tb + "input_module.heads.0.2.bias", H);
    if (p->n_in > 1) {
      MMK_TRY(pack_zipped(p, i, tb + "input_module.", "2.", t.in_lin, t.fs, st));
    } else {
      if (w) MMK_TRY(pack_rect(t.in_lin.Wp, t.in_lin.k_chunks, 0, 1, H, 0, t.fs, w, t.fs, 1, st));
      if (bb) MMK_TRY(pack_bias(t.in_lin.bias, 0, 1, H, bb, 0, st));
    }
    if (w) MMK_HIP(hipMemcpyAsync(t.win_raw, w, (size_t)H * t.fs * sizeof(float), hipMemcpyDeviceToDevice, st));
    if (bb) MMK_HIP(hipMemcpyAsync(t.bin_raw, bb, (size_t)H * sizeof(float), hipMemcpyDeviceToDevice, st));
    const float* wih = b.need(tb + "rnn.weight_ih_l0", (int64_t)G * H * H);
    const float* whh = b.need(tb + "rnn.weight_hh_l0", (int64_t)G * H * H);
    if (wih && w && t.fs <= 16 && p->fused_gru) {
      hipLaunchKernelGGL(srnn_compose_kernel, dim3((unsigned)(((int64_t)G * H * 16 + 255) / 256)), dim3(256), 0, st, wih, w, G * H, H, t.fs, t.v_comp);
      MMK_HIP(hipGetLastError());
    }
    const float* bih = bias ? b.need(tb + "rnn.bias_ih_l0", (int64_t)G * H) : nullptr;
    const float* bhh = bias ? b.need(tb + "rnn.bias_hh_l0", (int64_t)G * H) : nullptr;
    if (wih && w && bb && p->fused_gru && (!bias || (bih && bhh)) && (i == 0 || (prev_wu && prev_bu))) {
      // resident mode: the input half of the gates without its input Linear - and, below the top tier, without the up-sampler of the tier above
      const int fsp = (int)round_up(t.fs, 4), up_above = i > 0 ? p->tiers[i - 1].up : 0;
      hipLaunchKernelGGL(srnn_compose_v_kernel, dim3((unsigned)(((int64_t)G * H * fsp + 255) / 256)), dim3(256), 0, st, wih, w, G * H, H, t.fs, fsp, t.v_full);
      if (i == 0) {
        hipLaunchKernelGGL(srnn_compose_gconst_kernel, dim3((unsigned)((G * H + 255) / 256)), dim3(256), 0, st, wih, bb, (const float*)nullptr, bih, bhh, G * H, H, 1,
                           c.rnn_kind == 0 ? 1 : 0, t.gconst, t.gconst + G * H);
      } else {
        hipLaunchKernelGGL(srnn_compose_gconst_kernel, dim3((unsigned)((up_above * G * H + 255) / 256)), dim3(256), 0, st, wih, bb, prev_bu, bih, bhh, G * H, H, up_above,
                           c.rnn_kind == 0 ? 1 : 0, t.link_bias, t.gconst + G * H);
        MMK_HIP(hipMemcpyAsync(t.gconst, t.link_bias, (size_t)G * H * sizeof(float), hipMemcpyDeviceToDevice, st));     // slot 0: the tier multiplies it itself
        hipLaunchKernelGGL(srnn_compose_link_kernel, dim3((unsigned)(((int64_t)up_above * G * H * H + 255) / 256)), dim3(256), 0, st, wih, prev_wu, G * H, H, up_above, t.link_wp);
      }
      MMK_HIP(hipGetLastError());
      ++links_built;
    }
    if (c.rnn_kind == 1) {
      if (wih) MMK_TRY(pack_rect(t.gates.Wp, t.gates.k_chunks, 0, 1, G * H, 0, H, wih, H, 1, st));
      if (whh) MMK_TRY(pack_rect(t.gates_hh.Wp, t.gates_hh.k_chunks, 0, 1, G * H, 0, H, whh, H, 1, st));
      if (bih) MMK_TRY(pack_bias(t.gates.bias, 0, 1, G * H, bih, 0, st));
      if (bhh) MMK_TRY(pack_bias(t.gates_hh.bias, 0, 1, G * H, bhh, 0, st));
    } else {
      if (wih) MMK_TRY(pack_rect(t.gates.Wp, t.gates.k_chunks, 0, 1, G * H, t.gates.seg_chunk0[0], H, wih, H, 1, st));
      if (whh) MMK_TRY(pack_rect(t.gates.Wp, t.gates.k_chunks, 0, 1, G * H, t.gates.seg_chunk0[1], H, whh, H, 1, st));
      if (bih) MMK_TRY(pack_bias(t.gates.bias, 0, 1, G * H, bih, 0, st));
      if (bhh) MMK_TRY(pack_bias(t.gates.bias, 0, 1, G * H, bhh, 1, st));
    }
    for (size_t k = 0; k < t.deep.size(); ++k) {
      SrnnDeep& d = t.deep[k];
      const std::string sfx = "_l" + std::to_string(k + 1);
      const float* dih = b.need(tb + "rnn.weight_ih" + sfx, (int64_t)G * H * H);
      const float* dhh = b.need(tb + "rnn.weight_hh" + sfx, (int64_t)G * H * H);
      const float* dbi = bias ? b.need(tb + "rnn.bias_ih" + sfx, (int64_t)G * H) : nullptr;
      const float* dbh = bias ? b.need(tb + "rnn.bias_hh" + sfx, (int64_t)G * H) : nullptr;
      if (c.rnn_kind == 1) {
        if (dih) MMK_TRY(pack_rect(d.gates.Wp, d.gates.k_chunks, 0, 1, G * H, 0, H, dih, H, 1, st));
        if (dhh) MMK_TRY(pack_rect(d.gates_hh.Wp, d.gates_hh.k_chunks, 0, 1, G * H, 0, H, dhh, H, 1, st));
        if (dbi) MMK_TRY(pack_bias(d.gates.bias, 0, 1, G * H, dbi, 0, st));
        if (dbh) MMK_TRY(pack_bias(d.gates_hh.bias, 0, 1, G * H, dbh, 0, st));
      } else {
        if (dih) MMK_TRY(pack_rect(d.gates.Wp, d.gates.k_chunks, 0, 1, G * H, d.gates.seg_chunk0[0], H, dih, H, 1, st));
        if (dhh) MMK_TRY(pack_rect(d.gates.Wp, d.gates.k_chunks, 0, 1, G * H, d.gates.seg_chunk0[1], H, dhh, H, 1, st));
        if (dbi) MMK_TRY(pack_bias(d.gates.bias, 0, 1, G * H, dbi, 0, st));
        if (dbh) MMK_TRY(pack_bias(d.gates.bias, 0, 1, G * H, dbh, 1, st));
      }
    }
    const float* wu = b.need(tb + "up_sampler.fc.weight", (int64_t)H * t.up * H);
    const float* bu = b.need(tb + "up_sampler.fc.bias", (int64_t)H * t.up);
    if (wu) MMK_TRY(pack_rect(t.up_lin.Wp, t.up_lin.k_chunks, 0, 1, H * t.up, 0, H, wu, H, 1, st));
    if (bu) MMK_TRY(pack_bias(t.up_lin.bias, 0, 1, H * t.up, bu, 0, st));
    if (i == p->n_rnn_tiers - 1) { last_wu = wu; last_bu = bu; }
    prev_wu = wu; prev_bu = bu;
  }
  {
    // bottom tier: FramedConv1dIO -> heads.0 = Sequential(Linearizer, Unfold, Sequential(Flatten, Unsqueeze, Conv1dResampler))
    const int fsl = c.frame_size[c.n_tiers - 1];
    const std::string tb = "tiers." + std::to_string(c.n_tiers - 1) + ".input_module.heads.0.2.2.cv.";
    const float* w = b.need(tb + "weight", (int64_t)H * fsl);
    const float* bb = b.need(tb + "bias", H);
    if (p->n_in > 1) {
      MMK_TRY(pack_zipped(p, c.n_tiers - 1, "tiers." + std::to_string(c.n_tiers - 1) + ".input_module.", "2.2.cv.", p->bottom, fsl, st));
    } else {
      if (w) MMK_TRY(pack_rect(p->bottom.Wp, p->bottom.k_chunks, 0, 1, H, 0, fsl, w, fsl, 1, st));
      if (bb) MMK_TRY(pack_bias(p->bottom.bias, 0, 1, H, bb, 0, st));
    }
    if (w) MMK_HIP(hipMemcpyAsync(p->wb_raw, w, (size_t)H * fsl * sizeof(float), hipMemcpyDeviceToDevice, st));
    if (bb) MMK_HIP(hipMemcpyAsync(p->bb_raw, bb, (size_t)H * sizeof(float), hipMemcpyDeviceToDevice, st));
  }
  for (size_t i = 0; i < p->mlp.size(); ++i) {
    PackedLinear& m = p->mlp[i];
    const std::string kb = "output_modules.0.estimator.0.fc." + std::to_string(2 * i) + ".";
    const float* w = b.need(kb + "weight", (int64_t)m.N * m.segK[0]);
    const float* bb = b.need(kb + "bias", m.N);
    if (w) MMK_TRY(pack_rect(m.Wp, m.k_chunks, 0, 1, m.N, 0, m.segK[0], w, m.segK[0], 1, st));
    if (bb) MMK_TRY(pack_bias(m.bias, 0, 1, m.N, bb, 0, st));
    if (p->mlp.size() == 2) p->mlp_raw[i] = w;
    if (p->mlp.size() == 2 && i == 0 && w && bb && p->fused_bottom) {   // (wb_raw / bb_raw were copied above, on this stream)
      const int fsl = c.frame_size[c.n_tiers - 1];
      hipLaunchKernelGGL(srnn_compose_bottom_kernel, dim3(((fsl + 1) * m.N + 255) / 256), dim3(256), 0, st, w, bb, p->wb_raw, p->bb_raw, m.N,
                         H, fsl, p->a_comp, p->b_comp);
      MMK_HIP(hipGetLastError());
      p->bottom_composed = true;
      if (p->fused_gru && last_wu && last_bu && links_built == p->n_rnn_tiers) {
        // resident mode: the head's first layer through the last recurrent tier's up-sampler (srnn_resident.hip)
        const int S = p->tiers.back().up, Hm = m.N;
        hipLaunchKernelGGL(srnn_compose_cp_kernel, dim3((unsigned)(((int64_t)S * Hm * H + 255) / 256)), dim3(256), 0, st, w, last_wu, Hm, H, S,
                           p->cp_rpb, p->cp_tiles, p->cp0, p->cp_wp);
        hipLaunchKernelGGL(srnn_compose_bcs_kernel, dim3((S * Hm + 255) / 256), dim3(256), 0, st, w, bb, last_bu, p->bb_raw, Hm, H, S, p->bcs);
        MMK_HIP(hipGetLastError());
        p->resident_ready = true;
      }
    }
  }
  for (size_t k = 0; k < p->xheads.size(); ++k) {
    SrnnHead& h = p->xheads[k];
    for (size_t i = 0; i < h.mlp.size(); ++i) {
      PackedLinear& m = h.mlp[i];
      const std::string kb = "output_modules." + std::to_string(k + 1) + ".estimator.0.fc." + std::to_string(2 * i) + ".";
      const float* w = b.need(kb + "weight", (int64_t)m.N * m.segK[0]);
      const float* bb = b.need(kb + "bias", m.N);
      if (w) MMK_TRY(pack_rect(m.Wp, m.k_chunks, 0, 1, m.N, 0, m.segK[0], w, m.segK[0], 1, st));
      if (bb) MMK_TRY(pack_bias(m.bias, 0, 1, m.N, bb, 0, st));
    }
  }
  if (!b.missing().empty()) return fail(MMK_ERR_KEY, "srnn_commit: state_dict tensor %s", b.missing().c_str());
  if (!p->cap_stream) MMK_HIP(hipStreamCreateWithFlags(&p->cap_stream, hipStreamNonBlocking));
  MMK_HIP(hipMemsetAsync(p->tau, 0, 32 * sizeof(int64_t), st));     // position counter, error word, diagnostic stamps
  p->committed = true;
  return mmk_srnn_reset(p, stream);
}

static SrnnBottomArgs bottom_args(mmk_srnn_plan* p, const SrnnCall& call, int64_t tau_off, int64_t n_steps) {
  const mmk_srnn_config& c = p->cfg;
  const int H = p->H;
  SrnnTier& up = p->tiers[p->n_rnn_tiers - 1];
  SrnnBottomArgs a = {};
  a.B = call.M; a.H = H; a.Hm = c.mlp_hidden; a.Q = c.q_levels; a.n_out = c.q_levels + (c.learn_temp ? 1 : 0);
  a.learn_temp = c.learn_temp; a.min_temp = c.min_temp; a.class_size = (float)c.q_levels;
  a.fs = c.frame_size[c.n_tiers - 1]; a.up_slots = up.up;
  a.n_steps = (int32_t)n_steps;
  a.tau_ptr = p->tau; a.tau_off = tau_off;
  a.idx = const_cast<int64_t*>(call.idx); a.idx_rs = call.idx_rs;
  a.wb = p->wb_raw; a.bb = p->bb_raw; a.upper = up.out;
  a.fc0_wp = p->mlp[0].Wp; a.fc0_bias = p->mlp[0].bias; a.fc2_wp = p->mlp[1].Wp; a.fc2_bias = p->mlp[1].bias;
  a.fc0_raw = p->mlp_raw[0]; a.fc2_raw = p->mlp_raw[1];
  {
    const char* cenv = p->tune.get("MMK_SRNN_COMPOSED");
    if (p->bottom_composed && !(cenv && cenv[0] == '0')) { a.a_comp = p->a_comp; a.b_comp = p->b_comp; }
  }
  a.temperature = call.temperature; a.uniforms = call.uniforms; a.uni_ld = call.uni_ld; a.uni_off = call.uni_off;
  a.logits_out = p->logits; a.logits_ld = p->logits_ld;
  {
    const char* senv = diag_only("MMK_SRNN_STAMPS");
    a.stamps = (senv && senv[0] == '1') ? reinterpret_cast<unsigned long long*>(p->tau + 8) : nullptr;
  }
  return a;
}

// enqueue step t = *tau + tau_off, whose residue modulo frame_sizes[0] is `phase`
// bottom_steps: 0 = no bottom tier here (warm-up, or covered by an earlier fused launch), 1 = this step,
// > 1 = this and the following steps in one fused launch (no tier above fires inside the range)
static int emit_step(mmk_srnn_plan* p, const SrnnCall& call, int64_t tau_off, int phase, int bottom_steps, hipStream_t st) {
  const mmk_srnn_config& c = p->cfg;
  const int H = p->H, G = p->G, M = call.M;
  for (int i = 0; i < p->n_rnn_tiers; ++i) {
    SrnnTier& t = p->tiers[i];
    if (phase % t.fs != 0) continue;  // `if t % fs[i] == 0`, sample_rnn_v2.py:246
    if (p->fused_gru) {
      SrnnGruArgs g = {};
      g.B = M; g.H = H; g.fs = t.fs; g.div = t.fs; g.class_size = (float)c.q_levels;
      g.tau_ptr = p->tau; g.tau_off = tau_off;
      g.idx = call.idx; g.idx_rs = call.idx_rs; g.shift = call.shift;
      g.win_wp = t.in_lin.Wp; g.win_bias = t.in_lin.bias;
      {
        const char* cenv = p->tune.get("MMK_SRNN_COMPOSED");
        g.v_comp = (t.fs <= 16 && !(cenv && cenv[0] == '0')) ? t.v_comp : nullptr;
      }
      if (i > 0) {   // outputs[i-1][:, (t // fs[i]) % (fs[i-1] // fs[i])]      (:251)
        g.upper = p->tiers[i - 1].out;
        g.up_mod = p->tiers[i - 1].up;
      }
      if (c.rnn_kind == 1) {
        g.wih_wp = t.gates.Wp; g.wih_bias = t.gates.bias; g.whh_wp = t.gates_hh.Wp; g.whh_bias = t.gates_hh.bias;
        g.w_tile_chunks = t.gates.k_chunks;
      } else {   // LSTM: one packed matrix, K = [x | h]; the summed bias
        g.lstm = 1;
        g.wih_wp = t.gates.Wp; g.whh_wp = t.gates.Wp + (int64_t)t.gates.seg_chunk0[1] * 256; g.wih_bias = t.gates.bias;
        g.w_tile_chunks = t.gates.k_chunks; g.c = t.c;
      }
      g.h_ring = t.h; g.h_slot_stride = (int64_t)p->Bmax * H;
      g.cnt = t.cnt; g.done = t.done; g.h_gran = t.h_gran;
      {
        const char* senv = diag_only("MMK_SRNN_STAMPS");
        g.stamps = (senv && senv[0] == '1') ? reinterpret_cast<unsigned long long*>(p->tau + 16) : nullptr;
      }
      // the up-sampler rides in the same launch behind a grid-wide barrier when the whole grid is resident at once
      // (MMK_SRNN_FUSED_UP=0: its own launch)
      const char* fuenv = p->tune.get("MMK_SRNN_FUSED_UP");
      const bool fused_up = !(fuenv && fuenv[0] == '0') && srnn_gru_grid_resident(H, M);
      if (fused_up) {
        g.ups_wp = t.up_lin.Wp; g.ups_bias = t.up_lin.bias; g.ups_n_tiles = t.up_lin.n_tiles; g.ups_n = t.up_lin.N;
        g.ups_out = t.out; g.ups_out_ld = (int64_t)t.up * H;
        g.err = reinterpret_cast<int*>(p->tau + 4);      // sticky word, read by the next mmk_srnn_reset
      }
      MMK_TRY(launch_srnn_gru(g, st));
      if (fused_up) continue;
      // up-sampler on the slot the kernel has just published: its position counter is the tier's update counter
      LinearArgs a = {};
      t.up_lin.fill(a);
      a.seg[0].x = addr_time(t.h, (int64_t)p->Bmax * H, 0, 1, 2); a.seg[0].ld = H;
      a.M = M; a.tau_ptr = t.cnt; a.tau_off = 0;
      a.epilogue = EPI_STORE; a.act = ACT_NONE;
      a.out = addr_static(t.out); a.out_ld = (int64_t)t.up * H;
      MMK_TRY(launch_linear(a, st));
      continue;
    }
    {
      LinearArgs a = {};
      t.in_lin.fill(a);
      for (int m = 0; m < p->n_in; ++m) {       // ZipReduceVariables: K = [frame of input 0 | frame of input 1 | ...], weights folded in at commit
        a.seg[m].x = addr_time(call.xidx[m], 1, (int32_t)(call.shift - t.fs), 1, 0);
        a.seg[m].ld = call.xidx_rs[m];
        a.seg[m].kind = SEG_I64_LINEARIZED;
        a.seg[m].class_size = (float)p->in_class[m];
      }
      a.M = M; a.tau_ptr = p->tau; a.tau_off = tau_off;
      a.epilogue = EPI_STORE; a.act = ACT_NONE;
      a.out = addr_static(p->xbuf); a.out_ld = H;
      if (i > 0) {
        // outputs[i-1][:, (t // fs[i]) % (fs[i-1] // fs[i])]      (:251)
        SrnnTier& up = p->tiers[i - 1];
        a.has_add = 1;
        a.add = addr_time(up.out, H, 0, t.fs, up.fs / t.fs);
        a.add_ld = (int64_t)up.up * H;
      }
      MMK_TRY(launch_linear(a, st));
    }
    if (c.rnn_kind == 1) {
      LinearArgs a = {};
      t.gates.fill(a);
      a.seg[0].x = addr_static(p->xbuf); a.seg[0].ld = H;
      a.M = M; a.tau_ptr = p->tau; a.tau_off = tau_off;
      a.epilogue = EPI_STORE; a.act = ACT_NONE;
      a.out = addr_static(p->gi); a.out_ld = 3 * H;
      MMK_TRY(launch_linear(a, st));
      LinearArgs hh = {};
      t.gates_hh.fill(hh);
      hh.seg[0].x = addr_static(t.h); hh.seg[0].ld = H;
      hh.M = M; hh.tau_ptr = p->tau; hh.tau_off = tau_off;
      hh.epilogue = EPI_STORE; hh.act = ACT_NONE;
      hh.out = addr_static(p->gh); hh.out_ld = 3 * H;
      MMK_TRY(launch_linear(hh, st));
      MMK_TRY(launch_gru_cell(p->gi, p->gh, t.h, M, H, st));
    } else {
      LinearArgs a = {};
      t.gates.fill(a);
      a.seg[0].x = addr_static(p->xbuf); a.seg[0].ld = H;
      a.seg[1].x = addr_static(t.h); a.seg[1].ld = H;
      a.M = M; a.tau_ptr = p->tau; a.tau_off = tau_off;
      a.epilogue = EPI_STORE; a.act = ACT_NONE;
      a.out = addr_static(p->gi); a.out_ld = (int64_t)G * H;
      MMK_TRY(launch_linear(a, st));
      if (c.rnn_kind == 0)
        MMK_TRY(launch_lstm_cell(p->gi, 4 * H, nullptr, 0, t.h, H, t.c, H, nullptr, 0, M, H, st));
      else
        MMK_TRY(launch_rnn_tanh_cell(p->gi, t.h, M, H, st));
    }
    const float* top_h = t.h;
    for (auto& d : t.deep) {   // nn.LSTM / GRU / RNN with num_layers > 1: layer k's input is layer k-1's new state
      if (c.rnn_kind == 1) {
        LinearArgs a = {};
        d.gates.fill(a);
        a.seg[0].x = addr_static(top_h); a.seg[0].ld = H;
        a.M = M; a.tau_ptr = p->tau; a.tau_off = tau_off;
        a.epilogue = EPI_STORE; a.act = ACT_NONE;
        a.out = addr_static(p->gi); a.out_ld = 3 * H;
        MMK_TRY(launch_linear(a, st));
        LinearArgs hh = {};
        d.gates_hh.fill(hh);
        hh.seg[0].x = addr_static(d.h); hh.seg[0].ld = H;
        hh.M = M; hh.tau_ptr = p->tau; hh.tau_off = tau_off;
        hh.epilogue = EPI_STORE; hh.act = ACT_NONE;
        hh.out = addr_static(p->gh); hh.out_ld = 3 * H;
        MMK_TRY(launch_linear(hh, st));
        MMK_TRY(launch_gru_cell(p->gi, p->gh, d.h, M, H, st));
      } else {
        LinearArgs a = {};
        d.gates.fill(a);
        a.seg[0].x = addr_static(top_h); a.seg[0].ld = H;
        a.seg[1].x = addr_static(d.h); a.seg[1].ld = H;
        a.M = M; a.tau_ptr = p->tau; a.tau_off = tau_off;
        a.epilogue = EPI_STORE; a.act = ACT_NONE;
        a.out = addr_static(p->gi); a.out_ld = (int64_t)G * H;
        MMK_TRY(launch_linear(a, st));
        if (c.rnn_kind == 0)
          MMK_TRY(launch_lstm_cell(p->gi, 4 * H, nullptr, 0, d.h, H, d.c, H, nullptr, 0, M, H, st));
        else
          MMK_TRY(launch_rnn_tanh_cell(p->gi, d.h, M, H, st));
      }
      top_h = d.h;
    }
    {
      LinearArgs a = {};
      t.up_lin.fill(a);
      a.seg[0].x = addr_static(top_h); a.seg[0].ld = H;
      a.M = M; a.tau_ptr = p->tau; a.tau_off = tau_off;
      a.epilogue = EPI_STORE; a.act = ACT_NONE;
      a.out = addr_static(t.out); a.out_ld = (int64_t)t.up * H;
      MMK_TRY(launch_linear(a, st));
    }
  }
  if (bottom_steps <= 0) return MMK_OK;
  if (p->fused_bottom) {
    SrnnBottomArgs a = bottom_args(p, call, tau_off, bottom_steps);
    return launch_srnn_bottom(a, st);
  }
  {
    const int fsl = c.frame_size[c.n_tiers - 1];
    SrnnTier& up = p->tiers[p->n_rnn_tiers - 1];
    LinearArgs a = {};
    p->bottom.fill(a);
    for (int m = 0; m < p->n_in; ++m) {
      a.seg[m].x = addr_time(call.xidx[m], 1, (int32_t)(call.shift - fsl), 1, 0);
      a.seg[m].ld = call.xidx_rs[m];
      a.seg[m].kind = SEG_I64_LINEARIZED;
      a.seg[m].class_size = (float)p->in_class[m];
    }
    a.M = M; a.tau_ptr = p->tau; a.tau_off = tau_off;
    a.epilogue = EPI_STORE; a.act = ACT_NONE;
    a.out = addr_static(p->xbot); a.out_ld = H;
    a.has_add = 1;  // outputs[-1][:, (t % fs[-2]) - fs[-2]]   (:257)
    a.add = addr_time(up.out, H, 0, 1, up.fs);
    a.add_ld = (int64_t)up.up * H;
    MMK_TRY(launch_linear(a, st));
  }
  // one output module per target on the same hidden vector (:259); output k goes into input k (loops/generate.py:213-218)
  for (int k = 0; k < p->n_tgt; ++k) {
    const std::vector<PackedLinear>& mlp = k == 0 ? p->mlp : p->xheads[k - 1].mlp;
    float* logits = k == 0 ? p->logits : p->xheads[k - 1].logits;
    const int logits_ld = k == 0 ? p->logits_ld : p->xheads[k - 1].logits_ld;
    const int hidden = k == 0 ? c.mlp_hidden : p->xheads[k - 1].hidden;
    const float* x = p->xbot;
    int x_ld = H;
    for (size_t i = 0; i < mlp.size(); ++i) {
      const bool last = (i + 1 == mlp.size());
      LinearArgs a = {};
      mlp[i].fill(a);
      a.seg[0].x = addr_static(x); a.seg[0].ld = x_ld;
      a.M = M; a.tau_ptr = p->tau; a.tau_off = tau_off;
      a.epilogue = EPI_STORE; a.act = last ? (int)ACT_NONE : c.mlp_act;      // MLPIO.activation (modules/io.py:205)
      float* o = last ? logits : p->hid[i & 1];
      a.out = addr_static(o);
      a.out_ld = last ? logits_ld : hidden;
      MMK_TRY(launch_linear(a, st));
      x = o;
      x_ld = (int)a.out_ld;
    }
    SampleArgs s = {};
    s.logits = logits; s.ld = logits_ld; s.rows = M;
    s.n_classes = k == 0 ? c.q_levels : p->xheads[k - 1].q;
    s.has_temp_col = k == 0 ? c.learn_temp : p->xheads[k - 1].learn_temp;
    s.min_temp = k == 0 ? c.min_temp : p->xheads[k - 1].min_temp;
    s.temperature = call.temperature;
    s.uniforms = call.uniforms ? call.uniforms + (int64_t)k * M * call.uni_ld : nullptr;       // (n_targets, batch, n_steps)
    s.uniform_ld = call.uni_ld; s.uni_off = call.uni_off;
    s.out = const_cast<int64_t*>(call.xidx[k]); s.out_row_stride = call.xidx_rs[k]; s.out_tau_off = 0;
    s.tau_ptr = p->tau; s.tau_off = tau_off;
    MMK_TRY(launch_sample(s, st));
  }
  return MMK_OK;
}

// steps [first, first + count) relative to *tau; `phase` = residue of the first step modulo frame_sizes[0].
// With the fused bottom kernel the steps up to the next update of the tier above go into one launch.
static int emit_range(mmk_srnn_plan* p, const SrnnCall& call, int64_t first, int64_t count, int phase, bool with_bottom,
                      hipStream_t st) {
  const int period = p->cfg.frame_size[0];
  const int slots = p->tiers[p->n_rnn_tiers - 1].up;          // frame_sizes[-2]
  int64_t covered = 0;                                        // bottom steps already inside a fused launch
  for (int64_t s = 0; s < count; ++s) {
    const int ph = (int)((phase + s) % period);
    int bottom = 0;
    if (with_bottom) {
      if (!p->fused_bottom) {
        bottom = 1;
      } else if (covered == 0) {
        const int64_t until_update = slots - (ph % slots);
        bottom = (int)(until_update < count - s ? until_update : count - s);
        covered = bottom;
      }
    }
    MMK_TRY(emit_step(p, call, first + s, ph, bottom, st));
    if (covered > 0) --covered;
  }
  return MMK_OK;
}

// The graph of one period (frame_sizes[0] steps from residue phase0) for this call, captured when the cached one is another;
// `st`: the stream replays of the old graph may still be queued on
static int prepare_period_graph(mmk_srnn_plan* p, const SrnnCall& call, int64_t t_begin, int64_t n, bool with_bottom, hipStream_t st) {
  const int period = p->cfg.frame_size[0];
  if (n < 2 * period) return MMK_OK;
  const int phase0 = (int)(t_begin % period);
  // several periods per graph once the block is long: a replay boundary costs more than a kernel boundary inside a graph
  const int periods = n >= 16 * (int64_t)period ? 4 : 1;
  std::vector<int64_t> key = {call.M, (int64_t)(uintptr_t)call.idx, call.idx_rs, call.shift, with_bottom ? 1 : 0,
                              (int64_t)(uintptr_t)call.temperature, (int64_t)(uintptr_t)call.uniforms, call.uni_ld,
                              call.uni_off, phase0, periods};
  for (int m = 1; m < p->n_in; ++m) {
    key.push_back((int64_t)(uintptr_t)call.xidx[m]);
    key.push_back(call.xidx_rs[m]);
  }
  if (p->gc.exec && p->gc.key == key) return MMK_OK;
  MMK_HIP(hipStreamSynchronize(st));
  p->gc.reset();
  MMK_HIP(hipStreamBeginCapture(p->cap_stream, hipStreamCaptureModeThreadLocal));
  int rc = emit_range(p, call, 0, (int64_t)period * periods, phase0, with_bottom, p->cap_stream);
  if (rc == MMK_OK) rc = launch_bump(p->tau, (int64_t)period * periods, p->cap_stream);
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
  p->gc.steps = period * periods;
  return MMK_OK;
}

// n steps from the step *tau points at: whole periods as replays of the prepared graph, the rest eagerly
static int enqueue_steps(mmk_srnn_plan* p, const SrnnCall& call, int64_t t_begin, int64_t n, bool with_bottom, hipStream_t st) {
  const int period = p->cfg.frame_size[0];
  int64_t done = 0;
  if (n >= 2 * period) {
    const int64_t reps = n / p->gc.steps;
    for (int64_t r = 0; r < reps; ++r) MMK_HIP(hipGraphLaunch(p->gc.exec, st));
    done = reps * p->gc.steps;
  }
  if (n > done) {
    MMK_TRY(emit_range(p, call, 0, n - done, (int)((t_begin + done) % period), with_bottom, st));
    MMK_TRY(launch_bump(p->tau, n - done, st));
  }
  return MMK_OK;
}

// Resident mode (srnn_resident.hip: ONE launch per block, every tier resident) applies when the fused kernels' geometry does, the composed
// operands were built at commit, every workgroup of the launch has a CU of its own, and the block - from the next multiple of frame_sizes[0] on,
// where every tier updates - is at least one period long (MMK_SRNN_RESIDENT=0, or exec_mode 1 - the caller's redo path -: never)
static int resident_grid(mmk_srnn_plan* p, const SrnnCall& call, int64_t n_res, int* mt) {
  const mmk_srnn_config& c = p->cfg;
  const char* renv = p->tune.get("MMK_SRNN_RESIDENT");
  if ((renv && renv[0] == '0') || c.exec_mode == 1) return 0;
  if (!p->fused_bottom || !p->fused_gru || !p->resident_ready || n_res < c.frame_size[0]) return 0;
  const int S = p->tiers.back().up;
  if (S < 2 || !srnn_resident_supported(p->H, c.rnn_kind == 0, c.mlp_hidden, c.q_levels + (c.learn_temp ? 1 : 0), c.q_levels, c.frame_size[c.n_tiers - 1], S)) return 0;
  for (auto& t : p->tiers)
    if (t.fs > 128) return 0;                                 // the class ring holds 256 positions
  // (MMK_SRNN_SPARE_CUS: CUs the launch leaves alone; every workgroup of it must be resident at once, and a wait that times out - the CUs were
  //  held by something else for a second - is reported by mmk_srnn_sync_status and redone in turns by the caller)
  const char* senv = p->tune.get("MMK_SRNN_SPARE_CUS");
  return srnn_resident_grid(p->H, call.M, p->n_rnn_tiers, senv ? atoi(senv) : 0, mt);
}

static int run_resident(mmk_srnn_plan* p, const SrnnCall& call, int64_t t_begin, int64_t n, const int* mt, hipStream_t st, bool teacher = false) {
  const mmk_srnn_config& c = p->cfg;
  const int H = p->H, G = p->G, KC = H / 16, M = call.M;
  MMK_HIP(hipMemsetAsync(p->res_gran, 0, (size_t)p->res_gran_count * sizeof(unsigned long long), st));   // (a granule of an earlier block could carry the number this one waits for)
  if (!teacher) {
    hipLaunchKernelGGL(srnn_resident_init_kernel, dim3((unsigned)(((int64_t)M * 256 + 255) / 256)), dim3(256), 0, st, p->cls_gran, call.idx, call.idx_rs, M, t_begin);
    MMK_HIP(hipGetLastError());
  }
  SrnnResArgs a = {};
  a.teacher = teacher ? 1 : 0;
  a.shift = call.shift;
  a.B = M; a.H = H; a.n_tiers = p->n_rnn_tiers; a.lstm = c.rnn_kind == 0 ? 1 : 0;
  a.n_steps = (int32_t)n; a.t_begin = t_begin; a.class_size = (float)c.q_levels;
  int block0 = M;
  for (int i = 0; i < p->n_rnn_tiers; ++i) {
    SrnnTier& t = p->tiers[i];
    SrnnResTier& r = a.tier[i];
    const bool last = i == p->n_rnn_tiers - 1;
    r.fs = t.fs; r.up = t.up; r.up_mod = i > 0 ? p->tiers[i - 1].up : 0;
    r.n_tiles = last ? p->cp_tiles : G * (t.up - 1); r.rpb = last ? p->cp_rpb : 0;
    r.block0 = block0; r.mt = mt[i];
    block0 += KC * ((M + 16 * mt[i] - 1) / (16 * mt[i]));
    r.fsp = (int32_t)round_up(t.fs, 4);
    if (c.rnn_kind == 1) {
      r.whh_wp = t.gates_hh.Wp; r.w_tile_chunks = t.gates.k_chunks;
    } else {   // LSTM: one packed matrix, K = [x | h]
      r.whh_wp = t.gates.Wp + (int64_t)t.gates.seg_chunk0[1] * 256; r.w_tile_chunks = t.gates.k_chunks;
    }
    r.link_wp = i > 0 ? t.link_wp : nullptr;
    r.gconst = t.gconst; r.v_full = t.v_full;
    r.out_wp = last ? p->cp_wp : p->tiers[i + 1].link_wp; r.out_bias = last ? nullptr : p->tiers[i + 1].link_bias;
    r.h_ring = t.h; r.h_slot_stride = (int64_t)p->Bmax * H; r.c = t.c; r.cnt = t.cnt;
    r.h_gran = t.rh_gran; r.out_gran = t.rout_gran;
    r.upper_gran = i > 0 ? p->tiers[i - 1].rout_gran : nullptr;
    r.upper_h_gran = i > 0 ? p->tiers[i - 1].rh_gran : nullptr;
    r.prog = p->res_prog + i;
  }
  a.Hm = c.mlp_hidden; a.Q = c.q_levels; a.n_out = c.q_levels + (c.learn_temp ? 1 : 0); a.learn_temp = c.learn_temp; a.min_temp = c.min_temp;
  a.fsb = c.frame_size[c.n_tiers - 1]; a.S = p->tiers.back().up;
  a.idx = const_cast<int64_t*>(call.idx); a.idx_rs = call.idx_rs;
  a.cp0 = p->cp0; a.a_comp = p->a_comp; a.bcs = p->bcs;
  a.fc2_raw = p->mlp_raw[1]; a.fc2_bias = p->mlp[1].bias;
  a.temperature = call.temperature; a.uniforms = call.uniforms; a.uni_ld = call.uni_ld; a.uni_off = call.uni_off;
  a.logits_out = p->logits; a.logits_ld = p->logits_ld;
  a.cls_gran = p->cls_gran;
  a.err = reinterpret_cast<int*>(p->tau + 4);
  {
    const char* senv = diag_only("MMK_SRNN_STAMPS");
    a.stamps = (senv && senv[0] == '1') ? p->res_stamps : nullptr;
  }
  MMK_TRY(launch_srnn_resident(a, st));
  for (auto& t : p->tiers) {
    // no up-sampler ran inside the launch (its rows reach the tier below composed with that tier's gates, or with the head's first layer): once per
    // tier, on the state the launch left, for whoever continues between two updates
    LinearArgs u = {};
    t.up_lin.fill(u);
    u.seg[0].x = addr_time(t.h, (int64_t)p->Bmax * H, 0, 1, 2); u.seg[0].ld = H;
    u.M = M; u.tau_ptr = t.cnt; u.tau_off = 0;
    u.epilogue = EPI_STORE; u.act = ACT_NONE;
    u.out = addr_static(t.out); u.out_ld = (int64_t)t.up * H;
    MMK_TRY(launch_linear(u, st));
  }
  ++p->resident_blocks;
  return MMK_OK;
}

static int run_steps(mmk_srnn_plan* p, const SrnnCall& call, int64_t t_begin, int64_t n, bool with_bottom, hipStream_t st) {
  if (n <= 0) return MMK_OK;
  if (with_bottom) {
    const int period = p->cfg.frame_size[0];
    const int64_t head = (period - t_begin % period) % period;      // steps up to the next update of the top tier, run with the kernels in turns
    int mt[kResMaxTiers];
    if (n > head && resident_grid(p, call, n - head, mt) > 0) {
      if (head > 0) {
        MMK_TRY(launch_set_i64(p->tau, t_begin, st));
        MMK_TRY(emit_range(p, call, 0, head, (int)(t_begin % period), true, st));
      }
      MMK_TRY(run_resident(p, call, t_begin + head, n - head, mt, st));
      return launch_set_i64(p->tau, t_begin + n, st);
    }
  }
  if (!with_bottom && p->n_in == 1) {
    // The warm-up (sample_rnn_v2.py:229-234: generate_step over the prompt, outputs dropped) as ONE teacher-forced resident launch: the tiers with their
    // matrices in registers, windows from the prompt, no bottom tier - where the block starts on an update of the top tier and is whole periods long
    // (round 5: 155 launches of 17 us each for a 512-sample prompt at cfg 3)
    const int period = p->cfg.frame_size[0];
    const char* wenv = p->tune.get("MMK_SRNN_RESIDENT_WARMUP");
    int mt[kResMaxTiers];
    if (!(wenv && wenv[0] == '0') && t_begin % period == 0 && n % period == 0 && resident_grid(p, call, n, mt) > 0) {
      MMK_TRY(run_resident(p, call, t_begin, n, mt, st, true));
      --p->resident_blocks;      // (the counter says how many GENERATE blocks ran resident)
      ++p->resident_warmups;
      return launch_set_i64(p->tau, t_begin + n, st);
    }
  }
  MMK_TRY(prepare_period_graph(p, call, t_begin, n, with_bottom, st));
  MMK_TRY(launch_set_i64(p->tau, t_begin, st));
  return enqueue_steps(p, call, t_begin, n, with_bottom, st);
}

static int check_call(mmk_srnn_plan* p, int32_t batch, const int64_t* const* idx, const int64_t* idx_row_stride, SrnnCall& call) {
  if (!p) return fail(MMK_ERR_INVALID, "srnn: null plan");
  if (!p->committed) return fail(MMK_ERR_STATE, "srnn: plan not committed (bind weights, then mmk_srnn_commit)");
  if (batch < 1 || batch > p->Bmax) return fail(MMK_ERR_INVALID, "srnn: batch %d outside [1, %d]", batch, p->Bmax);
  if (!idx || !idx_row_stride) return fail(MMK_ERR_INVALID, "srnn: null input");
  call.M = batch;
  for (int m = 0; m < p->n_in; ++m) {
    if (!idx[m]) return fail(MMK_ERR_INVALID, "srnn: input %d is null (%d inputs expected)", m, p->n_in);
    call.xidx[m] = idx[m];
    call.xidx_rs[m] = idx_row_stride[m];
  }
  call.idx = call.xidx[0];
  call.idx_rs = call.xidx_rs[0];
  return MMK_OK;
}

extern "C" int mmk_srnn_warmup(mmk_srnn_plan* p, int32_t batch, const int64_t* idx, int64_t idx_row_stride,
                               int64_t prompt_len, mmk_stream_t stream) {
  if (p && p->n_in != 1) return fail(MMK_ERR_INVALID, "srnn_warmup: this network has %d inputs (mmk_srnn_warmup_multi)", p->n_in);
  return mmk_srnn_warmup_multi(p, batch, &idx, &idx_row_stride, prompt_len, stream);
}

extern "C" int mmk_srnn_warmup_multi(mmk_srnn_plan* p, int32_t batch, const int64_t* const* idx, const int64_t* idx_row_stride,
                                     int64_t prompt_len, mmk_stream_t stream) {
  SrnnCall call;
  MMK_TRY(check_call(p, batch, idx, idx_row_stride, call));
  const int64_t rf = p->cfg.frame_size[0];
  if (prompt_len < rf) return fail(MMK_ERR_INVALID, "srnn_warmup: prompt of %lld steps is shorter than rf=%lld", (long long)prompt_len, (long long)rf);
  const int64_t offset = prompt_len % rf;     // :230
  const int64_t stop = prompt_len - offset;   // self.prompt_length, :231
  call.shift = offset;
  // for t in range(rf, prompt_length): generate_step(window shifted by offset, t=t)   (:233-234)
  return run_steps(p, call, rf, stop - rf, false, (hipStream_t)stream);
}

extern "C" int mmk_srnn_generate(mmk_srnn_plan* p, int32_t batch, int64_t* idx, int64_t idx_row_stride, int64_t t0,
                                 int64_t n_steps, const float* temperature, const float* uniforms, mmk_stream_t stream) {
  if (p && p->n_in != 1) return fail(MMK_ERR_INVALID, "srnn_generate: this network has %d inputs (mmk_srnn_generate_multi)", p->n_in);
  return mmk_srnn_generate_multi(p, batch, &idx, &idx_row_stride, t0, n_steps, temperature, uniforms, stream);
}

extern "C" int mmk_srnn_generate_multi(mmk_srnn_plan* p, int32_t batch, int64_t* const* idx, const int64_t* idx_row_stride, int64_t t0,
                                       int64_t n_steps, const float* temperature, const float* uniforms, mmk_stream_t stream) {
  SrnnCall call;
  MMK_TRY(check_call(p, batch, idx, idx_row_stride, call));
  if (t0 < p->cfg.frame_size[0] || n_steps < 0) return fail(MMK_ERR_INVALID, "srnn_generate: t0 must be >= rf and n_steps >= 0");
  if (temperature && !uniforms) return fail(MMK_ERR_INVALID, "srnn_generate: temperature given without uniforms");
  call.shift = 0;
  call.temperature = temperature;
  call.uniforms = uniforms;
  call.uni_ld = n_steps;
  call.uni_off = -t0;
  return run_steps(p, call, t0, n_steps, true, (hipStream_t)stream);
}

extern "C" int mmk_srnn_inject_sync_error(mmk_srnn_plan* p, mmk_stream_t stream) {
  if (!p) return fail(MMK_ERR_INVALID, "srnn_inject_sync_error: null plan");
  if (!p->committed) return fail(MMK_ERR_STATE, "srnn_inject_sync_error: plan not committed");
  const int64_t code = 4;                 // the word a timed-out wait of the bottom kernel sets
  MMK_HIP(hipMemcpyAsync(p->tau + 4, &code, sizeof(code), hipMemcpyHostToDevice, (hipStream_t)stream));
  MMK_HIP(hipStreamSynchronize((hipStream_t)stream));
  return MMK_OK;
}

extern "C" int mmk_srnn_sync_status(mmk_srnn_plan* p, mmk_stream_t stream) {
  if (!p) return fail(MMK_ERR_INVALID, "srnn_sync_status: null plan");
  hipStream_t st = (hipStream_t)stream;
  MMK_HIP(hipStreamSynchronize(st));
  if (!p->committed) return MMK_OK;
  int64_t err = 0;
  MMK_HIP(hipMemcpy(&err, p->tau + 4, sizeof(err), hipMemcpyDeviceToHost));
  if (err != 0) {
    MMK_HIP(hipMemset(p->tau + 4, 0, sizeof(int64_t)));
    return fail(MMK_ERR_STATE, "srnn: a wait inside the tier / bottom kernels timed out (code %lld: 3 tier hand-over, 6 bottom role / 7 tier role of the resident kernel) - "
                "its workgroups were not all running side by side; the samples of this generation are invalid",
                (long long)err);
  }
  return MMK_OK;
}

extern "C" int64_t mmk_srnn_resident_blocks(const mmk_srnn_plan* p) { return p ? p->resident_blocks : 0; }
extern "C" int64_t mmk_srnn_resident_warmups(const mmk_srnn_plan* p) { return p ? p->resident_warmups : 0; }

extern "C" int mmk_srnn_last_logits(mmk_srnn_plan* p, int32_t batch, float* out, int64_t ld, mmk_stream_t stream) {
  if (!p || !out) return fail(MMK_ERR_INVALID, "srnn_last_logits: null argument");
  if (!p->committed) return fail(MMK_ERR_STATE, "srnn_last_logits: plan not committed");
  if (const char* senv = diag_only("MMK_SRNN_STAMPS"); senv && senv[0] == '1') {
    unsigned long long st[8];
    MMK_HIP(hipStreamSynchronize((hipStream_t)stream));
    MMK_HIP(hipMemcpy(st, p->tau + 8, sizeof(st), hipMemcpyDeviceToHost));
    const double n = st[7] ? (double)st[7] : 1.0;
    fprintf(stderr, "[mmk stamps] srnn bottom kernel, workgroup 0, us per launch over %llu launches: prologue=%.2f x=%.2f fc0=%.2f fc2=%.2f sampler=%.2f; shader clock %.0f MHz\n",
            st[7], st[0] * 1e-2 / n, st[1] * 1e-2 / n, st[2] * 1e-2 / n, st[3] * 1e-2 / n, st[4] * 1e-2 / n,
            st[6] ? 100.0 * (double)st[5] / (double)st[6] : 0.0);
    {
      unsigned long long rs[8 * (1 + kResMaxTiers)];
      MMK_HIP(hipMemcpy(rs, p->res_stamps, sizeof(rs), hipMemcpyDeviceToHost));
      if (rs[7]) {
        const double nb = (double)rs[7];
        fprintf(stderr, "[mmk stamps] srnn resident kernel, bottom role of clip 0, us per step over %llu steps: prologue(total)=%.2f wait for the state row=%.3f "
                "slot-0 product / wait for the composed row=%.3f hidden=%.3f fc2=%.3f draw=%.3f\n", rs[7], rs[0] * 1e-2, rs[1] * 1e-2 / nb, rs[2] * 1e-2 / nb,
                rs[3] * 1e-2 / nb, rs[4] * 1e-2 / nb, rs[5] * 1e-2 / nb);
        for (int i = 0; i < p->n_rnn_tiers; ++i) {
          const unsigned long long* r = rs + 8 * (1 + i);
          const double nu = r[7] ? (double)r[7] : 1.0;
          fprintf(stderr, "[mmk stamps] srnn resident kernel, tier %d workgroup 0, us per update over %llu updates: input half (state / rows of the tier above)=%.2f "
                  "window (wait for the classes)=%.2f cell=%.2f all-gather=%.2f output tiles=%.2f recurrent product=%.2f\n", i, r[7], r[0] * 1e-2 / nu, r[1] * 1e-2 / nu,
                  r[2] * 1e-2 / nu, r[3] * 1e-2 / nu, r[4] * 1e-2 / nu, r[5] * 1e-2 / nu);
        }
        MMK_HIP(hipMemset(p->res_stamps, 0, sizeof(rs)));
      }
    }
    MMK_HIP(hipMemcpy(st, p->tau + 16, sizeof(st), hipMemcpyDeviceToHost));
    const double ng = st[7] ? (double)st[7] : 1.0;
    fprintf(stderr, "[mmk stamps] srnn gru kernel, workgroup 0, us per launch over %llu launches: loads=%.2f wait for the bottom kernel=%.2f x=%.2f mfma=%.2f cell=%.2f grid barrier=%.2f up-sampler=%.2f\n",
            st[7], st[0] * 1e-2 / ng, st[4] * 1e-2 / ng, st[1] * 1e-2 / ng, st[2] * 1e-2 / ng, st[3] * 1e-2 / ng, st[5] * 1e-2 / ng, st[6] * 1e-2 / ng);
  }
  const int n = p->cfg.q_levels + (p->cfg.learn_temp ? 1 : 0);
  MMK_HIP(hipMemcpy2DAsync(out, ld * sizeof(float), p->logits, p->logits_ld * sizeof(float), n * sizeof(float), batch,
                           hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return MMK_OK;
}

extern "C" int mmk_srnn_last_logits_of(mmk_srnn_plan* p, int32_t target, int32_t batch, float* out, int64_t ld, mmk_stream_t stream) {
  if (!p || !out) return fail(MMK_ERR_INVALID, "srnn_last_logits_of: null argument");
  if (target == 0) return mmk_srnn_last_logits(p, batch, out, ld, stream);
  if (!p->committed) return fail(MMK_ERR_STATE, "srnn_last_logits_of: plan not committed");
  if (target < 0 || target >= p->n_tgt) return fail(MMK_ERR_INVALID, "srnn_last_logits_of: target %d of %d", target, p->n_tgt);
  const SrnnHead& h = p->xheads[target - 1];
  const int n = h.q + (h.learn_temp ? 1 : 0);
  MMK_HIP(hipMemcpy2DAsync(out, ld * sizeof(float), h.logits, h.logits_ld * sizeof(float), n * sizeof(float), batch,
                           hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return MMK_OK;
}

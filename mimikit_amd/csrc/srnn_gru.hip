// One SampleRNN GRU or LSTM tier update in ONE launch (gfx950).
//
// Reference: SampleRNNTier.forward (sample_rnn_v2.py:83-99) for a GRU tier with one layer:
//     x  = Linear(linearize(idx[t-fs:t])) (+ upper tier output slice)         input_module, K = fs (tiny)
//     gi = W_ih x + b_ih ; gh = W_hh h + b_hh                                   two (3H x H) products
//     r = s(gi_r + gh_r), z = s(gi_z + gh_z), n = tanh(gi_n + r gh_n), h' = (h - n) z + n      (ATen GRUCell)
// As separate launches this was four kernels (input linear, two gate GEMMs, cell) at 5-10 us each, each far
// below 1 us of work.  Here a workgroup owns 16 hidden units x 16 clips: it builds its 16 rows of x in LDS,
// streams the six 16-row weight tiles of its units (r, z, n of W_ih and W_hh; K split over 8 waves,
// v_mfma_f32_16x16x4_f32), reduces in LDS and runs the cell for its 256 (clip, unit) pairs - no exchange with
// other workgroups.  The hidden state is double buffered ([2][B][H], slot = update counter & 1): every
// workgroup reads the old slot, writes the new one, and the last workgroup to finish bumps the counter, which
// the up-sampling launch that follows uses as its position counter.
// LSTM tiers (the reference's default rnn_class) run through the same kernel: eight tiles (i, f, g, o of W_ih and W_hh,
// which the plan packs side by side along K), the ATen LSTMCell, and a cell state updated in place.
//
// Round 2.  (1) Second phase in the same launch: the tier's up-sampler, out = W_up h' + b.  The new state goes out as data-tagged
// granules {update number, value}; a workgroup polls the state of ITS OWN 16 clips (32 KB at H = 512; a light sentinel poll by one
// wave first) and multiplies its `up` column tiles, slot 0 - the row the tier below needs first - ahead of the others.
// (2) Composed mode (frame sizes <= 16): W_ih x = W_ih (b_in + upper) + (W_ih W_in) lin(window); the first product joins the
// work ahead of the poll, the second is a K = fs dot product per (clip, gate unit) in the cell.
#include <type_traits>

#include "mmk_common.h"
#include "srnn_gru.h"

namespace mmk {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) f32x4* gf32x4_ptr;

constexpr int kGruThreads = 512;
constexpr int kGruWaves = kGruThreads / 64;

template <int KC, bool LSTM>   // KC = H / 16 K-chunks; each of the 8 waves takes KC / 8 of them
__global__ __launch_bounds__(kGruThreads) void srnn_gru_kernel(const SrnnGruArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr int H = KC * 16;
  constexpr int CPW = KC / kGruWaves;              // chunks per wave
  constexpr int ldx = H + 4;
  constexpr int NG = LSTM ? 4 : 3;                 // gates; 2 NG weight tiles per workgroup
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ub = blockIdx.x % KC;                  // block of 16 hidden units
  const int m_first = (blockIdx.x / KC) * 16;      // first clip of this workgroup's row tile
  const int mg = min(16, a.B - m_first);
  const int64_t t = *a.tau_ptr + a.tau_off;
  const int64_t cnt = *a.cnt;

  // diagnostic (MMK_SRNN_STAMPS=1): 100 MHz wall-clock totals per phase of thread 0 of workgroup 0
  const bool stamping = a.stamps != nullptr && blockIdx.x == 0 && tid == 0;
  unsigned long long st_prev = stamping ? wall_clock64() : 0, st_acc[7] = {0, 0, 0, 0, 0, 0, 0};
  auto stamp = [&](int slot) {
    if (stamping) {
      const unsigned long long now = wall_clock64();
      st_acc[slot] += now - st_prev;
      st_prev = now;
    }
  };

  char* sp = smem_raw;
  float* xs = (float*)sp;    sp += 16 * ldx * 4;                  // x rows of the 16 clips
  float* hs = (float*)sp;    sp += 16 * ldx * 4;                  // h rows (old state)
  f32x4* red = (f32x4*)sp;   sp += 2 * NG * kGruWaves * 64 * 16;  // split-K partials: [tile][wave][lane]
  float* s_lin = (float*)sp;  sp += 16 * ((((a.fs + 15) / 16) * 16) + 4) * 4;   // linearized window [16][fs]
  float* vs = (float*)sp;                                         // composed mode: (W_ih W_in) rows of this workgroup's units [NG][16][16]

  // ---- weights first: 2 NG tiles x CPW chunks of this wave (nothing depends on them for a while) -----------------
  f32x4 w[2 * NG][CPW];
  {
    const int c0 = wave * CPW;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      gf32x4_ptr wi = (gf32x4_ptr)(uintptr_t)a.wih_wp + ((int64_t)(g * KC + ub) * a.w_tile_chunks + c0) * 64 + lane;
      gf32x4_ptr wh = (gf32x4_ptr)(uintptr_t)a.whh_wp + ((int64_t)(g * KC + ub) * a.w_tile_chunks + c0) * 64 + lane;
#pragma unroll
      for (int u = 0; u < CPW; ++u) {
        w[g][u] = wi[u * 64];
        w[NG + g][u] = wh[u * 64];
      }
    }
  }
  // ... and the first up-sampler tiles of this workgroup (second phase): nothing of them depends on this update.  The workgroup
  // up-samples ITS OWN clips: columns [j H + 16 ub, + 16) of the tier's (up H) outputs for every slot j < up (tile j KC + ub):
  // slot 0 - the row the tier below needs first - is every workgroup's first tile and goes out on its own
  constexpr int UB = (LSTM && KC == 32) ? 2 : 4;   // tiles per batch (registers, partial-sum buffer)
  const int up_tiles = a.ups_wp != nullptr ? a.ups_n_tiles / KC : 0;
  f32x4 wu0[UB][CPW];
  if (a.ups_wp != nullptr) {
#pragma unroll
    for (int j = 0; j < UB; ++j) {
      const int tile = min(j, up_tiles - 1) * KC + ub;
      gf32x4_ptr wsrc = (gf32x4_ptr)(uintptr_t)a.ups_wp + ((int64_t)tile * KC + wave * CPW) * 64 + lane;
#pragma unroll
      for (int u = 0; u < CPW; ++u) wu0[j][u] = wsrc[u * 64];
    }
  }
  // ... and the biases of the cell (thread = (clip, unit) pair) and of the first up-sampler batch (two outputs per thread)
  float cell_b[2 * NG], ups_b[2] = {0.f, 0.f};
  {
    const int unit = ub * 16 + (tid & 15);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      cell_b[g] = a.wih_bias ? a.wih_bias[g * H + unit] : 0.f;
      cell_b[NG + g] = (!LSTM && a.whh_bias) ? a.whh_bias[g * H + unit] : 0.f;
    }
    if (a.ups_wp != nullptr && a.ups_bias) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int j = min(k == 0 ? 0 : 1 + (tid >> 8), up_tiles - 1);      // batch {0}: tile 0; batch {1, 2, 3}: first round
        ups_b[k] = a.ups_bias[(j * KC + ub) * 16 + (tid & 15)];
      }
    }
  }
  const int kci = (a.fs + 15) / 16, ldl = kci * 16 + 4;
  const float* h_old = a.h_ring + (cnt & 1) * a.h_slot_stride;
  float* h_new = a.h_ring + ((cnt + 1) & 1) * a.h_slot_stride;
  // ---- the input projection's operands of this wave's column tiles (frame sizes <= 16: one K-chunk): requested now,
  //      with the gate weights, instead of one round trip per tile inside the x phase ------------------------------------
  constexpr int XT = KC / kGruWaves;                 // column tiles of x per wave
  const int xslot = a.up_mod > 0 ? (int)((t / a.div) % a.up_mod) : 0;     // outputs[i-1][:, (t // fs) % ...]   (:251)
  float x_up[XT][4], x_bias[XT];
  f32x4 x_w[XT];
  const bool x_hoisted = kci == 1;
  // Composed mode (frame sizes <= 16): W_ih x = W_ih (b_in + upper) + (W_ih W_in) lin(window).  The first product does not depend
  // on the newest classes and is multiplied ahead of the hand-over with the recurrent half; the second is a K = fs dot product
  // per (clip, gate unit) against the pre-multiplied matrix (srnn_plan.hip: compose, fp64 accumulation, one rounding), done in the
  // cell.  Same operands as the reference in another association.
  const bool composed = a.v_comp != nullptr && x_hoisted;
  if (x_hoisted) {
    const int q = lane >> 4, n = lane & 15;
#pragma unroll
    for (int j = 0; j < XT; ++j) {
      const int tile = wave + j * kGruWaves, col = tile * 16 + n;
#pragma unroll
      for (int r = 0; r < 4; ++r) {   // unconditional loads from clamped addresses
        const int m = min(4 * q + r, mg - 1);
        const float* src = a.upper ? a.upper + ((int64_t)(m_first + m) * a.up_mod + xslot) * H + col : a.win_bias + col;
        x_up[j][r] = *src;
      }
      x_bias[j] = a.win_bias[col];
      if (!composed) x_w[j] = ((gf32x4_ptr)(uintptr_t)a.win_wp)[(int64_t)tile * 64 + lane];
    }
  }
  // ---- old state rows -> LDS -------------------------------------------------------------------------------------
  for (int q = tid; q < 16 * (H / 4); q += kGruThreads) {
    const int m = q / (H / 4), c = (q - m * (H / 4)) * 4;
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (m < mg) v = *reinterpret_cast<const f32x4*>(h_old + (int64_t)(m_first + m) * H + c);
    *reinterpret_cast<f32x4*>(hs + m * ldx + c) = v;
  }
  if (composed) {
    const int q = lane >> 4, n = lane & 15;
#pragma unroll
    for (int j = 0; j < XT; ++j) {
      const int col = (wave + j * kGruWaves) * 16 + n;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = 4 * q + r;
        xs[m * ldx + col] = m < mg ? (a.upper ? x_bias[j] + x_up[j][r] : x_bias[j]) : 0.f;     // b_in + upper
      }
    }
    if (tid < NG * 16 * 4) {                                   // this workgroup's rows of W_ih W_in: [gate][unit][16], zero padded
      const int g = tid / 64, n = (tid >> 2) & 15, c4 = (tid & 3) * 4;
      *reinterpret_cast<f32x4*>(vs + (g * 16 + n) * 16 + c4) = *reinterpret_cast<const f32x4*>(a.v_comp + ((int64_t)g * H + ub * 16 + n) * 16 + c4);
    }
  }
  __syncthreads();
  // ---- the recurrent half of the products, W_hh h: it does not depend on the newest classes either ------------------
  f32x4 acc[2 * NG];
#pragma unroll
  for (int g = 0; g < 2 * NG; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
  {
    const float* hr = hs + (lane & 15) * ldx + wave * CPW * 16 + 4 * (lane >> 4);
    f32x4 hv[CPW];
#pragma unroll
    for (int u = 0; u < CPW; ++u) hv[u] = *reinterpret_cast<const f32x4*>(hr + u * 16);
#pragma unroll
    for (int u = 0; u < CPW; ++u) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int g = 0; g < NG; ++g) acc[NG + g] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[u][i], w[NG + g][u][i], acc[NG + g], 0, 0, 0);
      }
    }
  }
  // ---- the input half, W_ih x (composed mode: W_ih (b_in + upper)): NG 16 x 16 tiles, this wave's K range, then all partial sums
  auto input_half = [&]() {
    const int c0 = wave * CPW;
    const float* xr = xs + (lane & 15) * ldx + c0 * 16 + 4 * (lane >> 4);
    f32x4 xv[CPW];
#pragma unroll
    for (int u = 0; u < CPW; ++u) xv[u] = *reinterpret_cast<const f32x4*>(xr + u * 16);
#pragma unroll
    for (int u = 0; u < CPW; ++u) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[u][i], w[g][u][i], acc[g], 0, 0, 0);
      }
    }
#pragma unroll
    for (int g = 0; g < 2 * NG; ++g) red[(g * kGruWaves + wave) * 64 + lane] = acc[g];
  };
  if (composed) input_half();
  stamp(0);   // weights requested, old state in LDS, recurrent half multiplied
  // ---- the window, linearized (modules/io.py:106-112), zero padded to whole K-chunks ----------------------------
  for (int e = tid; e < 16 * kci * 16; e += kGruThreads) {
    const int m = e / (kci * 16), i = e - m * (kci * 16);
    float v = 0.f;
    if (m < mg && i < a.fs) {
      const int64_t pos = t + a.shift - a.fs + i;
      const int64_t cls = a.idx[(int64_t)(m_first + m) * a.idx_rs + pos];
      v = (((float)cls / a.class_size) - .5f) * 2.f;
    }
    s_lin[m * ldl + i] = v;
  }
  __syncthreads();
  stamp(4);   // window
  // ---- x = W_in lin + b_in (+ upper): 16 x 16 tiles over the waves, K = fs (same MFMA order as the launch path) ----
  if (composed) {
    // nothing: x never exists in this mode
  } else if (x_hoisted) {
    const int q = lane >> 4, n = lane & 15;
    const f32x4 xv = *reinterpret_cast<const f32x4*>(s_lin + n * ldl + 4 * q);
#pragma unroll
    for (int j = 0; j < XT; ++j) {
      const int col = (wave + j * kGruWaves) * 16 + n;
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[i], x_w[j][i], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = 4 * q + r;
        float v = acc[r] + x_bias[j];
        if (a.upper) v += x_up[j][r];
        xs[m * ldx + col] = m < mg ? v : 0.f;
      }
    }
  } else {
    const int slot = xslot;
    const int q = lane >> 4, n = lane & 15;
    for (int tile = wave; tile < KC; tile += kGruWaves) {
      const int col = tile * 16 + n;
      float upv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {   // unconditional loads from clamped addresses
        const int m = min(4 * q + r, mg - 1);
        const float* src = a.upper ? a.upper + ((int64_t)(m_first + m) * a.up_mod + slot) * H + col : a.win_bias + col;
        upv[r] = *src;
      }
      const float bias = a.win_bias[col];
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int c = 0; c < kci; ++c) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(s_lin + n * ldl + c * 16 + 4 * q);
        const f32x4 wv = ((gf32x4_ptr)(uintptr_t)a.win_wp)[((int64_t)tile * kci + c) * 64 + lane];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[i], wv[i], acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = 4 * q + r;
        float v = acc[r] + bias;
        if (a.upper) v += upv[r];
        xs[m * ldx + col] = m < mg ? v : 0.f;
      }
    }
  }
  if (!composed) __syncthreads();
  stamp(1);   // x
  if (!composed) {
    input_half();
    __syncthreads();
  }
  stamp(2);   // MFMAs (incl. the wait for the weights)
  // the new state: plain stores for the next update; when the up-sampler phase of this launch reads it from other XCDs it
  // also goes out as data-tagged granules {update number, value} (agent-scope stores, polled by the readers: one hop, where
  // a grid barrier behind acknowledged write-through stores took 2.6 us + the read, and an L2 write-back fence ~10 us)
  const bool fused_up = a.ups_wp != nullptr;
  auto store_state = [&](int64_t o, float v) {
    h_new[o] = v;
    if (fused_up)
      __hip_atomic_store(a.h_gran + o, ((unsigned long long)(unsigned)(cnt + 1) << 32) | __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  // ---- cell: one (clip, unit) pair per thread -----------------------------------------------------------------
  if (tid < 256) {
    const int m = tid >> 4, n = tid & 15;
    const int frag = ((m >> 2) * 16 + n) * 4 + (m & 3);      // (row m, col n) of a 16x16 accumulator image
    float s[2 * NG];
#pragma unroll
    for (int g = 0; g < 2 * NG; ++g) {
      const float* f = reinterpret_cast<const float*>(red + g * kGruWaves * 64) + frag;
      float v = 0.f;
#pragma unroll
      for (int wv = 0; wv < kGruWaves; ++wv) v += f[wv * 256];
      s[g] = v;
    }
    if (composed) {                                            // + (W_ih W_in) lin(window): K = fs <= 16
      const f32x4* l4 = reinterpret_cast<const f32x4*>(s_lin + m * ldl);
      const f32x4 l0 = l4[0], l1 = l4[1], l2 = l4[2], l3 = l4[3];
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const f32x4* v4 = reinterpret_cast<const f32x4*>(vs + (g * 16 + n) * 16);
        const f32x4 p = l0 * v4[0] + l1 * v4[1] + l2 * v4[2] + l3 * v4[3];
        s[g] += (p[0] + p[1]) + (p[2] + p[3]);
      }
    }
    if (m < mg) {
      const int unit = ub * 16 + n;
      const int64_t o = (int64_t)(m_first + m) * H + unit;
      if (LSTM) {
        // gates = (W_ih x + W_hh h) + (b_ih + b_hh); i, f, o = s(.), g = tanh(.); c' = f c + i g; h' = o tanh(c')
        float gt[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) gt[g] = (s[g] + s[NG + g]) + cell_b[g];
        const float ig = sigmoid_fast(gt[0]), fg = sigmoid_fast(gt[1]), cg = tanh_fast(gt[2]), og = sigmoid_fast(gt[3]);
        const float cn = fg * a.c[o] + ig * cg;
        a.c[o] = cn;
        store_state(o, og * tanh_fast(cn));
      } else {
        const float gi_r = s[0] + cell_b[0];
        const float gi_z = s[1] + cell_b[1];
        const float gi_n = s[2] + cell_b[2];
        const float gh_r = s[3] + cell_b[NG];
        const float gh_z = s[4] + cell_b[NG + 1];
        const float gh_n = s[5] + cell_b[NG + 2];
        const float r = sigmoid_fast(gh_r + gi_r);
        const float z = sigmoid_fast(gh_z + gi_z);
        const float nn = tanh_fast(gi_n + gh_n * r);
        const float hp = hs[m * ldx + unit];
        store_state(o, (hp - nn) * z + nn);
      }
    }
  }
  // ---- the last workgroup to finish publishes the new slot -----------------------------------------------------
  __syncthreads();
  stamp(3);   // cell
  if (stamping && !fused_up) {
    for (int i = 0; i < 5; ++i) a.stamps[i] += st_acc[i];
    a.stamps[7] += 1;
  }
  if (tid == 0) {
    if (!fused_up) __threadfence();
    const unsigned ticket = atomicAdd(a.done, 1u);
    if (ticket == gridDim.x - 1) {
      *a.done = 0;
      if (!fused_up) __threadfence();
      __hip_atomic_store(a.cnt, cnt + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (!fused_up) return;
  // ---- up-sampler: this workgroup's clips x its `up` column tiles, K split over the waves -------------------------------
  // (the rows of the own row tile come from the H / 16 workgroups of the tile - 32 KB at H = 512; every workgroup reading
  //  every clip's row for 16 columns was 4 x that and 6.4 us, measured)
  {
    const int c0 = wave * CPW;
    // The rows of the new state as granules, polled with agent-scope loads (sc1) until every tag is this update's.  Loads AND
    // the wait sit in one asm statement: the compiler does not know these are loads and would otherwise be free to copy the
    // destination registers too early.
    f32x4 hv[CPW];
    {
      typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
      const int m = lane & 15;
      const unsigned long long* hr = a.h_gran + (int64_t)(m_first + (m < mg ? m : mg - 1)) * H + c0 * 16 + 4 * (lane >> 4);   // clamped
      const unsigned epoch = (unsigned)(cnt + 1);
      u32x4v g[2 * CPW];
      unsigned spins = 0;
      // first a light poll - one wave, one granule per producing workgroup (its last element) - so that 64 K granules per
      // workgroup are not requested over and over while the producers' stores are still on their way; then everything
      if (wave == 0 && lane < KC) {
        const unsigned long long* sp = a.h_gran + (int64_t)(m_first + mg - 1) * H + lane * 16 + 15;
        while ((unsigned)(__hip_atomic_load(sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32) != epoch) {
          if (++spins > (1u << 20)) break;      // the full poll below reports it
          __builtin_amdgcn_s_sleep(1);
        }
        spins = 0;
      }
      __syncthreads();
      for (;;) {
        if constexpr (CPW == 1) {
          asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
                       : "=&v"(g[0]), "=&v"(g[1]) : "v"(hr) : "memory");
        } else if constexpr (CPW == 2) {
          asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %4, off offset:16 sc1\n\t"
                       "global_load_dwordx4 %2, %4, off offset:128 sc1\n\tglobal_load_dwordx4 %3, %4, off offset:144 sc1\n\ts_waitcnt vmcnt(0)"
                       : "=&v"(g[0]), "=&v"(g[1]), "=&v"(g[2]), "=&v"(g[3]) : "v"(hr) : "memory");
        } else {
          static_assert(CPW == 4, "H in {128, 256, 512}");
          asm volatile("global_load_dwordx4 %0, %8, off sc1\n\tglobal_load_dwordx4 %1, %8, off offset:16 sc1\n\t"
                       "global_load_dwordx4 %2, %8, off offset:128 sc1\n\tglobal_load_dwordx4 %3, %8, off offset:144 sc1\n\t"
                       "global_load_dwordx4 %4, %8, off offset:256 sc1\n\tglobal_load_dwordx4 %5, %8, off offset:272 sc1\n\t"
                       "global_load_dwordx4 %6, %8, off offset:384 sc1\n\tglobal_load_dwordx4 %7, %8, off offset:400 sc1\n\ts_waitcnt vmcnt(0)"
                       : "=&v"(g[0]), "=&v"(g[1]), "=&v"(g[2]), "=&v"(g[3]), "=&v"(g[4]), "=&v"(g[5]), "=&v"(g[6]), "=&v"(g[7])
                       : "v"(hr) : "memory");
        }
        bool all = true;
#pragma unroll
        for (int k = 0; k < 2 * CPW; ++k) all = all && g[k][1] == epoch && g[k][3] == epoch;
        if (all) break;
        // ~1 s: a workgroup of the grid never became resident (the launcher checks the grid against the CU count)
        if (++spins > (1u << 20) || (MMK_WAIT_ERR_LOOK && (spins & 255u) == 0 && a.err && __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
          if (a.err) atomicExch(a.err, 3);
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
#pragma unroll
      for (int u = 0; u < CPW; ++u)
        hv[u] = f32x4{__uint_as_float(g[2 * u][0]), __uint_as_float(g[2 * u][2]), __uint_as_float(g[2 * u + 1][0]), __uint_as_float(g[2 * u + 1][2])};
    }
    stamp(5);   // the new state of the own clips has arrived
    // batches of tiles: {0}, {1 .. UB - 1} (both from the registers filled at the top; PB = first preloaded tile of the
    // batch, a compile-time index), then UB at a time streamed (PB < 0)
    auto run_batch = [&](auto pb, int jb, int nb) {
      constexpr int PB = decltype(pb)::value;
#pragma unroll
      for (int j = 0; j < UB; ++j) {
        if (j < nb) {
          f32x4 wu[CPW];
          if constexpr (PB >= 0) {
#pragma unroll
            for (int u = 0; u < CPW; ++u) wu[u] = wu0[(PB + j) < UB ? (PB + j) : 0][u];
          } else {
            gf32x4_ptr wsrc = (gf32x4_ptr)(uintptr_t)a.ups_wp + ((int64_t)((jb + j) * KC + ub) * KC + c0) * 64 + lane;
#pragma unroll
            for (int u = 0; u < CPW; ++u) wu[u] = wsrc[u * 64];
          }
          f32x4 ua = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int u = 0; u < CPW; ++u) {
#pragma unroll
            for (int i = 0; i < 4; ++i) ua = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[u][i], wu[u][i], ua, 0, 0, 0);
          }
          red[(j * kGruWaves + wave) * 64 + lane] = ua;
        }
      }
      __syncthreads();
      for (int e = tid; e < nb * 256; e += kGruThreads) {       // 16 clips x 16 nb columns
        const int j = e >> 8, r = (e >> 4) & 15, n = e & 15;
        const int frag = ((r >> 2) * 16 + n) * 4 + (r & 3);     // (row r, col n) of a 16x16 accumulator image
        const float* f = reinterpret_cast<const float*>(red + j * kGruWaves * 64) + frag;
        float v = 0.f;
#pragma unroll
        for (int wv = 0; wv < kGruWaves; ++wv) v += f[wv * 256];
        const int col = ((jb + j) * KC + ub) * 16 + n;
        if (r < mg && col < a.ups_n) {
          float* dst = a.ups_out + (int64_t)(m_first + r) * a.ups_out_ld + col;
          const float bias = (PB == 0 && nb == 1) ? ups_b[0] : ((PB == 1 && e < kGruThreads) ? ups_b[1] : (a.ups_bias ? a.ups_bias[col] : 0.f));
          const float o = v + bias;
          *dst = o;
        }
      }
      __syncthreads();
    };
#ifndef MMK_SRNN_UP_ONE
#define MMK_SRNN_UP_ONE 0      // all preloaded tiles as ONE batch (one reduction round instead of two; slot 0 goes out later, the launch ends sooner)
#endif
    if (MMK_SRNN_UP_ONE) {
      run_batch(std::integral_constant<int, 0>{}, 0, min(up_tiles, UB));
    } else {
      run_batch(std::integral_constant<int, 0>{}, 0, 1);
      if (up_tiles > 1) run_batch(std::integral_constant<int, 1>{}, 1, min(up_tiles, UB) - 1);
    }
    for (int jb = UB; jb < up_tiles; jb += UB) run_batch(std::integral_constant<int, -1>{}, jb, min(UB, up_tiles - jb));
  }
  stamp(6);   // up-sampler
  if (stamping) {
    for (int i = 0; i < 7; ++i) a.stamps[i] += st_acc[i];
    a.stamps[7] += 1;
  }
}

size_t srnn_gru_lds_bytes(int H, int fs, bool lstm) {
  return (size_t)2 * 16 * (H + 4) * 4 + (size_t)(lstm ? 8 : 6) * kGruWaves * 64 * 16 + (size_t)16 * (((fs + 15) / 16) * 16 + 4) * 4 + (size_t)4 * 16 * 16 * 4;
}

// the fused up-sampler phase waits for every workgroup of the grid: they must all be resident (one 512-thread workgroup
// with > 80 KB of LDS per CU)
bool srnn_gru_grid_resident(int H, int B) {
  static const int n_cu = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    return n;
  }();
  return (H / 16) * ((B + 15) / 16) <= n_cu;
}

bool srnn_gru_supported(int H, int fs, bool lstm) {
  if (!(H == 128 || H == 256 || H == 512)) return false;
  return fs >= 1 && fs <= 256 && srnn_gru_lds_bytes(H, fs, lstm) <= 160 * 1024;
}

int launch_srnn_gru(const SrnnGruArgs& a, hipStream_t stream) {
  const bool lstm = a.lstm != 0;
  if (!srnn_gru_supported(a.H, a.fs, lstm)) return fail(MMK_ERR_UNSUPPORTED, "srnn tier kernel: geometry H=%d fs=%d", a.H, a.fs);
  const size_t lds = srnn_gru_lds_bytes(a.H, a.fs, lstm);
  dim3 grid((a.H / 16) * ((a.B + 15) / 16)), block(kGruThreads);
  if (a.ups_wp && !srnn_gru_grid_resident(a.H, a.B)) return fail(MMK_ERR_INVALID, "srnn tier kernel: fused up-sampler on a grid that is not resident at once");
#define MMK_GRU(KC_)                                                                              \
  do {                                                                                            \
    if (lstm) hipLaunchKernelGGL((srnn_gru_kernel<KC_, true>), grid, block, lds, stream, a);       \
    else hipLaunchKernelGGL((srnn_gru_kernel<KC_, false>), grid, block, lds, stream, a);           \
  } while (0)
  switch (a.H) {
    case 128: MMK_GRU(8); break;
    case 256: MMK_GRU(16); break;
    default: MMK_GRU(32); break;
  }
#undef MMK_GRU
  MMK_HIP(hipGetLastError());
  return MMK_OK;
}

}  // namespace mmk

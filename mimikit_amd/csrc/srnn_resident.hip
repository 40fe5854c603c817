// SampleRNN generate block as ONE launch: every recurrent tier, the bottom tier and the head stay resident for the whole block,
// workgroup roles by blockIdx, every weight in registers (gfx950).
//
// Reference: SampleRNN.generate_step (sample_rnn_v2.py:236-260) - tier i updates when t % fs[i] == 0 (:246-251,
// SampleRNNTier.forward :83-99: input Linear of the frame + the slot of the tier above, GRU / LSTM cell, LinearResampler), the
// bottom tier and the MLP head run every step (:252-260), the drawn class is written into the window of the next step.
//
// Why one launch.  With one launch per tier update (srnn_gru.hip) a tier launch of cfg 3 (frame sizes 16 / 4 / 1, H = 512, 64 clips)
// took 19 us, 7 of them re-loading 25 MB of gate matrices that had left the L2s since the update before, five launches per 16 steps:
// the tier stream alone was the whole 6 us step and the bottom kernel waited for it half of its time.  Here the matrices are loaded
// ONCE per block into the registers of the workgroups that use them, and the roles meet through data-tagged 8-byte granules
// {update number or position : 32, value : 32} (agent-scope stores, polled with agent-scope loads: one hop per hand-over, no flag):
//
//   blockIdx <  B                     bottom role, one workgroup per clip: hidden layer, fc2, draw - every step
//   tier i, KC x ceil(B / 16 MT)      tier role: 16 hidden units x 16 MT clips per workgroup, K split over the 8 waves (MFMA 16x16x4 f32)
//
// Same arithmetic in another association (as srnn_gru.hip / srnn_bottom.hip's composed modes, pinned by the same goldens and oracle
// tests): products of two weight matrices that meet without a non-linearity between them are multiplied at commit (fp64, rounded once):
//   W_ih x, x = W_in lin + b_in + up_j,  up_j = W_up,j h'_above + b_up,j  (the tier above's slot j):
//                         = (W_ih W_in) lin(window) + (W_ih W_up,j) h'_above + (W_ih (b_in + b_up,j) + b_ih)
//   fc0(conv(lin) + up_j) = (W0 W_up,j) h' + (W0 wb) lin(window) + (W0 (b_up,j + bb) + b0)
// No input Linear and no up-sampler ever runs: a tier (and the head) multiplies the STATE of the tier above.  For slot 0 - the slot on the
// chain: the tier above has just updated - the consumer itself multiplies (. W_up,0) h'_above as soon as that state is out, one hop behind
// the cell (the tier's workgroups from register tiles; the clip's bottom workgroup its 128 x 512 product); the rows of the slots j >= 1,
// needed one or more frames later, are made by the PRODUCING tier's workgroups meanwhile (whole gate rows for the tier below, streamed
// link tiles; hidden-layer rows for the head, register tiles).  What sits between the draw of a frame's last class and the next step's
// hidden layer is then: class granule -> one FMA + the cell -> state granules -> a product in the consumer.
//
// Hazards of running free (no launch boundary orders anything): a granule array is only ever rewritten by an update that cannot
// start before every reader of the previous value is through (class -> cell -> rows -> class is a cycle through every role), except
// the state granules, whose readers inside the tier (the all-gather of the next recurrent product) may lag: two buffers, by parity.
#include <type_traits>

#include "mmk_common.h"
#include "sampler256.h"
#include "srnn_resident.h"

namespace mmk {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) f32x4* gf32x4_ptr;
typedef unsigned long long u64;
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

constexpr int kResThreads = 512;
constexpr int kResWaves = kResThreads / 64;
constexpr unsigned kResSpinLimit = 1u << 20;     // ~1 s of polls: a role is not running beside this one

__device__ __forceinline__ u64 res_gload(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void res_gstore(u64* p, unsigned tag, unsigned bits) {
  __hip_atomic_store(p, ((u64)tag << 32) | (u64)bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#ifndef MMK_SRNN_RES_POLL_SLEEP
#define MMK_SRNN_RES_POLL_SLEEP 1      // s_sleep units between two looks at a granule that is not there yet
#endif
#ifndef MMK_SRNN_RES_ERR_LOOK
#define MMK_SRNN_RES_ERR_LOOK 0        // 1: a waiting wave looks at the error word every 256 polls; 0: its own time-out only - one exit less in every wait loop (cfg 3: 3.63 -> 3.44 us per step)
#endif
__device__ __forceinline__ void res_pause() { if (MMK_SRNN_RES_POLL_SLEEP > 0) __builtin_amdgcn_s_sleep(MMK_SRNN_RES_POLL_SLEEP); }
__device__ __forceinline__ bool res_give_up(unsigned& spins, int* err, int code) {
  if (++spins > kResSpinLimit) {
    if (err) atomicCAS(err, 0, code);
    return true;
  }
  if (!MMK_SRNN_RES_ERR_LOOK) return false;
  return (spins & 255u) == 0 && err && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;   // another wait has failed: do not pile up
}
// one granule, polled until its tag is `tag`
__device__ __forceinline__ unsigned res_wait(const u64* src, unsigned tag, int* err, int code) {
  u64 g = res_gload(src);
  unsigned spins = 0;
  while ((unsigned)(g >> 32) != tag) {
    if (res_give_up(spins, err, code)) break;
    res_pause();
    g = res_gload(src);
  }
  return (unsigned)g;
}

// A lane's share of a row of granules as an MFMA A operand: CPW chunks of 16 values, 4 consecutive ones per lane and chunk (two 16-byte
// agent-scope loads each), polled until every tag is `tag`.  Loads AND the wait sit in one asm statement: the compiler does not know
// these are loads and would otherwise be free to copy the destination registers too early.
template <int CPW>
__device__ __forceinline__ void res_poll_slice(const u64* hr, unsigned tag, f32x4 (&out)[CPW], int* err, int code) {
  u32x4v g[2 * CPW];
  unsigned spins = 0;
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
    // every tag == `tag`  <=>  their minimum and their maximum are: two running three-operand min / max chains and two compares instead of 4 CPW compares
    // and as many ands between the loads' return and the first product (the ring's poison check taught this: docs/history.md, round 6)
    unsigned tmin = g[0][1], tmax = g[0][1];
#pragma unroll
    for (int k = 0; k < 2 * CPW; ++k) {
      tmin = min(min(tmin, g[k][1]), g[k][3]);
      tmax = max(max(tmax, g[k][1]), g[k][3]);
    }
    if (tmin == tag && tmax == tag) break;
    if (res_give_up(spins, err, code)) break;
    res_pause();
  }
#pragma unroll
  for (int u = 0; u < CPW; ++u)
    out[u] = f32x4{__uint_as_float(g[2 * u][0]), __uint_as_float(g[2 * u][2]), __uint_as_float(g[2 * u + 1][0]), __uint_as_float(g[2 * u + 1][2])};
}

// two rows' slices in one round trip (H = 512: 16 loads in flight per lane); the tags of both are checked, either may be re-read
__device__ __forceinline__ void res_poll_slice_pair(const u64* h0, const u64* h1, unsigned tag, f32x4 (&o0)[4], f32x4 (&o1)[4], int* err, int code) {
  u32x4v g[16];
  unsigned spins = 0;
  for (;;) {
    asm volatile("global_load_dwordx4 %0, %16, off sc1\n\tglobal_load_dwordx4 %1, %16, off offset:16 sc1\n\t"
                 "global_load_dwordx4 %2, %16, off offset:128 sc1\n\tglobal_load_dwordx4 %3, %16, off offset:144 sc1\n\t"
                 "global_load_dwordx4 %4, %16, off offset:256 sc1\n\tglobal_load_dwordx4 %5, %16, off offset:272 sc1\n\t"
                 "global_load_dwordx4 %6, %16, off offset:384 sc1\n\tglobal_load_dwordx4 %7, %16, off offset:400 sc1\n\t"
                 "global_load_dwordx4 %8, %17, off sc1\n\tglobal_load_dwordx4 %9, %17, off offset:16 sc1\n\t"
                 "global_load_dwordx4 %10, %17, off offset:128 sc1\n\tglobal_load_dwordx4 %11, %17, off offset:144 sc1\n\t"
                 "global_load_dwordx4 %12, %17, off offset:256 sc1\n\tglobal_load_dwordx4 %13, %17, off offset:272 sc1\n\t"
                 "global_load_dwordx4 %14, %17, off offset:384 sc1\n\tglobal_load_dwordx4 %15, %17, off offset:400 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(g[0]), "=&v"(g[1]), "=&v"(g[2]), "=&v"(g[3]), "=&v"(g[4]), "=&v"(g[5]), "=&v"(g[6]), "=&v"(g[7]),
                   "=&v"(g[8]), "=&v"(g[9]), "=&v"(g[10]), "=&v"(g[11]), "=&v"(g[12]), "=&v"(g[13]), "=&v"(g[14]), "=&v"(g[15])
                 : "v"(h0), "v"(h1) : "memory");
    unsigned tmin = g[0][1], tmax = g[0][1];      // (as in res_poll_slice)
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      tmin = min(min(tmin, g[k][1]), g[k][3]);
      tmax = max(max(tmax, g[k][1]), g[k][3]);
    }
    if (tmin == tag && tmax == tag) break;
    if (res_give_up(spins, err, code)) break;
    res_pause();
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    o0[u] = f32x4{__uint_as_float(g[2 * u][0]), __uint_as_float(g[2 * u][2]), __uint_as_float(g[2 * u + 1][0]), __uint_as_float(g[2 * u + 1][2])};
    o1[u] = f32x4{__uint_as_float(g[8 + 2 * u][0]), __uint_as_float(g[8 + 2 * u][2]), __uint_as_float(g[9 + 2 * u][0]), __uint_as_float(g[9 + 2 * u][2])};
  }
}

// phase totals of thread 0 of a role's first workgroup, 100 MHz ticks: the diagnostic build only (-DMMK_DIAG) - in the product kernel the
// 64-bit totals would cost every lane 18 registers next to the resident weights
struct ResStamp {
#ifdef MMK_DIAG
  unsigned long long* dst;
  unsigned long long prev, acc[7];
  __device__ __forceinline__ void begin(unsigned long long* d) {
    dst = d;
    prev = d ? wall_clock64() : 0;
#pragma unroll
    for (int i = 0; i < 7; ++i) acc[i] = 0;
  }
  __device__ __forceinline__ void at(int slot) {
    if (dst) {
      const unsigned long long now = wall_clock64();
      acc[slot] += now - prev;
      prev = now;
    }
  }
  __device__ __forceinline__ void end(unsigned long long count) {
    if (dst) {
#pragma unroll
      for (int i = 0; i < 7; ++i) dst[i] += acc[i];
      dst[7] += count;
    }
  }
#else
  __device__ __forceinline__ void begin(unsigned long long*) {}
  __device__ __forceinline__ void at(int) {}
  __device__ __forceinline__ void end(unsigned long long) {}
#endif
};

// ---- tier role -------------------------------------------------------------------------------------------------------------
// KC = H / 16 unit blocks; MT row tiles of 16 clips per workgroup; HAS_UPPER: a tier above feeds this one; LAST: the tier right above
// the bottom tier.  Registers for the block: W_hh (NG tiles), the link tiles of slot 0 (HAS_UPPER, NG tiles: W_ih W_up[slot 0] of the
// tier above - this tier multiplies them with that tier's new STATE, one hop behind its cell, instead of waiting for its up-sampler),
// the first tiles of the rows composed with the head (LAST), the state of the own clips as MFMA operands.  What a tier makes for the
// tier below besides its state - W_ih,below (W_up[slot j] h' + b) for the slots j >= 1, whole gate rows - is streamed: those slots are
// only needed one or more frames of the tier below later.
template <int KC, bool LSTM, int MT, bool HAS_UPPER, bool LAST>
__device__ __forceinline__ void res_tier_role(const SrnnResArgs& a, const SrnnResTier& T, const int tier_index, char* smem_raw) {
  constexpr int H = KC * 16;
  constexpr int CPW = KC / kResWaves;              // K-chunks per wave
  constexpr int NG = LSTM ? 4 : 3;
  constexpr int GH = NG * H;
  constexpr int NP = MT >= 2 ? MT / 2 : 1;         // (clip, unit) pairs per thread in the cell
  constexpr int RED = NG * MT;                     // partial-sum images of 8 KB in LDS
  constexpr int BT = NG;                           // output tiles per batch (BT MT <= RED)
  constexpr int NRU = LAST ? (KC == 32 ? 1 : BT) : 0;   // output tiles kept in registers
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wg = blockIdx.x - T.block0;
  const int ub = wg % KC;                          // block of 16 hidden units
  const int m_first = (wg / KC) * 16 * MT;         // first clip
  const int mg = min(16 * MT, a.B - m_first);
  const int fs = T.fs, fsp = T.fsp, ldl = fsp + 4;
  const int c0 = wave * CPW;
  const int64_t BH = (int64_t)a.B * H;

  ResStamp st;
  st.begin((a.stamps && wg == 0 && tid == 0) ? a.stamps + 8 * (1 + tier_index) : nullptr);

  char* sp = smem_raw;
  f32x4* red = (f32x4*)sp;    sp += (size_t)RED * kResWaves * 64 * 16;
  float* s_lin = (float*)sp;  sp += (size_t)16 * MT * ldl * 4;
  float* vs = (float*)sp;                                          // (W_ih W_in) rows of this workgroup's units: [NG][16][fsp]

  // ---- once per block: every weight this workgroup multiplies at every update, into registers --------------------------------------
  f32x4 whh[NG][CPW], win0[HAS_UPPER ? NG : 1][CPW], wout[NRU > 0 ? NRU : 1][CPW];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    gf32x4_ptr wh = (gf32x4_ptr)(uintptr_t)T.whh_wp + ((int64_t)(g * KC + ub) * T.w_tile_chunks + c0) * 64 + lane;
#pragma unroll
    for (int u = 0; u < CPW; ++u) whh[g][u] = wh[u * 64];
    if constexpr (HAS_UPPER) {     // slot 0 of the link from the tier above: rows g H + 16 ub .. of W_ih W_up[0]
      gf32x4_ptr wi = (gf32x4_ptr)(uintptr_t)T.link_wp + ((int64_t)(g * KC + ub) * KC + c0) * 64 + lane;
#pragma unroll
      for (int u = 0; u < CPW; ++u) win0[g][u] = wi[u * 64];
    }
  }
  // output tiles of this workgroup: LAST - tile i of the rows composed with the head's first layer (ub n_tiles + i); otherwise the link
  // of the tier below, slot 1 + i / NG, gate i % NG: tile ((1 + i / NG) NG + i % NG) KC + ub of W_ih,below W_up
  auto tile_of = [&](int i) -> int64_t { return LAST ? (int64_t)ub * T.n_tiles + i : (int64_t)(NG + i) * KC + ub; };
  if constexpr (NRU > 0) {
#pragma unroll
    for (int i = 0; i < NRU; ++i) {
      gf32x4_ptr ws = (gf32x4_ptr)(uintptr_t)T.out_wp + (tile_of(min(i, T.n_tiles - 1)) * KC + c0) * 64 + lane;
#pragma unroll
      for (int u = 0; u < CPW; ++u) wout[i][u] = ws[u * 64];
    }
  }
  for (int e = tid; e < NG * 16 * fsp; e += kResThreads) {
    const int g = e / (16 * fsp), r = e - g * 16 * fsp, n = r / fsp, i = r - n * fsp;
    vs[e] = T.v_full[((int64_t)g * H + ub * 16 + n) * fsp + i];
  }
  // the cell's (clip, unit) pairs of this thread: constants, old state
  const int64_t cnt0 = *T.cnt;
  float cst_in[NP][NG], cst_hh[NP][NG], hprev[NP], cprev[NP];
  int p_m[NP], p_clip[NP];                         // row inside the workgroup (mt 16 + m), clip; p_clip < 0: no such pair / no such clip
  const int pn = tid & 15;
  const int unit = ub * 16 + pn;
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const int p = tid + k * kResThreads;
    const bool has = p < MT * 256;
    p_m[k] = has ? (p >> 4) : 0;
    p_clip[k] = (has && p_m[k] < mg) ? m_first + p_m[k] : -1;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      cst_in[k][g] = T.gconst[g * H + unit];       // top tier: W_ih b_in + b_ih; else: the constant of the link's slot 0
      cst_hh[k][g] = LSTM ? 0.f : T.gconst[(NG + g) * H + unit];
    }
    const int64_t o = (int64_t)(p_clip[k] < 0 ? m_first : p_clip[k]) * H + unit;
    hprev[k] = T.h_ring[(cnt0 & 1) * T.h_slot_stride + o];
    cprev[k] = LSTM ? T.c[o] : 0.f;
  }
  // the old state of the own clips, this wave's K range, as MFMA operands
  f32x4 hv[MT][CPW];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = mt * 16 + (lane & 15);
    const float* hr = T.h_ring + (cnt0 & 1) * T.h_slot_stride + (int64_t)(m_first + (m < mg ? m : mg - 1)) * H + c0 * 16 + 4 * (lane >> 4);
#pragma unroll
    for (int u = 0; u < CPW; ++u) hv[mt][u] = *reinterpret_cast<const f32x4*>(hr + u * 16);
  }
  // products of NG tiles with a slice of rows: partial sums of this wave's K range -> LDS
  auto gate_products = [&](const f32x4 (&w)[NG][CPW], const f32x4 (&x)[CPW], int mt) {
    f32x4 acc[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < CPW; ++u) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[u][i], w[g][u][i], acc[g], 0, 0, 0);
      }
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) red[((mt * NG + g) * kResWaves + wave) * 64 + lane] = acc[g];
  };
  // the sums over the waves for the cell's pairs (fixed order)
  auto gate_sums = [&](float (&s)[NP][NG]) {
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const int m = p_m[k] & 15, mt = p_m[k] >> 4;
      const int frag = ((m >> 2) * 16 + pn) * 4 + (m & 3);        // (row m, col n) of a 16x16 accumulator image
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const float* f = reinterpret_cast<const float*>(red + (mt * NG + g) * kResWaves * 64) + frag;
        float v = 0.f;
#pragma unroll
        for (int wv = 0; wv < kResWaves; ++wv) v += f[wv * 256];
        s[k][g] = v;
      }
    }
  };
  // Slices of rows of granules (this wave's K range of the clips of a row tile), preceded by a light poll - one granule per lane and row tile: the
  // last unit of a producing workgroup for one clip - so that the whole slices (64 KB per row tile at H = 512) are not requested over and over
  // while the producers' stores are still on their way; two row tiles per round trip where the registers allow (H = 512)
  auto row_of = [&](const u64* base, int mt) -> const u64* {
    const int m = mt * 16 + (lane & 15);
    return base + (int64_t)(m_first + (m < mg ? m : mg - 1)) * H;
  };
  auto light_poll = [&](const u64* base, unsigned tag, int mt) {
    (void)res_wait(row_of(base, mt) + (c0 + (lane >> 4) % CPW) * 16 + 15, tag, a.err, 7);
  };
  auto gather = [&](const u64* base, unsigned tag, int mt, f32x4 (&out)[CPW]) {
    light_poll(base, tag, mt);
    res_poll_slice<CPW>(row_of(base, mt) + c0 * 16 + 4 * (lane >> 4), tag, out, a.err, 7);
  };
  auto gather_own = [&](const u64* base, unsigned tag) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
      if (mt * 16 < mg) light_poll(base, tag, mt);
    if constexpr (CPW == 4 && MT >= 2) {
#pragma unroll
      for (int mt = 0; mt < MT; mt += 2) {
        if ((mt + 1) * 16 < mg) res_poll_slice_pair(row_of(base, mt) + c0 * 16 + 4 * (lane >> 4), row_of(base, mt + 1) + c0 * 16 + 4 * (lane >> 4), tag, hv[mt], hv[mt + 1], a.err, 7);
        else if (mt * 16 < mg) res_poll_slice<CPW>(row_of(base, mt) + c0 * 16 + 4 * (lane >> 4), tag, hv[mt], a.err, 7);
      }
    } else {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        if (mt * 16 < mg) res_poll_slice<CPW>(row_of(base, mt) + c0 * 16 + 4 * (lane >> 4), tag, hv[mt], a.err, 7);
    }
  };
  // W_hh h of the state the block starts from
  float s_hh[NP][NG];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
    if (mt * 16 < mg) gate_products(whh, hv[mt], mt);
  __syncthreads();
  gate_sums(s_hh);
  __syncthreads();

  const int n_upd = (a.n_steps + fs - 1) / fs;
  for (int upd = 0; upd < n_upd; ++upd) {
    const int64_t t = a.t_begin + (int64_t)upd * fs;
    const unsigned epoch = (unsigned)(t / fs) + 1u;
    const int64_t par = (int64_t)(epoch & 1u) * BH;
    // ---- the input half of the gates but for the window's part ------------------------------------------------------------------------
    float s_in[NP][NG];
    if constexpr (HAS_UPPER) {
      const int slot = (int)((t / fs) % T.up_mod);                             // outputs[i-1][:, (t // fs) % ...]   (:251)
      const unsigned uep = (unsigned)(t / ((int64_t)fs * T.up_mod)) + 1u;      // the update of the tier above this slot belongs to
      if (slot == 0) {
        // that tier's new state, one hop behind its cell, times W_ih W_up[0]
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          if (mt * 16 < mg) {
            f32x4 xv[CPW];
            gather(T.upper_h_gran + (int64_t)(uep & 1u) * BH, uep, mt, xv);
            gate_products(win0, xv, mt);
          }
        }
        __syncthreads();
        gate_sums(s_in);
#pragma unroll
        for (int k = 0; k < NP; ++k)
#pragma unroll
          for (int g = 0; g < NG; ++g) s_in[k][g] += cst_in[k][g];
      } else {
        // whole gate rows from the tier above (its streamed link tiles, constants included)
#pragma unroll
        for (int k = 0; k < NP; ++k) {
          const u64* src = T.upper_gran + ((int64_t)(p_clip[k] < 0 ? m_first : p_clip[k]) * T.up_mod + slot) * GH + unit;
#pragma unroll
          for (int g = 0; g < NG; ++g) s_in[k][g] = __uint_as_float(res_wait(src + g * H, uep, a.err, 7));
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < NP; ++k)
#pragma unroll
        for (int g = 0; g < NG; ++g) s_in[k][g] = cst_in[k][g];
    }
    st.at(0);
    // ---- the window, linearized (modules/io.py:106-112): the classes come from the bottom role as granules -------------------------
    for (int e = tid; e < 16 * MT * fsp; e += kResThreads) {
      const int m = e / fsp, i = e - m * fsp;
      float v = 0.f;
      if (m < mg && i < fs) {
        const int64_t pos = t - fs + i;
        const unsigned cls = a.teacher ? (unsigned)a.idx[(int64_t)(m_first + m) * a.idx_rs + pos + a.shift]
                                       : res_wait(a.cls_gran + (int64_t)(m_first + m) * 256 + (pos & 255), (unsigned)(pos + 1), a.err, 7);
        v = (((float)cls / a.class_size) - .5f) * 2.f;
      }
      s_lin[m * ldl + i] = v;
    }
    __syncthreads();
    st.at(1);
    if (a.teacher) {
      // Teacher-forced (the warm-up): no drawn class paces the tiers.  This update's rows of the tier above are read: say so; and before anything of this update
      // is published, the tier BELOW must have read everything of the update before (its rows are one buffer per slot, not two)
      if (tid == 0) atomicAdd(T.prog, 1ull);
      if constexpr (!LAST) {
        if (upd >= 1) {
          const SrnnResTier& TB = a.tier[tier_index + 1];
          const unsigned long long want = (unsigned long long)(KC * ((a.B + 16 * TB.mt - 1) / (16 * TB.mt))) * (unsigned long long)upd * (unsigned long long)T.up;
          unsigned spins = 0;
          while (res_gload(TB.prog) < want) {
            if (res_give_up(spins, a.err, 7)) break;
            __builtin_amdgcn_s_sleep(2);
          }
        }
      }
    }
    // ---- cell: (W_ih W_in) lin(window), gates, new state ------------------------------------------------------------------------
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const f32x4* l4 = reinterpret_cast<const f32x4*>(s_lin + p_m[k] * ldl);
      float gi[NG];
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const f32x4* v4 = reinterpret_cast<const f32x4*>(vs + (g * 16 + pn) * fsp);
        f32x4 p = l4[0] * v4[0];
        for (int c = 1; c < fsp / 4; ++c) p += l4[c] * v4[c];
        gi[g] = s_in[k][g] + ((p[0] + p[1]) + (p[2] + p[3]));
      }
      float hn;
      if (LSTM) {
        // gates = (W_ih x + b_ih + b_hh) + W_hh h; i, f, o = s(.), g = tanh(.); c' = f c + i g; h' = o tanh(c')
        const float ig = sigmoid_fast(gi[0] + s_hh[k][0]), fg = sigmoid_fast(gi[1] + s_hh[k][1]);
        const float cg = tanh_fast(gi[2] + s_hh[k][2]), og = sigmoid_fast(gi[NG - 1] + s_hh[k][NG - 1]);
        const float cn = fg * cprev[k] + ig * cg;
        cprev[k] = cn;
        hn = og * tanh_fast(cn);
      } else {
        const float gh_r = s_hh[k][0] + cst_hh[k][0], gh_z = s_hh[k][1] + cst_hh[k][1], gh_n = s_hh[k][2] + cst_hh[k][2];
        const float r = sigmoid_fast(gh_r + gi[0]);
        const float z = sigmoid_fast(gh_z + gi[1]);
        const float nn = tanh_fast(gi[2] + gh_n * r);
        hn = (hprev[k] - nn) * z + nn;
      }
      hprev[k] = hn;
      if (p_clip[k] >= 0) {
        const int64_t o = (int64_t)p_clip[k] * H + unit;
        res_gstore(T.h_gran + par + o, epoch, __float_as_uint(hn));
        T.h_ring[((cnt0 + upd + 1) & 1) * T.h_slot_stride + o] = hn;       // for whoever runs after this launch
      }
    }
    st.at(2);
    // ---- the new state of the own clips, this wave's K range (from the CPW workgroups that own those units); the first streamed output tiles
    //      are asked for ahead of it --------------------------------------------------------------------------------------------------------
    f32x4 wt[BT][CPW];                                                         // streamed tiles of a batch
    auto stream_batch = [&](int jb, int nb) {
      if (jb >= T.n_tiles) return;
#pragma unroll
      for (int j = 0; j < BT; ++j) {
        gf32x4_ptr ws = (gf32x4_ptr)(uintptr_t)T.out_wp + (tile_of(jb + (j < nb ? j : 0)) * KC + c0) * 64 + lane;
#pragma unroll
        for (int u = 0; u < CPW; ++u) wt[j][u] = ws[u * 64];
      }
    };
    // (LAST: the urgent tiles are in registers, further ones are asked for when their turn comes; four row tiles: their slices need the registers)
    if constexpr (!LAST && MT < 4) stream_batch(0, min(BT, T.n_tiles));
    gather_own(T.h_gran + par, epoch);
    if constexpr (!LAST && MT >= 4) stream_batch(0, min(BT, T.n_tiles));
    st.at(3);
    // ---- output tiles, batches of at most BT: partial sums -> LDS, summed over the waves, published as granules ------------------------
    // (PB: first register tile of the batch, or -1: the streamed tiles in wt; the tiles of the NEXT streamed batch are asked for before this one's sums)
    auto run_batch = [&](auto pb, int jb, int nb) {
      constexpr int PB = decltype(pb)::value;
      constexpr int NB = PB == 0 ? 1 : (PB > 0 ? (NRU - 1 > 0 ? NRU - 1 : 1) : BT);
      if constexpr (PB < 0 && LAST) stream_batch(jb, nb);
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        if (j < nb) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            if (mt * 16 < mg) {
              f32x4 ua = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
              for (int u = 0; u < CPW; ++u) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                  if constexpr (PB >= 0) ua = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[mt][u][i], wout[(PB + j) < NRU ? (PB + j) : 0][u][i], ua, 0, 0, 0);
                  else ua = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[mt][u][i], wt[j][u][i], ua, 0, 0, 0);
                }
              }
              red[((j * MT + mt) * kResWaves + wave) * 64 + lane] = ua;
            }
          }
        }
      }
      if constexpr (PB < 0 && !LAST) stream_batch(jb + BT, min(BT, T.n_tiles - jb - BT));
      __syncthreads();
      for (int e = tid; e < nb * MT * 256; e += kResThreads) {
        const int j = e / (MT * 256), r = (e >> 4) % (16 * MT), n = e & 15;
        if (r < mg) {
          const int m = r & 15;
          const int frag = ((m >> 2) * 16 + n) * 4 + (m & 3);
          const float* f = reinterpret_cast<const float*>(red + (j * MT + (r >> 4)) * kResWaves * 64) + frag;
          float v = 0.f;
#pragma unroll
          for (int wv = 0; wv < kResWaves; ++wv) v += f[wv * 256];
          const int64_t clip = m_first + r;
          if constexpr (!LAST) {
            const int row = (NG + jb + j) * H + ub * 16 + n;                   // slot (NG + jb + j) / NG, gate row (jb + j) % NG H + unit
            res_gstore(T.out_gran + clip * ((int64_t)T.up * GH) + row, epoch, __float_as_uint(v + T.out_bias[row]));
          } else {
            const int rr = (jb + j) * 16 + n;                                  // row of this unit block
            const int gidx = ub * T.rpb + rr;
            if (rr < T.rpb && gidx < (a.S - 1) * a.Hm)
              res_gstore(T.out_gran + clip * ((int64_t)a.S * a.Hm) + a.Hm + gidx, epoch, __float_as_uint(v));     // slot 1 + gidx / Hm, unit gidx % Hm
          }
        }
      }
      __syncthreads();
    };
    // (teacher-forced: the last recurrent tier's rows are for the bottom role, which does not run - the tier below a non-last tier does need its rows)
    const int n_tiles_run = (LAST && a.teacher) ? 0 : T.n_tiles;
    if constexpr (NRU > 0) {
      if (n_tiles_run > 0) run_batch(std::integral_constant<int, 0>{}, 0, 1);
      if constexpr (NRU > 1) {
        if (n_tiles_run > 1) run_batch(std::integral_constant<int, 1>{}, 1, min(n_tiles_run, NRU) - 1);
      }
    }
    for (int jb = NRU; jb < n_tiles_run; jb += BT) run_batch(std::integral_constant<int, -1>{}, jb, min(BT, n_tiles_run - jb));
    st.at(4);
    // ---- W_hh h' for the next update ----------------------------------------------------------------------------------------------------
    if (upd + 1 < n_upd) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        if (mt * 16 < mg) gate_products(whh, hv[mt], mt);
      __syncthreads();
      gate_sums(s_hh);
      __syncthreads();
    }
    st.at(5);
  }
  if (LSTM) {
#pragma unroll
    for (int k = 0; k < NP; ++k)
      if (p_clip[k] >= 0) T.c[(int64_t)p_clip[k] * H + unit] = cprev[k];
  }
  if (wg == 0 && tid == 0) *T.cnt = cnt0 + n_upd;
  st.end(n_upd);
}

// ---- bottom role -----------------------------------------------------------------------------------------------------------
// eight partial sums per lane, 16 lanes (one DPP row) that each hold a different K slice: lanes 2 c, 2 c + 1 of the row end with column c's
// total (own + mirror partner, + half-mirror partner, + the lane two further, + the neighbour: a fixed order) - as srnn_bottom.hip's
__device__ __forceinline__ float res_reduce_scatter8(const float (&v)[8], int ks) {
  auto mirror = [](float x) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x140, 0xf, 0xf, false)); };
  auto half_mirror = [](float x) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x141, 0xf, 0xf, false)); };
  const bool b3 = (ks & 8) != 0, b2 = (ks & 4) != 0, b1 = (ks & 2) != 0;
  float k4[4], k2[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) k4[i] = (b3 ? v[4 + i] : v[i]) + mirror(b3 ? v[i] : v[4 + i]);
#pragma unroll
  for (int i = 0; i < 2; ++i) k2[i] = (b2 ? k4[2 + i] : k4[i]) + half_mirror(b2 ? k4[i] : k4[2 + i]);
  float r = (b1 ? k2[1] : k2[0]) + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(b1 ? k2[0] : k2[1]), 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
  r += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(r), 0xB1, 0xf, 0xf, false));                                             // quad_perm [1,0,3,2]
  return r;
}

// One clip per workgroup, every step of the block (srnn_bottom.hip's one-clip kernel with the hidden layer's product composed
// through the up-sampler: see the top of this file).  Thread layout of both products: lane = 16 cgl + ks - the 16 lanes of a DPP row
// split K in 16 slices, the 4 rows of a wave and the 8 waves give 32 column groups.
template <int NF>   // NF = H / 16
__device__ __forceinline__ void res_bottom_role(const SrnnResArgs& a, char* smem_raw) {
  constexpr int H = NF * 16;
  constexpr int kHmMax = 128;
  constexpr int KS0 = H / 16;                 // inputs of the slot-0 product per thread
  constexpr int F0 = KS0 / 4;
  constexpr int kPad0 = KS0 + 4;              // LDS stride of a slice: 16 lanes x 16 bytes land in 16 different bank groups
  constexpr int KS2 = kHmMax / 16;            // fc2 inputs per thread
  constexpr int kPad2 = KS2 + 4;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Hm = a.Hm, n_out = a.n_out, S = a.S, fsb = a.fsb;
  const int clip = blockIdx.x;
  const int64_t t0 = a.t_begin;
  const int64_t BH = (int64_t)a.B * H;
  const SrnnResTier& TL = a.tier[a.n_tiers - 1];
  ResStamp st;
  st.begin((a.stamps && clip == 0 && tid == 0) ? a.stamps : nullptr);

  char* sp = smem_raw;
  float* xs = (float*)sp;    sp += 16 * kPad0 * 4;          // the state row, slice-padded: element k lives at (k / KS0) kPad0 + k % KS0
  float* hid = (float*)sp;   sp += 16 * kPad2 * 4;          // hidden units, slice-padded likewise (KS2)
  float* lbuf = (float*)sp;  sp += 1024 * 4;                // logits (n_out <= 1024)
  int* s_win = (int*)sp;     sp += 16 * 4;
  float* bcs = (float*)sp;   sp += (size_t)S * kHmMax * 4;  // constants of the hidden layer per slot
  float* acs = (float*)sp;   sp += 16 * kHmMax * 4;         // (W0 wb) per window position
  float* wx = (float*)sp;                                   // fc2 rows past the first 256 outputs (the temperature column): (n_out - 256, Hm)

  const int ks = lane & 15, cg = wave * 4 + (lane >> 4);    // K slice, column group (0 .. 31)
  f32x4 w0[4][F0];                                          // hidden units cg 4 + j of W0 W_up[slot 0], inputs ks KS0 ..
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int u = cg * 4 + j;
    const bool has = u < Hm;
    gf32x4_ptr src = (gf32x4_ptr)(uintptr_t)(a.cp0 + (int64_t)(has ? u : 0) * H + ks * KS0);
#pragma unroll
    for (int f = 0; f < F0; ++f) {
      const f32x4 v = src[f];
      w0[j][f] = has ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  f32x4 w2[8][2];                                           // output columns cg 8 + j (< 256), hidden units ks 8 ..
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = cg * 8 + j;
    const bool has = c < n_out;
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const int k = ks * KS2 + 4 * f;
      const bool in = has && k < Hm;
      const f32x4 v = *(gf32x4_ptr)(uintptr_t)(a.fc2_raw + (int64_t)(in ? c : 0) * Hm + (in ? k : 0));
      w2[j][f] = in ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  // (the reduce-scatter of fc2 leaves class cg 8 + c in lanes 2 c, 2 c + 1 of the row: the even one writes it)
  const float fc2_b = ((ks & 1) == 0 && cg * 8 + (ks >> 1) < n_out) ? a.fc2_bias[cg * 8 + (ks >> 1)] : 0.f;
  // (the matrices IN their registers before the step loop: a load the compiler still counts as pending at the loop's entry makes it wait
  //  inside every step, and such a wait also covers the step's own stores)
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int f = 0; f < F0; ++f) asm volatile("" : "+v"(w0[j][f]));
#pragma unroll
  for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(w2[j][0]), "+v"(w2[j][1]));
  const int n_extra = n_out > 256 ? n_out - 256 : 0;
  for (int i = tid; i < n_extra * Hm; i += kResThreads) wx[i] = a.fc2_raw[(int64_t)256 * Hm + i];
  for (int i = tid; i < 16 * kPad2; i += kResThreads) hid[i] = 0.f;
  for (int i = tid; i < S * kHmMax; i += kResThreads) { const int j = i / kHmMax, u = i - j * kHmMax; bcs[i] = u < Hm ? a.bcs[j * Hm + u] : 0.f; }
  for (int i = tid; i < 16 * kHmMax; i += kResThreads) { const int j = i / kHmMax, u = i - j * kHmMax; acs[i] = (j < fsb && u < Hm) ? a.a_comp[j * Hm + u] : 0.f; }
  if (tid < 16) s_win[tid] = tid < fsb ? (int)a.idx[(int64_t)clip * a.idx_rs + t0 - fsb + tid] : 0;
  const int xc = tid < H ? tid : 0;
  const int xs_at = (xc / KS0) * kPad0 + xc % KS0;
  // sum over the 16 lanes of a DPP row, every lane ends with the total (fixed order)
  auto row_sum = [](float v) -> float {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, false));   // row_half_mirror
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, false));   // row_mirror
    return v;
  };
  __syncthreads();
  const float a_c0 = (ks < 4 && cg * 4 + ks < Hm) ? acs[cg * 4 + ks] : 0.f;     // frames of one sample (the common case): the weight itself

  // (W0 W_up,0) . xs for this lane's K slice of its four hidden units, slices summed across the DPP row
  auto slot0_product = [&]() -> float {
    const f32x4* x4 = reinterpret_cast<const f32x4*>(xs + ks * kPad0);
    f32x4 acc[4];
    {
      const f32x4 xv = x4[0];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = xv * w0[j][0];
    }
#pragma unroll
    for (int f = 1; f < F0; ++f) {           // one input fragment at a time, four independent accumulation chains
      const f32x4 xv = x4[f];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] += xv * w0[j][f];
    }
    float tot[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) tot[j] = row_sum((acc[j][0] + acc[j][1]) + (acc[j][2] + acc[j][3]));
    return ks == 0 ? tot[0] : (ks == 1 ? tot[1] : (ks == 2 ? tot[2] : tot[3]));   // lane ks < 4 of a row: the total of unit cg 4 + ks
  };
  auto fc2_phase = [&]() {
    const f32x4* h4 = reinterpret_cast<const f32x4*>(hid + ks * kPad2);
    const f32x4 h0 = h4[0], h1 = h4[1];
    float part[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      f32x4 acc = h0 * w2[j][0];
      acc += h1 * w2[j][1];
      part[j] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    }
    const float mine = res_reduce_scatter8(part, ks);            // (class cg 8 + ks / 2, in two lanes)
    const int c = cg * 8 + (ks >> 1);
    if ((ks & 1) == 0 && c < n_out) lbuf[c] = mine + fc2_b;
    for (int r = wave; r < n_extra; r += kResWaves) {            // rows past 256: one wave each, lanes over k
      float p = 0.f;
      for (int k = lane; k < Hm; k += 64) p = fmaf(hid[(k / KS2) * kPad2 + k % KS2], wx[r * Hm + k], p);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) p += __shfl_xor(p, o);
      if (lane == 0) lbuf[256 + r] = p + a.fc2_bias[256 + r];
    }
  };
  // Greedy decode of 256 classes with a frame of one sample (cfg 3): EVERY wave picks the class itself from the logits in LDS (the same
  // deterministic pick eight times) and keeps it in a register - no barrier behind the draw
  const bool every_wave_picks = a.temperature == nullptr && fsb == 1 && a.Q == 256;
  int cur_cls = s_win[0];
  auto publish = [&](int64_t t, int result) {
    a.idx[(int64_t)clip * a.idx_rs + t] = result;
    res_gstore(a.cls_gran + (int64_t)clip * 256 + (t & 255), (unsigned)(t + 1), (unsigned)result);       // for the tier roles
  };
  auto sampler_phase = [&](int s, int64_t t) {
    if (every_wave_picks) {
      const int result = greedy_256(lbuf, a.learn_temp != 0, lbuf[256], a.min_temp, lane);
      cur_cls = result;
      if (wave == 0) {
        if (lane == 0) {
          s_win[0] = result;
          publish(t, result);
        }
        if (a.logits_out && s + 1 == a.n_steps)
          for (int c = lane; c < n_out; c += 64) a.logits_out[(int64_t)clip * a.logits_ld + c] = lbuf[c];
      }
    } else if (wave == 0) {
      const float* lg = lbuf;
      const int nc = a.Q;
      const int per = (nc + 63) / 64;
      if (a.logits_out && s + 1 == a.n_steps)
        for (int c = lane; c < n_out; c += 64) a.logits_out[(int64_t)clip * a.logits_ld + c] = lg[c];
      float denom = 1.f;
      if (a.learn_temp && (a.temperature != nullptr || nc != 256)) denom = fmaxf(sigmoidf_(lg[nc]), a.min_temp);   // mlp.py:60-62
      int result;
      if (a.temperature == nullptr) {
        float best = -INFINITY;
        int bi = 0x7fffffff;
        if (nc == 256) {
          bi = greedy_256(lg, a.learn_temp != 0, lg[nc], a.min_temp, lane);
        } else {
          for (int q = 0; q < per; ++q) {
            const int c = lane * per + q;
            if (c < nc) {
              const float v = a.learn_temp ? lg[c] / denom : lg[c];
              if (v > best || bi == 0x7fffffff) { best = v; bi = c; }
            }
          }
          bi = wave_argmax_first(best, bi);        // first maximum wins (torch.argmax)
        }
        result = bi;
      } else if (nc == 256) {
        result = sample_256(lg, a.learn_temp != 0, denom, a.temperature[clip], a.uniforms[(int64_t)clip * a.uni_ld + t + a.uni_off], lane);
      } else {
        const float T = a.temperature[clip];
        float mx = -INFINITY;
        for (int q = 0; q < per; ++q) {
          const int c = lane * per + q;
          if (c < nc) mx = fmaxf(mx, (a.learn_temp ? lg[c] / denom : lg[c]) / T);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        float local = 0.f;
        for (int q = 0; q < per; ++q) {
          const int c = lane * per + q;
          if (c < nc) local += expf((a.learn_temp ? lg[c] / denom : lg[c]) / T - mx);
        }
        float incl = local;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const float up = __shfl_up(incl, o);
          if (lane >= o) incl += up;
        }
        const float total = __shfl(incl, 63);
        const float target = a.uniforms[(int64_t)clip * a.uni_ld + t + a.uni_off] * total;
        float run = incl - local;
        int pick = 0x7fffffff, last_c = -1;
        for (int q = 0; q < per; ++q) {
          const int c = lane * per + q;
          if (c < nc) {
            const float e = expf((a.learn_temp ? lg[c] / denom : lg[c]) / T - mx);
            run += e;
            if (e > 0.f) last_c = c;
            if (pick == 0x7fffffff && run > target && e > 0.f) pick = c;
          }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          const int op = __shfl_xor(pick, o), ol = __shfl_xor(last_c, o);
          pick = op < pick ? op : pick;
          last_c = ol > last_c ? ol : last_c;
        }
        result = pick != 0x7fffffff ? pick : (last_c < 0 ? 0 : last_c);
      }
      if (lane < fsb) {
        const int keep = lane + 1 < fsb ? s_win[lane + 1] : result;
        s_win[lane] = keep;     // wave-synchronous shift: every lane read before any lane writes
      }
      if (lane == 0) publish(t, result);
    }
  };
  const int hid_u = cg * 4 + ks;                              // the hidden unit lane ks < 4 of a row finishes
  const int hid_at = (hid_u / KS2) * kPad2 + hid_u % KS2;
  const bool hid_lane = ks < 4 && hid_u < Hm;
  const u64* p_row = TL.out_gran + (int64_t)clip * ((int64_t)S * Hm) + hid_u;      // + slot Hm
  u64 p_g = 0;                                                 // the next step's composed row value, requested a phase ahead
  st.at(0);
  for (int s = 0; s < a.n_steps; ++s) {
    const int64_t t = t0 + s;
    const int slot = (int)(t % S);                             // outputs[-1][:, (t % fs[-2]) - fs[-2]]   (:257)
    const unsigned epoch = (unsigned)(t / S) + 1u;             // of the update this step's row belongs to
    float p_cur = 0.f;
    if (slot == 0) {
      // first step of a frame: the new state itself, one hop behind the tier's cell, times W0 W_up[slot 0]
      if (tid < H) xs[xs_at] = __uint_as_float(res_wait(TL.h_gran + (int64_t)(epoch & 1u) * BH + (int64_t)clip * H + xc, epoch, a.err, 6));
      __syncthreads();
      st.at(1);
      p_cur = slot0_product();
    } else if (hid_lane) {
      unsigned spins = 0;
      while ((unsigned)(p_g >> 32) != epoch) {
        if (res_give_up(spins, a.err, 6)) break;
        __builtin_amdgcn_s_sleep(1);
        p_g = res_gload(p_row + slot * Hm);
      }
      p_cur = __uint_as_float((unsigned)p_g);
    }
    st.at(2);
    // ---- hidden units: + (W0 wb) lin(window) + the slot's constant, Mish (MLPIO activation) ---------------------------------------
    if (hid_lane) {
      float pre = p_cur;
      if (fsb == 1) {
        pre = fmaf((((float)(every_wave_picks ? cur_cls : s_win[0]) / a.class_size) - .5f) * 2.f, a_c0, pre);   // Linearizer, modules/io.py:106-112
      } else {
        for (int i = 0; i < fsb; ++i) pre = fmaf((((float)s_win[i] / a.class_size) - .5f) * 2.f, acs[i * kHmMax + hid_u], pre);
      }
      hid[hid_at] = mish_fast(pre + bcs[slot * kHmMax + hid_u]);
    }
    __syncthreads();
    st.at(3);
    // ---- fc2; the next step's row is asked for meanwhile -------------------------------------------------------------------------
    if (hid_lane && s + 1 < a.n_steps && (t + 1) % S != 0) p_g = res_gload(p_row + (int)((t + 1) % S) * Hm);
    fc2_phase();
    __syncthreads();
    st.at(4);
    sampler_phase(s, t);
    if (!every_wave_picks) __syncthreads();
    st.at(5);
  }
  st.end(a.n_steps);
}

template <int KC, bool LSTM>
__global__ __launch_bounds__(kResThreads) void srnn_resident_kernel(const SrnnResArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int b = blockIdx.x;
  if (b < a.B) {
    if (!a.teacher) res_bottom_role<KC>(a, smem_raw);      // (teacher-forced: the tiers only - the warm-up drops the head's outputs)
    return;
  }
  int ti = 0;
#pragma unroll
  for (int i = 1; i < kResMaxTiers; ++i)
    if (i < a.n_tiers && b >= a.tier[i].block0) ti = i;
  const bool last = ti == a.n_tiers - 1;
  const int mt = a.tier[ti].mt;           // row tiles of 16 clips per workgroup: the tiers with time to spare take more, on fewer CUs
  if (ti == 0) {
    if (last) {
      if (mt == 1) res_tier_role<KC, LSTM, 1, false, true>(a, a.tier[0], 0, smem_raw);
      else res_tier_role<KC, LSTM, 2, false, true>(a, a.tier[0], 0, smem_raw);
    } else {
      if (mt == 1) res_tier_role<KC, LSTM, 1, false, false>(a, a.tier[0], 0, smem_raw);
      else if (mt == 2) res_tier_role<KC, LSTM, 2, false, false>(a, a.tier[0], 0, smem_raw);
      else res_tier_role<KC, LSTM, 4, false, false>(a, a.tier[0], 0, smem_raw);
    }
  } else if (last) {
    if (mt == 1) res_tier_role<KC, LSTM, 1, true, true>(a, a.tier[ti], ti, smem_raw);
    else res_tier_role<KC, LSTM, 2, true, true>(a, a.tier[ti], ti, smem_raw);
  } else {
    if (mt == 1) res_tier_role<KC, LSTM, 1, true, false>(a, a.tier[ti], ti, smem_raw);
    else res_tier_role<KC, LSTM, 2, true, false>(a, a.tier[ti], ti, smem_raw);
  }
}

// ---- host side -------------------------------------------------------------------------------------------------------------
static int res_cu_count() {
  static const int n_cu = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    return n;
  }();
  return n_cu;
}

bool srnn_resident_supported(int H, bool lstm, int Hm, int n_out, int Q, int fsb, int S) {
  if (!(H == 128 || H == 256 || H == 512)) return false;
  if (lstm && H == 512) return false;         // (four gates of both matrices + the state leave no registers for the products' operands)
  if (Hm < 16 || Hm > 128 || Hm % 16) return false;
  if (fsb < 1 || fsb > 16 || S < 2 || S > 64) return false;
  return n_out <= 1024 && Q <= 256 && n_out >= Q;
}

// Every workgroup of the launch waits for others: they must all be resident at once, one per CU (`spare_cus` are left to whatever else
// runs).  A tier's workgroup owns 16 units x 16 mt clips: the tiers start at mt = 1 and, while the launch does not fit, the topmost tier
// that can still grow doubles its row tiles (it updates least often: its products have the most time; the top tier up to 4 - unless it is
// also the last recurrent tier, whose role exists for 1 and 2 row tiles -, the others 2).
int srnn_resident_grid(int H, int B, int n_tiers, int spare_cus, int* mt_out) {
  const int n_cu = res_cu_count() - spare_cus, KC = H / 16;
  int mt[kResMaxTiers];
  for (int i = 0; i < n_tiers; ++i) mt[i] = 1;
  for (;;) {
    int grid = B;
    for (int i = 0; i < n_tiers; ++i) grid += KC * ((B + 16 * mt[i] - 1) / (16 * mt[i]));
    if (grid <= n_cu) {
      for (int i = 0; i < n_tiers && mt_out; ++i) mt_out[i] = mt[i];
      return grid;
    }
    int grow = -1;
    for (int i = 0; i < n_tiers && grow < 0; ++i)
      if (mt[i] < ((i == 0 && n_tiers > 1) ? 4 : 2) && 16 * mt[i] < B) grow = i;     // (a LAST tier is instantiated for 1 and 2 row tiles only)
    if (grow < 0) return 0;
    mt[grow] *= 2;
  }
}

size_t srnn_resident_lds_bytes(const SrnnResArgs& a) {
  const int NG = a.lstm ? 4 : 3;
  size_t tier = 0;
  for (int i = 0; i < a.n_tiers; ++i) {
    const int mt = a.tier[i].mt;
    const size_t b = (size_t)NG * mt * kResWaves * 64 * 16 + (size_t)16 * mt * (a.tier[i].fsp + 4) * 4 + (size_t)NG * 16 * a.tier[i].fsp * 4;
    tier = b > tier ? b : tier;
  }
  const int n_extra = a.n_out > 256 ? a.n_out - 256 : 0;
  const size_t bottom = (size_t)16 * (a.H / 16 + 4) * 4 + (size_t)16 * 12 * 4 + 1024 * 4 + 16 * 4 + (size_t)a.S * 128 * 4 + 16 * 128 * 4 + (size_t)n_extra * a.Hm * 4 + 64;
  size_t lds = tier > bottom ? tier : bottom;
  if (lds < 81 * 1024) lds = 81 * 1024;      // one workgroup per CU, whatever the registers would allow
  return lds;
}

int launch_srnn_resident(const SrnnResArgs& a, hipStream_t stream) {
  const bool lstm = a.lstm != 0;
  if (!srnn_resident_supported(a.H, lstm, a.Hm, a.n_out, a.Q, a.fsb, a.S)) return fail(MMK_ERR_UNSUPPORTED, "srnn resident kernel: geometry H=%d Hm=%d", a.H, a.Hm);
  if (a.n_tiers < 1 || a.n_tiers > kResMaxTiers) return fail(MMK_ERR_INVALID, "srnn resident kernel: %d recurrent tiers", a.n_tiers);
  const int KC = a.H / 16;
  int grid = a.B;
  for (int i = 0; i < a.n_tiers; ++i) {
    const int mt = a.tier[i].mt;
    if (!(mt == 1 || mt == 2 || (mt == 4 && i == 0 && a.n_tiers > 1))) return fail(MMK_ERR_INVALID, "srnn resident kernel: %d row tiles per workgroup of tier %d", mt, i);
    if (a.tier[i].block0 != grid) return fail(MMK_ERR_INVALID, "srnn resident kernel: tier %d starts at workgroup %d, %d expected", i, a.tier[i].block0, grid);
    grid += KC * ((a.B + 16 * mt - 1) / (16 * mt));
  }
  if (grid > res_cu_count()) return fail(MMK_ERR_INVALID, "srnn resident kernel: %d workgroups on %d CUs", grid, res_cu_count());
  const size_t lds = srnn_resident_lds_bytes(a);
  if (lds > 160 * 1024) return fail(MMK_ERR_UNSUPPORTED, "srnn resident kernel: %zu bytes of LDS", lds);
  dim3 g(grid), block(kResThreads);
#define MMK_RES(KC_)                                                                                   \
  do {                                                                                                 \
    if constexpr (KC_ < 32) {                                                                          \
      if (lstm) hipLaunchKernelGGL((srnn_resident_kernel<KC_, true>), g, block, lds, stream, a);        \
    }                                                                                                  \
    if (!lstm) hipLaunchKernelGGL((srnn_resident_kernel<KC_, false>), g, block, lds, stream, a);       \
  } while (0)
  switch (a.H) {
    case 128: MMK_RES(8); break;
    case 256: MMK_RES(16); break;
    default: MMK_RES(32); break;
  }
#undef MMK_RES
  MMK_HIP(hipGetLastError());
  return MMK_OK;
}

}  // namespace mmk

// XCD-pipelined, weight-stationary persistent WaveNet step kernel (gfx950).
//
// Why: both other persistent kernels give every XCD a clip group and make it walk ALL layers, so every XCD pulls the
// whole weight set through its fabric link once per step.  That link delivers ~0.9 TB/s per XCD (scripts/probes/
// stream_bw.hip: 62 MB in 70 us with nothing else going on), which alone is more than a step should take at
// BASELINE config 4 (30 x 256 channels: 45 MB per XCD and step for wavenet_persist.hip, 62 MB with the pre-multiplied
// matrices of wavenet_chain.hip), and the weight stream shares each CU's in-order memory pipe with the hand-off polls.
//
// Here the LAYERS are spread over the XCDs instead of the clips: stage x (= one XCD, verified from HW_REG_XCC_ID) owns
// iterations [x n_it, (x+1) n_it) of wavenet_chain.hip's one-hand-off-per-layer step, n_it <= 4, and keeps their
// weight tiles ON CHIP for the whole launch - three iterations' tiles in the matrix waves' registers, a fourth in LDS
// (256 KiB per workgroup at C = 256) - so no weight byte moves during generation.  The clip groups (<= 8 groups of
// <= 4 clips) travel through the stages like through a ring pipeline: stage x processes "visits" (step s, group g) in
// order, hands y / h / the running skip sums of the group to stage x + 1 with agent-scope granules, the last stage
// runs the head and hands the sampled classes back to stage 0.  With 8 groups in flight all 8 XCDs work at once; a
// step of a group is L + 1 iterations with an XCD-local exchange each, plus one cross-XCD exchange per stage.
//
// Inside a visit an iteration is wavenet_chain.hip's (same work split: 4 I/O waves, 8 matrix waves in the pairs
// h[t-d] | h[t] | y | [res ; skip], 4x4 MFMA blocks, same fixed-order partial sums, same epilogues), two workgroup
// barriers each:  B1 partial sums in LDS | B4 next operands in LDS.  History rings are private per workgroup and hold
// the layer inputs of ALL clip groups for the stage's own layers; the input of a stage's LAST layer comes back to the
// stage through a stage-local exchange buffer and joins its ring at the start of the next visit.
#include "wavenet_pipe.h"
#include "wavenet_handoff.h"
#include "sampler256.h"

namespace mmk {

constexpr int kPiWaves = 8;       // pairs: K segment h[t-d] | h[t] | y of the gate product, [res ; skip] product
constexpr int kPiThreads = 64 * kPiWaves;
constexpr int kPiMaxIt = 4;       // iterations per stage
constexpr int kPiRegSlots = 3;    // of which this many keep their weight tiles in the waves' registers, the rest in LDS

struct __attribute__((aligned(16))) PiEntry {
  unsigned ring_off;       // byte offset of the layer's history ring inside the workgroup's block
  unsigned dil, mask;
  unsigned flags;          // 1: gate product, 2: this workgroup has [res ; skip] rows in the iteration
};

// y and h granules of one exchange in one pass (wavenet_chain.hip's sweep): the threads of the first NTH / 64 waves keep
// four 16-byte loads in flight per round; ends with the workgroup barrier
template <int NTH>
__device__ __forceinline__ bool pipe_sweep_pair(const u64* gy, const u64* gh, bool with_h, int count, unsigned epoch, float* dy,
                                                float* dh, int* err_flag, int* s_fail) {
  const int tid = threadIdx.x;
  typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
  if (tid < NTH && tid * 4 < count) {
    const u64* py = gy + tid * 4;
    const u64* ph = with_h ? gh + tid * 4 : py;
    u32x4v y0, y1, h0, h1;
    unsigned spins = 0;
    bool ok = true;
    for (;;) {
      asm volatile(
          "global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %4, off offset:16 sc1\n\t"
          "global_load_dwordx4 %2, %5, off sc1\n\tglobal_load_dwordx4 %3, %5, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
          : "=&v"(y0), "=&v"(y1), "=&v"(h0), "=&v"(h1)
          : "v"(py), "v"(ph)
          : "memory");
      const bool all = y0[1] == epoch && y0[3] == epoch && y1[1] == epoch && y1[3] == epoch && h0[1] == epoch &&
                       h0[3] == epoch && h1[1] == epoch && h1[3] == epoch;
      if (all) break;
      ++spins;
      if (spins > kSpinLimit || ((spins & 255u) == 0 && __hip_atomic_load(err_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
        ok = false;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    if (!ok) {
      *s_fail = 1;
      atomicExch(err_flag, 1);
    }
    *reinterpret_cast<f32x4*>(dy) = f32x4{__uint_as_float(y0[0]), __uint_as_float(y0[2]), __uint_as_float(y1[0]), __uint_as_float(y1[2])};
    if (with_h)
      *reinterpret_cast<f32x4*>(dh) = f32x4{__uint_as_float(h0[0]), __uint_as_float(h0[2]), __uint_as_float(h1[0]), __uint_as_float(h1[2])};
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  return *s_fail == 0;
}

// Keeps a per-lane value opaque to loop-invariant code motion: with four unrolled iterations the compiler otherwise
// precomputes every LDS / granule address of every iteration before the visit loop (~90 VGPRs of loop invariants next
// to the resident weight tiles); recomputing them from one opaque offset costs a few VALU instructions per iteration.
__device__ __forceinline__ int opaque(int v) {
  asm volatile("" : "+v"(v));
  return v;
}

// four granules of one thread (no LDS, no barrier): returns the values, false after a timeout
__device__ __forceinline__ bool poll4(const u64* gp, unsigned epoch, f32x4& out, int* err_flag, int* s_fail) {
  typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
  u32x4v lo, hi;
  unsigned spins = 0;
  for (;;) {
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(lo), "=&v"(hi)
                 : "v"(gp)
                 : "memory");
    if (lo[1] == epoch && lo[3] == epoch && hi[1] == epoch && hi[3] == epoch) break;
    ++spins;
    if (spins > kSpinLimit || ((spins & 255u) == 0 && __hip_atomic_load(err_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
      *s_fail = 1;
      atomicExch(err_flag, 1);
      return false;
    }
    __builtin_amdgcn_s_sleep(1);
  }
  out = f32x4{__uint_as_float(lo[0]), __uint_as_float(lo[2]), __uint_as_float(hi[0]), __uint_as_float(hi[2])};
  return true;
}

// SAMPLER: 0 = greedy decode only, 1 = sampled decode only, 2 = decided at run time (the diagnostic build).  Two product kernels
// rather than one with both paths: this kernel has no register to spare (256 VGPRs, ~190 spilled SGPRs), and the sampled path
// compiled into the greedy kernel cost the greedy decode 13 % (363 instead of 420 k samples/s, measured).
template <int KC, int NIT, bool STAMPS, int SAMPLER>
__global__ __launch_bounds__(kPiThreads) void wavenet_pipe_kernel(const WnPipeArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr int CPW = KC / 2;          // K-chunks per wave: a pair of waves covers one K = C segment
  constexpr int NT = kPiThreads, nw = kPiWaves;
  constexpr int C = 16 * KC;
  constexpr int ldh = C + 4;
  constexpr int kRows = 4;
  constexpr int NREG = NIT < kPiRegSlots ? NIT : kPiRegSlots;
  constexpr bool kLdsSlot = NIT > kPiRegSlots;
  static_assert(NIT >= 1 && NIT <= kPiMaxIt, "iterations per stage");
  static_assert(kRows * (C / 4) <= NT / 2, "one ring piece / four granules per thread");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = a.L;
  const int n_stages = (L + 1 + NIT - 1) / NIT;
  int stage, j;
  {   // stage = the XCD this workgroup runs on, owner index = arrival order there; placement is verified, never assumed
    int* role = reinterpret_cast<int*>(smem_raw);
    if (tid == 0) {
      unsigned xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      xcc &= 0xfu;
      const unsigned mine = atomicAdd(&a.xcd_count[xcc & 7u], 1u);
      atomicAdd(&a.xcd_count[8], 1u);
      unsigned spins = 0;
      bool ok = xcc < 8u;
      while (__hip_atomic_load(&a.xcd_count[8], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
        if (++spins > kSpinLimit) { ok = false; break; }
        __builtin_amdgcn_s_sleep(2);
      }
      for (int x = 0; x < 8 && ok; ++x)
        ok = __hip_atomic_load(&a.xcd_count[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)a.Gn;
      if (!ok) atomicExch(a.err_flag, 2);
      role[0] = ok ? (int)xcc : -1;
      role[1] = (int)mine;
    }
    __syncthreads();
    stage = __builtin_amdgcn_readfirstlane(role[0]);
    j = __builtin_amdgcn_readfirstlane(role[1]);
    __syncthreads();
    if (stage < 0 || stage >= n_stages) return;
  }
  const int i0 = stage * NIT;                              // first iteration of the stage
  const int n_loc = min(NIT, L + 1 - i0);                  // its iterations
  const bool head_stage = (i0 + n_loc == L + 1);
  const bool first_stage = (stage == 0);

  // ---- LDS carve ------------------------------------------------------------------------------------
  const int wide = max(C, a.H1);
  const int ldy = wide + 4, ldl = a.n_logits_pad + 4;
  char* sp = smem_raw;
  float* hbuf = (float*)sp;   sp += 2 * kRows * ldh * 4;          // h_{i-1}[tau] / h_i[tau], by iteration parity
  float* hprev = (float*)sp;  sp += kRows * ldh * 4;              // h_i[tau - d_i]
  float* ybuf = (float*)sp;   sp += 2 * kRows * ldh * 4;          // y_{i-1} / y_i
  f32x4* red = (f32x4*)sp;    sp += nw * 64 * 16;                 // partial sums [wave][64]
  PiEntry* tab = (PiEntry*)sp;            sp += kPiMaxIt * 16;
  float* biasA = (float*)sp;              sp += kPiMaxIt * 16 * 4;
  float* biasB = (float*)sp;              sp += kPiMaxIt * 16 * 4;
  float* cndbuf = (float*)sp;             sp += 2 * 64 * 4;
  int* s_idx = (int*)sp;      sp += 16 * 4;
  int* s_fail = (int*)sp;     sp += 16;
  f32x4* wlds = (f32x4*)sp;   sp += kLdsSlot ? (size_t)nw * CPW * 64 * 16 : 0;   // the fourth iteration's tiles: [wave][fragment][lane]
  float* headbuf = (float*)sp; sp += 16 * ldy * 4;                // head stage only from here on
  float* lbuf = (float*)sp;   sp += 16 * ldl * 4;
  const int t_fc2 = a.n_logits_pad / 16, kc_fc2 = a.H1 / 16;
  const int nt2 = (head_stage && j < t_fc2) ? (t_fc2 - j + a.Gn - 1) / a.Gn : 0;
  f32x4* hw2 = (f32x4*)sp;    sp += (size_t)((t_fc2 + a.Gn - 1) / a.Gn) * kc_fc2 * 1024;
  float* hb2 = (float*)sp;

  const int D_q = lane >> 4, D_n = lane & 15;
  const bool owns_res = j < KC;                            // owners [0, C/16): residual rows of the [res ; hidden] matrix
  const bool owns_hid = j >= KC && j < KC + a.H1 / 16;     // the next H1/16: rows of fc0 . W_skip (the head's first Linear folded in)
  const bool has_cond = a.C1 > 0;

  for (int i = tid; i < 2 * kRows * ldh; i += NT) hbuf[i] = 0.f;
  for (int i = tid; i < kRows * ldh; i += NT) hprev[i] = 0.f;
  for (int i = tid; i < 2 * kRows * ldh; i += NT) ybuf[i] = 0.f;
  for (int i = tid; i < 128; i += NT) cndbuf[i] = 0.f;
  if (head_stage) {
    for (int i = tid; i < 16 * ldy; i += NT) headbuf[i] = 0.f;
    for (int i = tid; i < 16 * ldl; i += NT) lbuf[i] = 0.f;
  }
  if (tid < n_loc) {
    const int i = i0 + tid;
    const WnChainIter t = a.iters[i < L ? i : 0];
    const bool has_b = i >= 1 && (owns_res ? a.iters[i].prev_has_res != 0 : owns_hid);
    PiEntry e;
    e.ring_off = (unsigned)(t.ring_offset * 4);
    e.dil = (unsigned)t.dil; e.mask = (unsigned)t.ring_mask;
    e.flags = (i < L ? 1u : 0u) | (has_b ? 2u : 0u);
    tab[tid] = e;
  }
  for (int q = tid; q < n_loc * 16; q += NT) {
    const int i = i0 + (q >> 4), n = q & 15;
    const WnChainIter t = a.iters[i];
    biasA[q] = (i < L && t.A_bias) ? t.A_bias[j * 16 + n] : 0.f;
    const bool has_b = i >= 1 && (owns_res ? t.prev_has_res != 0 : owns_hid);
    const int btile = owns_res ? j : (j - KC + (t.prev_has_res ? KC : 0));
    biasB[q] = (has_b && t.B_bias) ? t.B_bias[btile * 16 + n] : 0.f;
  }
  for (int i = 0; i < nt2; ++i) {
    const f32x4* src = reinterpret_cast<const f32x4*>(a.fc2_wp) + (int64_t)(j + i * a.Gn) * kc_fc2 * 64;
    for (int q = tid; q < kc_fc2 * 64; q += NT) hw2[i * kc_fc2 * 64 + q] = src[q];
    if (tid < 16) hb2[i * 16 + tid] = a.fc2_bias[(j + i * a.Gn) * 16 + tid];
  }
  if (tid == 0) *s_fail = 0;

  // ---- the stage's weight tiles: in registers for the whole launch -----------------------------------
  const int role = wave >> 1, half = wave & 1;            // role 0: h[t-d], 1: h[t], 2: y, 3: [res ; skip]
  const int sm_k0 = ((lane >> 2) & 3) * 4 * CPW;          // first k of the lane's K sub-slice inside the wave's slice
  const int sm_n = (lane >> 4) * 4 + (lane & 3);          // its column inside the 16-column tile
  f32x4 wreg[NREG][CPW];
  {
    const int cbase = (role < 3 ? role * KC : 0) + half * CPW;
    const unsigned w_voff = (unsigned)(cbase * 64 + (sm_k0 / 4) * 16 + sm_n) * 16u;
#pragma unroll
    for (int q = 0; q < NIT; ++q) {
      // iterations without the product (i == L: no gate; i == 0 or a last layer's residual rows: no [res ; skip]) load any
      // valid tile: the result is never used
      const int i = min(i0 + q, L);
      const WnChainIter ta = a.iters[i < L ? i : 0];
      const bool has_b = i >= 1 && (owns_res ? a.iters[i].prev_has_res != 0 : owns_hid);
      const WnChainIter tb = a.iters[has_b ? i : 1];
      const int btile = owns_res ? j : ((owns_hid ? j - KC : 0) + (tb.prev_has_res ? KC : 0));
      const char* tile = role < 3 ? (const char*)(ta.A_wp + (int64_t)j * (3 * KC) * 256) : (const char*)(tb.B_wp + (int64_t)btile * KC * 256);
      const f32x4* src = reinterpret_cast<const f32x4*>(tile + w_voff);
      if (q < NREG) {
#pragma unroll
        for (int u = 0; u < CPW; ++u) wreg[q < NREG ? q : 0][u] = src[u * 16];
      } else {
#pragma unroll
        for (int u = 0; u < CPW; ++u) wlds[(wave * CPW + u) * 64 + lane] = src[u * 16];
      }
    }
  }
  __syncthreads();

  // ---- addressing ------------------------------------------------------------------------------------
  const int Bp = a.Gc * a.Mg;                              // clips a ring slot holds
  char* h_ring = (char*)(a.h_rings + (int64_t)(stage * a.Gn + j) * a.ring_floats_per_wg);
  const unsigned slot_bytes = (unsigned)Bp * C * 4;
  const unsigned group_bytes = (unsigned)a.Mg * C * 4;
  int* err = a.err_flag;
  const int e_m = lane >> 4, e_n = lane & 15;              // epilogue element of a lane (waves 0 and 1)
  const int frag = ((e_n >> 2) * 16 + 12 + (e_n & 3)) * 4 + e_m;
  const float gate_k = (e_n & 1) ? 1.f : 2.f;
  const float gate_scale = -gate_k * 1.4426950408889634f;
  const float gate_shift = 1.f - gate_k;
  const unsigned y_slot = (unsigned)(e_m * C + j * 8 + (e_n >> 1));
  const unsigned h_slot = (unsigned)(e_m * C + j * 16 + e_n);
  const int res_off = e_m * ldh + j * 16 + e_n;
  // Two kinds of waves besides the MFMA role every wave has.  vmcnt retires in order and counts stores too: a wave that
  // both stores (publishes, ring) and loads into registers ends up waiting for its store acknowledgements whenever the
  // compiler has to make sure an older load has landed.  So waves 0..3 ("I/O") publish, store and poll (the polls wait
  // for everything anyway), and waves 4..7 ("loaders") only ever load: the next iteration's delayed input and
  // conditioning terms.
  constexpr int NIO = NT / 2;
  const bool is_loader = tid >= NIO;
  const int lt = tid - NIO;                                // loader thread index
  const int sw_row = (tid * 4) / C, sw_col = tid * 4 - sw_row * C;    // an I/O thread's four granules in a sweep
  const int sw_off = sw_row * ldh + sw_col;
  const int x_off = (lane & 3) * ldh + half * CPW * 16 + sm_k0;
  const int64_t cond_clip = (int64_t)a.cond_steps * L * (2 * C);
  const int64_t tau0 = a.t0 - 1;
  const int64_t n_visits = a.n_steps * a.Gc;

  unsigned long long st_acc[18] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long st_prev = 0;
  auto stamp = [&](int slot) {
    if (STAMPS) {
      const unsigned long long now = wall_clock64();
      st_acc[slot] += now - st_prev;
      st_prev = now;
    }
  };
  const unsigned long long clk_start = STAMPS ? clock64() : 0, wall_start = STAMPS ? wall_clock64() : 0;

  auto sweep_yh = [&](const u64* gy, const u64* gh, bool with_h, int count, unsigned epoch, float* dy, float* dh) -> bool {
    const int so = opaque(sw_off);
    return pipe_sweep_pair<NIO>(gy, gh, with_h, count, epoch, dy + so, dh + so, err, s_fail);
  };

  // ---- head (wavenet_persist.hip's): the head stage's workgroups, every wave --------------------------
  auto head = [&](int64_t s, int64_t tau, int g, int mg, float skipacc) -> bool {
    const int m_first = g * a.Mg;
    u64* gran_hid = a.gran_hid + (int64_t)g * 16 * a.H1;
    u64* gran_logit = a.gran_logit + (int64_t)g * 16 * a.n_logits_pad;
    u64* gran_idx = a.gran_idx + (int64_t)g * 16;
    const unsigned he = (unsigned)(s + 1);
    const bool greedy = SAMPLER == 0 ? true : (SAMPLER == 1 ? false : a.temperature == nullptr);
    // sampled decode: this wave's clip's uniform and temperature are requested now - the draw at the end of the head would wait a
    // memory round trip for them on every sample's chain (the uniforms are streamed once, never cached)
    float u_pre = 0.f, T_pre = 1.f;
    if (SAMPLER != 0 && !greedy && j == 0 && wave < mg) {
      u_pre = a.uniforms[(int64_t)(m_first + wave) * a.uni_ld + s];
      T_pre = a.temperature[m_first + wave];
    }
    // the hidden units: every layer's fc0 . W_skip product has been accumulated on the way (skipacc); + fc0's bias, Mish
    if (wave == 1 && e_m < mg && owns_hid)
      gran_store<true>(gran_hid + e_m * a.H1 + (j - KC) * 16 + e_n, he, mish_fast(skipacc + a.fc0_bias[(j - KC) * 16 + e_n]));
    stamp(10);   // head: hidden units published
    if (j < t_fc2) {
      if (!sweep<NT>(gran_hid, mg * a.H1, he, nullptr, headbuf, a.H1, ldy, err, s_fail)) return false;
      stamp(11);   // head: wait for the hidden units
      const int per = (kc_fc2 + nw - 1) / nw;
      const int k0 = min(wave * per, kc_fc2), k1 = min(k0 + per, kc_fc2);
      for (int t = j, ti = 0; t < t_fc2; t += a.Gn, ++ti) {
        f32x4 v = reduce_waves(tile_mma(headbuf, ldy, hw2 + ti * kc_fc2 * 64, 0, k0, k1, lane), red, wave, lane, nw);
        if (wave == 0) {
          const float bias = hb2[ti * 16 + D_n];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int m = 4 * D_q + r;
            if (m < mg) gran_store<true>(gran_logit + m * a.n_logits_pad + t * 16 + D_n, he, v[r] + bias);
          }
        }
      }
    }
    stamp(12);   // head: fc2 tile + publish
    if (j == 0) {
      if (!sweep<NT>(gran_logit, mg * a.n_logits_pad, he, nullptr, lbuf, a.n_logits_pad, ldl, err, s_fail)) return false;
      stamp(13);   // head: wait for the logits
      const int nc = a.n_classes;
      const int per = (nc + 63) / 64;
      for (int m = wave; m < mg; m += nw) {
        const float* lg = lbuf + m * ldl;
        const int clip = m_first + m;
        const bool keep_logits = a.logits_out && s + 1 == a.n_steps;
        float denom = 1.f;
        if (a.learn_temp) denom = fmaxf(sigmoidf_(lg[nc]), a.min_temp);   // mlp.py:60-62
        int result;
        if (greedy) {
          float best = -INFINITY;
          int bi = 0x7fffffff;
          if (nc == 256) {   // four classes per lane in one 16-byte LDS read (this sits on every sample's chain)
            const f32x4 v4 = *reinterpret_cast<const f32x4*>(lg + lane * 4);
            const float inv_off = a.learn_temp ? denom : 1.f;
            float vv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) vv[q] = a.learn_temp ? v4[q] / inv_off : v4[q];
            best = vv[0]; bi = lane * 4;
#pragma unroll
            for (int q = 1; q < 4; ++q)
              if (vv[q] > best) { best = vv[q]; bi = lane * 4 + q; }
          } else
          for (int q = 0; q < per; ++q) {
            const int c = lane * per + q;
            if (c < nc) {
              const float v = a.learn_temp ? lg[c] / denom : lg[c];
              if (v > best || bi == 0x7fffffff) { best = v; bi = c; }
            }
          }
          auto take = [&](float ob, int oi) {     // first maximum wins (torch.argmax)
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
          };
#define MMK_DPP_STEP(CTRL)                                                                                           \
          take(__int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(best), CTRL, 0xf, 0xf, false)),            \
               __builtin_amdgcn_update_dpp(0, bi, CTRL, 0xf, 0xf, false))
          MMK_DPP_STEP(0xB1);
          MMK_DPP_STEP(0x4E);
          MMK_DPP_STEP(0x141);
          MMK_DPP_STEP(0x140);
#undef MMK_DPP_STEP
#pragma unroll
          for (int o = 16; o <= 32; o <<= 1) {
            const float ob = __shfl_xor(best, o);
            const int oi = __shfl_xor(bi, o);
            take(ob, oi);
          }
          result = bi;
        } else if (nc == 256) {
          result = sample_256(lg, a.learn_temp != 0, denom, m == wave ? T_pre : a.temperature[clip],
                              m == wave ? u_pre : a.uniforms[(int64_t)clip * a.uni_ld + s], lane);
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
          const float target = a.uniforms[(int64_t)clip * a.uni_ld + s] * total;
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
        if (lane == 0) {
          gran_store_u32<false>(gran_idx + m, he, (unsigned)result);   // to stage 0 (another XCD): agent scope
          a.idx[(int64_t)clip * a.idx_rs + tau + 1] = result;
        }
        if (keep_logits)
          for (int c = lane; c < nc + a.learn_temp; c += 64) a.logits_out[(int64_t)clip * a.logits_ld + c] = lg[c];
      }
      stamp(8);    // head: sampler
    }
    __syncthreads();
    return true;
  };


  // does the stage's last iteration produce a layer input (h) that the stage's own last layer must keep in its ring?
  const int i_last = i0 + n_loc - 1;
  const bool own_h = (i_last >= 1 && i_last <= L - 1) && !head_stage;
  const bool defer_own = a.Gc > 1;                         // (one group: its next visit reads that ring right away)
  float skipacc = 0.f;                                     // wave 1 of the skip-row owners
  int pend_g = -1;                                         // visit whose own h still has to join the ring
  int64_t pend_tau = 0;
  unsigned pend_epoch = 0;
  auto own_h_to_ring = [&](int pg, int64_t ptau, unsigned pepoch) -> bool {
    const int pm = min(a.Mg, a.B - pg * a.Mg);
    if (tid * 4 < pm * C) {
      f32x4 v;
      if (!poll4(a.gran_hown + (int64_t)(stage * a.Gc + pg) * 16 * C + tid * 4, pepoch, v, err, s_fail)) return false;
      const PiEntry e = tab[n_loc - 1];
      gf32x4_wptr dst = (gf32x4_wptr)(uintptr_t)(h_ring + e.ring_off + (u64)((unsigned)ptau & e.mask) * slot_bytes + (u64)pg * group_bytes);
      dst[tid] = v;
    }
    return true;
  };
  // delayed input and conditioning terms of iteration q of visit (s, g): requested one iteration ahead into registers
  f32x4 hp = f32x4{0.f, 0.f, 0.f, 0.f};
  float cnd = 0.f;
  // Where the request AFTER the next one reads: worked out in a loader wave's idle time (between B1 and B4), so that the top
  // of an iteration only issues two loads from ready-made addresses (the table look-up, the 64-bit address arithmetic and
  // their waits otherwise sit in front of the loader waves' MFMAs, and every other wave waits for them at B1).
  gf32x4_ptr nx_src = (gf32x4_ptr)(uintptr_t)a.zeros;
  gcfloat_ptr nx_cp = (gcfloat_ptr)(uintptr_t)a.zeros;
  auto compute_addr = [&](int q, int64_t s, int g) {
    const int m_first = g * a.Mg;
    const int mg = min(a.Mg, a.B - m_first);
    const int slot_f4 = mg * (C / 4);
    const int i = i0 + q;
    const PiEntry e = tab[q];
    const unsigned ring_off = sgpr(e.ring_off), dil = sgpr(e.dil), mask = sgpr(e.mask);
    const unsigned ntau = (unsigned)(tau0 + s);
    gf32x4_ptr src = (gf32x4_ptr)(uintptr_t)(h_ring + ring_off + (u64)((ntau - dil) & mask) * slot_bytes + (u64)g * group_bytes);
    nx_src = src + max(0, min(lt, slot_f4 - 1));
    const bool c_real = has_cond && i < L && lt >= 0 && lt < mg * 16;
    nx_cp = c_real ? (gcfloat_ptr)(uintptr_t)(a.condall + (int64_t)(m_first + (lt >> 4)) * cond_clip + (s * L + i) * (int64_t)(2 * C) +
                                              j * 16 + (lt & 15))
                   : (gcfloat_ptr)(uintptr_t)a.zeros;
  };
  auto issue_request = [&]() {
    hp = *nx_src;
    cnd = *nx_cp;
  };
  auto small_to_lds = [&](int cpar, int g) {
    const int mg = min(a.Mg, a.B - g * a.Mg);
    const int lq = opaque(lt);
    if (lq < mg * (C / 4)) *reinterpret_cast<f32x4*>(hprev + (lq / (C / 4)) * ldh + (lq % (C / 4)) * 4) = hp;
    if (lq < mg * 16) cndbuf[cpar * 64 + lq] = cnd;
  };
  if (is_loader) {
    compute_addr(0, 0, 0);
    issue_request();
    small_to_lds(0, 0);
    // the request issued at the top of the very first iteration is for the one after it
    if (n_loc > 1) compute_addr(1, 0, 0);
    else compute_addr(0, (a.Gc > 1 || n_visits < 2) ? 0 : 1, (a.Gc > 1 && n_visits > 1) ? 1 : 0);
  }

  for (int64_t v = 0; v < n_visits; ++v) {
    const int64_t s = v / a.Gc;
    const int g = (int)(v - s * a.Gc);
    const int64_t vn = v + 1;
    const int64_t sn = vn < n_visits ? vn / a.Gc : s;      // the visit after this one (past the end: this one again, unused)
    const int gn = vn < n_visits ? (int)(vn - sn * a.Gc) : g;
    const int64_t vn2 = v + 2;
    const int64_t sn2 = vn2 < n_visits ? vn2 / a.Gc : sn;  // and the one after that
    const int gn2 = vn2 < n_visits ? (int)(vn2 - sn2 * a.Gc) : gn;
    const int m_first = g * a.Mg;
    const int mg = min(a.Mg, a.B - m_first);
    const int64_t tau = tau0 + s;
    const unsigned tau_u = (unsigned)tau;
    const bool elem = e_m < mg;
    const int slot_f4 = mg * (C / 4);
    u64* gran_hl0 = a.gran_hl + (int64_t)(stage * a.Gc + g) * 2 * 16 * C;     // inside the stage, by iteration parity
    u64* gran_yl0 = a.gran_yl + (int64_t)(stage * a.Gc + g) * 2 * 16 * C;
    u64* gran_hx_out = a.gran_hx + (int64_t)(g * 2 + (stage & 1)) * 16 * C;     // to the next stage
    u64* gran_yx_out = a.gran_yx + (int64_t)(g * 2 + (stage & 1)) * 16 * C;
    u64* gran_sk_out = a.gran_skipfwd + (int64_t)(g * 2 + (stage & 1)) * 16 * C;
    const int pin = (stage + 1) & 1;                                            // (= (stage - 1) & 1: what the stage before wrote)
    u64* gran_hown = a.gran_hown + (int64_t)(stage * a.Gc + g) * 16 * C;
    if (STAMPS) st_prev = wall_clock64();
    // ---- the previous visit's last layer input joins its ring (exchange buffer of this stage only) ----
    if (pend_g >= 0) {
      if (!own_h_to_ring(pend_g, pend_tau, pend_epoch)) return;
      pend_g = -1;
    }
    // ---- inputs of the visit -----------------------------------------------------------------------
    if (first_stage) {
      if (s > 0) {
        if (tid < mg) {
          unsigned spins = 0;
          u64 w;
          for (;;) {
            w = gran_load(a.gran_idx + (int64_t)g * 16 + tid);
            if ((unsigned)(w >> 32) == (unsigned)s) break;
            ++spins;
            if (spins > kSpinLimit || ((spins & 255u) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
              *s_fail = 1;
              atomicExch(err, 1);
              break;
            }
            __builtin_amdgcn_s_sleep(1);
          }
          s_idx[tid] = (int)(unsigned)w;
        }
      } else {
        if (tid < mg) s_idx[tid] = (int)a.idx[(int64_t)(m_first + tid) * a.idx_rs + tau];
      }
      __syncthreads();                                       // P1
      if (*s_fail) return;
      {
        const PiEntry e = tab[0];
        gf32x4_wptr ring0 = (gf32x4_wptr)(uintptr_t)(h_ring + e.ring_off + (u64)(tau_u & e.mask) * slot_bytes + (u64)g * group_bytes);
        for (int q = tid; q < slot_f4; q += NT) {            // (slot_f4 <= 256: threads of the I/O waves)
          const int m = q / (C / 4), c = (q % (C / 4)) * 4;
          const int cls = s_idx[m];
          const float nanv = __builtin_nanf("");     // torch raises on an out-of-range class; stay memory-safe and visible
          const f32x4 val = (cls >= 0 && cls < a.q_levels) ? *reinterpret_cast<const f32x4*>(a.emb + (int64_t)cls * C + c)
                                                           : f32x4{nanv, nanv, nanv, nanv};
          *reinterpret_cast<f32x4*>(hbuf + m * ldh + c) = val;
          ring0[q] = val;
        }
      }
      __syncthreads();                                       // P2
    } else {
      // y_{i0-1}, h_{i0-1} of (s, g) from the stage before (another XCD)
      const int ip = i0 - 1;
      const unsigned ep = (unsigned)(s * (L + 1) + ip + 1);
      const int par = ip & 1;
      if (!sweep_yh(a.gran_yx + (int64_t)(g * 2 + pin) * 16 * C, a.gran_hx + (int64_t)(g * 2 + pin) * 16 * C, true, mg * C, ep,
                    ybuf + par * kRows * ldh, hbuf + par * kRows * ldh))
        return;                                              // ... P1
      if (wave == 1 && elem && owns_hid) {                   // the running sums of the hidden units' pre-activations come with them
        unsigned spins = 0;
        u64 w;
        const u64* gp = a.gran_skipfwd + (int64_t)(g * 2 + pin) * 16 * C + e_m * C + (j - KC) * 16 + e_n;
        for (;;) {
          w = gran_load(gp);
          if ((unsigned)(w >> 32) == ep) break;
          if (++spins > kSpinLimit) { *s_fail = 1; atomicExch(err, 1); break; }
          __builtin_amdgcn_s_sleep(1);
        }
        skipacc = __uint_as_float((unsigned)w);
      }
    }
    stamp(7);

#pragma unroll
    for (int q = 0; q < NIT; ++q) {
      if (q < n_loc) {
        const int i = i0 + q;
        const unsigned epoch = (unsigned)(s * (L + 1) + i + 1);
        const int par = i & 1;
        const int cpar = (int)((v * n_loc + q) & 1);
        const int hsel = i == 0 ? 0 : ((i - 1) & 1);         // where h_{i-1}[tau] lives (h_0 at i = 0)
        const unsigned flags = sgpr(tab[q].flags);
        const bool last_loc = (q == n_loc - 1);
        const bool cross = last_loc && !head_stage;          // the consumers of this iteration's outputs sit on another XCD
        u64* gran_y = gran_yl0 + par * 16 * C;
        u64* gran_h = gran_hl0 + par * 16 * C;
        // the next iteration's small operands (after the last one: the next visit's first iteration)
        // (measured on the product build, us per step: the two loads here 74.7; behind the wave's own operand reads 75.3; behind the
        //  MFMAs 75.7; in the idle time before B4, a whole iteration ahead, 79.9 - the exchange polls then queue behind them)
        if (is_loader) issue_request();                      // the next iteration's small operands (addresses ready-made)
        __builtin_amdgcn_sched_barrier(0);
        stamp(4);
        {
          const float* xsrc = (role == 0 ? hprev : (role == 1 ? hbuf + hsel * kRows * ldh : ybuf + ((i + 1) & 1) * kRows * ldh)) + opaque(x_off);
          f32x4 xv[CPW];
#pragma unroll
          for (int u = 0; u < CPW; ++u) xv[u] = *reinterpret_cast<const f32x4*>(xsrc + u * 4);
          __builtin_amdgcn_sched_barrier(0);
          if (STAMPS) {
            __builtin_amdgcn_s_waitcnt(0xC07F);              // lgkmcnt(0): the operands have arrived
            stamp(5);
          }
          f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int u = 0; u < CPW; ++u) {
            const f32x4 wv = q < NREG ? wreg[q < NREG ? q : 0][u] : wlds[(wave * CPW + u) * 64 + lane];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              if ((u * 4 + k) & 1) acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(xv[u][k], wv[k], acc1, 0, 0, 0);
              else acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(xv[u][k], wv[k], acc0, 0, 0, 0);
            }
          }
          f32x4 acc;
#pragma unroll
          for (int k = 0; k < 4; ++k) acc[k] = acc0[k] + acc1[k];
          red[opaque(tid)] = reduce_subslices(acc);
        }
        stamp(9);
        __syncthreads();                                     // B1: partial sums are in LDS
        stamp(0);
        if (wave == 0) {
          if ((flags & 1u) && elem) {
            const float* f = reinterpret_cast<const float*>(red) + opaque(frag);
            float pv[6];
#pragma unroll
            for (int w = 0; w < 6; ++w) pv[w] = f[w * 256];
            __builtin_amdgcn_sched_barrier(0);
            float acc = 0.f;
#pragma unroll
            for (int w = 0; w < 6; ++w) acc += pv[w];
            const float z = acc + cndbuf[cpar * 64 + opaque(lane)] + biasA[q * 16 + opaque(e_n)];
            const float act = fmaf(__frcp_rn(1.0f + __builtin_amdgcn_exp2f(z * gate_scale)), gate_k, gate_shift);
            // lane n takes lane n+1's value (row_shl:1): the even lane multiplies tanh(f) by its neighbour's sigmoid(g)
            const float other = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(act), 0x101, 0xf, 0xf, false));
            if (!(e_n & 1)) {
              const unsigned ys = (unsigned)opaque((int)y_slot);
              if (cross) gran_store<false>(gran_yx_out + ys, epoch, act * other);
              else gran_store<true>(gran_y + ys, epoch, act * other);
            }
          }
        } else if (wave == 1) {
          if ((flags & 2u) && elem) {
            const float* f = reinterpret_cast<const float*>(red) + opaque(frag);
            const float vb = (f[6 * 256] + f[7 * 256]) + biasB[q * 16 + opaque(e_n)];
            if (owns_res) {
              const float hn = hbuf[hsel * kRows * ldh + opaque(res_off)] + vb;     // h_i = h_{i-1} + R y + r
              const unsigned hs = (unsigned)opaque((int)h_slot);
              if (cross) {
                gran_store<false>(gran_hx_out + hs, epoch, hn);
                if (own_h) gran_store<true>(gran_hown + hs, epoch, hn);
              } else {
                gran_store<true>(gran_h + hs, epoch, hn);
              }
            } else {                                         // (flags & 2: a hidden-row owner)
              skipacc = (i == 1) ? vb : vb + skipacc;
            }
          }
          if (cross && elem && owns_hid) gran_store<false>(gran_sk_out + e_m * C + (j - KC) * 16 + e_n, epoch, skipacc);
          // (one iteration per stage: the embedding rows h_0 travel to the stage of layer 1 as every other layer input does)
          if (cross && i == 0 && elem && owns_res) gran_store<false>(gran_hx_out + h_slot, epoch, hbuf[res_off]);
        }
        stamp(1);
        // the next iteration's delayed input and conditioning terms -> LDS (hprev was last read before B1)
        if (is_loader) {
          small_to_lds((int)((v * n_loc + q + 1) & 1), last_loc ? gn : g);
          // address of the request the NEXT iteration's top will issue: for the iteration after that
          if (q + 2 < n_loc) compute_addr(q + 2, s, g);
          else if (q + 1 < n_loc) compute_addr(0, sn, gn);                 // next is the visit's last: then the next visit's first
          else if (n_loc > 1) compute_addr(1, sn, gn);                     // next is the next visit's first: then its second
          else compute_addr(0, sn2, gn2);                                  // one iteration per visit: the visit after the next
        }
        if (!is_loader && q >= 1 && i >= 2) {   // h_{i-1}[tau] (complete since the last sweep) joins the history ring of layer i-1
          const PiEntry ep = tab[q - 1];
          gf32x4_wptr dst = (gf32x4_wptr)(uintptr_t)(h_ring + sgpr(ep.ring_off) + (u64)(tau_u & sgpr(ep.mask)) * slot_bytes + (u64)g * group_bytes);
          const float* src = hbuf + hsel * kRows * ldh;
          const int tq = opaque(tid);
          if (tq < slot_f4) dst[tq] = *reinterpret_cast<const f32x4*>(src + (tq / (C / 4)) * ldh + (tq % (C / 4)) * 4);
        }
        stamp(2);
        if (!last_loc) {
          if (!sweep_yh(gran_y, gran_h, i >= 1, mg * C, epoch, ybuf + par * kRows * ldh, hbuf + par * kRows * ldh)) return;   // ... B4
        } else {
          __syncthreads();                                   // B4
        }
        stamp(3);
      }
    }
    if (own_h) {
      if (defer_own) {
        pend_g = g; pend_tau = tau; pend_epoch = (unsigned)(s * (L + 1) + i_last + 1);
      } else {
        if (!own_h_to_ring(g, tau, (unsigned)(s * (L + 1) + i_last + 1))) return;
      }
    }
    if (head_stage) {
      if (!head(s, tau, g, mg, skipacc)) return;
      stamp(6);
    }
  }
  if (pend_g >= 0) (void)own_h_to_ring(pend_g, pend_tau, pend_epoch);
  if (STAMPS && a.stamps && stage == a.stamp_stage && j == a.stamp_owner && tid == a.stamp_wave * 64) {
    st_acc[14] = clock64() - clk_start;
    st_acc[15] = wall_clock64() - wall_start;
    for (int i = 0; i < 16; ++i) a.stamps[i] = st_acc[i];
    a.stamps[16] = st_acc[16];
  }
}

int wn_pipe_iters_per_stage(int L) { return (L + 1 + 7) / 8; }

bool wn_pipe_supported(int C, int Mg, int Gc, int L) {
  return C >= 32 && C <= 256 && C % 32 == 0 && Mg >= 1 && Mg <= 4 && Gc >= 1 && Gc <= 8 && L >= 2 && wn_pipe_iters_per_stage(L) <= kPiMaxIt;
}

size_t wn_pipe_lds_bytes(const WnPipeArgs& a) {
  const int kc = a.C / 16, nw = kPiWaves;
  const int wide = a.C > a.H1 ? a.C : a.H1;
  const int ldh = a.C + 4, ldy = wide + 4, ldl = a.n_logits_pad + 4;
  return (size_t)5 * 4 * ldh * 4 + (size_t)nw * 64 * 16 + (size_t)kPiMaxIt * (16 + 128) + 2 * 64 * 4 + 16 * 4 + 16 +
         (wn_pipe_iters_per_stage(a.L) > kPiRegSlots ? (size_t)nw * (kc / 2) * 64 * 16 : 0) + (size_t)16 * ldy * 4 + (size_t)16 * ldl * 4 +
         (size_t)((a.n_logits_pad / 16 + a.Gn - 1) / a.Gn) * ((a.H1 / 16) * 1024 + 64);
}

int launch_wavenet_pipe(const WnPipeArgs& a, hipStream_t stream) {
  const int kc = a.C / 16;
  if (!wn_pipe_supported(a.C, a.Mg, a.Gc, a.L))
    return fail(MMK_ERR_UNSUPPORTED, "pipelined WaveNet kernel: C=%d, %d groups of %d clips, %d layers", a.C, a.Gc, a.Mg, a.L);
  const size_t lds = wn_pipe_lds_bytes(a);
  if (lds > 160 * 1024) return fail(MMK_ERR_UNSUPPORTED, "pipelined WaveNet kernel: %zu bytes of LDS needed", lds);
  dim3 grid(8 * a.Gn), block(kPiThreads);
  const int nit = a.n_it;
#define MMK_WNP3(KC_, NIT_)                                                                                            \
  do {                                                                                                                 \
    if (a.stamps) hipLaunchKernelGGL((wavenet_pipe_kernel<KC_, NIT_, true, 2>), grid, block, lds, stream, a);         \
    else if (a.temperature) hipLaunchKernelGGL((wavenet_pipe_kernel<KC_, NIT_, false, 1>), grid, block, lds, stream, a); \
    else hipLaunchKernelGGL((wavenet_pipe_kernel<KC_, NIT_, false, 0>), grid, block, lds, stream, a);                  \
  } while (0)
#define MMK_WNP2(KC_)                                                                                                  \
  do {                                                                                                                 \
    switch (nit) {                                                                                                     \
      case 1: MMK_WNP3(KC_, 1); break;                                                                                 \
      case 2: MMK_WNP3(KC_, 2); break;                                                                                 \
      case 3: MMK_WNP3(KC_, 3); break;                                                                                 \
      default: MMK_WNP3(KC_, 4); break;                                                                                \
    }                                                                                                                  \
  } while (0)
  switch (kc) {
    case 2: MMK_WNP2(2); break;
    case 4: MMK_WNP2(4); break;
    case 6: MMK_WNP2(6); break;
    case 8: MMK_WNP2(8); break;
    case 10: MMK_WNP2(10); break;
    case 12: MMK_WNP2(12); break;
    case 14: MMK_WNP2(14); break;
    default: MMK_WNP2(16); break;
  }
#undef MMK_WNP2
#undef MMK_WNP3
  MMK_HIP(hipGetLastError());
  return MMK_OK;
}

}  // namespace mmk

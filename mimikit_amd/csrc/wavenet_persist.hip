// Persistent WaveNet step kernel (gfx950): the whole `for t` loop of one generate call runs in ONE
// launch.  Per-step work is a chain of 2*L + 4 tiny dependent GEMMs; as separate launches each
// link costs >= 4-5 us of dispatch/drain on this chip (profiles/r01_v1_*), so the chain is kept
// on-chip and the links become data-tagged hand-offs through L2:
//
//   * clips are split in Gc independent groups (<= 16 clips each = one MFMA row tile); groups never
//     talk to each other;
//   * inside a group, workgroup j of Gn owns ONE 16-row tile of every layer's packed weight
//     matrices: tile j of A = [tap0 | tap1] (8 gated channels) and one tile of B = [res ; skip];
//   * a layer is:  phase A  z = W0.h_l[tau-d] + W1.h_l[tau] + (Wc.c)[tau] + b -> gate -> publish y slice
//                  phase B  wait y -> [res|skip] tile -> publish h_{l+1} slice / accumulate skip
//                           wait h_{l+1}
//     nothing else sits between the hand-offs: the delayed tap comes from a private per-workgroup ring
//     of past layer inputs (prefetched into LDS a layer ahead), and the conditioning products Wc.c of
//     all layers are computed for a whole block of positions by one GEMM BEFORE the launch (the
//     conditioning is known up front), so the chain carries only what depends on the previous sample;
//   * "publish" = 8-byte {epoch, value} granules, "wait" = every thread polls its own granules with
//     agent-scope relaxed atomic loads (sc1, L1 bypass) until all tags match
//     (cdna_hip_programming.md guideline 16, form R2: the data is the flag);
//   * head: skip sums -> fc0+Mish -> fc2 -> temperature/argmax|sample, three more hand-offs,
//     the sampled class is written to the caller's int64 tensor and handed to every workgroup
//     for the next step's embedding row.
//
// Same arithmetic as the launch path / the reference: fp32 MFMA fmaf chains, fixed reduction
// order, so results are run-to-run deterministic.  Every spin is bounded; a timeout raises
// err_flag and every workgroup leaves at its next hand-off.
#include "wavenet_persist.h"

namespace mmk {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;
// Pointers that come out of LDS tables lose their address space and compile to FLAT loads, which count against
// lgkmcnt as well: every later LDS wait would then also wait for the weight prefetch.  Force global loads.
typedef const __attribute__((address_space(1))) f32x4* gf32x4_ptr;
__device__ __forceinline__ gf32x4_ptr as_global(const void* p) { return (gf32x4_ptr)(uintptr_t)p; }

constexpr int kPersistThreads = 512;   // upper bound; actual = 64 * nw
constexpr unsigned kSpinLimit = 1u << 22;

// XCD == false: agent-scope store (sc1, write-through to memory; readable from every XCD).
// XCD == true : plain 8-byte store that stays in the producer's XCD L2; only used when every consumer
//               of the granule was VERIFIED (at kernel start, from HW_REG_XCC_ID) to run on the same XCD,
//               whose L2 is the coherence point for its CUs.  Consumers always load with sc1 (L1 bypass).
template <bool XCD>
__device__ __forceinline__ void gran_store(u64* p, unsigned epoch, float v) {
  const u64 x = ((u64)epoch << 32) | (u64)__float_as_uint(v);
  if (XCD)
    __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else
    __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool XCD>
__device__ __forceinline__ void gran_store_u32(u64* p, unsigned epoch, unsigned v) {
  const u64 x = ((u64)epoch << 32) | (u64)v;
  if (XCD)
    __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else
    __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 gran_load(const u64* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Poll `count` granules until every tag == epoch and write the values as floats into an LDS matrix
// (`cols` values per row, leading dimension ld; cols and ld are multiples of 4).  Each thread keeps four 8-byte
// loads in flight per round, so a sweep whose data is already there costs ONE round trip; the thread's first
// four granules sit at (row0, col0) when cols == cols0 (precomputed by the caller: no division on the path).
// Returns a workgroup-uniform success flag.
template <int NT>
__device__ __forceinline__ bool sweep(const u64* gran, int count, int cols, unsigned epoch, float* dst, int ld,
                                      int cols0, int row0, int col0, int* err_flag, int* s_fail) {
  const int tid = threadIdx.x;
  constexpr int nt = NT;   // blockDim.x, as a constant: reading it costs a global load from the dispatch packet
  for (int base = 0; base < count; base += nt * 4) {
    const int i0 = base + tid * 4;
    if (i0 < count) {      // count is a multiple of 16, i0 of 4: all four granules exist and share a row
      u64 v[4];
      unsigned spins = 0;
      bool ok = true;
      for (;;) {
        bool all = true;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = gran_load(gran + i0 + k);
#pragma unroll
        for (int k = 0; k < 4; ++k) all = all && ((unsigned)(v[k] >> 32) == epoch);
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
      int m, c;
      if (base == 0 && cols == cols0) {
        m = row0;
        c = col0;
      } else {
        m = i0 / cols;
        c = i0 - m * cols;
      }
      *reinterpret_cast<f32x4*>(dst + m * ld + c) =
          f32x4{__uint_as_float((unsigned)v[0]), __uint_as_float((unsigned)v[1]), __uint_as_float((unsigned)v[2]),
                __uint_as_float((unsigned)v[3])};
    }
  }
  // vmcnt(0) on EVERY path: waves without granules skip the polls, and without this the compiler has to assume
  // that older requests are still pending and waits for them in front of the next batch of requests instead
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  return *s_fail == 0;
}

// gate of the gated units with hardware exp/rcp: tanh(f) * sigmoid(g)   (wavenet_v2.py:151)
// |error| ~1e-7 absolute, far inside the logit tolerance; keeps the critical-path epilogue short
__device__ __forceinline__ float fast_sigmoid(float x) { return __frcp_rn(1.0f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) { return 1.0f - 2.0f * __frcp_rn(1.0f + __expf(2.0f * x)); }

// one wave's share of  X[16 x K] (LDS, ld) . W^T  for a 16-column tile: chunks [c0, c1) after chunk_base
__device__ __forceinline__ f32x4 tile_mma(const float* x, int ld, const f32x4* wp, int chunk_base, int c0, int c1,
                                          int lane) {
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  const int r = lane & 15, q = lane >> 4;
  for (int c = c0; c < c1; ++c) {
    const f32x4 w = wp[(int64_t)(chunk_base + c) * 64 + lane];
    const f32x4 xv = *reinterpret_cast<const f32x4*>(x + r * ld + c * 16 + 4 * q);
#pragma unroll
    for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[i], w[i], acc, 0, 0, 0);
  }
  return acc;
}

// same with the weight fragments already in registers (CPW chunks per wave, starting at chunk c0); all LDS
// reads are issued before the first MFMA so their latencies overlap
template <int CPW>
__device__ __forceinline__ f32x4 tile_mma_reg(const float* x, int ld, const f32x4 (&w)[CPW], int c0, int lane, f32x4 acc) {
  const int r = lane & 15, q = lane >> 4;
  f32x4 xv[CPW];
#pragma unroll
  for (int u = 0; u < CPW; ++u) xv[u] = *reinterpret_cast<const f32x4*>(x + r * ld + (c0 + u) * 16 + 4 * q);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int u = 0; u < CPW; ++u) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[u][i], w[u][i], acc, 0, 0, 0);
  }
  return acc;
}
// two products into one accumulator chain: x0 . w0 + x1 . w1
template <int CPW>
__device__ __forceinline__ f32x4 tile_mma_reg2(const float* x0, const f32x4 (&w0)[CPW], const float* x1,
                                               const f32x4 (&w1)[CPW], int ld, int c0, int lane) {
  const int r = lane & 15, q = lane >> 4;
  f32x4 xa[CPW], xb[CPW];
#pragma unroll
  for (int u = 0; u < CPW; ++u) {
    xa[u] = *reinterpret_cast<const f32x4*>(x0 + r * ld + (c0 + u) * 16 + 4 * q);
    xb[u] = *reinterpret_cast<const f32x4*>(x1 + r * ld + (c0 + u) * 16 + 4 * q);
  }
  __builtin_amdgcn_sched_barrier(0);   // keep the reads ahead of the MFMA chain (the scheduler sinks them otherwise)
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < CPW; ++u) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[u][i], w0[u][i], acc, 0, 0, 0);
  }
#pragma unroll
  for (int u = 0; u < CPW; ++u) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xb[u][i], w1[u][i], acc, 0, 0, 0);
  }
  return acc;
}

// cross-wave reduction in fixed order; result valid in wave 0 only (head phases)
__device__ __forceinline__ f32x4 reduce_waves(f32x4 acc, f32x4* red, int wave, int lane, int nw) {
  red[wave * 64 + lane] = acc;
  __syncthreads();
  f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
  if (wave == 0) {
    for (int w = 0; w < nw; ++w) {
      const f32x4 p = red[w * 64 + lane];
      v[0] += p[0]; v[1] += p[1]; v[2] += p[2]; v[3] += p[3];
    }
  }
  __syncthreads();
  return v;
}

template <int CPW, int NW, bool STAMPS, bool XCD>
__global__ __launch_bounds__(kPersistThreads) void wavenet_persist_kernel(const WnPersistArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int nw = NW;
  constexpr int NT = NW * 64;          // == NT
  int g = blockIdx.x / a.Gn;          // clip group
  int j = blockIdx.x % a.Gn;          // tile owner inside the group
  if (XCD) {
    // Roles follow the hardware: group = the XCD this workgroup actually runs on, owner index = arrival
    // order on that XCD.  Every workgroup registers, waits for all registrations, and the launch only
    // proceeds if each of the 8 XCDs hosts exactly Gn workgroups; otherwise err_flag = 2 and everyone
    // leaves (the host then falls back to agent-scope hand-offs).  Placement is verified, never assumed.
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
    g = role[0];
    j = role[1];
    __syncthreads();
    if (g < 0) return;
  }
  const int C = a.C, L = a.L;
  const int m_first = g * a.Mg;
  const int mg = min(a.Mg, a.B - m_first);   // clips of this group
  if (mg <= 0) return;

  // ---- LDS carve (all dynamic, 16-byte aligned pieces) -----------------------------------
  const int wide = max(max(C, a.S), a.H1);
  const int ldh = C + 4, ldy = wide + 4, ldl = a.n_logits_pad + 4;
  char* sp = smem_raw;
  float* hbuf = (float*)sp;   sp += 16 * ldh * 4;                         // layer input h_l[tau], rows = clips
  float* hprev0 = (float*)sp; sp += 2 * 16 * ldh * 4;                     // h_l[tau - d_l], double buffered by layer parity
  float* ybuf = (float*)sp;   sp += 16 * ldy * 4;                         // gated output / skip sums / hidden
  f32x4* red = (f32x4*)sp;    sp += 2 * nw * 64 * 16;                     // split-K partials: [A | B][nw][64]
  const float** lt_A = (const float**)sp; sp += L * 8;                    // per-layer table, copied once
  const float** lt_B = (const float**)sp; sp += L * 8;
  int64_t* lt_off = (int64_t*)sp;         sp += L * 8;
  int* lt_i = (int*)sp;                   sp += ((L * 4 * 4 + 15) / 16) * 16;   // dil, ring mask, has_res, btile
  float* biasA = (float*)sp;              sp += L * 16 * 4;
  float* biasB = (float*)sp;              sp += L * 16 * 4;
  int* s_idx = (int*)sp;      sp += 16 * 4;
  int* s_fail = (int*)sp;     sp += 16;
  float* lbuf = (float*)sp;                                               // logits for the sampler (owner 0 only)
  f32x4* redA = red, *redB = red + nw * 64;

  const int kcC = C / 16;                 // K-chunks of one tap / of B
  const int kcA = a.kcA;                  // chunks per A tile: 2*kcC (+ cond chunks, unused here)
  const int c0 = wave * CPW;              // this wave's chunk range inside a K = C product
  const int D_q = lane >> 4, D_n = lane & 15;
  const bool owns_res = j < kcC;          // owners [0, C/16) hold residual rows of B, the others skip rows
  const bool has_cond = a.C1 > 0;

  for (int i = tid; i < 3 * 16 * ldh; i += NT) hbuf[i] = 0.f;      // hbuf + both hprev buffers
  for (int i = tid; i < 16 * ldy; i += NT) ybuf[i] = 0.f;
  if (j == 0)
    for (int i = tid; i < 16 * ldl; i += NT) lbuf[i] = 0.f;
  for (int l = tid; l < L; l += NT) {
    const WnLayerTab t = a.layers[l];
    const int btile = owns_res ? j : (j - kcC + (t.has_res ? kcC : 0));
    lt_A[l] = t.A_wp;
    lt_B[l] = t.B_wp;
    lt_off[l] = t.ring_offset;
    lt_i[4 * l + 0] = t.dil;
    lt_i[4 * l + 1] = t.ring_mask;
    lt_i[4 * l + 2] = t.has_res;
    lt_i[4 * l + 3] = btile;
  }
  for (int i = tid; i < L * 16; i += NT) {
    const int l = i >> 4, n = i & 15;
    const WnLayerTab t = a.layers[l];
    const int btile = owns_res ? j : (j - kcC + (t.has_res ? kcC : 0));
    biasA[i] = t.A_bias ? t.A_bias[j * 16 + n] : 0.f;
    biasB[i] = (t.B_bias && (!owns_res || t.has_res)) ? t.B_bias[btile * 16 + n] : 0.f;
  }
  if (tid == 0) *s_fail = 0;
  __syncthreads();

  // ---- per-group exchange buffers, per-workgroup private history ring ---------------------------
  u64* gran_h = a.gran_h + (int64_t)g * 16 * C;
  u64* gran_y0 = a.gran_y + (int64_t)g * 2 * 16 * C;
  u64* gran_skip = a.gran_skip + (int64_t)g * 16 * a.S;
  u64* gran_hid = a.gran_hid + (int64_t)g * 16 * a.H1;
  u64* gran_logit = a.gran_logit + (int64_t)g * 16 * a.n_logits_pad;
  u64* gran_idx = a.gran_idx + (int64_t)g * 16;
  float* h_ring = a.h_rings + (int64_t)(g * a.Gn + j) * a.ring_floats_per_wg;   // follows the ROLE, not the block id
  const int slot_floats = a.Mg * C;       // one ring slot = the group's clips x C
  int* err = a.err_flag;

  // weight fragments: current layer and (prefetched one layer ahead) next layer
  f32x4 w_t1[CPW], w_t0[CPW], w_b[CPW];
  f32x4 n_t1[CPW], n_t0[CPW], n_b[CPW];
  auto load_A = [&](int l, f32x4 (&t1)[CPW], f32x4 (&t0)[CPW]) {
    gf32x4_ptr A = as_global(lt_A[l]) + (int64_t)j * kcA * 64 + lane;
#pragma unroll
    for (int u = 0; u < CPW; ++u) {
      t0[u] = A[(int64_t)(c0 + u) * 64];
      t1[u] = A[(int64_t)(kcC + c0 + u) * 64];
    }
  };
  auto load_B = [&](int l, f32x4 (&b)[CPW]) {
    if (!owns_res || lt_i[4 * l + 2]) {   // B of layer l: [res rows (if the layer has a residual conv) ; skip rows]
      gf32x4_ptr Bm = as_global(lt_B[l]) + (int64_t)lt_i[4 * l + 3] * kcC * 64 + lane;
#pragma unroll
      for (int u = 0; u < CPW; ++u) b[u] = Bm[(int64_t)(c0 + u) * 64];
    }
  };
  // Epilogues run one output ELEMENT per thread: element e = (clip m, column n) of the 16x16 tile for
  // e < mg*16, so only real clips cost transcendental work.  `frag` is the element's float index inside a
  // 64-lane x 4-register MFMA accumulator image.
  const bool elem = tid < mg * 16;
  const int e_m = tid >> 4, e_n = tid & 15;
  const int frag = ((e_m >> 2) * 16 + e_n) * 4 + (e_m & 3);
  auto sum_partials = [&](const f32x4* part) -> float {
    const float* f = reinterpret_cast<const float*>(part) + frag;
    float pv[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) pv[w] = f[w * 256];   // all reads in flight before the first add
    __builtin_amdgcn_sched_barrier(0);
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) v += pv[w];
    return v;
  };
  // history: float i = tid + k NT of a ring slot is (clip m0 + 2k, channel hc)   [NT == 2C]
  // the workgroup's h_l[tau - d_l] (mg x C floats) rides in up to 8 registers per thread while in flight
  constexpr int kHP = 8;
  const int mgC = mg * C;                         // <= kHP * NT by construction (launch check)
  const int hm0 = tid >= C ? 1 : 0, hc = tid - hm0 * C;
  float hp[kHP];
  auto ring_slot = [&](int l, int64_t pos) -> float* {
    return h_ring + lt_off[l] + (int64_t)(pos & lt_i[4 * l + 1]) * slot_floats;
  };
  auto hist_load = [&](int l, int64_t tau) {
    const float* src = ring_slot(l, tau - lt_i[4 * l]);
#pragma unroll
    for (int k = 0; k < kHP; ++k) {
      if (k * NT >= mgC) break;                   // uniform
      const int i = tid + k * NT;
      hp[k] = i < mgC ? src[i] : 0.f;
    }
  };
  auto hist_to_lds = [&](int l) {
    float* dst = hprev0 + (l & 1) * 16 * ldh;
#pragma unroll
    for (int k = 0; k < kHP; ++k) {
      if (k * NT >= mgC) break;
      if (tid + k * NT < mgC) dst[(hm0 + 2 * k) * ldh + hc] = hp[k];
    }
  };
  // h_l[tau] (LDS hbuf) -> ring of layer l; same thread <-> float mapping as hist_load, so a thread only ever
  // reads back its own stores
  auto ring_store = [&](int l, int64_t tau) {
    float* dst = ring_slot(l, tau);
#pragma unroll
    for (int k = 0; k < kHP; ++k) {
      if (k * NT >= mgC) break;
      if (tid + k * NT < mgC) dst[tid + k * NT] = hbuf[(hm0 + 2 * k) * ldh + hc];
    }
  };
  // conditioning product of (clip e_m, position, layer l, packed column 16 j + e_n), computed before the launch
  auto cond_at = [&](int l, int64_t s) -> float {
    return a.condall[(((int64_t)(m_first + e_m) * a.cond_steps + s) * L + l) * (2 * C) + j * 16 + e_n];
  };
  // first-round position of this thread's four granules in a sweep over mg x C values
  const int sw_row = (tid * 4) / C, sw_col = tid * 4 - sw_row * C;

  // diagnostic build only: 100 MHz wall-clock stamps of the phases of owner 1 of group 0, summed over steps and layers
  unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long st_prev = 0;
  auto stamp = [&](int slot) {
    if (STAMPS) {
      const unsigned long long now = wall_clock64();
      st_acc[slot] += now - st_prev;
      st_prev = now;
    }
  };
  const unsigned long long clk_start = STAMPS ? clock64() : 0, wall_start = STAMPS ? wall_clock64() : 0;
  float skipacc = 0.f;                           // element threads of skip-row owners
  const int64_t tau0 = a.t0 - 1;

  // Memory prefetch discipline.  vmcnt retires in order, so a poll issued behind a long-latency load only
  // returns after it.  Everything a layer needs from memory (its weight fragments, its delayed input, its
  // conditioning term) is therefore requested ONE LAYER AHEAD, at the top of the previous layer: right after a
  // sweep (whose last poll drained every older request) and right before that layer's MFMAs and epilogue, which
  // do not touch memory - the latency hides under compute and under the wait for the other workgroups.
  float cnd = 0.f, cnd_n = 0.f;
  hist_load(0, tau0);
  load_A(0, n_t1, n_t0);
  load_B(0, n_b);
  if (elem && has_cond) cnd_n = cond_at(0, 0);

  for (int64_t s = 0; s < a.n_steps; ++s) {
    const int64_t tau = tau0 + s;
    // ---- input 0: embedding row of the newest sample ---------------------------------------
    if (s > 0 && !a.teacher_forced) {
      // classes sampled by the previous step arrive as granules (epoch = s)
      if (tid < mg) {
        unsigned spins = 0;
        u64 v;
        for (;;) {
          v = gran_load(gran_idx + tid);
          if ((unsigned)(v >> 32) == (unsigned)s) break;
          ++spins;
          if (spins > kSpinLimit || ((spins & 255u) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
            *s_fail = 1;
            atomicExch(err, 1);
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
        s_idx[tid] = (int)(unsigned)v;
      }
    } else {
      if (tid < mg) s_idx[tid] = (int)a.idx[(int64_t)(m_first + tid) * a.idx_rs + tau];
    }
    __syncthreads();
    if (*s_fail) return;
    {
      float* ring0 = ring_slot(0, tau);
#pragma unroll
      for (int k = 0; k < kHP; ++k) {
        if (k * NT >= mgC) break;
        if (tid + k * NT < mgC) {
          const int cls = s_idx[hm0 + 2 * k];
          // torch raises on an out-of-range class; keep memory safe and make it visible (NaN row)
          const float v = (cls >= 0 && cls < a.q_levels) ? a.emb[(int64_t)cls * C + hc] : __builtin_nanf("");
          hbuf[(hm0 + 2 * k) * ldh + hc] = v;
          ring0[tid + k * NT] = v;
        }
      }
    }
    hist_to_lds(0);
#pragma unroll
    for (int u = 0; u < CPW; ++u) { w_t1[u] = n_t1[u]; w_t0[u] = n_t0[u]; w_b[u] = n_b[u]; }
    cnd = cnd_n;
    __syncthreads();
    if (STAMPS) st_prev = wall_clock64();

    for (int l = 0;; ++l) {   // leaves through the `last` branch at the bottom (keeps that path off the back edge)
      const unsigned epoch = (unsigned)(s * L + l + 1);
      const bool last = (l + 1 == L);
      const bool has_b = !owns_res || lt_i[4 * l + 2];
      // requests for the next layer (after the last layer: layer 0 of the next step).  Nothing is outstanding
      // here (a sweep or the step prologue just drained vmcnt); saying so keeps the compiler from guarding the
      // registers it recycles for the address arithmetic below with partial waits between the requests.
      __builtin_amdgcn_s_waitcnt(0x0F70);
      if (!last || s + 1 < a.n_steps) {
        const int nl = last ? 0 : l + 1;
        hist_load(nl, last ? tau + 1 : tau);
        if (elem && has_cond) cnd_n = cond_at(nl, last ? s + 1 : s);
        load_A(nl, n_t1, n_t0);
        load_B(nl, n_b);
      }
      stamp(8);   // requests issued
      // ---- phase A (critical): z = W0.h[tau-d] + W1.h[tau] + cond + b ; gate ; publish y -------------
      redA[wave * 64 + lane] = tile_mma_reg2<CPW>(hprev0 + (l & 1) * 16 * ldh, w_t0, hbuf, w_t1, ldh, c0, lane);
      __syncthreads();
      stamp(9);   // A: LDS reads + MFMA + barrier
      u64* gran_y = gran_y0 + (int64_t)(l & 1) * 16 * C;
      if (elem) {
        const float f = sum_partials(redA) + cnd + biasA[l * 16 + e_n];
        // even columns hold f, odd columns g of the same channel: one transcendental per thread
        const float act = (e_n & 1) ? fast_sigmoid(f) : fast_tanh(f);
        const float other = __shfl_down(act, 1);
        if (!(e_n & 1)) gran_store<XCD>(gran_y + e_m * C + j * 8 + (e_n >> 1), epoch, act * other);
      }
      // this layer's input joins its history ring while the other workgroups' y slices are on their way
      if (l > 0) ring_store(l, tau);
      stamp(0);   // A: epilogue + publish + ring store
      // ---- phase B (critical): wait y ; [res | skip] tile ----------------------------------------
      if (has_b) {
        if (!sweep<NT>(gran_y, mg * C, C, epoch, ybuf, ldy, C, sw_row, sw_col, err, s_fail)) return;
        stamp(1);   // wait y
        redB[wave * 64 + lane] = tile_mma_reg<CPW>(ybuf, ldy, w_b, c0, lane, f32x4{0.f, 0.f, 0.f, 0.f});
        __syncthreads();
        if (elem) {
          const float vb = sum_partials(redB) + biasB[l * 16 + e_n];
          if (owns_res)
            gran_store<XCD>(gran_h + e_m * C + j * 16 + e_n, epoch, hbuf[e_m * ldh + j * 16 + e_n] + vb);
          else
            skipacc = (l == 0) ? vb : vb + skipacc;
        }
      }
      stamp(2);   // phase B
      // ---- wait for the next layer's input; its delayed input moves from registers to LDS first ----
      if (!last) {
        hist_to_lds(l + 1);
        if (!sweep<NT>(gran_h, mg * C, C, epoch, hbuf, ldh, C, sw_row, sw_col, err, s_fail)) return;
#pragma unroll
        for (int u = 0; u < CPW; ++u) { w_t1[u] = n_t1[u]; w_t0[u] = n_t0[u]; w_b[u] = n_b[u]; }
        cnd = cnd_n;
        stamp(3);   // wait h'
      } else {
        __syncthreads();
        stamp(3);
        break;
      }
    }
    if (a.teacher_forced) continue;

    // ---- head -----------------------------------------------------------------------------------
    const unsigned he = (unsigned)(s + 1);
    if (elem && !owns_res) gran_store<XCD>(gran_skip + e_m * a.S + (j - kcC) * 16 + e_n, he, skipacc);
    // fc0 + Mish : tiles j, j+Gn, ... of H1/16
    const int t_fc0 = a.H1 / 16, kc_fc0 = a.S / 16;
    if (j < t_fc0) {
      if (!sweep<NT>(gran_skip, mg * a.S, a.S, he, ybuf, ldy, C, sw_row, sw_col, err, s_fail)) return;
      const int per = (kc_fc0 + nw - 1) / nw;
      const int k0 = min(wave * per, kc_fc0), k1 = min(k0 + per, kc_fc0);
      for (int t = j; t < t_fc0; t += a.Gn) {
        const f32x4* W = reinterpret_cast<const f32x4*>(a.fc0_wp) + (int64_t)t * kc_fc0 * 64;
        f32x4 v = reduce_waves(tile_mma(ybuf, ldy, W, 0, k0, k1, lane), redA, wave, lane, nw);
        if (wave == 0) {
          const float bias = a.fc0_bias[t * 16 + D_n];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int m = 4 * D_q + r;
            if (m < mg) gran_store<XCD>(gran_hid + m * a.H1 + t * 16 + D_n, he, mishf_(v[r] + bias));
          }
        }
      }
    }
    // fc2 : tiles of the (n_classes + temperature column) outputs
    const int t_fc2 = a.n_logits_pad / 16, kc_fc2 = a.H1 / 16;
    if (j < t_fc2) {
      if (!sweep<NT>(gran_hid, mg * a.H1, a.H1, he, ybuf, ldy, C, sw_row, sw_col, err, s_fail)) return;
      const int per = (kc_fc2 + nw - 1) / nw;
      const int k0 = min(wave * per, kc_fc2), k1 = min(k0 + per, kc_fc2);
      for (int t = j; t < t_fc2; t += a.Gn) {
        const f32x4* W = reinterpret_cast<const f32x4*>(a.fc2_wp) + (int64_t)t * kc_fc2 * 64;
        f32x4 v = reduce_waves(tile_mma(ybuf, ldy, W, 0, k0, k1, lane), redA, wave, lane, nw);
        if (wave == 0) {
          const float bias = a.fc2_bias[t * 16 + D_n];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int m = 4 * D_q + r;
            if (m < mg) gran_store<XCD>(gran_logit + m * a.n_logits_pad + t * 16 + D_n, he, v[r] + bias);
          }
        }
      }
    }
    // temperature column + argmax / inverse-CDF sample : owner 0 of the group, one clip per wave
    if (j == 0) {
      if (!sweep<NT>(gran_logit, mg * a.n_logits_pad, a.n_logits_pad, he, lbuf, ldl, C, sw_row, sw_col, err, s_fail)) return;
      const int nc = a.n_classes;
      const int per = (nc + 63) / 64;
      for (int m = wave; m < mg; m += nw) {
        const float* lg = lbuf + m * ldl;
        const int clip = m_first + m;
        if (a.logits_out)
          for (int c = lane; c < nc + a.learn_temp; c += 64) a.logits_out[(int64_t)clip * a.logits_ld + c] = lg[c];
        float denom = 1.f;
        if (a.learn_temp) denom = fmaxf(sigmoidf_(lg[nc]), a.min_temp);   // mlp.py:60-62
        int result;
        if (a.temperature == nullptr) {
          float best = -INFINITY;
          int bi = 0x7fffffff;
          for (int q = 0; q < per; ++q) {
            const int c = lane * per + q;
            if (c < nc) {
              const float v = a.learn_temp ? lg[c] / denom : lg[c];
              if (v > best || bi == 0x7fffffff) { best = v; bi = c; }
            }
          }
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o);
            const int oi = __shfl_xor(bi, o);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
          }
          result = bi;
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
          a.idx[(int64_t)clip * a.idx_rs + tau + 1] = result;
          gran_store_u32<XCD>(gran_idx + m, he, (unsigned)result);
        }
      }
    }
    __syncthreads();
    stamp(6);     // head (as seen by this workgroup)
  }
  if (STAMPS && a.stamps && g == 0 && j == 1 && tid == 0) {
    st_acc[12] = clock64() - clk_start;        // shader cycles
    st_acc[13] = wall_clock64() - wall_start;  // 100 MHz ticks
    for (int i = 0; i < 16; ++i) a.stamps[i] = st_acc[i];
  }
}

size_t wn_persist_lds_bytes(const WnPersistArgs& a, int nw) {
  const int wide = a.C > a.S ? (a.C > a.H1 ? a.C : a.H1) : (a.S > a.H1 ? a.S : a.H1);
  const int ldh = a.C + 4, ldy = wide + 4, ldl = a.n_logits_pad + 4;
  return (size_t)3 * 16 * ldh * 4 + (size_t)16 * ldy * 4 + (size_t)2 * nw * 64 * 16 + (size_t)a.L * 24 +
         (size_t)((a.L * 16 + 15) / 16) * 16 + (size_t)a.L * 128 + 16 * 4 + 16 + (size_t)16 * ldl * 4;
}

int launch_wavenet_persist(const WnPersistArgs& a, hipStream_t stream) {
  const int cpw = 2;                         // K-chunks of a K = C product per wave
  const int nw = a.C / (16 * cpw);
  if (nw < 1 || nw > 8 || a.C % 32) return fail(MMK_ERR_UNSUPPORTED, "persistent WaveNet: C=%d not in {32..256 step 32}", a.C);
  if ((int64_t)a.Mg * a.C > 8 * 64 * nw) return fail(MMK_ERR_UNSUPPORTED, "persistent WaveNet: %d clips per group do not fit the history prefetch", a.Mg);
  const size_t lds = wn_persist_lds_bytes(a, nw);
  if (lds > 160 * 1024) return fail(MMK_ERR_UNSUPPORTED, "persistent WaveNet: %zu bytes of LDS needed", lds);
  dim3 grid(a.Gc * a.Gn), block(64 * nw);
#define MMK_WNP(NW_)                                                                                               \
  do {                                                                                                             \
    if (a.stamps) {                                                                                                \
      if (a.xcd_local) hipLaunchKernelGGL((wavenet_persist_kernel<2, NW_, true, true>), grid, block, lds, stream, a);   \
      else hipLaunchKernelGGL((wavenet_persist_kernel<2, NW_, true, false>), grid, block, lds, stream, a);              \
    } else {                                                                                                       \
      if (a.xcd_local) hipLaunchKernelGGL((wavenet_persist_kernel<2, NW_, false, true>), grid, block, lds, stream, a);  \
      else hipLaunchKernelGGL((wavenet_persist_kernel<2, NW_, false, false>), grid, block, lds, stream, a);             \
    }                                                                                                              \
  } while (0)
  switch (nw) {
    case 1: MMK_WNP(1); break;
    case 2: MMK_WNP(2); break;
    case 3: MMK_WNP(3); break;
    case 4: MMK_WNP(4); break;
    case 5: MMK_WNP(5); break;
    case 6: MMK_WNP(6); break;
    case 7: MMK_WNP(7); break;
    default: MMK_WNP(8); break;
  }
#undef MMK_WNP
  MMK_HIP(hipGetLastError());
  return MMK_OK;
}

}  // namespace mmk

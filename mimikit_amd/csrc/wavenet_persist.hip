// Persistent WaveNet step kernel (gfx950): the whole `for t` loop of one generate call runs in ONE
// launch.  Per-step work is a chain of 2*L + 4 tiny dependent GEMMs; as separate launches each
// link costs >= 4-5 us of dispatch/drain on this chip (profiles/r01_v1_*), so the chain is kept
// on-chip and the links become data-tagged hand-offs through L2:
//
//   * clips are split in Gc independent groups (<= 16 clips each); groups never talk to each other;
//   * inside a group, workgroup j of Gn owns ONE 16-column tile of every layer's packed weight
//     matrices: tile j of A = [tap0 | tap1] (8 gated channels) and one tile of B = [res ; skip];
//   * a layer is:  phase A  z = W0.h_l[tau-d] + W1.h_l[tau] + (Wc.c)[tau] + b -> gate -> publish y slice
//                  phase B  wait y -> [res|skip] tile -> publish h_{l+1} slice / accumulate skip
//                           wait h_{l+1}
//     nothing else sits between the hand-offs: the delayed tap comes from a private per-workgroup ring
//     of past layer inputs (requested a layer ahead), and the conditioning products Wc.c of all layers are
//     computed for a whole block of positions by one GEMM BEFORE the launch (the conditioning is known up
//     front), so the chain carries only what depends on the previous sample;
//   * "publish" = 8-byte {epoch, value} granules, "wait" = I/O threads poll their own granules with
//     agent-scope relaxed atomic loads (sc1, L1 bypass) until all tags match
//     (cdna_hip_programming.md guideline 16, form R2: the data is the flag).  One poll set in flight per
//     thread, with a short sleep between rounds: back-to-back polling floods the memory pipe and costs
//     18 us per step on cfg4 (measured);
//   * a workgroup's waves are specialised (I/O waves / matrix waves, see the kernel) because the CU's memory
//     pipe is in order: polls must never queue behind the weight stream;
//   * groups of <= 4 clips use v_mfma_f32_4x4x1_16b_f32 blocks instead of 16-row tiles (no padding rows);
//   * head: skip sums -> fc0+Mish -> fc2 -> temperature/argmax|sample, three more hand-offs,
//     the sampled class is written to the caller's int64 tensor and handed to every workgroup
//     for the next step's embedding row.
//
// Same arithmetic as the launch path / the reference: fp32 MFMA fmaf chains, fixed reduction
// order, so results are run-to-run deterministic.  Every spin is bounded; a timeout raises
// err_flag and every workgroup leaves at its next hand-off.
#include "wavenet_persist.h"
#include "wavenet_handoff.h"

namespace mmk {

// A workgroup's per-layer facts, built once in LDS; one layer's entry is two 16-byte reads
struct __attribute__((aligned(16))) LtEntry {
  unsigned A_lo, A_hi;     // this workgroup's tile of the layer's packed A (byte address)
  unsigned B_lo, B_hi;     // its tile of packed B (the A tile again when it has none in this layer: fetched, unused)
  unsigned ring_off;       // byte offset of the layer's input-history ring inside the workgroup's block
  unsigned dil, mask;      // dilation, ring slots - 1
  unsigned has_b;
};

constexpr int kIoWaves = 4;                       // waves 0..3: hand-offs, epilogues, ring stores
constexpr int wn_cpw(int kc) { return kc % 4 == 0 ? 4 : 2; }          // K-chunks per matrix wave
constexpr int wn_threads(int kc) { return 64 * (kIoWaves + kc / wn_cpw(kc)); }

// Two kinds of waves.  vmcnt retires in order, so a poll issued behind a long-latency load only returns after
// it - and the weight stream of a layer (48 KiB per workgroup at C = 256, mostly from the memory-side cache)
// takes longer than the gap between two polls.  Therefore:
//   * I/O waves (0..3) never load anything but granules: they poll, move the arrived values into LDS, run the
//     per-element epilogues, publish, and store the layer inputs into the history ring;
//   * matrix waves (4..) never poll: they request the next layer's weight fragments, delayed input and
//     conditioning terms (one layer ahead, the fragment loads interleaved with the MFMA chain whose dependent
//     issue leaves the slots free), run the MFMAs, and pass the small operands on through LDS.
// Both kinds meet at the same six workgroup barriers per layer:
//   B1 phase-A partials in LDS | B1b y published | B2 y in LDS | B3 phase-B partials in LDS | B3b h' published |
//   B4 next input (+ delayed input) in LDS
template <int KC, bool SMALL, bool STAMPS, bool XCD>
__global__ __launch_bounds__(wn_threads(KC)) void wavenet_persist_kernel(const WnPersistArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr int CPW = wn_cpw(KC);      // K-chunks of a K = C product per matrix wave
  constexpr int NWM = KC / CPW;        // matrix waves
  constexpr int NMT = NWM * 64;        // matrix threads
  constexpr int NIO = kIoWaves * 64;   // I/O threads
  constexpr int nw = kIoWaves + NWM;
  constexpr int NT = nw * 64;          // == blockDim.x (reading the builtin costs a load from the dispatch packet)
  constexpr int C = 16 * KC;           // channels
  constexpr int kcC = KC;              // K-chunks of one tap / of B
  constexpr int ldh = C + 4;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_io = wave < kIoWaves;
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
    g = __builtin_amdgcn_readfirstlane(role[0]);
    j = __builtin_amdgcn_readfirstlane(role[1]);
    __syncthreads();
    if (g < 0) return;
  }
  const int L = a.L;
  const int m_first = g * a.Mg;
  const int mg = min(a.Mg, a.B - m_first);   // clips of this group
  if (mg <= 0) return;

  // ---- LDS carve (all dynamic, 16-byte aligned pieces) -----------------------------------
  const int wide = max(C, a.H1);
  const int ldy = wide + 4, ldl = a.n_logits_pad + 4;
  constexpr int kRed = (2 * NWM > nw ? 2 * NWM : nw) * 64;                // f32x4 slots: [A | B][NWM][64], head: [nw][64]
  char* sp = smem_raw;
  float* hbuf = (float*)sp;   sp += 16 * ldh * 4;                         // layer input h_l[tau], rows = clips
  float* hprev = (float*)sp;  sp += 16 * ldh * 4;                         // h_l[tau - d_l]
  float* ybuf = (float*)sp;   sp += 16 * ldy * 4;                         // gated output / skip sums / hidden
  f32x4* red = (f32x4*)sp;    sp += kRed * 16;                            // split-K partials
  LtEntry* ltab = (LtEntry*)sp;           sp += L * 32;                   // per-layer table, built once
  float* biasA = (float*)sp;              sp += L * 16 * 4;
  float* biasB = (float*)sp;              sp += L * 16 * 4;
  float* cndbuf = (float*)sp;             sp += 256 * 4;                  // conditioning term per output element
  int* sfx = (int*)sp;                    sp += ((L * 4 + 15) / 16) * 16;   // sfx[l] = sum of the dilations above layer l
  int* s_idx = (int*)sp;      sp += 16 * 4;
  int* s_fail = (int*)sp;     sp += 16;
  float* lbuf = (float*)sp;   sp += 16 * ldl * 4;                         // logits for the sampler (owner 0 only)
  // the head's weight tiles of this workgroup (fc0: tiles j, j + Gn, ..; fc2 likewise) and their biases, staged once
  // per launch: the layers' weight stream evicts them from L2 every step, and a miss costs ~1 us on the chain
  const int t_fc0 = a.H1 / 16, kc_fc0 = C / 16;
  const int t_fc2 = a.n_logits_pad / 16, kc_fc2 = a.H1 / 16;
  const int nt0 = j < t_fc0 ? (t_fc0 - j + a.Gn - 1) / a.Gn : 0, nt2 = j < t_fc2 ? (t_fc2 - j + a.Gn - 1) / a.Gn : 0;
  f32x4* hw0 = (f32x4*)sp;    sp += (size_t)((t_fc0 + a.Gn - 1) / a.Gn) * kc_fc0 * 1024;
  f32x4* hw2 = (f32x4*)sp;    sp += (size_t)((t_fc2 + a.Gn - 1) / a.Gn) * kc_fc2 * 1024;
  float* hb0 = (float*)sp;    sp += (size_t)((t_fc0 + a.Gn - 1) / a.Gn) * 64;
  float* hb2 = (float*)sp;
  f32x4* redA = red, *redB = red + NWM * 64;

  const int D_q = lane >> 4, D_n = lane & 15;
  const bool owns_res = j < kcC;          // owners [0, C/16) hold residual rows of B, the others skip rows
  const bool has_cond = a.C1 > 0;

  for (int i = tid; i < 2 * 16 * ldh; i += NT) hbuf[i] = 0.f;      // hbuf + hprev
  for (int i = tid; i < 16 * ldy; i += NT) ybuf[i] = 0.f;
  for (int i = tid; i < 256; i += NT) cndbuf[i] = 0.f;
  if (j == 0)
    for (int i = tid; i < 16 * ldl; i += NT) lbuf[i] = 0.f;
  for (int l = tid; l < L; l += NT) {
    const WnLayerTab t = a.layers[l];
    const bool hb = !owns_res || t.has_res;   // B of layer l: [res rows (if the layer has a residual conv) ; skip rows]
    const int btile = owns_res ? j : (j - kcC + (t.has_res ? kcC : 0));
    const uintptr_t Ap = (uintptr_t)(t.A_wp + (int64_t)j * a.kcA * 256);
    const uintptr_t Bp = hb ? (uintptr_t)(t.B_wp + (int64_t)btile * kcC * 256) : Ap;   // no B rows: any valid tile
    LtEntry e;
    e.A_lo = (unsigned)Ap; e.A_hi = (unsigned)(Ap >> 32);
    e.B_lo = (unsigned)Bp; e.B_hi = (unsigned)(Bp >> 32);
    e.ring_off = (unsigned)(t.ring_offset * 4);
    e.dil = (unsigned)t.dil; e.mask = (unsigned)t.ring_mask; e.has_b = hb ? 1u : 0u;
    ltab[l] = e;
  }
  for (int i = tid; i < L * 16; i += NT) {
    const int l = i >> 4, n = i & 15;
    const WnLayerTab t = a.layers[l];
    const int btile = owns_res ? j : (j - kcC + (t.has_res ? kcC : 0));
    biasA[i] = t.A_bias ? t.A_bias[j * 16 + n] : 0.f;
    biasB[i] = (t.B_bias && (!owns_res || t.has_res)) ? t.B_bias[btile * 16 + n] : 0.f;
  }
  for (int i = 0; i < nt0; ++i) {
    const f32x4* src = reinterpret_cast<const f32x4*>(a.fc0_wp) + (int64_t)(j + i * a.Gn) * kc_fc0 * 64;
    for (int q = tid; q < kc_fc0 * 64; q += NT) hw0[i * kc_fc0 * 64 + q] = src[q];
    if (tid < 16) hb0[i * 16 + tid] = a.fc0_bias[(j + i * a.Gn) * 16 + tid];
  }
  for (int i = 0; i < nt2; ++i) {
    const f32x4* src = reinterpret_cast<const f32x4*>(a.fc2_wp) + (int64_t)(j + i * a.Gn) * kc_fc2 * 64;
    for (int q = tid; q < kc_fc2 * 64; q += NT) hw2[i * kc_fc2 * 64 + q] = src[q];
    if (tid < 16) hb2[i * 16 + tid] = a.fc2_bias[(j + i * a.Gn) * 16 + tid];
  }
  if (tid == 0) *s_fail = 0;
  __syncthreads();
  if (tid == 0) {
    int acc = 0;
    for (int l = L - 1; l >= 0; --l) {
      sfx[l] = acc;
      acc += (int)ltab[l].dil;
    }
  }
  __syncthreads();

  // ---- per-group exchange buffers, per-workgroup private history ring ---------------------------
  u64* gran_h = a.gran_h + (int64_t)g * 16 * C;
  u64* gran_y0 = a.gran_y + (int64_t)g * 2 * 16 * C;
  u64* gran_skip = a.gran_skip + (int64_t)g * 16 * C;
  u64* gran_hid = a.gran_hid + (int64_t)g * 16 * a.H1;
  u64* gran_logit = a.gran_logit + (int64_t)g * 16 * a.n_logits_pad;
  u64* gran_idx = a.gran_idx + (int64_t)g * 16;
  char* h_ring = (char*)(a.h_rings + (int64_t)(g * a.Gn + j) * a.ring_floats_per_wg);   // follows the ROLE, not the block id
  const unsigned slot_bytes = (unsigned)a.Mg * C * 4;       // one ring slot = the group's clips x C
  const int slot_f4 = mg * (C / 4);                          // 16-byte pieces of a slot that hold real clips
  int* err = a.err_flag;

  // Epilogues run one output ELEMENT per I/O thread: element e = (clip m, column n) of the 16x16 tile for
  // e < mg*16, so only real clips cost transcendental work.  `frag` is the element's float index inside a
  // 64-lane x 4-register MFMA accumulator image.
  const bool elem = tid < mg * 16;
  const int e_m = tid >> 4, e_n = tid & 15;
  // 16x16 tiles: (row m, col n) sits in lane 16 (m / 4) + n, register m % 4; one partial per matrix wave.
  // 4x4 blocks  : (clip m, col n) sits in lane 16 (n / 4) + 12 + n % 4 (the lane of sub-slice 3, which holds the sum
  //               over the four sub-slices), register m; one partial per matrix wave.
  const int frag = SMALL ? ((e_n >> 2) * 16 + 12 + (e_n & 3)) * 4 + e_m : ((e_m >> 2) * 16 + e_n) * 4 + (e_m & 3);
  constexpr int kParts = NWM;
  auto sum_partials = [&](const f32x4* part) -> float {
    const float* f = reinterpret_cast<const float*>(part) + frag;
    float pv[kParts];
#pragma unroll
    for (int w = 0; w < kParts; ++w) pv[w] = f[w * 256];   // all reads in flight before the first add
    __builtin_amdgcn_sched_barrier(0);
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < kParts; ++w) v += pv[w];
    return v;
  };
  // gate: even packed columns hold f (tanh), odd columns g (sigmoid) of the same channel.  One code path for
  // both: tanh(x) = 2 sigmoid(2x) - 1, so act = k / (1 + exp(-k x)) - (k - 1) with k = 2 | 1   (wavenet_v2.py:151)
  const float gate_k = (e_n & 1) ? 1.f : 2.f;
  const float gate_scale = -gate_k * 1.4426950408889634f;
  const float gate_shift = 1.f - gate_k;
  // where this thread's results go (granule indices inside the group's exchange buffers)
  const unsigned y_slot = (unsigned)(e_m * C + j * 8 + (e_n >> 1));
  const unsigned h_slot = (unsigned)(e_m * C + j * 16 + e_n);
  const int res_off = e_m * ldh + j * 16 + e_n;          // residual input of the element (owners of residual rows)
  // first-round position of this thread's four granules in a sweep over mg x C values
  const int sw_row = (tid * 4) / C, sw_col = tid * 4 - sw_row * C;
  float* const sw_h = hbuf + sw_row * ldh + sw_col;
  float* const sw_y = ybuf + sw_row * ldy + sw_col;

  // ---- matrix-wave state ------------------------------------------------------------------------------
  const int mt = tid - NIO;                               // matrix thread index (negative on I/O waves)
  const int mwave = __builtin_amdgcn_readfirstlane(mt >> 6);
  const int c0 = mwave * CPW;                             // this wave's chunk range inside a K = C product
  // history: 16-byte piece q = mt + k NMT of a ring slot is (clip q / (C/4), channels 4 (q % (C/4)) ..)
  // Every matrix thread issues the SAME number of loads in every layer (addresses are clamped, never predicated):
  // with a conditional load on some path the compiler has to assume the shortest path when it counts vmcnt, and
  // the MFMAs end up waiting for requests issued just before them.
  constexpr int kClipsMax = SMALL ? 4 : 16;
  constexpr int kHP = (kClipsMax * (C / 4) + NMT - 1) / NMT;     // pieces per thread
  f32x4 hp[kHP];
  // conditioning product of (clip, position, layer, packed column 16 j + e_n), computed before the launch:
  // entry (step, layer) sits (step L + layer) 2C floats into the clip's block; element e = mt + k NMT belongs to
  // clip e / 16.  Without conditioning (or for a lane without an element) the address is a word of zeros.
  constexpr int kCP = (kClipsMax * 16 + NMT - 1) / NMT;          // elements per matrix thread
  const int64_t cond_clip = (int64_t)a.cond_steps * L * (2 * C);
  gcfloat_ptr cptr[kCP];
  int cstep[kCP];                                                // per lane: only real elements walk through condall
#pragma unroll
  for (int k = 0; k < kCP; ++k) {
    const int e = mt + k * NMT;
    const bool real = has_cond && e >= 0 && e < mg * 16;
    cptr[k] = real ? (gcfloat_ptr)(uintptr_t)(a.condall + (int64_t)(m_first + (e >> 4)) * cond_clip + j * 16 + (e & 15))
                   : (gcfloat_ptr)(uintptr_t)a.zeros;
    cstep[k] = real ? 2 * C : 0;
  }
  float cnd_n[kCP];
  // weight fragments: byte offset of this lane inside a tile; chunk c0 + u is u KiB further
  // (4x4 blocks: the lane's k sub-slice ks = lane / 16 starts 4 CPW ks floats into the wave's slice; its fragment u
  //  holds W[column lane % 16][4 consecutive k], which is element (q, n) of a chunk of the SAME packed matrix)
  constexpr int kFragStride = SMALL ? 16 : 64;               // f32x4 elements between a lane's consecutive fragments
  // lane = 16 cg + 4 ks + j: block lane / 4 = (column group cg, K sub-slice ks); the four sub-slices of a column group
  // share a row of 16 lanes, so their partial sums are added with two DPP row shifts instead of through LDS
  const int sm_k0 = ((lane >> 2) & 3) * 4 * CPW;             // first k of the lane's sub-slice inside the wave's slice
  const int sm_n = (lane >> 4) * 4 + (lane & 3);             // its column inside the 16-column tile
  const unsigned w_voff = SMALL ? (unsigned)((c0 + sm_k0 / 16) * 64 + ((sm_k0 % 16) / 4) * 16 + sm_n) * 16u
                                : (unsigned)(c0 * 64 + lane) * 16u;
  f32x4 w_t1[CPW], w_t0[CPW], w_b[CPW];
  // MFMA A-operand addresses (LDS floats): row = lane & 15, this wave's K range
  // (4x4 blocks: row = clip lane % 4, the lane's own 4 CPW consecutive k)
  constexpr int kXStride = SMALL ? 4 : 16;                   // floats between a lane's consecutive A-operand reads
  const int x_off = SMALL ? (lane & 3) * ldh + c0 * 16 + sm_k0 : (lane & 15) * ldh + c0 * 16 + 4 * (lane >> 4);
  const int xy_off = SMALL ? (lane & 3) * ldy + c0 * 16 + sm_k0 : (lane & 15) * ldy + c0 * 16 + 4 * (lane >> 4);
  unsigned cur_hasb = 0, nx_hasb = 0;                      // uniform: does this workgroup have B rows in the layer
  u64 rq_A = 0, rq_B = 0, rq_Bn = 0;
  // a request in two parts: `prepare` reads the layer's table entry into SGPRs and asks for the small pieces
  // (delayed input, conditioning terms); the weight fragments of the NEXT layer are loaded straight into the
  // registers of the fragment that the MFMA chain has just consumed (no second register set, no copies): one
  // 1-KiB load per four MFMAs in phase A, the B fragments right after phase B.  The texture path moves the
  // workgroup's 48 KiB of weights per layer (~0.3 us at 64 B/clk) under the matrix pipe's shadow, and the
  // compiler's own vmcnt bookkeeping makes each MFMA group wait for exactly its fragment.
  auto prepare = [&](int nl, unsigned ntau, int64_t cidx, bool with_cond) {   // cidx = step * L + layer of the request
    const u32x4 e0 = reinterpret_cast<const u32x4*>(ltab)[2 * nl];
    const u32x4 e1 = reinterpret_cast<const u32x4*>(ltab)[2 * nl + 1];
    rq_B = rq_Bn;                                             // B tile of the layer that starts now
    rq_A = ((u64)sgpr(e0[1]) << 32) | sgpr(e0[0]);           // A and B tiles of the next one
    rq_Bn = ((u64)sgpr(e0[3]) << 32) | sgpr(e0[2]);
    const unsigned ring_off = sgpr(e1[0]), dil = sgpr(e1[1]), mask = sgpr(e1[2]);
    nx_hasb = sgpr(e1[3]);
    gf32x4_ptr src = (gf32x4_ptr)(uintptr_t)(h_ring + ring_off + (u64)((ntau - dil) & mask) * slot_bytes);
#pragma unroll
    for (int k = 0; k < kHP; ++k) hp[k] = src[min(mt + k * NMT, slot_f4 - 1)];
#pragma unroll
    for (int k = 0; k < kCP; ++k) {
      cnd_n[k] = *(with_cond ? cptr[k] + (int64_t)cstep[k] * cidx : (gcfloat_ptr)(uintptr_t)a.zeros);
    }
  };
  auto frag_A = [&](int i) -> f32x4 {   // fragment i of the requested layer: tap 0 chunks, then tap 1 chunks
    gf32x4_ptr A0 = (gf32x4_ptr)(uintptr_t)(rq_A + w_voff);
    gf32x4_ptr A1 = (gf32x4_ptr)(uintptr_t)(rq_A + (u64)kcC * 1024 + w_voff);
    return i < CPW ? A0[i * kFragStride] : A1[(i - CPW) * kFragStride];
  };
  auto frag_B = [&](int i) -> f32x4 {
    gf32x4_ptr B0 = (gf32x4_ptr)(uintptr_t)(rq_B + w_voff);
    return B0[i * kFragStride];
  };
  // the requested layer's small operands -> LDS (their loads were issued before the weight loads, so the
  // compiler's wait leaves the weights in flight)
  auto small_to_lds = [&]() {
#pragma unroll
    for (int k = 0; k < kHP; ++k) {
      const int q = mt + k * NMT;
      if (q < slot_f4) *reinterpret_cast<f32x4*>(hprev + (q / (C / 4)) * ldh + (q % (C / 4)) * 4) = hp[k];
    }
#pragma unroll
    for (int k = 0; k < kCP; ++k)
      if (mt + k * NMT < mg * 16) cndbuf[mt + k * NMT] = cnd_n[k];
  };
  // diagnostic build only: 100 MHz wall-clock stamps of thread 0 (an I/O wave) of owner 1 of group 0, summed
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
  float skipacc = 0.f;                           // element threads of skip-row owners
  const int64_t tau0 = a.t0 - 1;

  if (!is_io) {   // layer 0 of the first step: requested and put in place right away (once per launch)
    prepare(0, (unsigned)tau0, 0, true);
#pragma unroll
    for (int u = 0; u < CPW; ++u) { w_t0[u] = frag_A(u); w_t1[u] = frag_A(CPW + u); }
    small_to_lds();
    cur_hasb = nx_hasb;
  }

  // Teacher-forced warm-up (filling the history rings from a prompt): position tau only has to go through the
  // layers whose output is still needed when generation starts at tf_end - the output of layer l at tau matters iff
  // tf_end - tau <= (sum of the dilations above l).  The top layer never runs, the bottom block sees the whole
  // prompt: 2/3 of the layer-positions for three equal blocks.  Lrun is uniform and never decreases.
  int Lrun = a.teacher_forced ? 1 : L;
  for (int64_t s = 0; s < a.n_steps; ++s) {
    const int64_t tau = tau0 + s;
    const unsigned tau_u = (unsigned)tau;
    if (a.teacher_forced)
      while (Lrun < L && (int64_t)sfx[Lrun] >= a.tf_end - tau) ++Lrun;
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
          if (spins > kSpinLimit || (MMK_WAIT_ERR_LOOK && (spins & 255u) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
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
    if (is_io) {
      const u32x4 e1 = reinterpret_cast<const u32x4*>(ltab)[1];
      gf32x4_wptr ring0 = (gf32x4_wptr)(uintptr_t)(h_ring + sgpr(e1[0]) + (u64)(tau_u & sgpr(e1[2])) * slot_bytes);
      for (int q = tid; q < slot_f4; q += NIO) {
        const int m = q / (C / 4), c = (q % (C / 4)) * 4;
        const int cls = s_idx[m];
        // torch raises on an out-of-range class; keep memory safe and make it visible (NaN row)
        const float nanv = __builtin_nanf("");
        const f32x4 v = (cls >= 0 && cls < a.q_levels) ? *reinterpret_cast<const f32x4*>(a.emb + (int64_t)cls * C + c)
                                                       : f32x4{nanv, nanv, nanv, nanv};
        *reinterpret_cast<f32x4*>(hbuf + m * ldh + c) = v;
        ring0[q] = v;
      }
    }
    __syncthreads();
    if (STAMPS && s > 0) stamp(7);   // wait for the sampled classes + embedding rows
    if (STAMPS) st_prev = wall_clock64();

    if (is_io) {
      // ================================ I/O waves ================================
      for (int l = 0;; ++l) {
        const unsigned epoch = (unsigned)(s * L + l + 1);
        const bool last = (l + 1 == L);
        const bool last_run = (l + 1 == Lrun);
        const u32x4 e1 = reinterpret_cast<const u32x4*>(ltab)[2 * l + 1];
        const unsigned ring_off = sgpr(e1[0]), ring_mask = sgpr(e1[2]), hasb = sgpr(e1[3]);
        __syncthreads();                                   // B1: phase-A partials are in LDS
        stamp(0);   // wait for phase A
        u64* gran_y = gran_y0 + (l & 1) * 16 * C;
        if (elem) {
          const float f = sum_partials(redA) + cndbuf[tid] + biasA[l * 16 + e_n];
          const float act = fmaf(__frcp_rn(1.0f + __builtin_amdgcn_exp2f(f * gate_scale)), gate_k, gate_shift);
          // lane i takes lane i+1's value (row_shl:1): the even lane multiplies tanh(f) by its neighbour's sigmoid(g)
          const float other = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(act), 0x101, 0xf, 0xf, false));
          if (!(e_n & 1)) gran_store<XCD>(gran_y + y_slot, epoch, act * other);
        }
        __syncthreads();                                   // B1b: y is on its way - the matrix waves may use the memory pipe
        // this layer's input joins its history ring while the other workgroups' y slices are on their way
        if (l > 0) {
          gf32x4_wptr dst = (gf32x4_wptr)(uintptr_t)(h_ring + ring_off + (u64)(tau_u & ring_mask) * slot_bytes);
          for (int q = tid; q < slot_f4; q += NIO)
            dst[q] = *reinterpret_cast<const f32x4*>(hbuf + (q / (C / 4)) * ldh + (q % (C / 4)) * 4);
        }
        stamp(1);   // epilogue A + publish + ring store
        if (hasb) {
          if (!sweep<NIO>(gran_y, mg * C, epoch, sw_y, ybuf, C, ldy, err, s_fail)) return;   // ... B2
          stamp(2);   // wait y
          __syncthreads();                                 // B3: phase-B partials are in LDS
          stamp(3);   // wait for phase B
          if (elem) {
            const float vb = sum_partials(redB) + biasB[l * 16 + e_n];
            if (owns_res)
              gran_store<XCD>(gran_h + h_slot, epoch, hbuf[res_off] + vb);
            else
              skipacc = (l == 0) ? vb : vb + skipacc;
          }
          __syncthreads();                                 // B3b: h' is on its way
          stamp(4);   // epilogue B + publish
        }
        if (!last) {
          if (!sweep<NIO>(gran_h, mg * C, epoch, sw_h, hbuf, C, ldh, err, s_fail)) return;   // ... B4
          stamp(5);   // wait h'
          if (last_run) {   // warm-up stops here: the input of the first layer that does not run still joins its ring
            const u32x4 en = reinterpret_cast<const u32x4*>(ltab)[2 * (l + 1) + 1];
            gf32x4_wptr dst = (gf32x4_wptr)(uintptr_t)(h_ring + sgpr(en[0]) + (u64)(tau_u & sgpr(en[2])) * slot_bytes);
            for (int q = tid; q < slot_f4; q += NIO)
              dst[q] = *reinterpret_cast<const f32x4*>(hbuf + (q / (C / 4)) * ldh + (q % (C / 4)) * 4);
            break;
          }
        } else {
          __syncthreads();                                 // B4
          break;
        }
      }
    } else {
      // =============================== matrix waves ===============================
      // A counted loop without exits: every iteration issues the same loads in the same order, so the compiler's
      // vmcnt bookkeeping is exact and an MFMA group waits for its own fragment only (with an exit or a skipped
      // load on some path it has to assume the shortest one, and phase A ends up waiting for the delayed-input
      // load issued just before it).  A hand-off timeout is noticed after the loop.
      for (int l = 0; l < Lrun; ++l) {
        const bool last = (l + 1 == Lrun);
        // The very last layer of a launch re-requests layer 0 (unused) so that the sequence stays branch-free.
        prepare(last ? 0 : l + 1, last ? tau_u + 1 : tau_u, last ? (s + 1) * L : s * L + l + 1, !last || s + 1 < a.n_steps);
#pragma unroll
        for (int u = 0; u < CPW; ++u) w_b[u] = frag_B(u);   // this layer's B fragments: needed at B2, 1.5 us from here
        __builtin_amdgcn_sched_barrier(0);
        stamp(8);    // requests issued
        // ---- phase A: z = W0.h[tau-d] + W1.h[tau]  (K = 2C, this wave's chunks) ----
        {
          const float* x0 = hprev + x_off;
          const float* x1 = hbuf + x_off;
          f32x4 xa[CPW], xb[CPW];
#pragma unroll
          for (int u = 0; u < CPW; ++u) {
            xa[u] = *reinterpret_cast<const f32x4*>(x0 + u * kXStride);
            xb[u] = *reinterpret_cast<const f32x4*>(x1 + u * kXStride);
          }
          __builtin_amdgcn_sched_barrier(0);
          f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int u = 0; u < CPW; ++u) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc = mma_step<SMALL>(xa[u][i], w_t0[u][i], acc);
          }
#pragma unroll
          for (int u = 0; u < CPW; ++u) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc = mma_step<SMALL>(xb[u][i], w_t1[u][i], acc);
          }
          redA[mwave * 64 + lane] = SMALL ? reduce_subslices(acc) : acc;
        }
        stamp(9);    // phase A: operand reads, (wait for the fragments), MFMAs with the next fragments' loads
        __syncthreads();                                   // B1
        __syncthreads();                                   // B1b: the I/O waves have published y
        // The workgroup shares one memory pipe, in order (requests AND returns): a burst of fragment loads in
        // front of a publish delays the publish, and a poll behind a slow load returns after it; and a layer's
        // 1.5 MB per XCD need about a microsecond of the XCD's fabric link.  So the stream is cut in three
        // 16-KiB pieces per workgroup, each issued right after a hand-off completed or a publish left:
        // this layer's B fragments at the layer's top, the next layer's tap-0 fragments behind publish y,
        // its tap-1 fragments behind publish h'.
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < CPW; ++u) w_t0[u] = frag_A(u);
        __builtin_amdgcn_sched_barrier(0);
        stamp(10);   // epilogue A of the I/O waves
        if (cur_hasb) {
          __syncthreads();                                 // B2: y is in LDS
          stamp(11);   // wait for y
          {
            f32x4 xv[CPW];
#pragma unroll
            for (int u = 0; u < CPW; ++u) xv[u] = *reinterpret_cast<const f32x4*>(ybuf + xy_off + u * kXStride);
            __builtin_amdgcn_sched_barrier(0);
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < CPW; ++u) {
#pragma unroll
              for (int i = 0; i < 4; ++i) acc = mma_step<SMALL>(xv[u][i], w_b[u][i], acc);
            }
            redB[mwave * 64 + lane] = SMALL ? reduce_subslices(acc) : acc;
          }
        }
        if (cur_hasb) {
          __syncthreads();                                 // B3
          __syncthreads();                                 // B3b: the I/O waves have published h'
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < CPW; ++u) w_t1[u] = frag_A(CPW + u);
        __builtin_amdgcn_sched_barrier(0);
        small_to_lds();
        stamp(12);   // phase B + B3 + small operands to LDS
        __syncthreads();                                   // B4
        stamp(13);   // wait for h'
        cur_hasb = nx_hasb;
      }
      if (*s_fail) return;
    }
    if (a.teacher_forced) continue;

    // ---- head -----------------------------------------------------------------------------------
    const unsigned he = (unsigned)(s + 1);
    if (elem && !owns_res) gran_store<XCD>(gran_skip + e_m * C + (j - kcC) * 16 + e_n, he, skipacc);
    // fc0 + Mish : tiles j, j+Gn, ... of H1/16
    if (j < t_fc0) {
      if (!sweep<NT>(gran_skip, mg * C, he, sw_y, ybuf, C, ldy, err, s_fail)) return;
      const int per = (kc_fc0 + nw - 1) / nw;
      const int k0 = min(wave * per, kc_fc0), k1 = min(k0 + per, kc_fc0);
      for (int t = j, ti = 0; t < t_fc0; t += a.Gn, ++ti) {
        f32x4 v = reduce_waves(tile_mma(ybuf, ldy, hw0 + ti * kc_fc0 * 64, 0, k0, k1, lane), redA, wave, lane, nw);
        if (wave == 0) {
          const float bias = hb0[ti * 16 + D_n];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int m = 4 * D_q + r;
            if (m < mg) gran_store<XCD>(gran_hid + m * a.H1 + t * 16 + D_n, he, mish_fast(v[r] + bias));
          }
        }
      }
    }
    stamp(16);    // head: wait for the skip sums + fc0
    // fc2 : tiles of the (n_classes + temperature column) outputs
    if (j < t_fc2) {
      if (!sweep<NT>(gran_hid, mg * a.H1, he, nullptr, ybuf, a.H1, ldy, err, s_fail)) return;
      const int per = (kc_fc2 + nw - 1) / nw;
      const int k0 = min(wave * per, kc_fc2), k1 = min(k0 + per, kc_fc2);
      for (int t = j, ti = 0; t < t_fc2; t += a.Gn, ++ti) {
        f32x4 v = reduce_waves(tile_mma(ybuf, ldy, hw2 + ti * kc_fc2 * 64, 0, k0, k1, lane), redA, wave, lane, nw);
        if (wave == 0) {
          const float bias = hb2[ti * 16 + D_n];
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
      if (!sweep<NT>(gran_logit, mg * a.n_logits_pad, he, nullptr, lbuf, a.n_logits_pad, ldl, err, s_fail)) return;
      const int nc = a.n_classes;
      const int per = (nc + 63) / 64;
      for (int m = wave; m < mg; m += nw) {
        const float* lg = lbuf + m * ldl;
        const int clip = m_first + m;
        // the caller can ask for the logits of the LAST step (mmk_wavenet_last_logits): stored after the class is out
        const bool keep_logits = a.logits_out && s + 1 == a.n_steps;
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
          // first maximum wins (torch.argmax): inside a row of 16 lanes through DPP (VALU), across rows through LDS permutes
          auto take = [&](float ob, int oi) {
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
          };
#define MMK_DPP_STEP(CTRL)                                                                                           \
          take(__int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(best), CTRL, 0xf, 0xf, false)),            \
               __builtin_amdgcn_update_dpp(0, bi, CTRL, 0xf, 0xf, false))
          MMK_DPP_STEP(0xB1);    // quad_perm [1,0,3,2]
          MMK_DPP_STEP(0x4E);    // quad_perm [2,3,0,1]
          MMK_DPP_STEP(0x141);   // row_half_mirror
          MMK_DPP_STEP(0x140);   // row_mirror
#undef MMK_DPP_STEP
#pragma unroll
          for (int o = 16; o <= 32; o <<= 1) {
            const float ob = __shfl_xor(best, o);
            const int oi = __shfl_xor(bi, o);
            take(ob, oi);
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
          gran_store_u32<XCD>(gran_idx + m, he, (unsigned)result);   // first: every workgroup of the group waits for it
          a.idx[(int64_t)clip * a.idx_rs + tau + 1] = result;
        }
        if (keep_logits)
          for (int c = lane; c < nc + a.learn_temp; c += 64) a.logits_out[(int64_t)clip * a.logits_ld + c] = lg[c];
      }
    }
    __syncthreads();
    stamp(6);     // head (as seen by this workgroup)
  }
  if (STAMPS && a.stamps && g == 0 && j == 1) {
    if (tid == 0) {
      st_acc[14] = clock64() - clk_start;        // shader cycles
      st_acc[15] = wall_clock64() - wall_start;  // 100 MHz ticks
      for (int i = 0; i < 8; ++i) a.stamps[i] = st_acc[i];
      a.stamps[14] = st_acc[14];
      a.stamps[15] = st_acc[15];
      a.stamps[16] = st_acc[16];
    }
    if (tid == NIO)
      for (int i = 8; i < 14; ++i) a.stamps[i] = st_acc[i];
  }
}

size_t wn_persist_lds_bytes(const WnPersistArgs& a) {
  const int kc = a.C / 16, nwm = kc / wn_cpw(kc), nw = kIoWaves + nwm;
  const int wide = a.C > a.H1 ? a.C : a.H1;
  const int ldh = a.C + 4, ldy = wide + 4, ldl = a.n_logits_pad + 4;
  const int red = (2 * nwm > nw ? 2 * nwm : nw) * 64;
  return (size_t)2 * 16 * ldh * 4 + (size_t)16 * ldy * 4 + (size_t)red * 16 + (size_t)a.L * 32 + (size_t)a.L * 128 +
         256 * 4 + (size_t)((a.L * 4 + 15) / 16) * 16 + 16 * 4 + 16 + (size_t)16 * ldl * 4 +
         (size_t)((a.H1 / 16 + a.Gn - 1) / a.Gn) * (kc * 1024 + 64) + (size_t)((a.n_logits_pad / 16 + a.Gn - 1) / a.Gn) * ((a.H1 / 16) * 1024 + 64);
}

int launch_wavenet_persist(const WnPersistArgs& a, hipStream_t stream) {
  const int kc = a.C / 16;
  if (kc < 2 || kc > 16 || a.C % 32 || a.S != a.C) return fail(MMK_ERR_UNSUPPORTED, "persistent WaveNet: C=%d not in {32..256 step 32}", a.C);
  if (a.Mg > 16) return fail(MMK_ERR_UNSUPPORTED, "persistent WaveNet: %d clips per group", a.Mg);
  const size_t lds = wn_persist_lds_bytes(a);
  if (lds > 160 * 1024) return fail(MMK_ERR_UNSUPPORTED, "persistent WaveNet: %zu bytes of LDS needed", lds);
  dim3 grid(a.Gc * a.Gn), block(wn_threads(kc));
#define MMK_WNP2(KC_, SM_)                                                                                             \
  do {                                                                                                                 \
    if (a.stamps) {                                                                                                    \
      if (a.xcd_local) hipLaunchKernelGGL((wavenet_persist_kernel<KC_, SM_, true, true>), grid, block, lds, stream, a);     \
      else hipLaunchKernelGGL((wavenet_persist_kernel<KC_, SM_, true, false>), grid, block, lds, stream, a);                \
    } else {                                                                                                           \
      if (a.xcd_local) hipLaunchKernelGGL((wavenet_persist_kernel<KC_, SM_, false, true>), grid, block, lds, stream, a);    \
      else hipLaunchKernelGGL((wavenet_persist_kernel<KC_, SM_, false, false>), grid, block, lds, stream, a);               \
    }                                                                                                                  \
  } while (0)
#define MMK_WNP(KC_)                   \
  do {                                 \
    if (small) MMK_WNP2(KC_, true);    \
    else MMK_WNP2(KC_, false);         \
  } while (0)
  // groups of at most 4 clips use 4x4 MFMA blocks instead of 16-row tiles (MMK_WN_SMALL=0 forces the tiles)
  const bool small = a.Mg <= 4 && !a.force_tiles;
  switch (kc) {
    case 2: MMK_WNP(2); break;
    case 4: MMK_WNP(4); break;
    case 6: MMK_WNP(6); break;
    case 8: MMK_WNP(8); break;
    case 10: MMK_WNP(10); break;
    case 12: MMK_WNP(12); break;
    case 14: MMK_WNP(14); break;
    default: MMK_WNP(16); break;
  }
#undef MMK_WNP
#undef MMK_WNP2
  MMK_HIP(hipGetLastError());
  return MMK_OK;
}

}  // namespace mmk

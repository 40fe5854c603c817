// Persistent WaveNet step kernel with ONE hand-off per layer on the per-sample chain (gfx950).
//
// wavenet_persist.hip pays two dependent all-to-all exchanges per layer: y (after the gate) and h' (after the
// residual 1x1).  Only the gate is non-linear, so the second one can leave the chain:
//
//     h_{l+1}[t] = h_l[t] + R_l y_l[t] + r_l                                           (wavenet_v2.py:172-176)
//     z_{l+1}[t] = W0 h_{l+1}[t-d] + W1 h_{l+1}[t] + c_{l+1}[t] + b                      (:141-150)
//                = W0 h_{l+1}[t-d] + W1 h_l[t] + (W1 R_l) y_l[t] + c_{l+1}[t] + (b + W1 r_l)
//
// The plan pre-multiplies W1 R_l and W1 r_l (fp64 accumulation, rounded to fp32 once; wavenet_plan.hip), and iteration
// l + 1 of a step multiplies the K = 3C operand [h_{l+1}[t-d] | h_l[t] | y_l[t]].  h_{l+1}[t] itself is still computed
// (the history ring and the next residual need it) but BESIDE the chain: it is published together with y_{l+1} and
// first read one iteration later.  A step is then L + 1 iterations,
//
//     iteration i:   gate product of layer i (i < L)            -> epilogue -> publish y_i
//                    [res ; skip] product of layer i-1 (i >= 1) -> publish h_i / accumulate the skip sums
//                    one sweep for y_i and h_i
//
// with one exchange wait each, instead of L layers x two.  Same operands as the reference in a different association
// (the pre-multiplied matrix), fp32 throughout; classes stay bit-exact on the margin-checked goldens.
//
// Work split inside a workgroup (768 threads): 4 I/O waves (hand-offs, epilogues, ring stores - they never touch the
// weight stream) and 8 matrix waves in four pairs: K segment h[t-d], K segment h[t], K segment y of the gate product,
// and the [res ; skip] product; each wave multiplies half a segment with v_mfma_f32_4x4x1_16b_f32 blocks (groups of
// <= 4 clips, as wavenet_persist.hip's SMALL mode).  Three workgroup barriers per iteration:
//     B1 partial sums in LDS | B1b published - the matrix waves may use the memory pipe | B4 next operands in LDS
// The weight stream of an iteration (64 KiB per workgroup at C = 256) is cut in two pieces so that neither a publish
// nor a poll queues behind it in the CU's in-order memory pipe: the first half of the NEXT tile is requested at the
// top of an iteration into staging registers (lands under the MFMA / epilogue phase), the second half right behind
// the publish (lands under the exchange wait).
//
// Everything else - granules, XCD-local placement check, private history rings, hoisted conditioning products, head,
// error handling - is wavenet_persist.hip's, and both kernels share the ring layout, so either can continue the
// other's state (the warm-up stays a prefill or the teacher-forced mode of wavenet_persist.hip).
#include "wavenet_chain.h"
#include "wavenet_handoff.h"
#include "sampler256.h"

namespace mmk {

constexpr int kChIo = 4;                            // waves 0..3
constexpr int kChMat = 8;                           // waves 4..11: pairs (h[t-d] | h[t] | y | [res ; skip])
constexpr int kChThreads = 64 * (kChIo + kChMat);

struct __attribute__((aligned(16))) ChEntry {
  unsigned A_lo, A_hi;     // this workgroup's tile of the iteration's gate matrix (byte address; always a valid tile)
  unsigned B_lo, B_hi;     // its tile of the [res ; skip] matrix of the layer below
  unsigned ring_off;       // byte offset of the layer's input-history ring inside the workgroup's block
  unsigned dil, mask;
  unsigned flags;          // 1: the iteration has a gate product, 2: this workgroup has [res ; skip] rows in it
};

// y and h granules of one iteration in one pass: every thread keeps up to eight 8-byte granules (four 16-byte loads)
// in flight per round.  Workgroup-uniform success flag; ends with the workgroup barrier B4.
template <int NTH>
__device__ __forceinline__ bool sweep_pair(const u64* gy, const u64* gh, bool with_h, int count, unsigned epoch, float* dy,
                                           float* dh, int* err_flag, int* s_fail) {
  const int tid = threadIdx.x;
  typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
  if (tid < NTH && tid * 4 < count) {
    const u64* py = gy + tid * 4;
    const u64* ph = with_h ? gh + tid * 4 : py;     // no h this iteration: read y twice (same tags)
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
      if (spins > kSpinLimit || (MMK_WAIT_ERR_LOOK && (spins & 255u) == 0 && __hip_atomic_load(err_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
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
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) on every path (see sweep())
  __syncthreads();
  return *s_fail == 0;
}

template <int KC, bool STAMPS, bool XCD>
__global__ __launch_bounds__(kChThreads) void wavenet_chain_kernel(const WnChainArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr int CPW = KC / 2;          // K-chunks per matrix wave: a pair of waves covers one K = C segment
  constexpr int H0 = (CPW + 1) / 2;    // fragments of the next tile requested at the top of an iteration (staged); the rest follows the publish
  constexpr int NIO = kChIo * 64, NT = kChThreads, nw = kChIo + kChMat;
  constexpr int NMT = kChMat * 64;
  constexpr int C = 16 * KC;
  constexpr int ldh = C + 4;
  constexpr int kRows = 4;             // clips per group
  static_assert(kRows * (C / 4) <= NMT && kRows * (C / 4) <= NIO * 4, "one ring piece per matrix thread");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_io = wave < kChIo;
  int g = blockIdx.x / a.Gn;
  int j = blockIdx.x % a.Gn;
  if (XCD) {   // roles follow the hardware placement, verified (see wavenet_persist.hip)
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
  const int mg = min(a.Mg, a.B - m_first);
  if (mg <= 0) return;

  // ---- LDS carve ------------------------------------------------------------------------------------
  const int wide = max(C, a.H1);
  const int ldy = wide + 4, ldl = a.n_logits_pad + 4;
  char* sp = smem_raw;
  float* hbuf = (float*)sp;   sp += 2 * kRows * ldh * 4;          // h_{i-1}[tau] / h_i[tau], by iteration parity
  float* hprev = (float*)sp;  sp += kRows * ldh * 4;              // h_i[tau - d_i]
  float* ybuf = (float*)sp;   sp += 2 * kRows * ldh * 4;          // y_{i-1} / y_i
  f32x4* red = (f32x4*)sp;    sp += nw * 64 * 16;                 // partial sums: [matrix wave][64]; head: [wave][64]
  ChEntry* tab = (ChEntry*)sp;            sp += (L + 1) * 32;
  float* biasA = (float*)sp;              sp += (L + 1) * 16 * 4;
  float* biasB = (float*)sp;              sp += (L + 1) * 16 * 4;
  float* cndbuf = (float*)sp;             sp += 2 * 64 * 4;       // conditioning term per gate element, by iteration parity
  int* s_idx = (int*)sp;      sp += 16 * 4;
  int* s_fail = (int*)sp;     sp += 16;
  float* headbuf = (float*)sp; sp += 16 * ldy * 4;                // skip sums / hidden units for the head's 16-row tiles
  float* lbuf = (float*)sp;   sp += 16 * ldl * 4;
  const int t_fc0 = a.H1 / 16, kc_fc0 = C / 16;
  const int t_fc2 = a.n_logits_pad / 16, kc_fc2 = a.H1 / 16;
  const int nt0 = j < t_fc0 ? (t_fc0 - j + a.Gn - 1) / a.Gn : 0, nt2 = j < t_fc2 ? (t_fc2 - j + a.Gn - 1) / a.Gn : 0;
  f32x4* hw0 = (f32x4*)sp;    sp += (size_t)((t_fc0 + a.Gn - 1) / a.Gn) * kc_fc0 * 1024;
  f32x4* hw2 = (f32x4*)sp;    sp += (size_t)((t_fc2 + a.Gn - 1) / a.Gn) * kc_fc2 * 1024;
  float* hb0 = (float*)sp;    sp += (size_t)((t_fc0 + a.Gn - 1) / a.Gn) * 64;
  float* hb2 = (float*)sp;

  const int D_q = lane >> 4, D_n = lane & 15;
  const bool owns_res = j < KC;           // owners [0, C/16) hold residual rows of B, the others skip rows
  const bool has_cond = a.C1 > 0;

  for (int i = tid; i < 2 * kRows * ldh; i += NT) hbuf[i] = 0.f;
  for (int i = tid; i < kRows * ldh; i += NT) hprev[i] = 0.f;
  for (int i = tid; i < 2 * kRows * ldh; i += NT) ybuf[i] = 0.f;
  for (int i = tid; i < 16 * ldy; i += NT) headbuf[i] = 0.f;
  for (int i = tid; i < 128; i += NT) cndbuf[i] = 0.f;
  for (int i = tid; i < 16 * ldl; i += NT) lbuf[i] = 0.f;
  for (int i = tid; i <= L; i += NT) {
    // iteration i: gate tile of layer i (i == L: layer 0's, the tile the next step starts with - a prefetch),
    //              [res ; skip] tile of layer i - 1 (none: the tile of the next iteration that has one)
    const WnChainIter ta = a.iters[i < L ? i : 0];
    const bool has_b = i >= 1 && (!owns_res || a.iters[i].prev_has_res);
    int ib = i;
    if (!has_b) ib = (i == 0 || i + 1 > L) ? 1 : i + 1;   // (i == L for residual-row owners: back to iteration 1's tile)
    const WnChainIter tb = a.iters[ib];
    const int btile = owns_res ? j : (j - KC + (tb.prev_has_res ? KC : 0));
    const uintptr_t Ap = (uintptr_t)(ta.A_wp + (int64_t)j * (3 * KC) * 256);
    const uintptr_t Bp = (uintptr_t)(tb.B_wp + (int64_t)btile * KC * 256);
    ChEntry e;
    e.A_lo = (unsigned)Ap; e.A_hi = (unsigned)(Ap >> 32);
    e.B_lo = (unsigned)Bp; e.B_hi = (unsigned)(Bp >> 32);
    e.ring_off = (unsigned)(ta.ring_offset * 4);
    e.dil = (unsigned)ta.dil; e.mask = (unsigned)ta.ring_mask;
    e.flags = (i < L ? 1u : 0u) | (has_b ? 2u : 0u);
    tab[i] = e;
  }
  for (int q = tid; q < (L + 1) * 16; q += NT) {
    const int i = q >> 4, n = q & 15;
    const WnChainIter t = a.iters[i];
    biasA[q] = (i < L && t.A_bias) ? t.A_bias[j * 16 + n] : 0.f;
    const bool has_b = i >= 1 && (!owns_res || t.prev_has_res);
    const int btile = owns_res ? j : (j - KC + (t.prev_has_res ? KC : 0));
    biasB[q] = (has_b && t.B_bias) ? t.B_bias[btile * 16 + n] : 0.f;
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

  // ---- per-group exchange buffers, per-workgroup private history ring ---------------------------
  u64* gran_h0 = a.gran_h + (int64_t)g * 2 * 16 * C;
  u64* gran_y0 = a.gran_y + (int64_t)g * 2 * 16 * C;
  u64* gran_skip = a.gran_skip + (int64_t)g * 16 * C;
  u64* gran_hid = a.gran_hid + (int64_t)g * 16 * a.H1;
  u64* gran_logit = a.gran_logit + (int64_t)g * 16 * a.n_logits_pad;
  u64* gran_idx = a.gran_idx + (int64_t)g * 16;
  char* h_ring = (char*)(a.h_rings + (int64_t)(g * a.Gn + j) * a.ring_floats_per_wg);
  const unsigned slot_bytes = (unsigned)a.Mg * C * 4;
  const int slot_f4 = mg * (C / 4);
  int* err = a.err_flag;

  // epilogue element of a lane (waves 0 and 1): (clip e_m, column e_n) of the 16-column tile
  const int e_m = lane >> 4, e_n = lane & 15;
  const bool elem = e_m < mg;
  // (clip m, col n) of a 4x4-block accumulator image sits in lane 16 (n / 4) + 12 + n % 4 (the sub-slice-3 lane holds
  // the sum over the four K sub-slices), register m
  const int frag = ((e_n >> 2) * 16 + 12 + (e_n & 3)) * 4 + e_m;
  const float gate_k = (e_n & 1) ? 1.f : 2.f;      // tanh(x) = 2 sigmoid(2x) - 1: one code path for both gate halves
  const float gate_scale = -gate_k * 1.4426950408889634f;
  const float gate_shift = 1.f - gate_k;
  const unsigned y_slot = (unsigned)(e_m * C + j * 8 + (e_n >> 1));
  const unsigned h_slot = (unsigned)(e_m * C + j * 16 + e_n);
  const int res_off = e_m * ldh + j * 16 + e_n;
  const int sw_row = (tid * 4) / C, sw_col = tid * 4 - sw_row * C;    // the thread's four granules in a sweep
  const int sw_off = sw_row * ldh + sw_col;

  // ---- matrix-wave state ------------------------------------------------------------------------------
  const int mt = tid - NIO;
  const int mwave = __builtin_amdgcn_readfirstlane(mt >> 6);
  const int role = mwave >> 1, half = mwave & 1;          // role 0: h[t-d], 1: h[t], 2: y, 3: [res ; skip]
  const int sm_k0 = ((lane >> 2) & 3) * 4 * CPW;          // first k of the lane's K sub-slice inside the wave's slice
  const int sm_n = (lane >> 4) * 4 + (lane & 3);          // its column inside the 16-column tile
  const int cbase = (role < 3 ? role * KC : 0) + half * CPW;          // first chunk of the wave's slice inside the tile
  const unsigned w_voff = (unsigned)(cbase * 64 + (sm_k0 / 4) * 16 + sm_n) * 16u;
  const int x_off = (lane & 3) * ldh + half * CPW * 16 + sm_k0;
  const int64_t cond_clip = (int64_t)a.cond_steps * L * (2 * C);
  const bool c_real = has_cond && mt >= 0 && mt < mg * 16;
  gcfloat_ptr cptr = c_real ? (gcfloat_ptr)(uintptr_t)(a.condall + (int64_t)(m_first + (mt >> 4)) * cond_clip + j * 16 + (mt & 15))
                            : (gcfloat_ptr)(uintptr_t)a.zeros;
  const int cstep = c_real ? 2 * C : 0;
  f32x4 w_cur[CPW], w_stage[H0];
  f32x4 hp = f32x4{0.f, 0.f, 0.f, 0.f};
  float cnd = 0.f;
  u64 tile_next = 0;
  // Requests at the top of an iteration, for the NEXT one (ni): its delayed input (from the private ring), its
  // conditioning terms, and the first half of its weight tile (into staging registers).  Every matrix thread issues
  // the same loads in every iteration (clamped addresses, never predicated): the compiler's vmcnt bookkeeping stays exact.
  auto request_next = [&](int ni, unsigned ntau, int64_t cidx, bool with_cond) {
    const u32x4 e0 = reinterpret_cast<const u32x4*>(tab)[2 * ni];
    const u32x4 e1 = reinterpret_cast<const u32x4*>(tab)[2 * ni + 1];
    const u64 At = ((u64)sgpr(e0[1]) << 32) | sgpr(e0[0]);
    const u64 Bt = ((u64)sgpr(e0[3]) << 32) | sgpr(e0[2]);
    tile_next = role < 3 ? At : Bt;
    const unsigned ring_off = sgpr(e1[0]), dil = sgpr(e1[1]), mask = sgpr(e1[2]);
    gf32x4_ptr src = (gf32x4_ptr)(uintptr_t)(h_ring + ring_off + (u64)((ntau - dil) & mask) * slot_bytes);
    hp = src[min(mt, slot_f4 - 1)];
    cnd = *(with_cond ? cptr + (int64_t)cstep * cidx : (gcfloat_ptr)(uintptr_t)a.zeros);
    gf32x4_ptr wsrc = (gf32x4_ptr)(uintptr_t)(tile_next + w_voff);
#pragma unroll
    for (int u = 0; u < H0; ++u) w_stage[u] = wsrc[u * 16];
  };
  auto request_rest = [&]() {
    gf32x4_ptr wsrc = (gf32x4_ptr)(uintptr_t)(tile_next + w_voff);
#pragma unroll
    for (int u = H0; u < CPW; ++u) w_cur[u] = wsrc[u * 16];
  };
  auto small_to_lds = [&](int ni) {
    if (mt < slot_f4) *reinterpret_cast<f32x4*>(hprev + (mt / (C / 4)) * ldh + (mt % (C / 4)) * 4) = hp;
    if (mt < mg * 16) cndbuf[(ni & 1) * 64 + mt] = cnd;
  };
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
  const int64_t tau0 = a.t0 - 1;

  if (!is_io) {   // iteration 0 of the first step: everything requested and put in place right away (once per launch)
    request_next(0, (unsigned)tau0, 0, true);
    request_rest();
#pragma unroll
    for (int u = 0; u < H0; ++u) w_cur[u] = w_stage[u];
    small_to_lds(0);
  }

  // ---- head (as wavenet_persist.hip): skip sums -> fc0 + Mish -> fc2 -> temperature / argmax | sample; every wave
  // of the workgroup runs it (same barriers on both kinds of waves).  Returns false after a hand-off timeout.
  auto head = [&](int64_t s, int64_t tau, float skipacc) -> bool {
    const unsigned he = (unsigned)(s + 1);
    if (is_io && wave == 1 && elem && !owns_res) gran_store<XCD>(gran_skip + e_m * C + (j - KC) * 16 + e_n, he, skipacc);
    const int sw_row_y = (tid * 4) / C;
    float* const sw_head = headbuf + sw_row_y * ldy + (tid * 4 - sw_row_y * C);
    if (j < t_fc0) {
      if (!sweep<NT>(gran_skip, mg * C, he, sw_head, headbuf, C, ldy, err, s_fail)) return false;
      const int per = (kc_fc0 + nw - 1) / nw;
      const int k0 = min(wave * per, kc_fc0), k1 = min(k0 + per, kc_fc0);
      for (int t = j, ti = 0; t < t_fc0; t += a.Gn, ++ti) {
        f32x4 v = reduce_waves(tile_mma(headbuf, ldy, hw0 + ti * kc_fc0 * 64, 0, k0, k1, lane), red, wave, lane, nw);
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
    stamp(16);
    if (j < t_fc2) {
      if (!sweep<NT>(gran_hid, mg * a.H1, he, nullptr, headbuf, a.H1, ldy, err, s_fail)) return false;
      const int per = (kc_fc2 + nw - 1) / nw;
      const int k0 = min(wave * per, kc_fc2), k1 = min(k0 + per, kc_fc2);
      for (int t = j, ti = 0; t < t_fc2; t += a.Gn, ++ti) {
        f32x4 v = reduce_waves(tile_mma(headbuf, ldy, hw2 + ti * kc_fc2 * 64, 0, k0, k1, lane), red, wave, lane, nw);
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
    if (j == 0) {
      if (!sweep<NT>(gran_logit, mg * a.n_logits_pad, he, nullptr, lbuf, a.n_logits_pad, ldl, err, s_fail)) return false;
      const int nc = a.n_classes;
      const int per = (nc + 63) / 64;
      for (int m = wave; m < mg; m += nw) {
        const float* lg = lbuf + m * ldl;
        const int clip = m_first + m;
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
          auto take = [&](float ob, int oi) {     // first maximum wins (torch.argmax)
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
        } else if (nc == 256) {
          result = sample_256(lg, a.learn_temp != 0, denom, a.temperature[clip], a.uniforms[(int64_t)clip * a.uni_ld + s], lane);
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
    return true;
  };
  // The two kinds of waves run their own copy of the step loop (same sequence of workgroup barriers in both): what one kind
  // keeps in registers is not live in the other's code.
  if (is_io) {
    float skipacc = 0.f;
    for (int64_t s = 0; s < a.n_steps; ++s) {
      const int64_t tau = tau0 + s;
      const unsigned tau_u = (unsigned)tau;
      // ---- input 0: embedding row of the newest sample -> hbuf[0] (h_0[tau]) and ring 0 --------------
      if (s > 0) {
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
        const u32x4 e1 = reinterpret_cast<const u32x4*>(tab)[1];
        gf32x4_wptr ring0 = (gf32x4_wptr)(uintptr_t)(h_ring + sgpr(e1[0]) + (u64)(tau_u & sgpr(e1[2])) * slot_bytes);
        for (int q = tid; q < slot_f4; q += NIO) {
          const int m = q / (C / 4), c = (q % (C / 4)) * 4;
          const int cls = s_idx[m];
          const float nanv = __builtin_nanf("");     // torch raises on an out-of-range class; stay memory-safe and visible
          const f32x4 v = (cls >= 0 && cls < a.q_levels) ? *reinterpret_cast<const f32x4*>(a.emb + (int64_t)cls * C + c)
                                                         : f32x4{nanv, nanv, nanv, nanv};
          *reinterpret_cast<f32x4*>(hbuf + m * ldh + c) = v;
          ring0[q] = v;
        }
      }
      __syncthreads();
      if (STAMPS && s > 0) stamp(7);   // wait for the sampled classes + embedding rows
      if (STAMPS) st_prev = wall_clock64();

      // ================================ I/O waves ================================
      for (int i = 0; i <= L; ++i) {
        const unsigned epoch = (unsigned)(s * (L + 1) + i + 1);
        const int par = i & 1;
        const int hsel = i == 0 ? 0 : ((i - 1) & 1);            // where h_{i-1}[tau] lives (h_0 at i = 0)
        const u32x4 e1 = reinterpret_cast<const u32x4*>(tab)[2 * i + 1];
        const unsigned flags = sgpr(e1[3]);
        u64* gran_y = gran_y0 + par * 16 * C;
        u64* gran_h = gran_h0 + par * 16 * C;
        __syncthreads();                                   // B1: partial sums are in LDS
        stamp(0);
        if (wave == 0) {
          if ((flags & 1u) && elem) {
            const float* f = reinterpret_cast<const float*>(red) + frag;
            float pv[6];
#pragma unroll
            for (int w = 0; w < 6; ++w) pv[w] = f[w * 256];
            __builtin_amdgcn_sched_barrier(0);
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 6; ++w) v += pv[w];
            const float z = v + cndbuf[par * 64 + lane] + biasA[i * 16 + e_n];
            const float act = fmaf(__frcp_rn(1.0f + __builtin_amdgcn_exp2f(z * gate_scale)), gate_k, gate_shift);
            // lane n takes lane n+1's value (row_shl:1): the even lane multiplies tanh(f) by its neighbour's sigmoid(g)
            const float other = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(act), 0x101, 0xf, 0xf, false));
            if (!(e_n & 1)) gran_store<XCD>(gran_y + y_slot, epoch, act * other);
          }
        } else if (wave == 1) {
          if ((flags & 2u) && elem) {
            const float* f = reinterpret_cast<const float*>(red) + frag;
            const float vb = (f[6 * 256] + f[7 * 256]) + biasB[i * 16 + e_n];
            if (owns_res)
              gran_store<XCD>(gran_h + h_slot, epoch, hbuf[hsel * kRows * ldh + res_off] + vb);   // h_i = h_{i-1} + R y + r
            else
              skipacc = (i == 1) ? vb : vb + skipacc;
          }
        }
        __syncthreads();                                   // B1b: published - the matrix waves may use the memory pipe
        stamp(1);
        if (i >= 2) {   // h_{i-1}[tau] (complete since the last sweep) joins the history ring of layer i-1
          const u32x4 ep = reinterpret_cast<const u32x4*>(tab)[2 * (i - 1) + 1];
          gf32x4_wptr dst = (gf32x4_wptr)(uintptr_t)(h_ring + sgpr(ep[0]) + (u64)(tau_u & sgpr(ep[2])) * slot_bytes);
          const float* src = hbuf + hsel * kRows * ldh;
          for (int q = tid; q < slot_f4; q += NIO)
            dst[q] = *reinterpret_cast<const f32x4*>(src + (q / (C / 4)) * ldh + (q % (C / 4)) * 4);
        }
        stamp(2);
        if (i < L) {
          const bool with_h = i >= 1;                      // h_i exists for 1 <= i <= L-1 (chain mode: every such layer has a residual)
          if (!sweep_pair<NIO>(gran_y, gran_h, with_h, mg * C, epoch, ybuf + par * kRows * ldh + sw_off,
                               hbuf + par * kRows * ldh + sw_off, err, s_fail))
            return;                                        // ... B4
        } else {
          __syncthreads();                                 // B4
        }
        stamp(3);
      }
      if (!head(s, tau, skipacc)) return;
      stamp(6);
    }
  } else {
    for (int64_t s = 0; s < a.n_steps; ++s) {
      const int64_t tau = tau0 + s;
      const unsigned tau_u = (unsigned)tau;
      __syncthreads();                                     // (the I/O waves take in the sampled classes ...
      if (*s_fail) return;
      __syncthreads();                                     //  ... and fetch the embedding rows)
      // =============================== matrix waves ===============================
      for (int i = 0; i <= L; ++i) {
        const bool lastit = (i == L);
        const int ni = lastit ? 0 : i + 1;
        request_next(ni, lastit ? tau_u + 1 : tau_u, lastit ? (s + 1) * L : s * L + ni, ni < L && (!lastit || s + 1 < a.n_steps));
        __builtin_amdgcn_sched_barrier(0);
        stamp(8);
        {
          const int hsel = i == 0 ? 0 : ((i - 1) & 1);
          const float* xsrc = role == 0 ? hprev : (role == 1 ? hbuf + hsel * kRows * ldh : ybuf + ((i + 1) & 1) * kRows * ldh);
          f32x4 xv[CPW];
#pragma unroll
          for (int u = 0; u < CPW; ++u) xv[u] = *reinterpret_cast<const f32x4*>(xsrc + x_off + u * 4);
          __builtin_amdgcn_sched_barrier(0);
          f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int u = 0; u < CPW; ++u) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              if ((u * 4 + k) & 1) acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(xv[u][k], w_cur[u][k], acc1, 0, 0, 0);
              else acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(xv[u][k], w_cur[u][k], acc0, 0, 0, 0);
            }
          }
          f32x4 acc;
#pragma unroll
          for (int k = 0; k < 4; ++k) acc[k] = acc0[k] + acc1[k];
          red[mwave * 64 + lane] = reduce_subslices(acc);
        }
        stamp(9);
        __syncthreads();                                   // B1
        small_to_lds(ni);
        stamp(10);
        __syncthreads();                                   // B1b: the I/O waves have published
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < H0; ++u) w_cur[u] = w_stage[u];
        request_rest();
        __builtin_amdgcn_sched_barrier(0);
        stamp(11);
        __syncthreads();                                   // B4: y_i and h_i are in LDS
        stamp(12);
      }
      if (*s_fail) return;
      if (!head(s, tau, 0.f)) return;
    }
  }
  if (STAMPS && a.stamps && g == 0 && j == 1) {
    if (tid == 0) {
      st_acc[14] = clock64() - clk_start;
      st_acc[15] = wall_clock64() - wall_start;
      for (int i = 0; i < 8; ++i) a.stamps[i] = st_acc[i];
      a.stamps[14] = st_acc[14];
      a.stamps[15] = st_acc[15];
      a.stamps[16] = st_acc[16];
    }
    if (tid == NIO)
      for (int i = 8; i < 14; ++i) a.stamps[i] = st_acc[i];
  }
}

bool wn_chain_supported(int C, int Mg, int L) { return C >= 32 && C <= 256 && C % 32 == 0 && Mg >= 1 && Mg <= 4 && L >= 2; }

size_t wn_chain_lds_bytes(const WnChainArgs& a) {
  const int kc = a.C / 16, nw = kChIo + kChMat;
  const int wide = a.C > a.H1 ? a.C : a.H1;
  const int ldh = a.C + 4, ldy = wide + 4, ldl = a.n_logits_pad + 4;
  return (size_t)5 * 4 * ldh * 4 + (size_t)nw * 64 * 16 + (size_t)(a.L + 1) * (32 + 128) + 2 * 64 * 4 + 16 * 4 + 16 +
         (size_t)16 * ldy * 4 + (size_t)16 * ldl * 4 +
         (size_t)((a.H1 / 16 + a.Gn - 1) / a.Gn) * (kc * 1024 + 64) + (size_t)((a.n_logits_pad / 16 + a.Gn - 1) / a.Gn) * ((a.H1 / 16) * 1024 + 64);
}

int launch_wavenet_chain(const WnChainArgs& a, hipStream_t stream) {
  const int kc = a.C / 16;
  if (!wn_chain_supported(a.C, a.Mg, a.L)) return fail(MMK_ERR_UNSUPPORTED, "chain WaveNet kernel: C=%d, %d clips per group, %d layers", a.C, a.Mg, a.L);
  const size_t lds = wn_chain_lds_bytes(a);
  if (lds > 160 * 1024) return fail(MMK_ERR_UNSUPPORTED, "chain WaveNet kernel: %zu bytes of LDS needed", lds);
  dim3 grid(a.Gc * a.Gn), block(kChThreads);
#define MMK_WNC(KC_)                                                                                                    \
  do {                                                                                                                 \
    if (a.stamps) {                                                                                                    \
      if (a.xcd_local) hipLaunchKernelGGL((wavenet_chain_kernel<KC_, true, true>), grid, block, lds, stream, a);       \
      else hipLaunchKernelGGL((wavenet_chain_kernel<KC_, true, false>), grid, block, lds, stream, a);                  \
    } else {                                                                                                           \
      if (a.xcd_local) hipLaunchKernelGGL((wavenet_chain_kernel<KC_, false, true>), grid, block, lds, stream, a);      \
      else hipLaunchKernelGGL((wavenet_chain_kernel<KC_, false, false>), grid, block, lds, stream, a);                 \
    }                                                                                                                  \
  } while (0)
  switch (kc) {
    case 2: MMK_WNC(2); break;
    case 4: MMK_WNC(4); break;
    case 6: MMK_WNC(6); break;
    case 8: MMK_WNC(8); break;
    case 10: MMK_WNC(10); break;
    case 12: MMK_WNC(12); break;
    case 14: MMK_WNC(14); break;
    default: MMK_WNC(16); break;
  }
#undef MMK_WNC
  MMK_HIP(hipGetLastError());
  return MMK_OK;
}

}  // namespace mmk

// WaveNet generation as a pipeline of layer stages whose visits are GROUPS OF 16 CLIPS on the matrix pipe (gfx950): the large-batch form of
// wavenet_spipe.hip.  BASELINE config 4 (30 layers x 256 channels, conditioned) with more clips per GPU than the one-clip ring serves at its
// beat (~1.06 us per clip and stage: 128 clips = 136 us per step, 256 = 272).
//
// Reference: WaveNet.forward / WNLayer.forward (wavenet_v2.py:131-182, :276-293), MLP head and CategoricalSampler (networks/mlp.py:58-63,
// modules/targets.py:37-52).  The arithmetic is the stage pipeline's one-hand-off form, taken one stage further back:
//   x_s   = x_{s-1} + R_{s-1} y_{s-1} + br_{s-1}                                                       (off the chain)
//   z_s   = W0_s x_s[t - d_s] + Wc_s c[t] + b                                                           (a visit ahead)
//         + W1_s x_{s-2} + (W1_s R_{s-2}) y_{s-2}                                                       (early: what the stage BELOW received)
//         + (W1_s R_{s-1}) y_{s-1}                                                                      (on the chain)
//   y_s   = tanh(z_f) sigmoid(z_g)
//   hid  += (fc0 W_skip_{s-1}) y_{s-1}                                                                  (off the chain)
// with W1 R and fc0 W_skip composed at commit (fp64 accumulation, one rounding).  Only the last product of z_s, the gate and the store of y_s lie
// between the arrival of y_{s-1} and the departure of y_s: x_s is needed by the stage after next only (and by this layer's history ring).
//
// Shape.  A layer is a stage of 8 CUs (4 stages per XCD, roles from where a workgroup runs, as in the stage pipeline); a CU owns 32 units:
// 64 gate rows, 32 residual rows, 16 of the head's hidden units - 94,208 weights, in registers for the whole launch as A operands of
// v_mfma_f32_16x16x4_f32 (192 per lane in waves 0 - 3, 176 in waves 4 - 7).  A message is x | y | running hidden pre-activations of one group of 16
// clips (40 KB) in the B operand's layout; a stage reads the message of the stage below too (its x and y parts), so x and y are stored with plain
// stores only where the next TWO stages sit on this XCD.
//   waves 0-3  per visit: the early products of a gate tile (rows: f of 8 units | g of the same 8) from the stage below's message - gathered and
//              multiplied while y_{s-1} is still on its way -, then a quarter each of y_{s-1} (looked at until no word is the poison word), 64
//              products, the tile of the known terms added, the gate (the g half sits 32 lanes up: v_permlane32_swap), y_s stored;
//   waves 4-7  off the chain: a residual tile x a K half of y_{s-1} (the halves meet in LDS; waves 6, 7 add x_{s-1} - their own look at the message's
//              x part - and store x_s to the next stage's message and to the layer's history ring), the hidden units' tile (K quarters, added up by
//              wave 4 and handed on beside the message), and the NEXT visit's known terms W0 x_s[t - d] + Wc c[t] + b: the rows (history ring, conditioning)
//              come by LDS-DMA, asked for a visit ahead into a double-buffered image (no registers, no staging instructions); d = 1: the stage's own
//              newest message instead.
// Messages are the stage pipeline's: raw floats against a poison word, four generations per (stage, group), the producer re-poisons its own
// words two steps ahead.  The head stage's CU p serves groups p, p + 8, ..: hidden tile, fc2 tiles and the temperature row as matrix
// products, one wave per clip for the draw, the embedded classes as stage 0's next message.
//
// A group's step is a trip of L + 1 visits of ~3.5 us (the exchange ~2, the products on the chain ~1.5): 109 us per step from 7 to ~10 groups;
// beyond, a stage's ~9.5 us per visit (368 matrix instructions per SIMD = 4.9 us at the pipe's rate, plus what the two waves of a SIMD wait for) is the beat.
#define MMK_FAST_RCP 1
#include "wavenet_bpipe.h"
#include "sampler256.h"

namespace mmk {

namespace {

typedef float f32x4b __attribute__((ext_vector_type(4)));
typedef unsigned u32x4b __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

constexpr int kC = 256, kH1 = 128, kQ = 256;
constexpr int kThreads = 512;
constexpr int kG = kBpGroup;
constexpr unsigned kSpin = 1u << 22;
constexpr int kLgStride = 260;
#ifndef MMK_BP_ABL
#define MMK_BP_ABL 0      // timing experiments (results wrong): 1 no known-term products, 2 no hidden-unit products, 4 no early products, 8 no early gather either, 16 no residual products
#endif
#ifndef MMK_BP_LOOK_SLEEP
#define MMK_BP_LOOK_SLEEP 0      // s_sleep units (64 cycles) between two looks at a message part that is not complete (a look is a whole L2 round trip: 0 / 2 / 6 -> 106.9 / 107.4 / 108.0 us per step at 128 clips)
#endif
#ifndef MMK_BP_CHAIN_PRIO
#define MMK_BP_CHAIN_PRIO 2
#endif

// word of element (k, n) - channel / unit k, clip n of the group - in a B-operand image of 16 n: the lane (k % 4 / 1 .. = K sub-step, n) of
// k-step k / 4 reads it in one 16-byte read together with the three k-steps beside it
__device__ __forceinline__ int pos_of(int k, int n) { return (((k >> 4) * 64) + (k & 3) * 16 + n) * 4 + ((k >> 2) & 3); }

struct BpLds {
  float ye[8192];                 // what the stage below received, x_{s-2} | y_{s-2}: the B images of the early products
  float yl[2][4096];              // [visit parity] the newest message's y_{s-1} (stage 0: x, the embedded class): B image of the products on the chain
  float tc[2][8192];              // [visit parity] delayed x | conditioning row of a visit, written by LDS-DMA a visit ahead: element (channel 4 kk + e, clip n) at kk 64 + 4 n + e
  float biasT[2][4][256];         // [visit parity][gate tile]: the known terms, in the product's output layout [register][lane]
  float rpart[2][256];            // residual tiles: the first K half's partial
  float hpart[4][256];            // hidden-unit tile: the K quarters' partials
  unsigned arr[4];                // chain wave w: visits whose quarter of the newest y it has staged
  unsigned earr[4], edone[4];     // ... whose quarter of the early x image it has staged (its quarter of the early y image: earry, before that) / whose early products it has issued
  unsigned earry[4];
  unsigned done[8];               // wave: visits whose yl image it reads no more
  unsigned rp[2], rp_used[2];     // residual tile: second half's partial written / taken
  unsigned hp[4], hp_used[1];
  unsigned rows[4], bdone[4];     // helper: visits whose rows it has staged / whose known-term products it has finished
  unsigned bias_ready[4], bias_used[4];
};

__device__ __forceinline__ void sig(unsigned* p, unsigned v, int lane) {
  __atomic_signal_fence(__ATOMIC_SEQ_CST);
  if (lane == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  __atomic_signal_fence(__ATOMIC_SEQ_CST);
}
template <int N>
__device__ __forceinline__ bool wait_min(const unsigned* p, unsigned want, int32_t* err, int tag = 0) {
  unsigned spins = 0;
  for (;;) {
    unsigned m = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
    for (int i = 1; i < N; ++i) m = min(m, __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    if (m >= want) break;
    __builtin_amdgcn_s_sleep(1);
    if (++spins > kSpin || (MMK_WAIT_ERR_LOOK && (spins & 4095u) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
      atomicCAS(err, 0, 1);
      return false;
    }
  }
  __atomic_signal_fence(__ATOMIC_SEQ_CST);
  return true;
}

__device__ __forceinline__ void msg_put(unsigned* p, unsigned v, bool local) {
  if (local) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // stays in this XCD's L2
  else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned bits_of(float v) { return min(__float_as_uint(v), 0xFFFFFFFEu); }     // (never the poison word: wavenet_spipe.hip)
__device__ __forceinline__ bool clean(const u32x4b& v) { return v[0] != kSpPoison && v[1] != kSpPoison && v[2] != kSpPoison && v[3] != kSpPoison; }
// the poison word is the largest unsigned: "no word of these is poison" as maxima and ONE compare (16 compares and their ands were ~35 instructions between a look's
// return and the copy of its message into LDS)
// (ONE running maximum: the kernel sits at its 256 registers, a tree's temporaries spill into the visit loop)
__device__ __forceinline__ bool clean4(const u32x4b& a, const u32x4b& b, const u32x4b& c, const u32x4b& d) {
  unsigned mx = max(max(a[0], a[1]), a[2]);
  mx = max(max(mx, a[3]), b[0]); mx = max(max(mx, b[1]), b[2]); mx = max(max(mx, b[3]), c[0]); mx = max(max(mx, c[1]), c[2]);
  mx = max(max(mx, c[3]), d[0]); mx = max(max(mx, d[1]), d[2]); mx = max(mx, d[3]);
  return mx != kSpPoison;
}

// four 16-byte loads past the L1, 1 KB apart (a wave's 4 KB of a message)
__device__ __forceinline__ void load4_sc1(const unsigned* p, u32x4b& r0, u32x4b& r1, u32x4b& r2, u32x4b& r3) {
  asm volatile(
      "global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %4, off offset:1024 sc1\n\tglobal_load_dwordx4 %2, %4, off offset:2048 sc1\n\t"
      "global_load_dwordx4 %3, %4, off offset:3072 sc1\n\ts_waitcnt vmcnt(0)"
      : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
      : "v"(p)
      : "memory");
}
// diagnostic build: time (10 ns ticks) spent in the phases of a visit, totals per launch (mmk_wavenet_sync_status prints them).  Chain wave 0 of the stamped
// stage's CU 0: [0] early operands seen and staged (incl. the wait), [1] early products issued, [2] image free, [3] y_{s-1} seen and staged (incl. the wait), [4] all
// quarters, [5] products issued, [6] known terms there, [7] gate + stores.  Helper wave 4: [0] y_{s-1} staged, [1] residual tile + x_s stored, [2] hidden units + handed on,
// [3] the rows' wait, [4] rows in the image, [5] all rows, [6] products + tile free, [7] tile written
struct BpStamp {
#ifdef MMK_DIAG
  u64 acc[16] = {}, last = 0;
  bool on = false;
  // trace: the wall-clock time of every mark of three consecutive visits in the middle of the launch, for two SIMD pairs of the stamped CU (chain waves 0 and 2,
  // helper waves 4 and 6: a.stamps[160 + 24 role + 8 visit + mark]) - who waits for whom inside a visit
  unsigned long long* tr = nullptr;
  bool tv = false;
  int lane0 = 1;
  __device__ __forceinline__ void start(int64_t v = -1, int64_t v0 = 0) {
    if (on) last = __builtin_amdgcn_s_memrealtime();
    tv = tr != nullptr && v >= v0 && v < v0 + 3;
    if (tv) tr += 0;
    vrel = (int)(v - v0);
  }
  int vrel = 0;
  __device__ __forceinline__ void mark(int k) {
    if (on || tv) {
      const u64 t = __builtin_amdgcn_s_memrealtime();
      if (on) { acc[k] += t - last; last = t; }
      if (tv && lane0 == 0) tr[8 * vrel + k] = t;
    }
  }
  __device__ __forceinline__ void flush(unsigned long long* dst, int lane) const {
    if (on && lane == 0)
      for (int k = 0; k < 16; ++k) dst[k] = acc[k];
  }
#else
  bool on = false;
  unsigned long long* tr = nullptr;
  int lane0 = 1;
  __device__ __forceinline__ void start(int64_t = -1, int64_t = 0) {}
  __device__ __forceinline__ void mark(int) {}
  __device__ __forceinline__ void flush(unsigned long long*, int) const {}
#endif
};

// 16 bytes per lane from global memory straight into LDS (no registers): lane l's bytes land at dst + 16 l
__device__ __forceinline__ void glds16(const float* src, float* dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}

__device__ __forceinline__ void glds16_sc1(const float* src, float* dst) {      // ... past this CU's L1 (cache policy bit 4 = sc1: agent scope)
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 16);
}

__device__ __forceinline__ f32x4b mfma4(float a, float b, f32x4b c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// acc0 / acc1 += A registers w[0 .. 4 N) x the N 16-byte B reads b[0], b[64], ..: the B read two ahead is asked for before the current one's four
// products (read, wait, eight products, read again left the matrix pipe idle for an LDS round trip in every eight: 41 cycles per product, not 32)
template <int N>
__device__ __forceinline__ void mfma_sweep(const float* w, const f32x4b* b, f32x4b& acc0, f32x4b& acc1) {
  f32x4b q0 = b[0], q1 = N > 1 ? b[64] : b[0];
#pragma unroll
  for (int k = 0; k < N; ++k) {
    f32x4b q2 = q1;
    if (k + 2 < N) q2 = b[(k + 2) * 64];
    __builtin_amdgcn_sched_barrier(0);      // (the scheduler otherwise sinks the read behind the products it was meant to run under)
    acc0 = mfma4(w[4 * k + 0], q0[0], acc0);
    acc1 = mfma4(w[4 * k + 1], q0[1], acc1);
    acc0 = mfma4(w[4 * k + 2], q0[2], acc0);
    acc1 = mfma4(w[4 * k + 3], q0[3], acc1);
    q0 = q1;
    q1 = q2;
  }
}

// the same over an image in the LDS-DMA's order: k-step kk of lane (e, n) is the float at kk 64 + 4 n + e
template <int N>
__device__ __forceinline__ void mfma_sweep_lin(const float* w, const float* b, f32x4b& acc0, f32x4b& acc1) {
  float q[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) q[i] = b[(i < N ? i : 0) * 64];
#pragma unroll
  for (int k = 0; k < N; k += 2) {
    float n0 = q[4], n1 = q[5];
    if (k + 6 < N) n0 = b[(k + 6) * 64];
    if (k + 7 < N) n1 = b[(k + 7) * 64];
    __builtin_amdgcn_sched_barrier(0);
    acc0 = mfma4(w[k], q[0], acc0);
    acc1 = mfma4(w[k + 1], q[1], acc1);
    q[0] = q[2]; q[1] = q[3]; q[2] = q[4]; q[3] = q[5]; q[4] = n0; q[5] = n1;
  }
}

// a wave's 4 KB of a message into an LDS image, looked at until no word of it is the poison word (every look is the whole part: one trip to
// the L2 once it is there)
__device__ __forceinline__ bool gather(const unsigned* src, float* dst, int lane, int32_t* err, int tag = 0) {
  u32x4b r[4];
  unsigned spins = 0;
  for (;;) {
    load4_sc1(src + 4 * lane, r[0], r[1], r[2], r[3]);
    if (__all(clean4(r[0], r[1], r[2], r[3]))) break;
    if (MMK_BP_LOOK_SLEEP > 0) __builtin_amdgcn_s_sleep(MMK_BP_LOOK_SLEEP);
    if (++spins > kSpin || (MMK_WAIT_ERR_LOOK && (spins & 1023u) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
      atomicMax(err, 0x10000 | tag);      // (diagnosis: which look never saw its message)
      return false;
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) *reinterpret_cast<u32x4b*>(dst + j * 256 + 4 * lane) = r[j];
  return true;
}

// ------------------------------------------------------------------------------------------------------------------------------------
// layer stage, waves 0-3: gate tile w of the CU = rows [f of units 32 p + 8 w .. + 7 | g of the same units], K = [x_{s-2} | y_{s-2} | y_{s-1}]
// ------------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void chain_role(const WnBpipeArgs& a, BpLds& S, int stage, int p, int w, int lane) {
  float wa[kBpChainRegs];      // [W1 (x_{s-2}) | W1 R_{s-2} (y_{s-2}) | W1 R_{s-1} (y_{s-1})] of the tile's rows
  {
    const float* img = a.img + ((int64_t)stage * 8 + p) * kBpCuFloats + (int64_t)w * kBpChainRegs * 64 + lane;
#pragma unroll
    for (int i = 0; i < kBpChainRegs; ++i) wa[i] = img[i * 64];
#pragma unroll
    for (int i = 0; i < kBpChainRegs; ++i) asm volatile("" : "+v"(wa[i]));
  }
  const int G = (a.B + kG - 1) / kG;
  const int n = lane & 15, q = lane >> 4;
  // (x and y are read by the next stage AND the one after it: plain stores only where both sit on this XCD)
  const bool local_next = ((stage + 2) >> 2) == (stage >> 2);
  const int64_t stage_words = (int64_t)G * kSpSlots * kBpMsgWords;      // (the blocks of a launch are laid out for ITS groups: the plan poisons only those)
  const unsigned* inbox = a.msg + (int64_t)stage * stage_words;
  const unsigned* below = a.msg + (int64_t)(stage > 0 ? stage - 1 : 0) * stage_words;      // what the stage below received
  unsigned* outbox = a.msg + (int64_t)(stage + 1) * stage_words;
  const int late_off = stage == 0 ? 0 : 4096;      // stage 0's newest operand is the embedded class (the x part of its message)
  const int64_t V = a.n_steps * G;
  int t = 0, g = 0;
  __builtin_amdgcn_s_setprio(MMK_BP_CHAIN_PRIO);      // (ahead of the helper wave of the same SIMD, whose products are off the chain)
  BpStamp st;
  st.on = a.stamps != nullptr && stage == a.stamp_stage && p == 0 && w == 0;
  if (a.stamps != nullptr && stage == a.stamp_stage && p == 0 && (w == 0 || w == 2)) st.tr = a.stamps + 160 + 24 * (w >> 1);
  st.lane0 = lane;
  for (int64_t v = 0; v < V; ++v) {
    const int slot = t & 3, buf = (int)(v & 1);
    const unsigned uv = (unsigned)v;
    st.start(v, V / 2);
    f32x4b acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    // ---- early: W1 x_{s-2} + (W1 R_{s-2}) y_{s-2} - what the stage BELOW received for this visit, there a whole visit before y_{s-1} ---------------
    // (asked for by LDS-DMA behind the previous visit's early products instead: 128 clips 109 -> 119 us per step - the look at y_{s-1} then waits for it too)
    if (stage >= 1) {
      const unsigned* em = below + ((int64_t)g * kSpSlots + slot) * kBpMsgWords;
      if (v >= 1 && !wait_min<4>(S.edone, uv, a.err_flag, 64 * 1 + stage)) return;
      // y_{s-2} first: it left its stage a visit before x_{s-2} does (x_{s-2} = x_{s-3} + R y_{s-3} is made off the chain, behind y_{s-2}'s own products),
      // so these 64 products run while x_{s-2} is still on its way
      if (!(MMK_BP_ABL & 8) && !gather(em + 4096 + 1024 * w, S.ye + 4096 + 1024 * w, lane, a.err_flag, 64 * 15 + stage)) return;
      sig(&S.earry[w], uv + 1, lane);
      if (!wait_min<4>(S.earry, uv + 1, a.err_flag, 64 * 20 + stage)) return;
      if (!(MMK_BP_ABL & 12)) mfma_sweep<16>(wa + 64, reinterpret_cast<const f32x4b*>(S.ye + 4096) + lane, acc0, acc1);
      if (!(MMK_BP_ABL & 8) && !gather(em + 1024 * w, S.ye + 1024 * w, lane, a.err_flag, 64 * 14 + stage)) return;
      sig(&S.earr[w], uv + 1, lane);
      if (!wait_min<4>(S.earr, uv + 1, a.err_flag, 64 * 2 + stage)) return;
      st.mark(0);
      if (!(MMK_BP_ABL & 12)) mfma_sweep<16>(wa, reinterpret_cast<const f32x4b*>(S.ye) + lane, acc0, acc1);
      sig(&S.edone[w], uv + 1, lane);
      st.mark(1);
#ifdef MMK_DIAG
      if (a.stamps && g == 0 && t + 1 == (int)a.n_steps && p == 0 && w == 0 && lane == 0) a.stamps[128 + stage] = __builtin_amdgcn_s_memrealtime();      // early products out
#endif
    }
    // ---- on the chain: (W1 R_{s-1}) y_{s-1} -------------------------------------------------------------------------------------------------------
    if (v >= 2 && !wait_min<8>(S.done, uv - 1, a.err_flag, 64 * 3 + stage)) return;
    st.mark(2);
    if (!gather(inbox + ((int64_t)g * kSpSlots + slot) * kBpMsgWords + late_off + 1024 * w, S.yl[buf] + 1024 * w, lane, a.err_flag, 64 * 16 + stage)) return;
    st.mark(3);
    sig(&S.arr[w], uv + 1, lane);
    if (!wait_min<4>(S.arr, uv + 1, a.err_flag, 64 * 4 + stage)) return;
    st.mark(4);
#ifdef MMK_DIAG
    // the chain's time line: group 0 of the launch's last step, wall clock (100 MHz, one counter for the chip) when y was complete here ...
    if (a.stamps && g == 0 && t + 1 == (int)a.n_steps && p == 0 && w == 0 && lane == 0) a.stamps[64 + 2 * stage] = __builtin_amdgcn_s_memrealtime();
#endif
    mfma_sweep<16>(wa + 128, reinterpret_cast<const f32x4b*>(S.yl[buf]) + lane, acc0, acc1);
    st.mark(5);
    sig(&S.done[w], uv + 1, lane);
    if (!wait_min<1>(&S.bias_ready[w], uv + 1, a.err_flag, 64 * 5 + stage)) return;
    st.mark(6);
    float z[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) z[i] = (acc0[i] + acc1[i]) + S.biasT[buf][w][i * 64 + lane];
    sig(&S.bias_used[w], uv + 1, lane);
    // tanh(f) sigmoid(g) (wavenet_v2.py:151): the g rows of a unit sit 32 lanes above its f rows (v_permlane32_swap: the upper half's values in both halves)
    unsigned* dst = outbox + ((int64_t)g * kSpSlots + slot) * kBpMsgWords + 4096 + pos_of(32 * p + 8 * w + 4 * (q & 1), n);
    unsigned* psn = outbox + ((int64_t)g * kSpSlots + ((t + 2) & 3)) * kBpMsgWords + 4096 + pos_of(32 * p + 8 * w + 4 * (q & 1), n);
    unsigned yb4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(z[i]), __float_as_uint(z[i]), false, false);
      yb4[i] = bits_of(tanh_fast(z[i]) * sigmoid_fast(__uint_as_float(sw[1])));
    }
    // (channel k + i of a lane: word + 64 i - pos_of; the data words first, the poison for two steps ahead behind them)
    if (q < 2) {
      if (local_next) {
#pragma unroll
        for (int i = 0; i < 4; ++i) __hip_atomic_store(dst + 64 * i, yb4[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
        for (int i = 0; i < 4; ++i) __hip_atomic_store(psn + 64 * i, kSpPoison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) __hip_atomic_store(dst + 64 * i, yb4[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int i = 0; i < 4; ++i) __hip_atomic_store(psn + 64 * i, kSpPoison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    st.mark(7);
#ifdef MMK_DIAG
    if (a.stamps && g == 0 && t + 1 == (int)a.n_steps && p == 0 && w == 0 && lane == 0) a.stamps[64 + 2 * stage + 1] = __builtin_amdgcn_s_memrealtime();      // ... and when y_s had been stored
#endif
    if (++g == G) { g = 0; ++t; }
  }
  st.flush(a.stamps, lane);
}

// ------------------------------------------------------------------------------------------------------------------------------------
// layer stage, waves 4-7
// ------------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void helper_role(const WnBpipeArgs& a, BpLds& S, int stage, int p, int h, int lane) {
  float wb[128], wr[32], wh[16];
  {
    const float* cu = a.img + ((int64_t)stage * 8 + p) * kBpCuFloats;
    const float* ib = cu + (4 * kBpChainRegs + h * 128) * 64 + lane;
    const float* ir = cu + (4 * kBpChainRegs + 4 * 128 + h * 32) * 64 + lane;
    const float* ih = cu + (4 * kBpChainRegs + 4 * 128 + 4 * 32 + h * 16) * 64 + lane;
#pragma unroll
    for (int i = 0; i < 128; ++i) wb[i] = ib[i * 64];
#pragma unroll
    for (int i = 0; i < 32; ++i) wr[i] = ir[i * 64];
#pragma unroll
    for (int i = 0; i < 16; ++i) wh[i] = ih[i * 64];
#pragma unroll
    for (int i = 0; i < 128; ++i) asm volatile("" : "+v"(wb[i]));
#pragma unroll
    for (int i = 0; i < 32; ++i) asm volatile("" : "+v"(wr[i]));
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(wh[i]));
  }
  float bz[4], br[4];
  const int r2 = h & 1, kh = h >> 1;      // the residual tile (channels 32 p + 16 r2 ..) and K half of y this wave multiplies
  {
    const float* cst = a.cst + ((int64_t)stage * 8 + p) * kBpCstFloats;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      bz[i] = cst[h * 256 + i * 64 + lane];
      br[i] = cst[(4 + r2) * 256 + i * 64 + lane];
    }
  }
  const int G = (a.B + kG - 1) / kG;
  const int n = lane & 15, q = lane >> 4;
  const bool local_next = ((stage + 2) >> 2) == (stage >> 2);      // x_s: read by the next two stages
  const bool local_hid = ((stage + 1) >> 2) == (stage >> 2);       // the hidden units' sum: by the next one
  const int64_t stage_words = (int64_t)G * kSpSlots * kBpMsgWords;      // (the blocks of a launch are laid out for ITS groups: the plan poisons only those)
  const unsigned* inbox = a.msg + (int64_t)stage * stage_words;
  unsigned* outbox = a.msg + (int64_t)(stage + 1) * stage_words;
  float* ring = a.hist[stage];
  const int ring_mask = a.ring[stage] - 1, d = a.dil[stage];
  const int64_t slot_stride = (int64_t)a.Bmax * kC;
  const int C1 = a.C1;
  const int64_t V = a.n_steps * G;
  const bool late_rows = d >= 2 && (d - 1) * G < 2;
  BpStamp st;
  st.on = a.stamps != nullptr && stage == a.stamp_stage && p == 0 && h == 0;
  if (a.stamps != nullptr && stage == a.stamp_stage && p == 0 && (h == 0 || h == 2)) st.tr = a.stamps + 160 + 48 + 24 * (h >> 1);
  st.lane0 = lane;

  // everything of visit v1's z that does not depend on its message: W0 x_s[t - d] + Wc c[t] + b, tile h.  The rows are asked for (ask_rows: LDS-DMA, no
  // registers) at the top of the visit before and waited for after its other work: a ring row a ring ago and a conditioning row come from HBM
  auto ask_rows = [&](int64_t v1, int t1, int g1) {
    float* img = S.tc[v1 & 1];
    const int clip = min(kG * g1 + n, a.B - 1);      // (the lanes of clips that do not exist read the last one's rows: their columns are never stored)
    if (!(d == 1 && t1 >= 1)) {
      const int64_t tp = a.t0 - 1 + t1 - d;
      if (tp >= 0) {
        // lane (q, n): clip n, channels 16 (4 h + jj) + 4 q .. + 3 - the k-step 4 (4 h + jj) + q of the image
        const float* src = ring + (tp & ring_mask) * slot_stride + (int64_t)clip * kC + 64 * h + 4 * q;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) glds16_sc1(src + 16 * jj, img + (4 * h + jj) * 256);      // (written by the sibling CUs: never this CU's L1 copy of the slot)
      } else {      // in front of the sequence: zeros
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) *reinterpret_cast<f32x4b*>(img + (4 * h + jj) * 256 + 4 * lane) = f32x4b{0.f, 0.f, 0.f, 0.f};
      }
    }
    if (C1 > 0) {
      const float* crow = a.cproj + ((int64_t)clip * a.cond_steps + t1) * C1 + 64 * h + 4 * q;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
        if (64 * h + 16 * jj < C1) glds16(crow + 16 * jj, img + 4096 + (4 * h + jj) * 256);      // (C1 in whole 16s; the blocks above it stay zero)
    }
  };
  auto make_bias = [&](int64_t v1, int t1, int g1) -> bool {
    const unsigned u1 = (unsigned)v1;
    float* img = S.tc[v1 & 1];
    st.mark(3);
    if (d == 1 && t1 >= 1) {
      // the stage's own message of the step before (x part, this wave's quarter): 16-byte groups (channels 16 (4 h + j) + 4 q' + e for q' = 0 .. 3, clip n)
      const unsigned* src = outbox + ((int64_t)g1 * kSpSlots + ((t1 - 1) & 3)) * kBpMsgWords + 1024 * h;
      u32x4b r[4];
      unsigned spins = 0;
      for (;;) {
        load4_sc1(src + 4 * lane, r[0], r[1], r[2], r[3]);
        if (__all(clean4(r[0], r[1], r[2], r[3]))) break;
        __builtin_amdgcn_s_sleep(2);
        if (++spins > kSpin || (MMK_WAIT_ERR_LOOK && (spins & 1023u) == 0 && __hip_atomic_load(a.err_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
          atomicMax(a.err_flag, 0x10000 | (64 * 17 + stage));
          return false;
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) img[(16 * h + 4 * j + qq) * 64 + 4 * n + q] = __uint_as_float(r[j][qq]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the rows asked for a visit ago
    st.mark(4);
    sig(&S.rows[h], u1 + 1, lane);
    if (!wait_min<4>(S.rows, u1 + 1, a.err_flag, 64 * 7 + stage)) return false;
    st.mark(5);
    f32x4b acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const float* tb = img + 4 * n + q;
    if (!(MMK_BP_ABL & 1)) {
      mfma_sweep_lin<64>(wb, tb, acc0, acc1);
      if (C1 > 0) mfma_sweep_lin<64>(wb + 64, tb + 4096, acc0, acc1);
    }
    if (v1 >= 2 && !wait_min<1>(&S.bias_used[h], u1 - 1, a.err_flag, 64 * 8 + stage)) return false;
    st.mark(6);
#pragma unroll
    for (int i = 0; i < 4; ++i) S.biasT[v1 & 1][h][i * 64 + lane] = (acc0[i] + acc1[i]) + bz[i];
    sig(&S.bias_ready[h], u1 + 1, lane);
    return true;
  };

  // (conditioning blocks above C1 are never written: zero once)
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    *reinterpret_cast<f32x4b*>(S.tc[0] + 4096 + (4 * h + jj) * 256 + 4 * lane) = f32x4b{0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<f32x4b*>(S.tc[1] + 4096 + (4 * h + jj) * 256 + 4 * lane) = f32x4b{0.f, 0.f, 0.f, 0.f};
  }
  ask_rows(0, 0, 0);
  if (!make_bias(0, 0, 0)) return;
  int t = 0, g = 0;
  for (int64_t v = 0; v < V; ++v) {
    const int slot = t & 3, buf = (int)(v & 1);
    const unsigned uv = (unsigned)v;
    const int64_t tau = a.t0 - 1 + t;
    st.start(v, V / 2);
    int t1 = t, g1 = g + 1;
    if (g1 == G) { g1 = 0; ++t1; }
    // The ring row x_s[t1 - d] of visit v + 1 was stored by ALL 8 CUs of this stage (d G) visits before it.  What orders this CU's read behind a sibling's
    // store is the chain itself: y_{s-1} of visit w is here => the class of w's step was drawn => every CU had handed on its hidden units' sum of the
    // step before, which its waves 6 / 7 do behind their ring store.  At the top of visit v that covers the rows up to visit v - 1 - G: enough where
    // (d - 1) G >= 2; a launch of ONE group on the d = 2 stage (row of visit v - 1) asks behind y_{s-1} of visit v instead (covers v - G) and its
    // siblings complete their ring stores before they hand on (the release below)
    if (v + 1 < V && !late_rows) ask_rows(v + 1, t1, g1);
    if (!wait_min<4>(S.arr, uv + 1, a.err_flag, 64 * 9 + stage)) return;
    if (v + 1 < V && late_rows) ask_rows(v + 1, t1, g1);
    st.mark(0);
    const f32x4b* yb = reinterpret_cast<const f32x4b*>(S.yl[buf]) + lane;
    // ---- x_s = x_{s-1} + R y_{s-1} + br, off the chain (the next stage but one multiplies with it, a visit from now): residual tile r2, K half kh ----
    {
      f32x4b rc0 = {0.f, 0.f, 0.f, 0.f}, rc1 = {0.f, 0.f, 0.f, 0.f};
      // where a step is one group's trip (up to 10 groups): behind the 64 products of this SIMD's chain wave - the pipe takes the two waves' products in the order
      // they were issued, and those 64 are on the chain (128 clips 109.1 -> 107.4 us per step; at 16 groups x_s would leave too late: 141 -> 144)
      if (G <= 10 && !wait_min<1>(&S.done[h], uv + 1, a.err_flag, 64 * 18 + stage)) return;
      if (!(MMK_BP_ABL & 16)) mfma_sweep<8>(wr, yb + 8 * kh * 64, rc0, rc1);
      if (kh == 0) {      // (the second halves' waves publish: wave 4 has the hidden units' hand-over to do)
        if (v >= 1 && !wait_min<1>(&S.rp_used[r2], uv, a.err_flag, 64 * 10 + stage)) return;
#pragma unroll
        for (int i = 0; i < 4; ++i) S.rpart[r2][i * 64 + lane] = rc0[i] + rc1[i];
        sig(&S.rp[r2], uv + 1, lane);
      } else {
        // x_{s-1}: my 4 words per lane of the newest message's x part (published behind its y), looked at until they are there
        const unsigned* src = inbox + ((int64_t)g * kSpSlots + slot) * kBpMsgWords;
        float xprev[4];
        unsigned spins = 0;
        for (;;) {
          unsigned wv[4];
          bool ok = true;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            wv[i] = __hip_atomic_load(src + (((2 * p + r2) * 64) + i * 16 + n) * 4 + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = ok && wv[i] != kSpPoison;
          }
          if (__all(ok)) {
#pragma unroll
            for (int i = 0; i < 4; ++i) xprev[i] = __uint_as_float(wv[i]);
            break;
          }
          __builtin_amdgcn_s_sleep(2);
          if (++spins > kSpin || (MMK_WAIT_ERR_LOOK && (spins & 1023u) == 0 && __hip_atomic_load(a.err_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
            atomicCAS(a.err_flag, 0, 0x20000 | stage);
            return;
          }
        }
        if (!wait_min<1>(&S.rp[r2], uv + 1, a.err_flag, 64 * 11 + stage)) return;
        float xs[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) xs[i] = xprev[i] + ((S.rpart[r2][i * 64 + lane] + (rc0[i] + rc1[i])) + br[i]);
        sig(&S.rp_used[r2], uv + 1, lane);
        unsigned* dx = outbox + ((int64_t)g * kSpSlots + slot) * kBpMsgWords;
        unsigned* px = outbox + ((int64_t)g * kSpSlots + ((t + 2) & 3)) * kBpMsgWords;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int word = (((2 * p + r2) * 64) + i * 16 + n) * 4 + q;
          msg_put(dx + word, bits_of(xs[i]), local_next);
          msg_put(px + word, kSpPoison, local_next);
        }
        // the layer's input at tau into its ring (later taps; the launch path, should the batch be redone there)
        const int clip = kG * g + n;
        if (clip < a.B)
          *reinterpret_cast<f32x4b*>(ring + (tau & ring_mask) * slot_stride + (int64_t)clip * kC + 32 * p + 16 * r2 + 4 * q) = f32x4b{xs[0], xs[1], xs[2], xs[3]};
        if (G == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");      // (one group: the row is read a visit from now - see late_rows; the wait is off the beat there)
      }
    }
    st.mark(1);
    // ---- hid_s = hid_{s-1} + (fc0 W_skip_{s-1}) y_{s-1}: units 16 p .., K quarter h -------------------------------------------------------
    float hs[4] = {0.f, 0.f, 0.f, 0.f};
    {
      f32x4b acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k4 = 0; k4 < ((MMK_BP_ABL & 2) ? 0 : 4); ++k4) {
        const f32x4b b = yb[(4 * h + k4) * 64];
        acc0 = mfma4(wh[4 * k4 + 0], b[0], acc0);
        acc1 = mfma4(wh[4 * k4 + 1], b[1], acc1);
        acc0 = mfma4(wh[4 * k4 + 2], b[2], acc0);
        acc1 = mfma4(wh[4 * k4 + 3], b[3], acc1);
      }
      sig(&S.done[4 + h], uv + 1, lane);
      if (h != 0) {
        if (v >= 1 && !wait_min<1>(S.hp_used, uv, a.err_flag, 64 * 12 + stage)) return;
#pragma unroll
        for (int i = 0; i < 4; ++i) S.hpart[h][i * 64 + lane] = acc0[i] + acc1[i];
        sig(&S.hp[h], uv + 1, lane);
      } else {
        if (!wait_min<3>(S.hp + 1, uv + 1, a.err_flag, 64 * 13 + stage)) return;
#pragma unroll
        for (int i = 0; i < 4; ++i) hs[i] = ((acc0[i] + acc1[i]) + S.hpart[1][i * 64 + lane]) + (S.hpart[2][i * 64 + lane] + S.hpart[3][i * 64 + lane]);
        sig(S.hp_used, uv + 1, lane);
      }
    }
    // wave 4: the hidden units' sum handed on.  Where the stages run at their beat (more than 10 groups) BEHIND the next visit's known terms - chain wave 0 waits
    // for those, the sum is needed by the head only, and every stage adds its part a visit's length behind its y, so the hand-over's own latency shows once per
    // step, in front of the head (256 clips 153 -> 144 us per step); where a step is one group's trip, in front of them (128 clips 109.5 against 111.2)
    auto hand_on = [&]() -> bool {
      if (h == 0) {
        if (stage >= 1) {      // the sum so far: 4 words per lane of the message, looked at until they are there
          const unsigned* src = inbox + ((int64_t)g * kSpSlots + slot) * kBpMsgWords + 8192;
          unsigned spins = 0;
          for (;;) {
            unsigned wv[4];
            bool ok = true;
  #pragma unroll
            for (int i = 0; i < 4; ++i) {
              wv[i] = __hip_atomic_load(src + ((p * 64) + i * 16 + n) * 4 + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              ok = ok && wv[i] != kSpPoison;
            }
            if (__all(ok)) {
  #pragma unroll
              for (int i = 0; i < 4; ++i) hs[i] += __uint_as_float(wv[i]);
              break;
            }
            __builtin_amdgcn_s_sleep(2);
            if (++spins > kSpin || (MMK_WAIT_ERR_LOOK && (spins & 1023u) == 0 && __hip_atomic_load(a.err_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
              atomicCAS(a.err_flag, 0, 0x20000 | stage);
              return false;
            }
          }
        }
        unsigned* dst = outbox + ((int64_t)g * kSpSlots + slot) * kBpMsgWords + 8192;
        unsigned* psn = outbox + ((int64_t)g * kSpSlots + ((t + 2) & 3)) * kBpMsgWords + 8192;
  #pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int word = ((p * 64) + i * 16 + n) * 4 + q;
          msg_put(dst + word, bits_of(hs[i]), local_hid);
          msg_put(psn + word, kSpPoison, local_hid);
        }
      }
      return true;
    };
    st.mark(2);
    const bool late_hand_on = G > 10;
    if (!late_hand_on && !hand_on()) return;
    if (v + 1 < V && !make_bias(v + 1, t1, g1)) return;
    st.mark(7);
    if (late_hand_on && !hand_on()) return;
    t = t1; g = g1;
  }
  st.flush(a.stamps + 16, lane);
}

// ------------------------------------------------------------------------------------------------------------------------------------
// head stage: CU p serves groups p, p + 8, ..
// ------------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void head_role(const WnBpipeArgs& a, unsigned char* lds_raw, int p) {
  float* yimg = reinterpret_cast<float*>(lds_raw);      // 4096: y of the last layer
  float* hin = yimg + 4096;                             // 2048: the hidden units' sum so far
  float* hidb = hin + 2048;                             // 2048: Mish(hidden), B image of fc2
  float* lg = hidb + 2048;                              // 16 x kLgStride logits
  int* cls = reinterpret_cast<int*>(lg + kG * kLgStride);
  int* s_fail = cls + kG;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, q = lane >> 4;
  float w0[64], w2[64], wt[32];
  // A operands: lane (q, m) of k-step kk holds row m, column 4 kk + q
#pragma unroll
  for (int kk = 0; kk < 64; ++kk) w0[kk] = a.head_w0[(int64_t)(16 * w + n) * kC + 4 * kk + q];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int kk = 0; kk < 32; ++kk) w2[32 * j + kk] = a.fc2_w[(int64_t)(16 * (w + 8 * j) + n) * kH1 + 4 * kk + q];
#pragma unroll
  for (int kk = 0; kk < 32; ++kk) wt[kk] = (w == 0 && n == 0 && a.learn_temp) ? a.fc2_w[(int64_t)kQ * kH1 + 4 * kk + q] : 0.f;
  float b0[4], b2[2][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    b0[i] = a.head_b0[16 * w + 4 * q + i];
    b2[0][i] = a.fc2_b[16 * w + 4 * q + i];
    b2[1][i] = a.fc2_b[16 * (w + 8) + 4 * q + i];
  }
  const float bt = a.learn_temp ? a.fc2_b[kQ] : 0.f;
#pragma unroll
  for (int i = 0; i < 64; ++i) { asm volatile("" : "+v"(w0[i])); asm volatile("" : "+v"(w2[i])); }
#pragma unroll
  for (int i = 0; i < 32; ++i) asm volatile("" : "+v"(wt[i]));
  if (tid == 0) *s_fail = 0;
  const int G = (a.B + kG - 1) / kG;
  const int64_t stage_words = (int64_t)G * kSpSlots * kBpMsgWords;      // (the blocks of a launch are laid out for ITS groups: the plan poisons only those)
  const unsigned* inbox = a.msg + (int64_t)a.L * stage_words;
  unsigned* outbox = a.msg;      // stage 0's (another XCD: written through)
  // the embedded classes of cls[] as stage 0's message of step s1: x = E[class], y = 0
  auto emit = [&](int g, int s1) {
    u64* dst = reinterpret_cast<u64*>(outbox + ((int64_t)g * kSpSlots + (s1 & 3)) * kBpMsgWords);
    u64* psn = reinterpret_cast<u64*>(outbox + ((int64_t)g * kSpSlots + ((s1 + 2) & 3)) * kBpMsgWords);
    const u64 pp = ((u64)kSpPoison << 32) | kSpPoison;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int g4 = tid + r * kThreads;                                   // 16-byte group of the x image: clip g4 % 16, channels k0 + 4 e
      const int k0 = (g4 >> 6) * 16 + ((g4 >> 4) & 3), c = cls[g4 & 15];
      const float* e = a.emb + (int64_t)c * kC + k0;
      __hip_atomic_store(dst + 2 * g4, ((u64)bits_of(e[4]) << 32) | bits_of(e[0]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(dst + 2 * g4 + 1, ((u64)bits_of(e[12]) << 32) | bits_of(e[8]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(dst + 2048 + 2 * g4, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(dst + 2048 + 2 * g4 + 1, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(psn + 2 * g4, pp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(psn + 2 * g4 + 1, pp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(psn + 2048 + 2 * g4, pp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(psn + 2048 + 2 * g4 + 1, pp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  for (int g = p; g < G; g += 8) {
    if (tid < kG) {
      const int c = kG * g + tid;
      int k = c < a.B ? (int)a.idx[(int64_t)c * a.idx_rs + a.t0 - 1] : 0;
      cls[tid] = k < 0 ? 0 : (k >= kQ ? kQ - 1 : k);
    }
    __syncthreads();
    emit(g, 0);
    __syncthreads();
  }
  for (int s = 0; s < (int)a.n_steps; ++s) {
    const int64_t tau = a.t0 - 1 + s;
    for (int g = p; g < G; g += 8) {
      // ---- y of the last layer and the hidden units' sum: 6144 words, 3 KB per wave ----------------------------------------------------
      {
        const unsigned* src = inbox + ((int64_t)g * kSpSlots + (s & 3)) * kBpMsgWords + 4096 + 768 * w + 4 * lane;
        u32x4b r[3];
        unsigned spins = 0;
        for (;;) {
          asm volatile("global_load_dwordx4 %0, %3, off sc1\n\tglobal_load_dwordx4 %1, %3, off offset:1024 sc1\n\tglobal_load_dwordx4 %2, %3, off offset:2048 sc1\n\t"
                       "s_waitcnt vmcnt(0)"
                       : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2])
                       : "v"(src)
                       : "memory");
          if (__all(clean4(r[0], r[1], r[2], r[2]))) break;
          __builtin_amdgcn_s_sleep(2);
          if (++spins > kSpin || (MMK_WAIT_ERR_LOOK && (spins & 1023u) == 0 && __hip_atomic_load(a.err_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
            atomicCAS(a.err_flag, 0, 0x20000 | a.L);
            *s_fail = 1;
            break;
          }
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) *reinterpret_cast<u32x4b*>(yimg + 768 * w + j * 256 + 4 * lane) = r[j];
      }
      __syncthreads();
      if (*s_fail) return;
      // ---- hidden units 16 w ..: (fc0 W_skip of the last layer) y + the sum so far + bias, Mish -----------------------------------------
      {
        f32x4b acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        const f32x4b* yb = reinterpret_cast<const f32x4b*>(yimg) + lane;
#pragma unroll
        for (int k4 = 0; k4 < 16; ++k4) {
          const f32x4b b = yb[k4 * 64];
          acc0 = mfma4(w0[4 * k4 + 0], b[0], acc0);
          acc1 = mfma4(w0[4 * k4 + 1], b[1], acc1);
          acc0 = mfma4(w0[4 * k4 + 2], b[2], acc0);
          acc1 = mfma4(w0[4 * k4 + 3], b[3], acc1);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int word = ((w * 64) + i * 16 + n) * 4 + q;      // pos_of(16 w + 4 q + i, n)
          hidb[word] = mish_fast(((acc0[i] + acc1[i]) + hin[word]) + b0[i]);
        }
      }
      __syncthreads();
      // ---- logits: class tiles w and w + 8, the temperature row on wave 0 ---------------------------------------------------------------
      {
        const f32x4b* hb = reinterpret_cast<const f32x4b*>(hidb) + lane;
        f32x4b q0 = {0.f, 0.f, 0.f, 0.f}, q1 = {0.f, 0.f, 0.f, 0.f}, qt = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k4 = 0; k4 < 8; ++k4) {
          const f32x4b b = hb[k4 * 64];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            q0 = mfma4(w2[4 * k4 + e], b[e], q0);
            q1 = mfma4(w2[32 + 4 * k4 + e], b[e], q1);
          }
          if (w == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) qt = mfma4(wt[4 * k4 + e], b[e], qt);
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          lg[n * kLgStride + 16 * w + 4 * q + i] = q0[i] + b2[0][i];
          lg[n * kLgStride + 16 * (w + 8) + 4 * q + i] = q1[i] + b2[1][i];
        }
        if (w == 0 && q == 0) lg[n * kLgStride + kQ] = qt[0] + bt;
      }
      __syncthreads();
      // ---- the draw: a wave per clip -------------------------------------------------------------------------------------------------------
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int cn = w + 8 * r, c = kG * g + cn;
        const float* l = lg + cn * kLgStride;
        int result = 0;
        if (c < a.B) {
          if (a.logits_out && s + 1 == (int)a.n_steps)
            for (int k = lane; k < kQ + (a.learn_temp ? 1 : 0); k += 64) a.logits_out[(int64_t)c * a.logits_ld + k] = l[k];
          if (a.temperature == nullptr) {
            result = greedy_256(l, a.learn_temp != 0, l[kQ], a.min_temp, lane);
          } else {
            float denom = 1.f;
            if (a.learn_temp) denom = fmaxf(sigmoidf_(l[kQ]), a.min_temp);       // mlp.py:60-62
            result = sample_256(l, a.learn_temp != 0, denom, a.temperature[c], a.uniforms[(int64_t)c * a.uni_ld + s], lane);
          }
          if (lane == 0) a.idx[(int64_t)c * a.idx_rs + tau + 1] = result;
        }
        if (lane == 0) cls[cn] = result;
      }
      __syncthreads();
      if (s + 1 < (int)a.n_steps) emit(g, s + 1);
      __syncthreads();
    }
  }
}

__global__ __launch_bounds__(kThreads) void wavenet_bpipe_kernel(const WnBpipeArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char bpipe_lds[];
  BpLds& S = *reinterpret_cast<BpLds*>(bpipe_lds);
  __shared__ int s_role;
  const int tid = threadIdx.x;
  // roles from where the workgroup RUNS: XCD x hosts stages 4 x .. 4 x + 3, eight workgroups each, in arrival order (wavenet_spipe.hip)
  if (tid == 0) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    id &= 0xf;
    const unsigned ticket = atomicAdd(a.xcd_count + (id & 7), 1u);
    int role = -1;
    if (id >= 8 || ticket >= 32) atomicExch(a.err_flag, 2);
    else role = (int)(id * 32 + ticket);
    s_role = role;
  }
  if (tid < (int)((sizeof(BpLds) - offsetof(BpLds, arr)) / 4)) (&S.arr[0])[tid] = 0;
  __syncthreads();
  const int role = s_role;
  if (role < 0) return;
  const int stage = role >> 3, p = role & 7;
  if (stage > a.L) return;
  if (stage == a.L) {
    head_role(a, bpipe_lds, p);
    return;
  }
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (wave < 4) chain_role(a, S, stage, p, wave, lane);
  else helper_role(a, S, stage, p, wave - 4, lane);
}

// ------------------------------------------------------------------------------------------------------------------------------------
// commit: the A-operand images.  Lane (q, m) of register kk holds row m of the tile, column 4 kk + q.  Composed entries (W1 R, fc0 W_skip)
// are fp64 dot products of length 256, rounded once.
// ------------------------------------------------------------------------------------------------------------------------------------
__device__ double dot_strided(const float* arow, int64_t a_stride, const float* bcol, int64_t b_stride, int n) {
  double acc = 0.0;
  for (int c = 0; c < n; ++c) acc += (double)arow[c * a_stride] * (double)bcol[c * b_stride];
  return acc;
}
// raw row of conv_dil (2C, C, 2) for row m of gate tile (p, w): f rows of units 32 p + 8 w .. + 7, then the g rows of the same units
__device__ __forceinline__ int gate_row(int p, int w, int m) { return m < 8 ? 32 * p + 8 * w + m : kC + 32 * p + 8 * w + (m - 8); }

__global__ __launch_bounds__(256) void bpipe_image_kernel(const WnSpRaw* __restrict__ raw, int L, int C1, const float* __restrict__ f0, float* __restrict__ img,
                                                          float* __restrict__ cst) {
  const int64_t n_img = (int64_t)L * 8 * kBpCuFloats, n_cst = (int64_t)L * 8 * kBpCstFloats;
  for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < n_img + n_cst; id += (int64_t)gridDim.x * blockDim.x) {
    if (id < n_img) {
      const int s = (int)(id / (8 * (int64_t)kBpCuFloats)), p = (int)((id / kBpCuFloats) % 8);
      const int r = (int)(id % kBpCuFloats), lane = r & 63, reg = r >> 6, q = lane >> 4, m = lane & 15;
      const WnSpRaw rw = raw[s];
      const bool below = s >= 1 && raw[s - 1].wr != nullptr;
      float out = 0.f;
      const bool below2 = s >= 2 && raw[s - 2].wr != nullptr;
      constexpr int kCh = 4 * kBpChainRegs;
      if (reg < kCh) {                       // gate tile w x [x_{s-2} | y_{s-2} | y_{s-1}] (stage 0: the last third x the embedded class)
        const int w = reg / kBpChainRegs, kk = reg % kBpChainRegs, part = kk >> 6, k = 4 * (kk & 63) + q, nrow = gate_row(p, w, m);
        const float* w1row = rw.wd + (int64_t)nrow * kC * 2 + 1;      // W1[n][.]: stride 2
        if (part == 0) {
          if (s >= 1) out = w1row[2 * k];                                                                                        // W1[n][k]
        } else if (part == 1) {
          if (below2) out = (float)dot_strided(w1row, 2, raw[s - 2].wr + k, kC, kC);                                           // (W1 R_{s-2})[n][k]
        } else {
          if (s == 0) out = w1row[2 * k];
          else if (below) out = (float)dot_strided(w1row, 2, raw[s - 1].wr + k, kC, kC);                                       // (W1 R_{s-1})[n][k]
        }
      } else if (reg < kCh + 512) {          // gate tile h x [x_s[t - d] | c[t]]
        const int h = (reg - kCh) >> 7, k = 4 * ((reg - kCh) & 127) + q, nrow = gate_row(p, h, m);
        if (k < kC) out = rw.wd[((int64_t)nrow * kC + k) * 2 + 0];                                                                // W0[n][k]
        else if (rw.w1 && k - kC < C1) out = rw.w1[(int64_t)nrow * C1 + (k - kC)];
      } else if (reg < kCh + 512 + 128) {    // residual tile r2, K half kh of y
        const int h = (reg - kCh - 512) >> 5, kk = (reg - kCh - 512) & 31, r2 = h & 1, kh = h >> 1;
        if (below) out = raw[s - 1].wr[(int64_t)(32 * p + 16 * r2 + m) * kC + 128 * kh + 4 * kk + q];
      } else {                               // hidden units 16 p .., K quarter h of y
        const int h = (reg - kCh - 640) >> 4, kk = (reg - kCh - 640) & 15;
        if (s >= 1) out = (float)dot_strided(f0 + (int64_t)(16 * p + m) * kC, 1, raw[s - 1].ws + 64 * h + 4 * kk + q, kC, kC);    // (fc0 W_skip)[u][k]
      }
      img[id] = out;
    } else {
      const int64_t t = id - n_img;
      const int s = (int)(t / (8 * kBpCstFloats)), p = (int)((t / kBpCstFloats) % 8);
      const int r = (int)(t % kBpCstFloats), tile = r >> 8, i = (r >> 6) & 3, lane = r & 63, q = lane >> 4;
      const WnSpRaw rw = raw[s];
      const bool below = s >= 1 && raw[s - 1].wr != nullptr;
      float out = 0.f;
      if (tile < 4) {      // b_dil + b_1x1, then tap 1 . b_res of the layer below: the order the one-hand-off kernels add them in
        const int nrow = gate_row(p, tile, 4 * q + i);
        out = (rw.bd ? rw.bd[nrow] : 0.f) + (rw.b1 ? rw.b1[nrow] : 0.f);
        if (below && raw[s - 1].br) out += (float)dot_strided(rw.wd + (int64_t)nrow * kC * 2 + 1, 2, raw[s - 1].br, 1, kC);
        if (s >= 2 && raw[s - 2].wr && raw[s - 2].br) out += (float)dot_strided(rw.wd + (int64_t)nrow * kC * 2 + 1, 2, raw[s - 2].br, 1, kC);      // x_{s-1} = x_{s-2} + R y + br
      } else if (below && raw[s - 1].br) {
        out = raw[s - 1].br[32 * p + 16 * (tile - 4) + 4 * q + i];
      }
      cst[t] = out;
    }
  }
}

}  // namespace

bool wn_bpipe_supported(int C, int S, int H1, int n_classes, int L, int n_cond, int cond_dim, int batch) {
  // (the stage pipeline's networks; cond_dim: the widths of all conditioning inputs together, whole 16-byte groups of a row)
  const bool cond_ok = n_cond == 0 || (n_cond >= 1 && n_cond <= 2 && cond_dim > 0 && cond_dim <= kC && cond_dim % 16 == 0);
  return C == kC && S == kC && H1 >= 1 && H1 <= kH1 && n_classes >= 2 && n_classes <= kQ && cond_ok && L >= 1 && L <= kSpMaxLayers && batch >= 1 &&
         batch <= kBpMaxClips;
}
int64_t wn_bpipe_img_floats(int L) { return (int64_t)L * 8 * kBpCuFloats; }
int64_t wn_bpipe_cst_floats(int L) { return (int64_t)L * 8 * kBpCstFloats; }
int64_t wn_bpipe_msg_words(int L, int Bmax) { return (int64_t)(L + 1) * ((Bmax + kG - 1) / kG) * kSpSlots * kBpMsgWords; }

int wn_bpipe_build_image(const WnSpRaw* raw_dev, int L, int C1, const float* f0, float* img, float* cst, hipStream_t stream) {
  if (L < 1 || L > kSpMaxLayers || C1 < 0 || C1 > kC) return fail(MMK_ERR_UNSUPPORTED, "wavenet batched stage pipeline: L = %d, C1 = %d", L, C1);
  hipLaunchKernelGGL(bpipe_image_kernel, dim3(4096), dim3(256), 0, stream, raw_dev, L, C1, f0, img, cst);
  MMK_HIP(hipGetLastError());
  return MMK_OK;
}

int launch_wavenet_bpipe(const WnBpipeArgs& a, hipStream_t stream) {
  if (a.n_steps <= 0 || a.B <= 0) return MMK_OK;
  if (a.B > kBpMaxClips || a.L > kSpMaxLayers) return fail(MMK_ERR_UNSUPPORTED, "wavenet batched stage pipeline: %d clips, %d layers", a.B, a.L);
  const size_t head_lds = (size_t)(4096 + 2048 + 2048 + kG * kLgStride) * sizeof(float) + (kG + 4) * sizeof(int);
  const size_t lds = sizeof(BpLds) > head_lds ? sizeof(BpLds) : head_lds;
  MMK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wavenet_bpipe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(wavenet_bpipe_kernel, dim3(256), dim3(kThreads), lds, stream, a);
  MMK_HIP(hipGetLastError());
  return MMK_OK;
}

}  // namespace mmk

// WaveNet generation as a PIPELINE OF LAYER STAGES (gfx950): every layer is a stage of C / 32 = 8 CUs that hold the layer's matrices in
// registers for the whole launch; the clips of the batch stream through the stages ONE AT A TIME, so the chip holds up to 32 clips in
// flight, each at a different layer.  BASELINE config 4: 30 layers x 256 channels, conditioned, 32 clips per GPU.
//
// Reference: WaveNet.forward / WNLayer.forward (wavenet_v2.py:131-182, :276-293), MLP head and CategoricalSampler (networks/mlp.py:58-63,
// modules/targets.py:37-52).  Same arithmetic as the other step kernels in the one-hand-off form of wavenet_chain.hip:
//   x_s     = x_{s-1} + R_{s-1} y_{s-1} + br_{s-1}                                  (layer s's input: :172-176 of the layer below)
//   z_s     = W0_s x_s[t - d_s] + W1_s x_{s-1} + (W1_s R_{s-1}) y_{s-1} + cond_s + b  (:141-150; tap 1 through the residual sum)
//   y_s     = tanh(z_f) sigmoid(z_g)                                                 (:151)
//   hid    += (fc0 W_skip_s) y_s                                                     (:165-171 folded into the head's first Linear)
// with W1 R and fc0 W_skip pre-multiplied at commit (fp64 accumulation, one rounding).
//
// Why this shape.  A step of one clip is a chain of L + 1 dependent all-to-all exchanges; arithmetic (0.75 GFLOP per step and batch)
// and bytes are a tenth of it.  the XCD-pipelined kernel of rounds 2 - 4 (removed) spread a layer over 32 CUs and moved groups of 4 clips: 2.3 us per layer, of which
// two workgroup barriers, the K-split reduction through LDS and a 32-way gather are most.  Here
//   * a layer lives on 8 CUs (the whole net on 30 x 8 + 8 = 248 of the 256 CUs, 4 stages per XCD: 7 of 8 exchanges stay in one L2);
//   * a visit is ONE clip: per wave 16 gate rows x 512 inputs = 128 weights per lane, all in registers, packed-fp32 FMAs out of a
//     broadcast LDS read, a quad reduction by DPP, the gate, one store - no MFMA (a 1-row tile wastes it), no workgroup barrier;
//   * messages are raw floats checked against a poison word (a NaN no layer produces): no tags, half the bytes; the producer
//     re-poisons its own words two steps ahead (same wave, same address: ordered), so there is nothing to acknowledge;
//   * the four chain waves of a CU gather a message together (a quarter each, straight into a shared LDS image, one counter per
//     quarter): the L2 sees each message once per CU instead of once per wave;
//   * everything that does not depend on the newest sample - the delayed-tap product W0 x[t - d], the conditioning term, biases -
//     is prepared one step ahead by four helper waves (the second wave of every SIMD) and handed over through LDS; they also carry
//     the head's running hidden pre-activations from stage to stage beside the chain.
// Delayed taps come from the launch path's history rings in global memory (so warm-up = the prefill scattered into those rings,
// and a timed-out batch can be redone on the launch path); a layer with d = 1 reads its own previous output message instead.
// the gates' and the head's reciprocals on v_rcp_f32 (1 ulp): the correctly rounded division is ten instructions on every stage's
// critical path (55.64 -> 55.28 us per cfg-4 step)
#define MMK_FAST_RCP 1
#include <type_traits>

#include "wavenet_spipe.h"
#include "sampler256.h"

namespace mmk {

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4s __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

constexpr int kC = 256, kH1 = 128, kQ = 256;
constexpr int kThreads = 512;
constexpr int kCuPerStage = kC / 32;          // 8
constexpr int kWavesPerStage = kC / 8;        // 32 chain waves (and as many helpers)
constexpr int kMsgFloats = 2 * kC;            // 512
#ifndef MMK_SP_EARLY_Y
#define MMK_SP_EARLY_Y 1
#endif
#ifndef MMK_SP_HOIST_ADDR
#define MMK_SP_HOIST_ADDR 1
#endif
#ifndef MMK_SP_XSLICE20
#define MMK_SP_XSLICE20 1      // the LDS image of a message as 16 slices of 16 channels + 4 floats of padding (the chain lanes' 16-byte reads of 16 different slices then
                               // cover the 64 banks once); 0: round 3's blocks of 32 channels + 4, whose slices collide pairwise (31 % of the LDS-active cycles at 128 clips:
                               // profiles/r06_v2_pmc_sq_single_clips128_summary.csv)
#endif
constexpr int kPadBlk = 36;                   // 32 channels + 4 floats of padding: the K slices of a broadcast read fall on different banks
constexpr int kXSlice = 20;                   // 16 channels + 4 floats of padding
constexpr int kHalf = MMK_SP_XSLICE20 ? 16 * kXSlice : (kC / 32) * kPadBlk;    // 320 (288) floats: one padded vector of C channels
constexpr int kXyRing = 8;
constexpr int kChainRegs = 44, kHelperRegs = 32;   // float4 registers per lane in the stage images
constexpr unsigned kSpinLimit = 1u << 22;
#ifndef MMK_SP_POLL_GAP
#define MMK_SP_POLL_GAP 1
#endif
constexpr int kPollGap = MMK_SP_POLL_GAP;     // s_sleep units (64 cycles) between two looks at a message that has not arrived
#ifndef MMK_SP_LDS_SLEEP
#define MMK_SP_LDS_SLEEP 1      // (round 4, helpers look and stage: helpers 1 / chain waves 2 -> 50.85 us per cfg-4 step, 5 / 5 -> 51.45, 1 / 10 -> 51.4, 0 / 5 -> 51.0)
#endif
#ifndef MMK_SP_CHAIN_SLEEP
#define MMK_SP_CHAIN_SLEEP 2    // the same inside the chain waves' wait for their message (the helpers' waits keep MMK_SP_LDS_SLEEP)
#endif
constexpr int kChainSleep = MMK_SP_CHAIN_SLEEP;
constexpr int kLdsSleep = MMK_SP_LDS_SLEEP;   // s_sleep units inside the spins on LDS counters: with the chain waves' wait as three FLAT loads 0 / 1 / 3 / 6 / 10 -> 55.7 / 55.6 / 55.3 / 55.9 / 56.5 us per step; as ds_reads (55.3 -> 54.4) 0 / 1 / 2 / 3 / 5 / 7 / 10 / 15 -> 54.3 / 54.9 / 54.6 / 54.3 / 53.9 / 54.1 / 54.2 / 54.7
#ifndef MMK_SP_SLOT_SHIFT
#define MMK_SP_SLOT_SHIFT 0     // empty slots in front of layer 0 (which layers are the first and the last of an XCD).  Measured on cfg 4: 1 -> 58.8 us
                                // per step against 50.9 (layer 10, a dilation-1 stage, is then the LAST of its XCD; with 0 layers 0 and 20 are the first of theirs)
#endif
constexpr int kSlotShift = MMK_SP_SLOT_SHIFT;
#ifndef MMK_SP_LAG
#define MMK_SP_LAG 1           // the biases run four iterations behind the messages ...
#endif
#ifndef MMK_SP_LAG_CLIPS
#define MMK_SP_LAG_CLIPS 40    // ... from this many clips on (>= 8).  Measured on one box, cfg 4, us per step: 32 clips 53.2 with the lag, 52.0 without; 64 clips
                               // (with the early looks) 92.1 with it, 97.3 without: the lag pays where the clips queue up, and costs where one clip's latency binds
#endif
#ifndef MMK_SP_CHAIN_PRIO
#define MMK_SP_CHAIN_PRIO 3
#endif
#ifndef MMK_SP_ROWS8
#define MMK_SP_ROWS8 1         // chain products as 8 gate rows x a K slice of 16 per lane (half the LDS reads, one v_permlane16_swap level more): cfg 4, 32 clips 44.9 -> 44.2 us per step
#endif
#ifndef MMK_SP_BACKUP
#define MMK_SP_BACKUP 0        // the helper halfway between two look duties looks for the current message too, half a round trip behind
#endif
#ifndef MMK_SP_G_LOCAL
#define MMK_SP_G_LOCAL 6       // s_sleep units (64 clocks) by which the second pair of eyes trails: message from this XCD ...
#endif
#ifndef MMK_SP_G_REMOTE
#define MMK_SP_G_REMOTE 16     // ... and from another one
#endif
#ifndef MMK_SP_DUTYFIRST
#define MMK_SP_DUTYFIRST 1     // the helper whose look duty is next looks for that message before it makes the current iteration's bias: cfg 4, 32 clips 52.4 -> 44.9 us per step
#endif
#ifndef MMK_SP_EARLY_DEPTH
#define MMK_SP_EARLY_DEPTH 3   // ... up to this many visits ahead of the message that is being waited for
#endif
#ifndef MMK_SP_EARLY
#define MMK_SP_EARLY 0         // a helper looks for its next message already while it waits for the one before (staged by another helper): cfg 4, 64 clips 99.2 -> 92.1 us
                               // per step before the next-duty helper looked first; with that, 32 clips 44.9 without the early looks, 46.4 - 47.4 with them
#endif
#ifndef MMK_SP_CHAIN_SLEEP_BY_MODE
#define MMK_SP_CHAIN_SLEEP_BY_MODE 1
#endif
#ifndef MMK_SP_POLL_GAP_BY_MODE
#define MMK_SP_POLL_GAP_BY_MODE 1
#endif
#ifndef MMK_SP_MFMA_BIAS
#define MMK_SP_MFMA_BIAS 1     // four-behind mode: the biases of FOUR consecutive visits as one batch of v_mfma_f32_4x4x1 (helper_role)
#endif
#ifndef MMK_SP_LAG1
#define MMK_SP_LAG1 1          // where the delayed input cannot be asked for early (dilation 1 and 2), its rows are staged TWO iterations after the request instead of one
#endif
#ifndef MMK_SP_BIASSHIFT
#define MMK_SP_BIASSHIFT 1     // the bias of a visit is multiplied one iteration after its rows were staged (no helper waits for another one's staging inside an iteration)
#endif
#ifndef MMK_SP_NOARRWAIT
#define MMK_SP_NOARRWAIT 1     // helpers off duty do not wait for the current message (16 clips or more: the rings' own counters bound how far they run ahead)
#endif
#ifndef MMK_SP_ABL
#define MMK_SP_ABL 0           // timing builds only (results are wrong): the diagnostic build's dbg bits 1 / 2 / 4 / 8 / 16 / 32 / 128 / 256 as a compile-time mask of the product kernel
#endif
#define SP_ABL(bit) ((STAMPS && (a.dbg & (bit))) || (MMK_SP_ABL & (bit)))
#ifndef MMK_SP_WAKEUP
#define MMK_SP_WAKEUP 1        // the looking helper wakes the chain waves out of their s_sleep when it has staged a message
#endif
__device__ __forceinline__ int pad_of(int ch) { return MMK_SP_XSLICE20 ? (ch >> 4) * kXSlice + (ch & 15) : (ch >> 5) * kPadBlk + (ch & 31); }
static_assert(!MMK_SP_XSLICE20 || MMK_SP_ROWS8, "the 20-float slices belong to the 8-row form of the chain products");

__device__ __forceinline__ float dpp_quad_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
  return v;
}
__device__ __forceinline__ float dpp_half_mirror_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_mirror_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, false));
}

__device__ __forceinline__ float dpp_mirror(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, false)); }
__device__ __forceinline__ float dpp_half_mirror(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, false)); }

// Four partial sums per lane, 16 lanes (one DPP row) that each hold a different K slice: add them up across the row and leave column c's
// total in lanes 4 c .. 4 c + 3 of the row (lane l of the wave ends with column l / 4 of the wave's 16).  Fixed order: own + mirror
// partner, + half-mirror partner, then the quad.  A lane reads 1/4 of the inputs it would need with one column per lane: the LDS, which
// all four SIMDs share, is what bounds a visit otherwise.
__device__ __forceinline__ float row_reduce_scatter4(float v0, float v1, float v2, float v3, int ks) {
  const bool hi = (ks & 8) != 0, q4 = (ks & 4) != 0;
  float t0 = hi ? v2 : v0, t1 = hi ? v3 : v1;
  const float u0 = hi ? v0 : v2, u1 = hi ? v1 : v3;
  t0 += dpp_mirror(u0);
  t1 += dpp_mirror(u1);
  float w = q4 ? t1 : t0;
  const float sd = q4 ? t0 : t1;
  w += dpp_half_mirror(sd);
  return dpp_quad_sum(w);
}
// two partial sums per lane: lanes 0-7 of the row end with column 0's total, lanes 8-15 with column 1's
__device__ __forceinline__ float row_reduce_scatter2(float v0, float v1, int ks) {
  const bool hi = (ks & 8) != 0;
  float t = hi ? v1 : v0;
  const float u = hi ? v0 : v1;
  t += dpp_mirror(u);
  return dpp_quad_sum(dpp_half_mirror_add(t));
}

// eight partial sums per lane: lanes 2 c, 2 c + 1 of the row end with column c's total (own + mirror partner, + half-mirror partner, + the lane
// two further, + the neighbour)
__device__ __forceinline__ float row_reduce_scatter8(const float (&v)[8], int ks) {
  const bool b3 = (ks & 8) != 0, b2 = (ks & 4) != 0, b1 = (ks & 2) != 0;
  float k4[4], k2[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) k4[i] = (b3 ? v[4 + i] : v[i]) + dpp_mirror(b3 ? v[i] : v[4 + i]);
#pragma unroll
  for (int i = 0; i < 2; ++i) k2[i] = (b2 ? k4[2 + i] : k4[i]) + dpp_half_mirror(b2 ? k4[i] : k4[2 + i]);
  float r = (b1 ? k2[1] : k2[0]) + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(b1 ? k2[0] : k2[1]), 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
  r += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(r), 0xB1, 0xf, 0xf, false));                                             // quad_perm [1,0,3,2]
  return r;
}

__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

// layers + head + the empty slots have to fit the chip's 32 slots
__device__ __forceinline__ int slot_shift(const WnSpipeArgs& a) { return a.L + 1 + kSlotShift <= 32 ? kSlotShift : 0; }

// LDS counters: written by one lane of one wave, read by all; LDS serves a wave's operations in issue order, so data written before
// the counter is visible to whoever has read the new counter value.  The signal fences only pin the compiler's order.
__device__ __forceinline__ void lds_signal(unsigned* p, unsigned v, int lane) {
  __atomic_signal_fence(__ATOMIC_SEQ_CST);
  if (lane == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  __atomic_signal_fence(__ATOMIC_SEQ_CST);
}
// (atomic loads, not volatile ones: the compiler leaves a volatile access through a generic pointer a FLAT instruction - which reaches
//  LDS through the CU's vector-memory pipe, behind every global load and store in flight, and is waited for with vmcnt(0))
__device__ __forceinline__ unsigned lds_min4(const unsigned* p) {
  const unsigned v0 = __hip_atomic_load(p + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP), v1 = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  const unsigned v2 = __hip_atomic_load(p + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP), v3 = __hip_atomic_load(p + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  return min(min(v0, v1), min(v2, v3));
}
// wait until all four counters reach `want`; false after ~1 s (the other waves of the workgroup have failed or the kernel is wedged)
__device__ __forceinline__ bool lds_wait4(const unsigned* p, unsigned want, int32_t* err) {
  unsigned spins = 0;
  while (lds_min4(p) < want) {
    if (kLdsSleep > 0) __builtin_amdgcn_s_sleep(kLdsSleep);
    if (++spins > kSpinLimit || ((spins & 4095u) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
      atomicExch(err, 1);
      return false;
    }
  }
  __atomic_signal_fence(__ATOMIC_SEQ_CST);
  return true;
}
__device__ __forceinline__ bool lds_wait1(const unsigned* p, unsigned want, int32_t* err) {
  unsigned spins = 0;
  while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < want) {
    if (kLdsSleep > 0) __builtin_amdgcn_s_sleep(kLdsSleep);
    if (++spins > kSpinLimit || ((spins & 4095u) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
      atomicExch(err, 1);
      return false;
    }
  }
  __atomic_signal_fence(__ATOMIC_SEQ_CST);
  return true;
}

// message words: plain stores stay in the producer XCD's L2 (consumer on the same XCD, whose L2 is the coherence point of its CUs);
// otherwise written through (sc1).  Consumers always read past their L1 (sc1).
__device__ __forceinline__ void msg_store(unsigned* p, unsigned v, bool local) {
  if (local) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the bits of a value that goes into a message: 0xFFFFFFFF means "not arrived" there, and a NaN with exactly that payload CAN come out of
// the arithmetic (NaNs keep the payload of their source: a checkpoint or a conditioning input that carries such a NaN would make a consumer
// wait for a word that has arrived - until the 1-s timeout and the redo on the launch path).  One integer minimum maps it to the
// neighbouring NaN.
__device__ __forceinline__ unsigned msg_bits(float v) { return min(__float_as_uint(v), 0xFFFFFFFEu); }
__device__ __forceinline__ unsigned msg_load(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

constexpr int kRowRing = 4;                   // (= the number of helper waves: helper v mod 4 stages visit v) visits whose rows (delayed input | conditioning) the loading helper keeps staged
constexpr int kRowSlice = 20;                 // a K slice of 16 floats + 4 of padding: the 16 slices' 16-byte reads fall on different banks
constexpr int kRowPad = 16 * kRowSlice;
constexpr int kBRow = 528;                    // 512 floats + 16: the four visits' rows start 64 B apart modulo the 256 B of the banks
struct Lds {
  float xy[kXyRing][2 * kHalf];               // the newest messages of this stage: [x padded | y padded]
  float rows[kRowRing][2][kRowPad];           // [visit][x_s[t - d] | c[t]]: what the biases of the next visits are multiplied with
  float rowsb[2][4][kBRow];                   // the same rows of a BATCH of four visits, [x_s[t - d] (256) | c[t] (256)] + padding, two batches (MMK_SP_MFMA_BIAS)
  unsigned arrived[4];                        // [v mod 4]: v + 1 once the message of visit v is staged into xy (by helper v mod 4)
  unsigned hdone[4];                          // per chain wave: visits whose xy image it no longer needs
  unsigned rows_ready[4];                     // [v mod 4]: v + 1 once the rows of visit v are staged (by helper v mod 4: every word has ONE writer, so it only grows)
  unsigned ready[4];                          // per helper: biases prepared (visit count)
  unsigned hidin_ready[4];                    // [v mod 4]: v + 1 once the hidden units' hand-over of the stage below is staged for visit v
  float hidin[kXyRing][16];                   // that hand-over: this CU's 16 units
  float bias[];                               // [chain wave][step parity][Bcap clips][gate row]: everything of z that is known a step ahead (dynamic LDS: 512 B per clip)
};
__device__ __forceinline__ int bias_off(int q, int parity, int c, int j, int Bcap) { return ((q * 2 + parity) * Bcap + c) * 16 + j; }
__host__ __device__ __forceinline__ int bias_cap(int B) { return (B + 7) & ~7; }

struct Stamps {
  u64 t_wait = 0, t_compute = 0, t_post = 0, t_bias = 0, visits = 0, polls = 0;
};

// ------------------------------------------------------------------------------------------------------------------------------------
// chain waves (waves 0-3 of a layer stage's workgroup): gate rows 16 W .. 16 W + 15, residual channels 8 W .. 8 W + 7, W = 4 p + q
// ------------------------------------------------------------------------------------------------------------------------------------
// the two things a chain wave waits for in LDS before it computes visit v, read together: the message staged by the polling helper, this
// wave's bias prepared
#ifndef MMK_SP_TIGHT_WAIT
#define MMK_SP_TIGHT_WAIT 1
#endif
// up to 2^24 looks (~1 s) at ONE LDS counter in a loop of six instructions (read, wait, compare, branch out / count, branch back); returns the last value read.  The compiler's
// form of the loop below - two reads, a minimum, the time-out's bookkeeping and three exits - leaves ~10 scalar instructions between the look that sees the message
// and the first LDS read of the visit, on every visit's chain.
__device__ __forceinline__ unsigned lds_spin_ge(const unsigned* p, unsigned want) {
  const unsigned addr = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) unsigned*)p;
  unsigned val, cnt;
  asm volatile(
      "s_mov_b32 %1, 0x1000000\n"
      "1:\n\t"
      "ds_read_b32 %0, %2\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "v_cmp_le_u32 vcc, %3, %0\n\t"
      "s_cbranch_vccnz 2f\n\t"
      "s_sub_u32 %1, %1, 1\n\t"
      "s_cmp_lg_u32 %1, 0\n\t"
      "s_cbranch_scc1 1b\n"
      "2:"
      : "=&v"(val), "=&s"(cnt)
      : "v"(addr), "s"(want)
      : "vcc", "scc", "memory");
  return (unsigned)__builtin_amdgcn_readfirstlane((int)val);      // (every lane read the same word: a scalar for the caller's branch)
}

template <int SLEEP>
__device__ __forceinline__ bool chain_wait(const Lds& S, int q, unsigned v, int32_t* err) {
  unsigned spins = 0;
#if MMK_SP_TIGHT_WAIT
  if (SLEEP == 0) {      // (the trip-bound regime: no pause between two looks)
    // the bias of a visit is made a step ahead: it is there, or it is waited for first - then ONE counter is looked at
    while (__hip_atomic_load(&S.ready[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < v + 1) {
      if (++spins > kSpinLimit || ((spins & 4095u) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
        atomicExch(err, 1);
        return false;
      }
    }
    // (ONE loop with the whole time-out in it and one test behind it: an outer loop that looked at the error word every 4096 polls left a dozen scalar instructions
    //  and three taken branches between the poll that sees the message and the visit's first LDS read; a wave that gives up still raises the error word, the others
    //  run into their own time-out at about the same moment)
    if (__builtin_expect(lds_spin_ge(&S.arrived[v & 3], v + 1) < v + 1, 0)) {
      atomicExch(err, 1);
      return false;
    }
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    return true;
  }
#endif
  for (;;) {
    const unsigned arr = __hip_atomic_load(&S.arrived[v & 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const unsigned rd = __hip_atomic_load(&S.ready[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (arr >= v + 1 && rd >= v + 1) break;
    if (SLEEP > 0) __builtin_amdgcn_s_sleep(SLEEP);
    if (++spins > kSpinLimit || ((spins & 4095u) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
      atomicExch(err, 1);
      return false;
    }
  }
  __atomic_signal_fence(__ATOMIC_SEQ_CST);
  return true;
}

// (LAG4 - 40 clips or more, the ring goes at the stages' beat: the chain waves sleep between two looks at their counters and leave the issue
//  slots to the helpers, 64 clips 71.9 -> 71.2 us per step; with fewer clips the step is one clip's trip and they look without a pause, 32 clips 42.0 -> 41.5)
template <bool STAMPS, bool LAG4>
__device__ void chain_role(const WnSpipeArgs& a, Lds& S, int stage, int p, int q, int lane) {
  const int W = 4 * p + q;
  // products: a lane holds 4 gate rows x one K slice of 32 of [x | y] (and 2 residual rows x 16 of y); the rows' totals come out of a
  // reduce-scatter over the DPP row, so that lane l ends with gate row l / 4 and residual channel l / 8 of the wave
  const int ks = lane & 15;
  const int j = lane >> 2, j8 = lane >> 3;
  f32x4s wz[32], wr[8], wh[4];
  {
    const f32x4s* img = reinterpret_cast<const f32x4s*>(a.img_chain) + ((int64_t)stage * kWavesPerStage + W) * kChainRegs * 64 + lane;
#pragma unroll
    for (int i = 0; i < 32; ++i) wz[i] = img[i * 64];
#pragma unroll
    for (int i = 0; i < 8; ++i) wr[i] = img[(32 + i) * 64];
#pragma unroll
    for (int i = 0; i < 4; ++i) wh[i] = img[(40 + i) * 64];
  }
  float bx = a.cst_chain[((int64_t)stage * kWavesPerStage + W) * 64 + lane];
  // IN their registers before the visit loop: a load the compiler still counts as pending at the loop's entry makes it wait inside
  // every visit - vmcnt(40) ... vmcnt(0) along the products - and the last of those waits also cover the visit's own stores, the
  // written-through message of the previous visit included
#pragma unroll
  for (int i = 0; i < 32; ++i) asm volatile("" : "+v"(wz[i]));
#pragma unroll
  for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(wr[i]));
#pragma unroll
  for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(wh[i]));
  asm volatile("" : "+v"(bx));
  // The head's hidden pre-activations hid_s = hid_{s-1} + (fc0 W_skip_{s-1}) y_{s-1}: this wave's 4 units (4 W + lane / 16), K = 256 over the
  // 16 lanes of a row - with the 16 y values a lane holds for the residual product anyway, BEHIND the publish (off the chain).  They are
  // added up INSIDE an XCD only (a hand-over from another XCD is a 0.8-us load): the first stage of an XCD starts a new sum, the last one
  // writes the XCD's sum where the head collects the (at most eight) of them.
  const int64_t hid_words = (int64_t)a.Bmax * kSpSlots * kH1;
  const int grp = (stage + slot_shift(a)) >> 2;
  const bool hid_chain_in = stage >= 2 && ((stage - 1 + slot_shift(a)) >> 2) == grp;
  const bool hid_last = stage == a.L - 1 || ((stage + 1 + slot_shift(a)) >> 2) != grp;
  const bool hid_local = hid_last ? grp == ((a.L + slot_shift(a)) >> 2) : true;      // (the head's own XCD: its L2 is the meeting point)
  unsigned* hid_out = (hid_last ? a.hidgrp + (int64_t)grp * hid_words : a.hidmsg + (int64_t)(stage + 1) * hid_words) + 4 * W + (lane >> 4);
  const bool g_row = (j & 1) != 0;
  const float gate_scale = g_row ? -1.4426950408889634f : -2.8853900817779268f;
  const float gate_k = g_row ? 1.f : 2.f, gate_shift = g_row ? 0.f : -1.f;
  const bool local_next = ((stage + 1 + slot_shift(a)) >> 2) == ((stage + slot_shift(a)) >> 2);
  const int64_t stage_words = (int64_t)a.Bmax * kSpSlots * kMsgFloats;
  unsigned* msg_out = a.msg + (int64_t)(stage + 1) * stage_words;
#if MMK_SP_ROWS8
  const int kso = ((lane & 16) ? kHalf : 0) + pad_of(16 * ks);        // K slice lane & 31 of 16: x[16 ks ..] in the even rows of 16 lanes, y[16 ks ..] in the odd ones
#else
  const int kso = (ks < 8 ? 0 : kHalf) + (ks & 7) * kPadBlk;           // K slice ks: x[32 ks ..] for ks < 8, y[32 (ks - 8) ..] above
#endif
  const int xr_off = kHalf + pad_of(16 * ks);                          // y[16 ks ..]
  const int xin_off = pad_of(8 * W + j8);
  const bool pub_lane = (lane & 7) < 2;
  const int pub_off = W * 16 + (lane & 1) * 8 + (lane >> 3);
  const int ring_mask = a.ring[stage] - 1;
  float* hist = a.hist[stage];
  const int64_t slot_stride = (int64_t)a.Bmax * kC;
  const int B = a.B, n_steps = (int)a.n_steps, Bcap = bias_cap(a.B);
  Stamps st;
  u64 t0c = 0;
  unsigned v = 0;
  for (int s = 0; s < n_steps; ++s) {
    const int64_t tau = a.t0 - 1 + s;
    const int slot = s & 3, pslot = (s + 2) & 3;
    for (int c = 0; c < B; ++c, ++v) {
      if (STAMPS && !(a.dbg & 64)) t0c = __builtin_amdgcn_s_memtime();
#if MMK_SP_HOIST_ADDR
      // where this visit reads and writes, worked out BEFORE the wait for its message (the compiler puts these eight scalar / vector instructions behind the wait,
      // in front of the first LDS read - on every visit's chain)
      typedef const __attribute__((address_space(3))) float* lds_cf;
      typedef const __attribute__((address_space(3))) f32x4s* lds_cf4;
      const int o_img = (int)(v & (kXyRing - 1)) * 2 * kHalf;
      lds_cf p_z = (lds_cf)(&S.xy[0][0] + o_img + kso), p_y = (lds_cf)(&S.xy[0][0] + o_img + xr_off), p_in = (lds_cf)(&S.xy[0][0] + o_img + xin_off);
      lds_cf p_b = (lds_cf)(&S.bias[bias_off(q, s & 1, c, j, Bcap)]);
      int64_t o_dst = ((int64_t)c * kSpSlots + slot) * kMsgFloats + pub_off;
      asm volatile("" : "+v"(p_z), "+v"(p_y), "+v"(p_in), "+v"(p_b), "+v"(o_dst));
#endif
      if (!chain_wait<MMK_SP_CHAIN_SLEEP_BY_MODE ? (LAG4 ? kChainSleep : 0) : kChainSleep>(S, q, v, a.err_flag)) return;
      __builtin_amdgcn_s_setprio(MMK_SP_CHAIN_PRIO);              // (low while it spins: the helper wave of this SIMD gets the issue slots)
      if (STAMPS && !(a.dbg & 64)) {
        const u64 t = __builtin_amdgcn_s_memtime(); st.t_wait += t - t0c; t0c = t;
        if (a.stamps && c == 0 && s + 1 == n_steps && p == 0 && q == 0 && lane == 0) a.stamps[112 + stage] = __builtin_amdgcn_s_memrealtime();
      }
      if SP_ABL(16) {          // (diagnostic build, timing only: the chain waves do nothing - what the helpers' loop takes alone)
        lds_signal(&S.hdone[q], v + 1, lane);
        st.visits += 1;
        continue;
      }
      // what the helper prepared a step ahead: W0 x[t - d] + conditioning + biases
#if MMK_SP_HOIST_ADDR
      const float bzv = *p_b;
      const float* xb = &S.xy[0][0] + o_img;
#else
      const float bzv = S.bias[bias_off(q, s & 1, c, j, Bcap)];
      const float* xb = S.xy[v & (kXyRing - 1)];
#endif
#if MMK_SP_ROWS8
      // ---- z = [W1 | W1 R] . [x ; y]: 8 gate rows x ONE K slice of 16 per lane: 4 reads of 4 inputs (half of what 4 rows x 32 inputs read:
      //      the four chain waves' reads share one LDS), 64 packed FMAs; the two rows of 16 lanes that hold the same 8 gate rows swap
      //      halves (v_permlane16_swap: even rows keep gate rows 0-3, odd rows 4-7), then the row's reduce-scatter as before -----------------
      f32x2 acc8[8];
#pragma unroll
      for (int cc = 0; cc < 8; ++cc) acc8[cc] = f32x2{0.f, 0.f};
#if MMK_SP_EARLY_Y
      // the y slice of the residual product and the layer's own input are asked for NOW, with the gate products' reads: left to the compiler they go out behind the
      // 64 gate products (it re-uses their registers) and are waited for at once - an LDS round trip in front of the residual product, on every visit's chain
      f32x4s xv4[4], yv4[4];
#if MMK_SP_HOIST_ADDR
#pragma unroll
      for (int i = 0; i < 4; ++i) xv4[i] = ((lds_cf4)p_z)[i];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i) yv4[i] = ((lds_cf4)p_y)[i];
      const float xin_early = *p_in;
#else
#pragma unroll
      for (int i = 0; i < 4; ++i) xv4[i] = *reinterpret_cast<const f32x4s*>(xb + kso + i * 4);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i) yv4[i] = *reinterpret_cast<const f32x4s*>(xb + xr_off + i * 4);
      const float xin_early = xb[xin_off];
#endif
      __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#if MMK_SP_EARLY_Y
        const f32x4s xv = xv4[i];
#else
        const f32x4s xv = *reinterpret_cast<const f32x4s*>(xb + kso + i * 4);
#endif
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) {
          acc8[cc] = fma2(f32x2{wz[cc * 4 + i][0], wz[cc * 4 + i][1]}, f32x2{xv[0], xv[1]}, acc8[cc]);
          acc8[cc] = fma2(f32x2{wz[cc * 4 + i][2], wz[cc * 4 + i][3]}, f32x2{xv[2], xv[3]}, acc8[cc]);
        }
      }
#else
      // ---- z = [W1 | W1 R] . [x ; y]: 8 reads of 4 inputs, 64 packed FMAs (4 rows x 32 inputs per lane) ---------------------------------
      f32x2 acc[4][2];
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) acc[cc][0] = acc[cc][1] = f32x2{0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const f32x4s xv = *reinterpret_cast<const f32x4s*>(xb + kso + i * 4);
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          acc[cc][0] = fma2(f32x2{wz[cc * 8 + i][0], wz[cc * 8 + i][1]}, f32x2{xv[0], xv[1]}, acc[cc][0]);
          acc[cc][1] = fma2(f32x2{wz[cc * 8 + i][2], wz[cc * 8 + i][3]}, f32x2{xv[2], xv[3]}, acc[cc][1]);
        }
      }
#endif
      // ---- the layer's own input x_s = x_{s-1} + (R y + br): 2 channels x 16 inputs per lane ---------------------------------------------
      f32x2 rac[2][2];
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) rac[cc][0] = rac[cc][1] = f32x2{0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#if MMK_SP_EARLY_Y && MMK_SP_ROWS8
        const f32x4s yv = yv4[i];
#else
        const f32x4s yv = *reinterpret_cast<const f32x4s*>(xb + xr_off + i * 4);
#endif
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
          rac[cc][0] = fma2(f32x2{wr[cc * 4 + i][0], wr[cc * 4 + i][1]}, f32x2{yv[0], yv[1]}, rac[cc][0]);
          rac[cc][1] = fma2(f32x2{wr[cc * 4 + i][2], wr[cc * 4 + i][3]}, f32x2{yv[2], yv[3]}, rac[cc][1]);
        }
      }
#if MMK_SP_EARLY_Y && MMK_SP_ROWS8
      const float xin = xin_early;
#else
      const float xin = xb[xin_off];
#endif
      float zc[4];
#if MMK_SP_ROWS8
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc8[cc][0] + acc8[cc][1]), __float_as_uint(acc8[4 + cc][0] + acc8[4 + cc][1]), false, false);
        zc[cc] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
      }
#else
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) zc[cc] = (acc[cc][0][0] + acc[cc][0][1]) + (acc[cc][1][0] + acc[cc][1][1]);
#endif
      float z = row_reduce_scatter4(zc[0], zc[1], zc[2], zc[3], ks);
      const float xs = row_reduce_scatter2((rac[0][0][0] + rac[0][0][1]) + (rac[0][1][0] + rac[0][1][1]),
                                           (rac[1][0][0] + rac[1][0][1]) + (rac[1][1][0] + rac[1][1][1]), ks);
      const float xnew = xin + (xs + bx);
      z += bzv;
      // tanh(f) sigmoid(g) (wavenet_v2.py:151) with the hardware exp2 / rcp as in the other step kernels; the g row sits four lanes up
      const float act = fmaf(mmk_rcp(1.0f + __builtin_amdgcn_exp2f(z * gate_scale)), gate_k, gate_shift);
      const float other = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(act), 0x104, 0xf, 0xf, false));   // row_shl:4
      const float y = act * other;
      // ---- publish 8 x | 8 y, re-poison the same words two steps ahead, keep x_s for the delayed taps -------------------------------------
      if (pub_lane) {
#if MMK_SP_HOIST_ADDR
        unsigned* dst = msg_out + o_dst;
#else
        unsigned* dst = msg_out + ((int64_t)c * kSpSlots + slot) * kMsgFloats + pub_off;
#endif
        msg_store(dst, msg_bits((lane & 1) ? y : xnew), local_next);
      }
      if (STAMPS && !(a.dbg & 64)) {
        const u64 t = __builtin_amdgcn_s_memtime(); st.t_compute += t - t0c; t0c = t;
        // wall clock (100 MHz, one counter for the chip) of clip 0's publish in the last step, per stage: the chain's time line
        if (a.stamps && c == 0 && s + 1 == n_steps && p == 0 && q == 0 && lane == 0) a.stamps[16 + stage] = __builtin_amdgcn_s_memrealtime();
        if (a.stamps && s == n_steps / 2 && p == 0 && q == 0 && lane == 0 && c < 32) a.stamps[256 + stage * 32 + c] = __builtin_amdgcn_s_memrealtime();        // published, every clip
        if (a.stamps && s == n_steps / 2 && c == 5 && q == 0 && lane == 0) a.stamps[256 + 2048 + stage * 8 + p] = __builtin_amdgcn_s_memrealtime();               // published, every CU of the stage, clip 5
      }
      __builtin_amdgcn_s_setprio(0);
      if (pub_lane) msg_store(msg_out + ((int64_t)c * kSpSlots + pslot) * kMsgFloats + pub_off, kSpPoison, local_next);
      if ((lane & 7) == 0) hist[(tau & ring_mask) * slot_stride + (int64_t)c * kC + 8 * W + j8] = xnew;
      // ---- the hidden units' sum, handed on ---------------------------------------------------------------------------------------------------
      if (stage >= 1) {
        f32x2 hc[2] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};
        if (!SP_ABL(128))      // (diagnostic build, timing only: dbg 128 no hidden-unit products, dbg 256 nor the wait for the sum so far)
#pragma unroll
        for (int i = 0; i < 4; ++i) {      // (the y slice of the residual product, read again: 16 registers kept across the gate would not fit)
          const f32x4s yv = *reinterpret_cast<const f32x4s*>(xb + xr_off + i * 4);
          hc[0] = fma2(f32x2{wh[i][0], wh[i][1]}, f32x2{yv[0], yv[1]}, hc[0]);
          hc[1] = fma2(f32x2{wh[i][2], wh[i][3]}, f32x2{yv[2], yv[3]}, hc[1]);
        }
        const float hs = dpp_mirror_add(dpp_half_mirror_add(dpp_quad_sum((hc[0][0] + hc[0][1]) + (hc[1][0] + hc[1][1]))));
        // the sum so far (the stage below handed it on behind its own publish) was fetched by the helper that staged this visit's message:
        // the chain waves load NOTHING from memory - a load's data is waited for with a count that also covers the stores before it,
        // i.e. every visit would wait for its own publish to be acknowledged (a written-through one: ~1 us)
        float hin = 0.f;
        if (hid_chain_in && !SP_ABL(256)) {
          if (!lds_wait1(&S.hidin_ready[v & 3], v + 1, a.err_flag)) return;
          hin = S.hidin[v & (kXyRing - 1)][4 * q + (lane >> 4)];
        }
        if (ks == 0) {
          msg_store(hid_out + ((int64_t)c * kSpSlots + slot) * kH1, msg_bits(hin + hs), hid_local);
          msg_store(hid_out + ((int64_t)c * kSpSlots + pslot) * kH1, kSpPoison, hid_local);
        }
      }
      lds_signal(&S.hdone[q], v + 1, lane);        // this wave is through with the LDS image of visit v
      if (STAMPS && !(a.dbg & 64)) { const u64 t = __builtin_amdgcn_s_memtime(); st.t_post += t - t0c; }
      if (STAMPS) st.visits += 1;
    }
  }
  if (STAMPS && a.stamps && stage == a.stamp_stage && p == 0 && q == 0 && lane == 0) {
    a.stamps[0] = st.t_wait; a.stamps[1] = st.t_compute; a.stamps[2] = st.t_post; a.stamps[3] = st.visits;
  }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// helper waves (waves 4-7): the CU's eyes - they look for the messages and stage them - and, for chain wave h of the same workgroup, the
// part of z known a step ahead
// ------------------------------------------------------------------------------------------------------------------------------------
// One iteration per visit `it` = (clip c, step s):
//   1. the visit's message.  Helper it mod 4 LOOKS for it (lane l: floats 8 l .. 8 l + 7 = 8 x or 8 y channels of producing wave l / 2; past
//      this CU's L1) until no word is poison, stages it into the LDS image the chain waves read, tells them (S.arrived, s_wakeup) and
//      sends the first look at the message of visit it + 4; the other three wait for S.arrived - as evidence, they read nothing of it.
//   2. the bias of visit it - 1 + B from the rows asked for one or three iterations ago (staged in LDS by the helper that asked)
//   3. the helper that looked asks for the rows of the bias of visit it + B [+ 2]: the conditioning row c[t] and this stage's input x_s[t - d]
// Why the helpers look and not a chain wave (round 3, and this round's first builds): a wave's memory operations retire in order, so a
// look waits behind every store the same wave sent before it.  The chain waves publish - and the last stage of an XCD writes THROUGH to
// memory, ~1 us per store: with a chain wave looking, such a stage saw its message 0.2 us later than the others (two stores a visit), and 1.5
// us later once the chain waves also handed on the hidden units' sums (four).  The helpers store nothing outside LDS.  Their own slow loads
// (a row 512 steps old comes from HBM) are asked for right AFTER a look duty, four visits before the next one.  And four lookers in turn
// have four messages' looks in flight: a stage whose message comes from another XCD (~0.8 us per look) is no longer held to one visit
// per round trip when the clips queue up (64 and more clips in the ring).
// Nothing a step ahead is waited for where it is asked for.  That matters most for the rows of a dilation-1 layer, which are this stage's
// OWN output of visit `it` (published ~0.8 us after the message arrived): waiting for them inside the iteration that asked made such a
// stage's helpers take 1.66 us per visit whatever the chain did - the beat of the whole ring, 32 or 128 clips alike (64 clips: 106 us a step).
// What makes a ring row safe to read: every wave works through the visits in one order, and the arrival of message (c, s) says that step
// s - 1 of clip c has left the head - every visit up to (c, s - 1) is complete on every CU of this stage.  Row t_{s+1} - d of clip c was
// written by visit (c, s + 1 - d): covered iff d >= 2.  d = 1 polls the stage's own newest message instead (its ring entry has no
// arrival check).  Step 0's biases (rows the warm-up wrote) are prepared before the loop.
// LAG4: the launch's biases run four iterations behind the messages (40 clips or more).  A template argument, not a run-time flag: with
// both forms of the loop in one body the 32-clip step took 44.0 us instead of 42.2 (the compiler's schedule of the common part changes)
template <bool STAMPS, bool LAG4>
__device__ void helper_role(const WnSpipeArgs& a, Lds& S, int stage, int p, int h, int lane) {
  const int W = 4 * p + h;
  const int ks = lane & 15;                     // K slice of 16 of the delayed input and the conditioning row (4 gate rows per lane)
  const int j = lane >> 2;
  // MF (four-behind mode): the biases of four consecutive visits are ONE batch of v_mfma_f32_4x4x1: 16 blocks = 4 groups of 4 gate rows x
  // 4 K residues; lane 4 b + i carries row 4 (b & 3) + i as the A operand and visit i of the batch as the B operand, block b's share of K
  // is the floats 16 m + 4 (b >> 2) .. + 3 of every 16.  128 products per batch and helper on the matrix pipe (which the chain wave of the
  // SIMD does not use) instead of 4 x 64 packed FMAs + 4 reduce-scatters on the vector ALU it shares with it, and one hand-shake per four visits.
  constexpr bool MF = LAG4 && (MMK_SP_MFMA_BIAS != 0);
  f32x4s w0[16], wc[16];
  {
    const f32x4s* img = reinterpret_cast<const f32x4s*>(a.img_helper) + ((int64_t)stage * kWavesPerStage + W) * kHelperRegs * 64;
    if constexpr (MF) {
      // the same image, read in the matrix operand's order: register m (e = 0 .. 3) = W[row 4 g + i][half 256 + 16 (m & 15) + 4 ksub + e]
      const int g = (lane >> 2) & 3, i = lane & 3, ksub = lane >> 4;
#pragma unroll
      for (int m = 0; m < 16; ++m) w0[m] = img[(i * 4 + ksub) * 64 + 16 * g + m];
#pragma unroll
      for (int m = 0; m < 16; ++m) wc[m] = img[(16 + i * 4 + ksub) * 64 + 16 * g + m];
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) w0[i] = img[i * 64 + lane];
#pragma unroll
      for (int i = 0; i < 16; ++i) wc[i] = img[(16 + i) * 64 + lane];
    }
  }
  float bz = a.cst_helper[((int64_t)stage * kWavesPerStage + W) * 64 + lane];
  f32x4s bz4 = f32x4s{0.f, 0.f, 0.f, 0.f};          // MF: the constants of rows 4 g .. 4 g + 3 (lane 4 j of the table holds row j's)
  if constexpr (MF) {
    const float* cst = a.cst_helper + ((int64_t)stage * kWavesPerStage + W) * 64 + 16 * ((lane >> 2) & 3);
    bz4 = f32x4s{cst[0], cst[4], cst[8], cst[12]};
    asm volatile("" : "+v"(bz4));
  }
  // (in their registers before the loop, as in the chain role)
#pragma unroll
  for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(w0[i]));
#pragma unroll
  for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(wc[i]));
  asm volatile("" : "+v"(bz));
  const int d = a.dil[stage], ring_mask = a.ring[stage] - 1;
  const int64_t slot_stride = (int64_t)a.Bmax * kC;
  const int64_t stage_words = (int64_t)a.Bmax * kSpSlots * kMsgFloats;
  const int B = a.B, Bcap = bias_cap(a.B);
  const int n_visits = (int)a.n_steps * B;
  const int64_t hid_words = (int64_t)a.Bmax * kSpSlots * kH1;
  const bool hid_chain_in = stage >= 2 && ((stage - 1 + slot_shift(a)) >> 2) == ((stage + slot_shift(a)) >> 2);      // (as in the chain role)
  // The rows a bias is multiplied with - this stage's delayed input and the projected conditioning row, 1 KB each - come into the CU ONCE:
  // helper v mod 4 asks for those of visit v (lane l: floats 4 l .. 4 l + 3 of each) and stages them in LDS, all four helpers read their
  // K slices from there.  (Every lane asking for its own 2 x 64 bytes - the form of round 3 - moves 32 KB per visit through the CU's
  // address unit, 64 bytes per clock: ~500 clocks per visit, in front of every look, every publish and every hand-over of that CU.)
  // A helper has one request in flight: asked for three iterations before its visit's bias is due where the row is old enough for that
  // (d >= 3: the arrival that has been seen then covers it, and a row 512 steps old comes from HBM), one iteration before otherwise.
  const int ahead = (d - 2) * a.B >= 2 ? 2 : 0;      // (visit it + B + 2's row was written by visit it + B + 2 - d B; proven complete: up to it - B)
  // With eight clips or more the biases run FOUR iterations behind: the helper that has looked in iteration `it` asks for the rows of visit
  // it + B - 1 and makes that bias in its next duty iteration, it + 4 (still B - 5 visits before the chain wave needs it).  Then nothing
  // in a helper's loop waits for memory in the steady state - neither for a conditioning row from HBM nor, in a dilation-1 stage, for the
  // stage's own output of the visit before (which the chain waves may still be working on when the next message is already staged:
  // with the bias of visit it + B made in iteration it + 1, the helpers - and with them the staging of the next messages - went at the
  // pace of chain visit + look round trip + bias products: 1.7 us per visit in front of stages 10 and 20, the ring's beat).
  // (one iteration behind where the rows cannot be asked for ahead of time - dilation 1 and 2: a row asked for at the end of the duty
  //  iteration and staged at the start of the next one is waited for, ~0.3 us in front of everything that helper does in that iteration;
  //  in a free run - nothing waits for messages - those stages take 1.60 - 1.66 us per visit against 1.54, and the slowest stage is the ring's beat)
  const int lag = LAG4 ? 4 : ((MMK_SP_LAG1 && B >= 8 && ahead == 0) ? 1 : 0);
  const int roff = lag == 4 ? B - 1 : B + ahead;          // the rows of visit it + roff are asked for in iteration it
  // who asks for (and stages) the rows of visit v: the helper that looks in iteration v - B + 1 (lag 4) or v - B - ahead (no lag), right after
  // its look duty
  auto rows_mine = [&](int v) { return (((unsigned)(v - roff)) & 3u) == (unsigned)h; };
  // ---- looking for messages -----------------------------------------------------------------------------------------------------------------
  const __amdgpu_buffer_rsrc_t inbox = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(a.msg + (int64_t)stage * stage_words), 0, -1, 0x00020000);
  const int look_off = 32 * lane;                                       // bytes inside one message
  const int st_off = ((lane & 1) ? kHalf : 0) + pad_of(8 * (lane >> 1));
  auto look = [&](int byte_off, u32x4s& lo, u32x4s& hi) {               // 32 bytes per lane past this CU's L1 (sc1), counted by the compiler
    asm volatile("" ::: "memory");              // (every look is a load of its own: two looks at one address must not be merged into one)
    lo = __builtin_amdgcn_raw_buffer_load_b128(inbox, byte_off, 0, 16);
    hi = __builtin_amdgcn_raw_buffer_load_b128(inbox, byte_off + 16, 0, 16);
  };
  // (diagnostic build, dbg 4: nothing waits for a message - every stage runs at the pace of its own work; results are wrong, the
  //  per-stage times say what a stage's service time is when its inbox is never empty)
  constexpr int kGap = MMK_SP_POLL_GAP_BY_MODE ? (LAG4 ? kPollGap : 0) : kPollGap;
  const bool freerun = SP_ABL(4);
  if (STAMPS && a.stamps && freerun && p == 0 && h == 0 && lane == 0) a.stamps[182 + stage] = __builtin_amdgcn_s_memrealtime();
  auto landed = [&](const u32x4s& lo, const u32x4s& hi) {
    if (freerun) return true;
    // (the poison word is the largest unsigned: one maximum of the eight words and one compare - as eight compares and their ands the check was ~20 instructions
    //  between a look's return and the staging of its message)
    const unsigned m = max(max(max(lo[0], lo[1]), max(lo[2], lo[3])), max(max(hi[0], hi[1]), max(hi[2], hi[3])));
    return __all(m != kSpPoison) != 0;
  };
  u32x4s pre_lo = u32x4s{0, 0, 0, 0}, pre_hi = u32x4s{0, 0, 0, 0};
  u64 n_polls = 0;
  const __amdgpu_buffer_rsrc_t ring = __builtin_amdgcn_make_buffer_rsrc(a.hist[stage], 0, -1, 0x00020000);      // rows other CUs write: past L1 (sc1)
  const __amdgpu_buffer_rsrc_t own = __builtin_amdgcn_make_buffer_rsrc(a.msg + (int64_t)(stage + 1) * stage_words, 0, -1, 0x00020000);
  const bool cond_lane = 4 * lane < a.C1;       // (C1 is a multiple of 16)
  u64 hs_t[6] = {0, 0, 0, 0, 0, 0}, hs_t0 = 0;      // diagnostic build: cycles in the phases of an iteration
  auto hstamp = [&](int k) {
    if (STAMPS && !(a.dbg & 64)) { const u64 t = __builtin_amdgcn_s_memtime(); hs_t[k] += t - hs_t0; hs_t0 = t; }
  };
  // request the rows of the bias of visit (c2, s2).  x: this stage's input at position t_{s2} - d, from the history ring (zeros
  // in front of the sequence) - or, d = 1 past step 0, from this stage's newest message: channels 4 l .. + 3 are words (l / 2) 16 +
  // (l & 1) 4 of it (per producing wave 8 x | 8 y).  c: my four floats of the projected conditioning row c[t] (LinearIO of input 1,
  // modules/io.py:115-122): the layer's 1x1 product with it (wavenet_v2.py:140-150) is multiplied here, beside the delayed-tap product
  auto request_rows = [&](int s2, int c2, bool from_own, u32x4s& xr, f32x4s& cr, bool with_cond = true) {
    if (from_own) {
      const int off = (((c2 * kSpSlots + ((s2 - 1) & 3)) * kMsgFloats) + (lane >> 1) * 16 + (lane & 1) * 4) * 4;
      xr = __builtin_amdgcn_raw_buffer_load_b128(own, off, 0, 16);
    } else {
      const int64_t tp = SP_ABL(2) ? a.t0 - 1 + s2 - 2 : a.t0 - 1 + s2 - d;     // (dbg 2: diagnostic build, timing only)
      if (tp >= 0) xr = __builtin_amdgcn_raw_buffer_load_b128(ring, (int)(((tp & ring_mask) * slot_stride + (int64_t)c2 * kC + 4 * lane) * 4), 0, 16);
      else xr = u32x4s{0, 0, 0, 0};
    }
    // (a repeated look at the stage's own message does not ask for the conditioning row again: that one comes from HBM, and the look
    //  behind it would wait for it - a wave's loads return in order)
    if (!with_cond) return;
    if (cond_lane && !SP_ABL(1)) cr = *reinterpret_cast<const f32x4s*>(a.cproj + ((int64_t)c2 * a.cond_steps + s2) * a.C1 + 4 * lane);
    else cr = f32x4s{0.f, 0.f, 0.f, 0.f};
  };
  // the requested rows into the LDS slot of visit v3, once every helper is through with the visit that used the slot before
  auto stage_rows = [&](unsigned v3, const u32x4s& xr, const f32x4s& cr) -> bool {
    if constexpr (MF) {      // into the batch's buffer, once every helper is through with the batch that used it before
      const unsigned kb = v3 >> 2;
      if (kb >= 2 && !lds_wait4(S.ready, 4 * kb - 4, a.err_flag)) return false;
      float* dst = &S.rowsb[kb & 1][v3 & 3][4 * lane];
      *reinterpret_cast<f32x4s*>(dst) = f32x4s{__uint_as_float(xr[0]), __uint_as_float(xr[1]), __uint_as_float(xr[2]), __uint_as_float(xr[3])};
      *reinterpret_cast<f32x4s*>(dst + 256) = cr;
      lds_signal(&S.rows_ready[v3 & 3], v3 + 1, lane);
      return true;
    }
    if (v3 >= (unsigned)kRowRing && !lds_wait4(S.ready, v3 - kRowRing + 1, a.err_flag)) return false;
    float* dst = &S.rows[v3 & (kRowRing - 1)][0][kRowSlice * (lane >> 2) + 4 * (lane & 3)];
    *reinterpret_cast<f32x4s*>(dst) = f32x4s{__uint_as_float(xr[0]), __uint_as_float(xr[1]), __uint_as_float(xr[2]), __uint_as_float(xr[3])};
    *reinterpret_cast<f32x4s*>(dst + kRowPad) = cr;
    lds_signal(&S.rows_ready[v3 & (kRowRing - 1)], v3 + 1, lane);
    return true;
  };
  // the bias of visit v3 = (c2, s2): W0 x_s[t - d] + W_1x1 c[t] + constants, into the LDS image the chain wave reads
  auto bias_of = [&](unsigned v3, int s2, int c2) -> bool {
    if (!lds_wait1(&S.rows_ready[v3 & (kRowRing - 1)], v3 + 1, a.err_flag)) return false;
    if SP_ABL(8) {             // (diagnostic build, timing only: no bias products - what the rest of the helpers' loop takes)
      lds_signal(&S.ready[h], v3 + 1, lane);
      return true;
    }
    const float* xs = &S.rows[v3 & (kRowRing - 1)][0][kRowSlice * ks];
    f32x2 acc[4][2];
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) acc[cc][0] = acc[cc][1] = f32x2{0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4s xf = *reinterpret_cast<const f32x4s*>(xs + 4 * i);
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        acc[cc][0] = fma2(f32x2{w0[cc * 4 + i][0], w0[cc * 4 + i][1]}, f32x2{xf[0], xf[1]}, acc[cc][0]);
        acc[cc][1] = fma2(f32x2{w0[cc * 4 + i][2], w0[cc * 4 + i][3]}, f32x2{xf[2], xf[3]}, acc[cc][1]);
      }
    }
    if (a.C1 > 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f32x4s cf = *reinterpret_cast<const f32x4s*>(xs + kRowPad + 4 * i);
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
          acc[cc][0] = fma2(f32x2{wc[cc * 4 + i][0], wc[cc * 4 + i][1]}, f32x2{cf[0], cf[1]}, acc[cc][0]);
          acc[cc][1] = fma2(f32x2{wc[cc * 4 + i][2], wc[cc * 4 + i][3]}, f32x2{cf[2], cf[3]}, acc[cc][1]);
        }
      }
    }
    float zc[4];
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) zc[cc] = (acc[cc][0][0] + acc[cc][0][1]) + (acc[cc][1][0] + acc[cc][1][1]);
    const float t = row_reduce_scatter4(zc[0], zc[1], zc[2], zc[3], ks);
    if ((lane & 3) == 0) S.bias[bias_off(h, s2 & 1, c2, j, Bcap)] = t + bz;
    lds_signal(&S.ready[h], v3 + 1, lane);
    return true;
  };
  // MF: a quarter of batch kb's products (chunk 0 waits for the batch's rows and clears the sums, chunk 3 adds up the K residues, adds the
  // constants and hands the four biases to the chain waves); only visits in [v_lo, v_hi) exist
  f32x4s macc[4] = {f32x4s{0.f, 0.f, 0.f, 0.f}, f32x4s{0.f, 0.f, 0.f, 0.f}, f32x4s{0.f, 0.f, 0.f, 0.f}, f32x4s{0.f, 0.f, 0.f, 0.f}};      // four sums in turn: a product does not wait for the one before
  auto mf_chunk = [&](int kb, int ch, int v_lo, int v_hi) -> bool {
    if (ch == 0) {
      const int vv = 4 * kb + (lane & 3);
      const bool need = vv >= v_lo && vv < v_hi;
      unsigned spins = 0;
      while (!__all(!need || __hip_atomic_load(&S.rows_ready[lane & 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= (unsigned)vv + 1)) {
        if (kLdsSleep > 0) __builtin_amdgcn_s_sleep(kLdsSleep);
        if (++spins > kSpinLimit || ((spins & 4095u) == 0 && __hip_atomic_load(a.err_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
          atomicExch(a.err_flag, 1);
          return false;
        }
      }
      __atomic_signal_fence(__ATOMIC_SEQ_CST);
#pragma unroll
      for (int e = 0; e < 4; ++e) macc[e] = f32x4s{0.f, 0.f, 0.f, 0.f};
    }
    // (the chunk as a compile-time constant: a register array indexed by a run-time value lives in scratch memory)
    auto products = [&](auto chc) {
      constexpr int CH = decltype(chc)::value;
      const float* xb = &S.rowsb[kb & 1][lane & 3][(CH >> 1) * 256 + 4 * (lane >> 4)];
#pragma unroll
      for (int mm = 0; mm < 8; ++mm) {
        constexpr int m0 = (CH & 1) * 8;
        const f32x4s xv = *reinterpret_cast<const f32x4s*>(xb + 16 * (m0 + mm));
        const f32x4s wv = (CH >> 1) ? wc[m0 + mm] : w0[m0 + mm];
#pragma unroll
        for (int e = 0; e < 4; ++e) macc[e] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv[e], xv[e], macc[e], 0, 0, 0);
      }
    };
    if (!SP_ABL(8))
    switch (ch) {
      case 0: products(std::integral_constant<int, 0>{}); break;
      case 1: products(std::integral_constant<int, 1>{}); break;
      case 2: if (a.C1 > 0) products(std::integral_constant<int, 2>{}); break;
      default: if (a.C1 > 0) products(std::integral_constant<int, 3>{}); break;
    }
    if (ch == 3) {
      f32x4s tot;
      const f32x4s msum = (macc[0] + macc[1]) + (macc[2] + macc[3]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {          // the four K residues sit 16, 32 and 48 lanes apart
        const auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(msum[r]), __float_as_uint(msum[r]), false, false);
        const float y = __uint_as_float(s16[0]) + __uint_as_float(s16[1]);
        const auto s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(y), __float_as_uint(y), false, false);
        tot[r] = __uint_as_float(s32[0]) + __uint_as_float(s32[1]);
      }
      const int vv = 4 * kb + (lane & 3);
      if (lane < 16 && vv >= v_lo && vv < v_hi) {
        const int s2 = vv / B, c2 = vv - s2 * B;
        *reinterpret_cast<f32x4s*>(&S.bias[bias_off(h, s2 & 1, c2, 4 * (lane >> 2), Bcap)]) = tot + bz4;
      }
      lds_signal(&S.ready[h], (unsigned)(4 * kb + 4 < v_hi ? 4 * kb + 4 : v_hi), lane);
    }
    return true;
  };
  u32x4s xr = u32x4s{0, 0, 0, 0};
  f32x4s cr = f32x4s{0.f, 0.f, 0.f, 0.f};
  // ---- step 0: the rows the warm-up wrote, one clip after the other -------------------------------------------------------------------------
  if constexpr (MF) {
    for (int kb = 0; 4 * kb < B; ++kb) {
      if (4 * kb + h < B) {
        request_rows(0, 4 * kb + h, false, xr, cr);
        if (!stage_rows((unsigned)(4 * kb + h), xr, cr)) return;
      }
      for (int ch = 0; ch < 4; ++ch)
        if (!mf_chunk(kb, ch, 0, B)) return;
    }
  } else {
    for (int c0 = 0; c0 < B; ++c0) {
      if (rows_mine(c0)) {
        request_rows(0, c0, false, xr, cr);
        if (!stage_rows((unsigned)c0, xr, cr)) return;
      }
      if (!bias_of((unsigned)c0, 0, c0)) return;
    }
  }
  int s = 0, c = 0;                               // visit it = (c, s)
  int sp = 0, cp = B - 1 - lag;                   // visit it - 1 - lag + B = (cp, sp): the bias that is due in iteration it
  int sa = roff / B, ca = roff % B;               // visit it + roff = (ca, sa): the rows asked for in iteration it
  int sl = 4 / B, cl = 4 % B;                     // visit it + 4 = (cl, sl): the message whose first look goes out in iteration it
  if (lag == 0 && ahead > 0)                      // (what iterations -2 and -1 would have asked for: positions the warm-up wrote)
    for (int k = 0; k < ahead; ++k) {
      const int vr = B + k;
      if (vr < n_visits && rows_mine(vr)) request_rows(vr / B, vr % B, false, xr, cr);
    }
  // A first look four visits ahead is a look at a slot of ANOTHER clip (or of this step + 1) only with four clips or more: the slot
  // (clip, step mod 4) was poisoned when step - 2 of that clip was published, which lies behind the visit that is being staged.  With
  // fewer clips, four visits ahead is up to four STEPS ahead - a slot that still holds the message of four steps ago.
  const bool lookahead = B >= 4;
  bool staged_next = false;                       // my next message is staged already (in the iteration before the duty)
#ifndef MMK_SP_SHIFT_ALL
#define MMK_SP_SHIFT_ALL 0      // (experiment: the shift below 40 clips too - valid from 9 clips on only)
#endif
  constexpr int shift = (MMK_SP_BIASSHIFT && (LAG4 || MMK_SP_SHIFT_ALL)) ? 1 : 0;      // the biases one iteration behind their rows' staging (with the four-iteration lag: 64 clips 84.9 -> 82.7 us per step)
  int spb = 0, cpb = 0;                           // (the visit staged in the iteration before)
  // this stage's message comes from another XCD (or, stage 0, from the head): a look is a ~0.8-us round trip there, ~0.3 inside an XCD
  const bool remote_in = stage == 0 ? ((a.L + slot_shift(a)) >> 2) != (slot_shift(a) >> 2) : ((stage - 1 + slot_shift(a)) >> 2) != ((stage + slot_shift(a)) >> 2);
  if (lookahead && h < n_visits) look(((((h % B) * kSpSlots + ((h / B) & 3)) * kMsgFloats) * 4) + look_off, pre_lo, pre_hi);   // the first look at "my" first message
  else pre_lo[0] = kSpPoison;
  for (int it = 0; it < n_visits; ++it) {
    if (STAMPS && !(a.dbg & 64)) hs_t0 = __builtin_amdgcn_s_memtime();
    const bool duty = (it & 3) == h;
    // ---- 1. the visit's message ------------------------------------------------------------------------------------------------------------
    // a landed message (in the look registers) into the LDS image of visit vv = (cc, ss); the chain waves are told
    // ... and behind it the hidden units' hand-over of the stage below for the same visit (this CU's 16 units), for the chain waves' sums
    auto fetch_hid = [&](int vv, int cc, int ss) -> bool {
      if (!hid_chain_in) return true;
      if SP_ABL(32) {          // (diagnostic build, timing only: the hidden sums' hand-over is not fetched)
        lds_signal(&S.hidin_ready[vv & 3], (unsigned)vv + 1, lane);
        return true;
      }
      const unsigned* src = a.hidmsg + (int64_t)stage * hid_words + ((int64_t)cc * kSpSlots + (ss & 3)) * kH1 + 16 * p + (lane & 15);
      unsigned w, spins = 0;
      for (;;) {
        w = msg_load(src);
        if (__all(w != kSpPoison) || freerun) break;
        if (++spins > kSpinLimit || ((spins & 1023u) == 0 && __hip_atomic_load(a.err_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
          atomicExch(a.err_flag, 1);
          return false;
        }
      }
      if (lane < 16) S.hidin[vv & (kXyRing - 1)][lane] = __uint_as_float(w);
      lds_signal(&S.hidin_ready[vv & 3], (unsigned)vv + 1, lane);
      return true;
    };
    auto stage_message = [&](int vv, int cc, int ss, const u32x4s& m_lo, const u32x4s& m_hi) {
      float* dst = &S.xy[vv & (kXyRing - 1)][st_off];
      *reinterpret_cast<f32x4s*>(dst) = f32x4s{__uint_as_float(m_lo[0]), __uint_as_float(m_lo[1]), __uint_as_float(m_lo[2]), __uint_as_float(m_lo[3])};
      *reinterpret_cast<f32x4s*>(dst + 4) = f32x4s{__uint_as_float(m_hi[0]), __uint_as_float(m_hi[1]), __uint_as_float(m_hi[2]), __uint_as_float(m_hi[3])};
      if (STAMPS && a.stamps && cc == 0 && ss + 1 == (int)a.n_steps && p == 0 && lane == 0) a.stamps[64 + stage] = __builtin_amdgcn_s_memrealtime();
      if (STAMPS && a.stamps && ss == (int)a.n_steps / 2 && p == 0 && lane == 0 && cc < 32) a.stamps[256 + 1024 + stage * 32 + cc] = __builtin_amdgcn_s_memrealtime();   // seen, every clip, the launch's middle step
      lds_signal(&S.arrived[vv & 3], (unsigned)vv + 1, lane);
#if MMK_SP_WAKEUP
      asm volatile("s_wakeup");            // the other waves of the workgroup out of their s_sleep: they look at the counter again at once
#endif
    };
    if (duty) {
      if (!staged_next) {
        // (the LDS image of visit it - 8 is overwritten: every chain wave has to be through with it)
        if (it >= kXyRing - 2 && !lds_wait4(S.hdone, (unsigned)it - (kXyRing - 2) + 1, a.err_flag)) return;
        const int off = ((c * kSpSlots + (s & 3)) * kMsgFloats) * 4 + look_off;
        // (one exit besides the message: the time-out.  With the error word looked at every 1024 polls the loop left by three ways and carried their flags
        //  to the staging - a wave that gives up still raises the word, the others run into their own time-out at about the same moment)
        unsigned spins = 0;
        while (!landed(pre_lo, pre_hi)) {
          if (++spins > kSpinLimit) {
            atomicExch(a.err_flag, 1);
            return;
          }
          if (kGap > 0 && spins > 1) __builtin_amdgcn_s_sleep(kGap);
          look(off, pre_lo, pre_hi);
        }
        if (STAMPS) n_polls += spins;
        stage_message(it, c, s, pre_lo, pre_hi);
        if (!fetch_hid(it, c, s)) return;
      }
      staged_next = false;
      if (lookahead && it + 4 < n_visits) look(((cl * kSpSlots + (sl & 3)) * kMsgFloats) * 4 + look_off, pre_lo, pre_hi);
      else pre_lo[0] = kSpPoison;
    } else if (MMK_SP_EARLY && (MMK_SP_EARLY == 1 || remote_in) && it + ((h - it) & 3) < n_visits && (((h - it) & 3) <= MMK_SP_EARLY_DEPTH) && (((h - it) & 3) == 1 || lookahead)) {
      // not my look duty: while message `it` is not staged (by its helper), I look for MY next one already (1 - 3 visits ahead) - when the
      // clips queue up it is there, and a look at another XCD's memory is a ~0.8-us round trip that would otherwise start only when the
      // messages before it have been staged one after the other: the four helpers' looks then run side by side
      const int nahead = (h - it) & 3, nd = it + nahead;
      const int cn = c + nahead >= B ? c + nahead - B : c + nahead, sn = c + nahead >= B ? s + 1 : s;
      const int off = ((cn * kSpSlots + (sn & 3)) * kMsgFloats) * 4 + look_off;
      unsigned spins = 0;
      while (__hip_atomic_load(&S.arrived[it & 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < (unsigned)it + 1) {
        if (!staged_next) {
          if (landed(pre_lo, pre_hi)) {
            if (nd < kXyRing - 2 || lds_min4(S.hdone) + (kXyRing - 2) >= (unsigned)nd + 1) {
              stage_message(nd, cn, sn, pre_lo, pre_hi);
              if (!fetch_hid(nd, cn, sn)) return;
              staged_next = true;
            } else if (kLdsSleep > 0) __builtin_amdgcn_s_sleep(kLdsSleep);
          } else {
            look(off, pre_lo, pre_hi);
          }
        } else if (kLdsSleep > 0) __builtin_amdgcn_s_sleep(kLdsSleep);
        if (++spins > kSpinLimit || ((spins & 1023u) == 0 && __hip_atomic_load(a.err_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
          atomicExch(a.err_flag, 1);
          return;
        }
      }
      __atomic_signal_fence(__ATOMIC_SEQ_CST);
    } else if (MMK_SP_BACKUP && (MMK_SP_BACKUP == 1 || remote_in) && ((it + 2) & 3) == h && it >= 2) {
      // Halfway between two look duties: a SECOND pair of eyes on message `it`, half a look's round trip behind the helper on duty, in
      // registers of its own (the first look at this helper's next message is in flight in the others).  A CU then looks twice per round
      // trip, and what a stage waits for is the slowest of its eight CUs: the mean of that maximum shrinks with the looks' period.
      // Whoever sees the message first stages it; the other one finds S.arrived set, or stages the same bytes once more.
      if (it >= kXyRing - 2 && !lds_wait4(S.hdone, (unsigned)it - (kXyRing - 2) + 1, a.err_flag)) return;
      const int off = ((c * kSpSlots + (s & 3)) * kMsgFloats) * 4 + look_off;
      u32x4s bk_lo, bk_hi;
      unsigned spins = 0;
      if (remote_in) __builtin_amdgcn_s_sleep(MMK_SP_G_REMOTE); else __builtin_amdgcn_s_sleep(MMK_SP_G_LOCAL);
      while (__hip_atomic_load(&S.arrived[it & 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < (unsigned)it + 1) {
        look(off, bk_lo, bk_hi);
        if (landed(bk_lo, bk_hi)) {
          stage_message(it, c, s, bk_lo, bk_hi);
          if (!fetch_hid(it, c, s)) return;
          break;
        }
        if (++spins > kSpinLimit || ((spins & 1023u) == 0 && __hip_atomic_load(a.err_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
          atomicExch(a.err_flag, 1);
          return;
        }
        if (kPollGap > 0) __builtin_amdgcn_s_sleep(kPollGap);
      }
      __atomic_signal_fence(__ATOMIC_SEQ_CST);
    } else if (!(MMK_SP_NOARRWAIT && (LAG4 || B >= 16)) && !lds_wait1(&S.arrived[it & 3], (unsigned)it + 1, a.err_flag)) return;
    hstamp(1);
    // ---- 2. the bias of visit it - 1 - lag + B (its rows were asked for one, three or four iterations ago) -------------------------------------
    if (it >= 1 + lag && it - 1 - lag + B < n_visits) {
      const unsigned v3 = (unsigned)(it - 1 - lag + B);
      if (rows_mine((int)v3)) {
        if (d == 1) {
          unsigned spins = 0;
          while (!freerun && !__all(xr[0] != kSpPoison && xr[1] != kSpPoison && xr[2] != kSpPoison && xr[3] != kSpPoison)) {
            if (++spins > kSpinLimit || ((spins & 1023u) == 0 && __hip_atomic_load(a.err_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
              atomicExch(a.err_flag, 1);
              return;
            }
            request_rows(sp, cp, true, xr, cr, false);
          }
        }
        if (!stage_rows(v3, xr, cr)) return;
      }
    }
    hstamp(4);
#if MMK_SP_DUTYFIRST
    // My look duty is the NEXT visit: I look for that message NOW and make this iteration's bias afterwards (it has B - 2 visits of
    // slack).  Otherwise all four helpers start their products when message `it` is staged, and nobody looks for message it + 1 until
    // the helper whose duty it is has finished them: a stage then takes (products + look) per visit however few clips wait - with 32
    // clips in 31 stages that, not the latency of one clip's trip, set the step (41 us up to 24 clips, + 1.3 us for every further one).
    if (B >= 4 && ((it + 1) & 3) == h && it + 1 < n_visits && !staged_next) {      // (fewer clips: the bias may be the one the next message waits for)
      const int vn = it + 1;
      const int cn = c + 1 == B ? 0 : c + 1, sn = c + 1 == B ? s + 1 : s;
      if (vn >= kXyRing - 2 && !lds_wait4(S.hdone, (unsigned)vn - (kXyRing - 2) + 1, a.err_flag)) return;
      const int off = ((cn * kSpSlots + (sn & 3)) * kMsgFloats) * 4 + look_off;
      unsigned spins = 0;
      while (!landed(pre_lo, pre_hi)) {
        if (++spins > kSpinLimit) {
          atomicExch(a.err_flag, 1);
          return;
        }
        if (kGap > 0 && spins > 1) __builtin_amdgcn_s_sleep(kGap);
        look(off, pre_lo, pre_hi);
      }
      if (STAMPS) n_polls += spins;
      stage_message(vn, cn, sn, pre_lo, pre_hi);
      if (!fetch_hid(vn, cn, sn)) return;
      staged_next = true;
    }
#endif
    if constexpr (MF) {
      // the batch before the one that visit it + B - 5 (staged in this iteration) belongs to: a quarter of its products per iteration
      const int u = it + B - 5, kbc = (u >> 2) - 1;
      if (it >= 5 && kbc >= (B >> 2) && 4 * kbc < n_visits) {
        if (!mf_chunk(kbc, u & 3, B, n_visits)) return;
        hstamp(5);
      }
    } else if (shift) {
      if (it >= 2 + lag && it - 2 - lag + B < n_visits) {
        if (!bias_of((unsigned)(it - 2 - lag + B), spb, cpb)) return;
        hstamp(5);
      }
      spb = sp; cpb = cp;
    } else if (it >= 1 + lag && it - 1 - lag + B < n_visits) {
      const unsigned v3 = (unsigned)(it - 1 - lag + B);
      if (!bias_of(v3, sp, cp)) return;
      hstamp(5);
    }
    // ---- 3. the rows of the bias of visit it + roff (the helper that looked: its next look duty is four visits away) ------------------------------
    if (duty && sa >= 1 && it + roff < n_visits) request_rows(sa, ca, d == 1, xr, cr);
    if (++ca == B) { ca = 0; ++sa; }
    if (++cp == B) { cp = 0; ++sp; }
    if (++cl == B) { cl = 0; ++sl; }
    hstamp(0);
    if (++c == B) { c = 0; ++s; }
  }
  if (STAMPS && a.stamps && freerun && p == 0 && h == 0 && lane == 0) a.stamps[144 + stage] = __builtin_amdgcn_s_memrealtime();
  if (STAMPS && a.stamps && stage == a.stamp_stage && p == 0 && h == 0 && lane == 0) {
    for (int k = 0; k < 6; ++k) a.stamps[6 + k] = hs_t[k];
    a.stamps[4] = 4 * n_polls;          // (helper 0 looks for a quarter of the visits)
  }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// head stage: workgroup p serves the clips c = p (mod 8): last skip product, Mish, second Linear, [temperature], argmax / draw,
// and the next step's embedded sample as the message for stage 0
// ------------------------------------------------------------------------------------------------------------------------------------
// The head's two products like a layer stage's: a lane holds a few rows x ONE K slice and the rows' totals come out of a reduce-scatter over
// the DPP row - 4 units x 16 inputs of y, then 8 logits x 8 hidden units.  (With a quarter / a half of a row's K per lane - the first
// form - the 512 threads read 128 KB of LDS per product, which the LDS serves in ~1000 clocks: both products were bound by that.)
constexpr int kYsSlice = 20;      // a 16-float K slice of y + 4 floats of padding: the 16 slices' 16-byte reads fall on different banks
constexpr int kHidSlice = 12;     // an 8-float K slice of the hidden units + 4

template <bool STAMPS>
__device__ void head_role(const WnSpipeArgs& a, int p) {
  __shared__ __attribute__((aligned(16))) float ys[16 * kYsSlice], hid[16 * kHidSlice], lg[kQ + 4];
  __shared__ int s_fail;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int o = tid >> 2, kq = tid & 3;           // the hidden unit this lane ends up with (4 lanes each), and which of the four it is
  const int rg = tid >> 4, ks = tid & 15;         // row group (4 hidden units / 8 logits) and K slice
  f32x4s w0[16], w2[16];                          // w0[cc * 4 + i]: unit 4 rg + cc, inputs 16 ks + 4 i ..; w2[cc * 2 + i]: logit 8 rg + cc, units 8 ks + 4 i ..
  float wt[2] = {0.f, 0.f}, bt = 0.f;
#pragma unroll
  for (int cc = 0; cc < 4; ++cc)
#pragma unroll
    for (int i = 0; i < 4; ++i) w0[cc * 4 + i] = *reinterpret_cast<const f32x4s*>(a.head_w0 + (int64_t)(4 * rg + cc) * kC + 16 * ks + 4 * i);
#pragma unroll
  for (int cc = 0; cc < 8; ++cc)
#pragma unroll
    for (int i = 0; i < 2; ++i) w2[cc * 2 + i] = *reinterpret_cast<const f32x4s*>(a.fc2_w + (int64_t)(8 * rg + cc) * kH1 + 8 * ks + 4 * i);
  const float b0 = a.head_b0[o];
  const float b2 = a.fc2_b[8 * rg + (ks >> 1)];
  if (a.learn_temp) {
    wt[0] = a.fc2_w[(int64_t)kQ * kH1 + lane];
    wt[1] = a.fc2_w[(int64_t)kQ * kH1 + 64 + lane];
    bt = a.fc2_b[kQ];
  }
  // (in their registers before the step loop, as in the chain role)
#pragma unroll
  for (int k = 0; k < 16; ++k) { asm volatile("" : "+v"(w0[k])); asm volatile("" : "+v"(w2[k])); }
  asm volatile("" : "+v"(wt[0]), "+v"(wt[1]), "+v"(bt));
  if (tid == 0) s_fail = 0;
  const int L = a.L;
  const int64_t stage_words = (int64_t)a.Bmax * kSpSlots * kMsgFloats;
  const int64_t hid_words = (int64_t)a.Bmax * kSpSlots * kH1;
  const unsigned* msg_in = a.msg + (int64_t)L * stage_words;
  unsigned* msg_out = a.msg;                                               // stage 0's inbox (another XCD: written through)
  // the XCDs' sums of hidden pre-activations (layers 0 .. L - 2): groups g_first .. g_last, thread (o, kq) adds up those with g = g_first + kq (mod 4)
  const int g_first = (1 + slot_shift(a)) >> 2, g_last = (L - 1 + slot_shift(a)) >> 2;
  const bool last_mine = L >= 2 && ((g_last - g_first) & 3) == kq;
  u64 hs_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, hs_t0 = 0;      // diagnostic build: cycles in the phases of a visit (workgroup 0)
  auto hstamp = [&](int k) {
    if (STAMPS && !(a.dbg & 64)) { const u64 t = __builtin_amdgcn_s_memtime(); hs_t[k] += t - hs_t0; hs_t0 = t; }
  };
  // wave 0: the embedded class as stage 0's message of step s1 (x = E[class], y = 0), in the producers' layout (per 8 channels: 8 x | 8 y)
  auto publish_class = [&](int c, int s1, int cls) {
    cls = cls < 0 ? 0 : (cls >= kQ ? kQ - 1 : cls);
    const f32x4s e = *reinterpret_cast<const f32x4s*>(a.emb + (int64_t)cls * kC + 4 * lane);
    const int off = (lane >> 1) * 16 + (lane & 1) * 4;
    u64* dx = reinterpret_cast<u64*>(msg_out + ((int64_t)c * kSpSlots + (s1 & 3)) * kMsgFloats + off);
    u64* dp = reinterpret_cast<u64*>(msg_out + ((int64_t)c * kSpSlots + ((s1 + 2) & 3)) * kMsgFloats + off);
    __hip_atomic_store(dx + 0, ((u64)msg_bits(e[1]) << 32) | msg_bits(e[0]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(dx + 1, ((u64)msg_bits(e[3]) << 32) | msg_bits(e[2]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(dx + 4, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(dx + 5, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const u64 pp = ((u64)kSpPoison << 32) | kSpPoison;
    __hip_atomic_store(dp + 0, pp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(dp + 1, pp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(dp + 4, pp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(dp + 5, pp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  // one word of an XCD's sum, looked at until it is there (every lane of the wave: the producers of a row are equally late)
  auto hid_word = [&](const unsigned* src, bool mine) -> float {
    unsigned w1 = 0, spins = 0;
    for (;;) {
      if (mine) w1 = msg_load(src);
      if (__all(!mine || w1 != kSpPoison) || SP_ABL(4)) break;
      if (++spins > kSpinLimit || ((spins & 1023u) == 0 && __hip_atomic_load(a.err_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
        atomicExch(a.err_flag, 1);
        s_fail = 1;
        return 0.f;
      }
      __builtin_amdgcn_s_sleep(2);
    }
    return mine ? __uint_as_float(w1) : 0.f;
  };
  if (wave == 0)
    for (int c = p; c < a.B; c += kCuPerStage) publish_class(c, 0, (int)a.idx[(int64_t)c * a.idx_rs + a.t0 - 1]);
  __syncthreads();
  if (STAMPS && a.stamps && (a.dbg & 4) && p == 0 && tid == 0) a.stamps[182 + a.L] = __builtin_amdgcn_s_memrealtime();
  for (int s = 0; s < (int)a.n_steps; ++s) {
    const int64_t tau = a.t0 - 1 + s;
    const int slot = s & 3;
    for (int c = p; c < a.B; c += kCuPerStage) {
      if (STAMPS) hs_t0 = __builtin_amdgcn_s_memtime();
      // ---- the sums of the XCDs below the last one: complete several stages before y gets here, collected while it travels ---------------
      float hacc = 0.f;
      const unsigned* hsrc = a.hidgrp + ((int64_t)c * kSpSlots + slot) * kH1 + o;
      if (L >= 2) {
        for (int g0 = g_first; g0 < g_last; g0 += 4) {                   // (uniform trip count; a thread's group is g0 + kq)
          const bool mine = g0 + kq < g_last;
          hacc += hid_word(hsrc + (int64_t)(g0 + kq) * hid_words, mine);
        }
      }
      hstamp(0);
      // the sum of the last XCD below (or beside) the head arrives about when y does: waves 1 - 7 look for it while wave 0 looks for y,
      // wave 0's lanes take their word of it in the same looks as y - nothing of it is left to wait for behind the barrier
      const unsigned* hlast = hsrc + (int64_t)g_last * hid_words;
      if (wave != 0 && L >= 2) hacc += hid_word(hlast, last_mine);
      if (wave == 0) {             // y of the last layer: channels 4 lane .. + 3
        const unsigned* src = msg_in + ((int64_t)c * kSpSlots + slot) * kMsgFloats + (lane >> 1) * 16 + 8 + (lane & 1) * 4;
        const bool want_h = L >= 2 && last_mine;
        u32x4s w4;
        unsigned wh1 = 0, spins = 0;
        for (;;) {
          if (want_h) wh1 = msg_load(hlast);
          asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(w4) : "v"(src) : "memory");
          const unsigned mx = max(max(max(w4[0], w4[1]), max(w4[2], w4[3])), want_h ? wh1 : 0u);      // (poison is the largest unsigned: one maximum, one compare)
          if (__all(mx != kSpPoison) || SP_ABL(4)) break;
          if (++spins > kSpinLimit) {
            atomicExch(a.err_flag, 1);
            s_fail = 1;
            break;
          }
        }
        if (want_h) hacc += __uint_as_float(wh1);
        *reinterpret_cast<f32x4s*>(ys + (lane >> 2) * kYsSlice + 4 * (lane & 3)) =
            f32x4s{__uint_as_float(w4[0]), __uint_as_float(w4[1]), __uint_as_float(w4[2]), __uint_as_float(w4[3])};
        if (STAMPS && a.stamps && c == 0 && s + 1 == (int)a.n_steps && lane == 0) a.stamps[64 + L] = __builtin_amdgcn_s_memrealtime();
      }
      __syncthreads();
      if (s_fail) return;
      hstamp(1);
      // ---- hidden units: (fc0 W_skip of the last layer) y: 4 units x 16 inputs per lane, totals by the row's reduce-scatter ---------------
      float hsum;
      {
        f32x2 acc[4][2];
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) acc[cc][0] = acc[cc][1] = f32x2{0.f, 0.f};
        const float* yk = ys + ks * kYsSlice;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const f32x4s yv = *reinterpret_cast<const f32x4s*>(yk + 4 * i);
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) {
            acc[cc][0] = fma2(f32x2{w0[cc * 4 + i][0], w0[cc * 4 + i][1]}, f32x2{yv[0], yv[1]}, acc[cc][0]);
            acc[cc][1] = fma2(f32x2{w0[cc * 4 + i][2], w0[cc * 4 + i][3]}, f32x2{yv[2], yv[3]}, acc[cc][1]);
          }
        }
        float zc[4];
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) zc[cc] = (acc[cc][0][0] + acc[cc][0][1]) + (acc[cc][1][0] + acc[cc][1][1]);
        hsum = row_reduce_scatter4(zc[0], zc[1], zc[2], zc[3], ks) + dpp_quad_sum(hacc);      // (unit 4 rg + ks / 4 = o, in its four lanes)
      }
      if (kq == 0) hid[(o >> 3) * kHidSlice + (o & 7)] = mish_fast(hsum + b0);
      __syncthreads();
      hstamp(2);
      // ---- logits: 8 classes x 8 hidden units per lane ---------------------------------------------------------------------------------------
      {
        const float* hk = hid + ks * kHidSlice;
        const f32x4s hv0 = *reinterpret_cast<const f32x4s*>(hk), hv1 = *reinterpret_cast<const f32x4s*>(hk + 4);
        float qs[8];
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) {
          f32x2 q = fma2(f32x2{w2[cc * 2][0], w2[cc * 2][1]}, f32x2{hv0[0], hv0[1]}, f32x2{0.f, 0.f});
          q = fma2(f32x2{w2[cc * 2][2], w2[cc * 2][3]}, f32x2{hv0[2], hv0[3]}, q);
          q = fma2(f32x2{w2[cc * 2 + 1][0], w2[cc * 2 + 1][1]}, f32x2{hv1[0], hv1[1]}, q);
          q = fma2(f32x2{w2[cc * 2 + 1][2], w2[cc * 2 + 1][3]}, f32x2{hv1[2], hv1[3]}, q);
          qs[cc] = q[0] + q[1];
        }
        const float qv = row_reduce_scatter8(qs, ks);                 // (class 8 rg + ks / 2, in two lanes)
        if ((ks & 1) == 0) lg[8 * rg + (ks >> 1)] = qv + b2;
      }
      if (wave == 0 && a.learn_temp) {
        float tv = fmaf(wt[0], hid[(lane >> 3) * kHidSlice + (lane & 7)], wt[1] * hid[(8 + (lane >> 3)) * kHidSlice + (lane & 7)]);
        tv = dpp_mirror_add(dpp_half_mirror_add(dpp_quad_sum(tv)));        // every lane: its row's sum (DPP; six ds_bpermute round trips before)
        tv = (readlane_f(tv, 0) + readlane_f(tv, 16)) + (readlane_f(tv, 32) + readlane_f(tv, 48));
        if (lane == 0) lg[kQ] = tv + bt;
      }
      __syncthreads();
      hstamp(3);
      if (wave == 0) {
        if (a.logits_out && s + 1 == (int)a.n_steps)
          for (int k = lane; k < kQ + (a.learn_temp ? 1 : 0); k += 64) a.logits_out[(int64_t)c * a.logits_ld + k] = lg[k];
        int result;
        if (a.temperature == nullptr) {
          // argmax of logits / max(sigmoid(t), min_temp) (mlp.py:60-62, targets.py:37-52): a division by one positive number keeps the
          // order, so the maximum of the raw logits is the answer - unless the division rounds an EARLIER, slightly smaller logit onto
          // the maximum's quotient (first-maximum rule).  Only then (some other logit within 4 ulp of the maximum) divide and compare.
          const f32x4s v4 = *reinterpret_cast<const f32x4s*>(lg + lane * 4);
          const float m = wave_max_dpp(fmaxf(fmaxf(v4[0], v4[1]), fmaxf(v4[2], v4[3])));
          // the first maximum (targets.py / torch.argmax): per k the lanes that hold it as a scalar mask, the lowest such lane, the smallest
          // 4 lane + k - scalar instructions beside the vector unit instead of a second trip through the DPP rows
          result = 0x7fffffff;
          bool odd = false, near = false;                         // NaN logits (argmax takes the first), or a logit within 4 ulp of the maximum
          const float lim = m - fmaxf(fabsf(m) * 4.8e-7f, 1e-37f);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const unsigned long long mk = __ballot(v4[k] == m);
            if (mk) result = min(result, 4 * (int)__builtin_ctzll(mk) + k);
            odd = odd || v4[k] != v4[k];
            near = near || (v4[k] != m && v4[k] >= lim);
          }
          if (__any(odd)) {
            int cand = 0x7fffffff;
#pragma unroll
            for (int k = 3; k >= 0; --k)
              if (v4[k] != v4[k]) cand = lane * 4 + k;
            result = wave_min_dpp(cand);
          } else if (a.learn_temp && __any(near)) {
            const float denom = fmaxf(sigmoidf_(lg[kQ]), a.min_temp);
            float vv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) vv[k] = v4[k] / denom;
            float best = vv[0];
            int bi = lane * 4;
#pragma unroll
            for (int k = 1; k < 4; ++k)
              if (vv[k] > best) { best = vv[k]; bi = lane * 4 + k; }
            result = wave_argmax_first(best, bi);
          }
          result = result > kQ - 1 ? kQ - 1 : result;
        } else {
          float denom = 1.f;
          if (a.learn_temp) denom = fmaxf(sigmoidf_(lg[kQ]), a.min_temp);       // mlp.py:60-62
          result = sample_256(lg, a.learn_temp != 0, denom, a.temperature[c], a.uniforms[(int64_t)c * a.uni_ld + s], lane);
        }
        hstamp(4);
        if (s + 1 < (int)a.n_steps) publish_class(c, s + 1, result);
        if (a.stamps && c == 0 && lane == 0 && (s + 1 == (int)a.n_steps || s + 2 == (int)a.n_steps))
          a.stamps[16 + a.L + (s + 2 == (int)a.n_steps ? 1 : 2)] = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) a.idx[(int64_t)c * a.idx_rs + tau + 1] = result;
        hstamp(5);
      }
      // (no barrier here: the next visit's first shared write - ys, by wave 0 - comes after wave 0 has read lg, and nobody reads ys or
      //  hid of this visit past the barrier above)
    }
  }
  if (STAMPS && a.stamps && p == 0 && tid == 0)
    for (int k = 0; k < 6; ++k) a.stamps[176 + k] = hs_t[k];
  if (STAMPS && a.stamps && (a.dbg & 4) && p == 0 && tid == 0) a.stamps[144 + a.L] = __builtin_amdgcn_s_memrealtime();
}

#include "wavenet_spipe_pair.inc"

template <bool STAMPS, bool LAG4>
__global__ __launch_bounds__(kThreads) void wavenet_spipe_kernel(const WnSpipeArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char spipe_lds[];      // Lds + the bias image of bias_cap(B) clips (launch_wavenet_spipe sizes it)
  Lds& S = *reinterpret_cast<Lds*>(spipe_lds);
  __shared__ int s_role;
  const int tid = threadIdx.x;
  // Roles come from where the workgroup RUNS: XCD x hosts stages 4 x .. 4 x + 3, eight workgroups each, in arrival order.  A launch
  // that does not put 32 workgroups on every XCD reports error 2 (the caller falls back to the launch path).
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
  if (tid < 20) (&S.arrived[0])[tid] = 0;       // arrived, hdone, rows_ready, ready, hidin_ready are adjacent
  __syncthreads();
  const int role = s_role;
  if (role < 0) return;
  // slot = role / 8 of the chip's 32 (four per XCD); the first kSlotShift slots stay empty, so that layer l sits in slot l + kSlotShift
  const int stage = (role >> 3) - slot_shift(a), p = role & 7;
  if (stage < 0 || stage > a.L) return;
  if (stage == a.L) {
    head_role<STAMPS>(a, p);
    return;
  }
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (wave < 4) {
    chain_role<STAMPS, LAG4>(a, S, stage, p, wave, lane);
  } else {
    helper_role<STAMPS, LAG4>(a, S, stage, p, wave - 4, lane);
  }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// commit: the per-lane register images.  One thread per float4; composed entries (W1 R, fc0 W_skip) are fp64 dot products of length C.
// ------------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int gate_raw_row(int W, int j) { return (j & 1) ? kC + 8 * W + (j >> 1) : 8 * W + (j >> 1); }

__device__ double dot_cols(const float* arow, int64_t a_stride, const float* bcol, int64_t b_stride, int n) {
  double acc = 0.0;
  for (int c = 0; c < n; ++c) acc += (double)arow[c * a_stride] * (double)bcol[c * b_stride];
  return acc;
}

__global__ __launch_bounds__(256) void spipe_image_kernel(const WnSpRaw* __restrict__ raw, int L, int C1, const float* __restrict__ f0, const float* __restrict__ fb0,
                                                          float* __restrict__ img_chain, float* __restrict__ img_helper, float* __restrict__ cst_chain,
                                                          float* __restrict__ cst_helper, float* __restrict__ head_w0, float* __restrict__ head_b0) {
  const int64_t n_chain = (int64_t)L * kWavesPerStage * kChainRegs * 64, n_helper = (int64_t)L * kWavesPerStage * kHelperRegs * 64;
  const int64_t n_cst = (int64_t)L * kWavesPerStage * 64, n_hw = (int64_t)kH1 * kC / 4, n_hb = kH1;
  const int64_t total = n_chain + n_helper + 2 * n_cst + n_hw + n_hb;
  for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (int64_t)gridDim.x * blockDim.x) {
    if (id < n_chain) {
      const int lane = (int)(id & 63), qi = (int)((id >> 6) % kChainRegs), W = (int)((id / (kChainRegs * 64)) % kWavesPerStage), s = (int)(id / (kChainRegs * 64 * kWavesPerStage));
      const WnSpRaw r = raw[s];
      float out[4] = {0.f, 0.f, 0.f, 0.f};
      if (qi < 32) {
#if MMK_SP_ROWS8
        // register cc * 4 + i: gate row 8 (lane / 32) + cc of the wave, inputs 16 (lane & 15) + 4 i .. of x (even rows of 16 lanes) or y (odd rows).
        // After the swap of halves even rows hold the sums of cc = 0 .. 3, odd rows of cc = 4 .. 7, and the row's reduce-scatter leaves lane l
        // with gate row 8 (l / 32) + 4 (l / 16 & 1) + (l & 15) / 4 = l / 4, as in the other form.
        const int ks = lane & 15, cc = qi >> 2, i = qi & 3, n = gate_raw_row(W, 8 * (lane >> 5) + cc);
        const bool xpart = (lane & 16) == 0;
        for (int e = 0; e < 4; ++e) {
          const int k = 16 * ks + 4 * i + e;
#else
        // register cc * 8 + i: gate row 4 (lane / 16) + cc of the wave, inputs 32 (ks & 7) + 4 i .. of x (ks < 8) or y
        const int ks = lane & 15, cc = qi >> 3, i = qi & 7, n = gate_raw_row(W, 4 * (lane >> 4) + cc);
        const bool xpart = ks < 8;
        for (int e = 0; e < 4; ++e) {
          const int k = 32 * (ks & 7) + 4 * i + e;
#endif
          if (xpart) out[e] = r.wd[((int64_t)n * kC + k) * 2 + 1];                                                        // W1[n][k]
          else if (s >= 1 && raw[s - 1].wr) out[e] = (float)dot_cols(r.wd + (int64_t)n * kC * 2 + 1, 2, raw[s - 1].wr + k, kC, kC);   // (W1 R)[n][k]
        }
      } else if (qi < 40) {
        if (s >= 1 && raw[s - 1].wr) {      // register 32 + cc * 4 + i: residual channel 2 (lane / 16) + cc of the wave, inputs 16 ks + 4 i .. of y
          const int ks = lane & 15, cc = (qi - 32) >> 2, i = (qi - 32) & 3;
          for (int e = 0; e < 4; ++e) out[e] = raw[s - 1].wr[(int64_t)(8 * W + 2 * (lane >> 4) + cc) * kC + 16 * ks + 4 * i + e];
        }
      } else if (s >= 1) {                  // register 40 + i: hidden unit 4 W + lane / 16 of the head's first Linear, inputs 16 ks + 4 i .. of y
        const int j4 = lane >> 4, ks16 = lane & 15, i = qi - 40, hrow = 4 * W + j4;
        for (int e = 0; e < 4; ++e) out[e] = (float)dot_cols(f0 + (int64_t)hrow * kC, 1, raw[s - 1].ws + 16 * ks16 + 4 * i + e, kC, kC);   // (fc0 W_skip)[h][k]
      }
      reinterpret_cast<f32x4s*>(img_chain)[id] = f32x4s{out[0], out[1], out[2], out[3]};
    } else if (id < n_chain + n_helper) {
      const int64_t t = id - n_chain;
      const int lane = (int)(t & 63), qi = (int)((t >> 6) % kHelperRegs), W = (int)((t / (kHelperRegs * 64)) % kWavesPerStage), s = (int)(t / (kHelperRegs * 64 * kWavesPerStage));
      const WnSpRaw r = raw[s];
      float out[4] = {0.f, 0.f, 0.f, 0.f};
      if (qi < 16) {      // register cc * 4 + i: gate row 4 (lane / 16) + cc, delayed inputs 16 ks + 4 i ..
        const int ks = lane & 15, cc = qi >> 2, i = qi & 3, n = gate_raw_row(W, 4 * (lane >> 4) + cc);
        for (int e = 0; e < 4; ++e) out[e] = r.wd[((int64_t)n * kC + 16 * ks + 4 * i + e) * 2 + 0];                        // W0[n][k]
      } else {      // register 16 + cc * 4 + i: the same rows of the conditioning 1x1 convolution, inputs 16 ks + 4 i .. (zeros above C1)
        const int ks = lane & 15, cc = (qi - 16) >> 2, i = (qi - 16) & 3, n = gate_raw_row(W, 4 * (lane >> 4) + cc);
        for (int e = 0; e < 4; ++e) {
          const int k = 16 * ks + 4 * i + e;
          out[e] = (r.w1 && k < C1) ? r.w1[(int64_t)n * C1 + k] : 0.f;
        }
      }
      reinterpret_cast<f32x4s*>(img_helper)[t] = f32x4s{out[0], out[1], out[2], out[3]};
    } else if (id < n_chain + n_helper + n_cst) {
      const int64_t t = id - n_chain - n_helper;
      const int lane = (int)(t & 63), W = (int)((t >> 6) % kWavesPerStage), s = (int)(t / (64 * kWavesPerStage));
      float bx = 0.f;
      if (s >= 1 && raw[s - 1].wr && raw[s - 1].br) bx = raw[s - 1].br[8 * W + (lane >> 3)];
      cst_chain[t] = bx;
    } else if (id < n_chain + n_helper + 2 * n_cst) {
      const int64_t t = id - n_chain - n_helper - n_cst;
      const int lane = (int)(t & 63), W = (int)((t >> 6) % kWavesPerStage), s = (int)(t / (64 * kWavesPerStage));
      const WnSpRaw r = raw[s];
      const int n = gate_raw_row(W, lane >> 2);
      // b_dil + b_1x1, then tap 1 . b_res of the layer below: the order the one-hand-off kernels add them in
      float bz = (r.bd ? r.bd[n] : 0.f) + (r.b1 ? r.b1[n] : 0.f);
      if (s >= 1 && raw[s - 1].wr && raw[s - 1].br) bz += (float)dot_cols(r.wd + (int64_t)n * kC * 2 + 1, 2, raw[s - 1].br, 1, kC);
      cst_helper[t] = bz;
    } else if (id < n_chain + n_helper + 2 * n_cst + n_hw) {
      const int64_t t = id - n_chain - n_helper - 2 * n_cst;
      const int hrow = (int)(t / (kC / 4)), k0 = (int)(t % (kC / 4)) * 4;
      float out[4];
      for (int e = 0; e < 4; ++e) out[e] = (float)dot_cols(f0 + (int64_t)hrow * kC, 1, raw[L - 1].ws + k0 + e, kC, kC);
      reinterpret_cast<f32x4s*>(head_w0)[t] = f32x4s{out[0], out[1], out[2], out[3]};
    } else {
      const int hrow = (int)(id - (n_chain + n_helper + 2 * n_cst + n_hw));
      double acc = fb0 ? (double)fb0[hrow] : 0.0;
      for (int l = 0; l < L; ++l)
        if (raw[l].bs) acc += dot_cols(f0 + (int64_t)hrow * kC, 1, raw[l].bs, 1, kC);
      head_b0[hrow] = (float)acc;
    }
  }
}

}  // namespace

bool wn_spipe_supported(int C, int S, int H1, int n_classes, int L, int n_cond, int cond_dim, int batch) {
  // (the helper waves multiply 16 K slices of 16 conditioning channels: at most 256 of them, in whole slices)
  // (cond_dim: the widths of ALL conditioning inputs together - the plan hands their projections over side by side, as one row)
  // (a head of fewer hidden units or classes runs as the 128 x 256 one: the plan pads its matrices with zero rows / columns and gives the
  //  missing classes a bias of -inf)
  const bool cond_ok = n_cond == 0 || (n_cond >= 1 && n_cond <= 2 && cond_dim > 0 && cond_dim <= kC && cond_dim % 16 == 0);
  return C == kC && S == kC && H1 >= 1 && H1 <= kH1 && n_classes >= 2 && n_classes <= kQ && cond_ok && L >= 1 && L <= kSpMaxLayers && batch >= 1 &&
         batch <= kSpMaxClips;
}
int64_t wn_spipe_img_chain_floats(int L, int C) { return (int64_t)L * (C / 8) * kChainRegs * 64 * 4; }
int64_t wn_spipe_img_helper_floats(int L, int C) { return (int64_t)L * (C / 8) * kHelperRegs * 64 * 4; }
int64_t wn_spipe_cst_floats(int L, int C) { return (int64_t)L * (C / 8) * 64; }
int64_t wn_spipe_msg_words(int L, int C, int Bmax) { return (int64_t)(L + 1) * Bmax * kSpSlots * 2 * C; }
int64_t wn_spipe_hidmsg_words(int L, int Bmax) { return (int64_t)(L + 1) * Bmax * kSpSlots * kH1; }
int64_t wn_spipe_hidgrp_words(int Bmax) { return (int64_t)8 * Bmax * kSpSlots * kH1; }

int wn_spipe_build_image(const WnSpRaw* raw_dev, int L, int C, int C1, const float* f0, const float* fb0, float* img_chain, float* img_helper,
                         float* cst_chain, float* cst_helper, float* head_w0, float* head_b0, hipStream_t stream) {
  if (C != kC || L < 1 || L > kSpMaxLayers || C1 < 0 || C1 > kC) return fail(MMK_ERR_UNSUPPORTED, "wavenet stage pipeline: C = %d, L = %d, C1 = %d", C, L, C1);
  hipLaunchKernelGGL(spipe_image_kernel, dim3(2048), dim3(256), 0, stream, raw_dev, L, C1, f0, fb0, img_chain, img_helper, cst_chain, cst_helper, head_w0,
                     head_b0);
  MMK_HIP(hipGetLastError());
  return MMK_OK;
}

int launch_wavenet_spipe(const WnSpipeArgs& a, hipStream_t stream) {
  if (a.n_steps <= 0 || a.B <= 0) return MMK_OK;
  if (a.B > kSpMaxClips || a.L > kSpMaxLayers || a.C != kC) return fail(MMK_ERR_UNSUPPORTED, "wavenet stage pipeline: %d clips, %d layers, %d channels", a.B, a.L, a.C);
  const size_t lds = sizeof(Lds) + (size_t)4 * 2 * bias_cap(a.B) * 16 * sizeof(float);
  {      // (more than 64 KB of dynamic LDS has to be asked for; per launch: the attribute belongs to the current device)
    const size_t lds_max = sizeof(Lds) + (size_t)4 * 2 * kSpMaxClips * 16 * sizeof(float);
    MMK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wavenet_spipe_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
    MMK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wavenet_spipe_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
#ifdef MMK_DIAG
    MMK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wavenet_spipe_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
    MMK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wavenet_spipe_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
#endif
  }
  // two clips per visit (wavenet_spipe_pair.inc) where the ring goes at the stages' beat: a.pair 1 asks for it, 0 refuses it, < 0 leaves it to the clip count
  // (diagnostic build: the pair form takes the two wall-clock stamps per visit only; the phase stamps belong to the one-clip form: MMK_WN_SPIPE_PAIR=0)
  const bool pair = wn_spipe_pair_form(a.B, a.pair);
  if (pair) {
    const size_t lds2 = sizeof(Lds2) + (size_t)4 * 2 * bias_cap(a.B) * 16 * sizeof(float);
    MMK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wavenet_spipe_pair_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(sizeof(Lds2) + (size_t)4 * 2 * kSpMaxClips * 16 * sizeof(float))));
    hipLaunchKernelGGL(wavenet_spipe_pair_kernel, dim3(256), dim3(kThreads), lds2, stream, a);
    MMK_HIP(hipGetLastError());
    return MMK_OK;
  }
  const bool lag4 = MMK_SP_LAG && a.B >= MMK_SP_LAG_CLIPS && a.B >= 12;      // (the biases four iterations behind the messages: helper_role)
#ifdef MMK_DIAG
  if (a.stamps) {      // the stamped instantiation (and its timing switches) exist in the diagnostic build only
    if (lag4) hipLaunchKernelGGL((wavenet_spipe_kernel<true, true>), dim3(256), dim3(kThreads), lds, stream, a);
    else hipLaunchKernelGGL((wavenet_spipe_kernel<true, false>), dim3(256), dim3(kThreads), lds, stream, a);
    MMK_HIP(hipGetLastError());
    return MMK_OK;
  }
#endif
  if (lag4) hipLaunchKernelGGL((wavenet_spipe_kernel<false, true>), dim3(256), dim3(kThreads), lds, stream, a);
  else hipLaunchKernelGGL((wavenet_spipe_kernel<false, false>), dim3(256), dim3(kThreads), lds, stream, a);
  MMK_HIP(hipGetLastError());
  return MMK_OK;
}

}  // namespace mmk

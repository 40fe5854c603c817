// WaveNet generation as a PIPELINE OF WORKGROUPS THAT OWN WHOLE LAYERS (gfx950) - for networks small enough that a layer's matrices
// fit a fraction of a CU's registers: C = S = 64 channels, kernel 2, gated, MLP head 64 -> 128 -> 256 (+ temperature)
// (BASELINE config 2: ten layers, dilations 1 .. 512, 8 clips), with up to two conditioning inputs - whose products sum_j conv_1x1_j(c_j)
// (wavenet_v2.py:141-147) the plan forms for a block of positions ahead of the launch, as for the other persistent kernels: a thread reads
// its gate row's term with the delayed taps, off the step's chain.
//
// Reference: WaveNet.forward / WNLayer.forward (wavenet_v2.py:120-180, :277-296), MLP head and CategoricalSampler (networks/mlp.py:58-63,
// modules/targets.py:37-52) - the same arithmetic as the other step kernels, on the launch path's packed matrices.
//
// The chain kernel (wavenet_chain.hip) spreads every layer over C / 8 workgroups and pays one cross-CU exchange per layer: 11 dependent
// exchanges of ~1.5 us bound a step of this network at 18 us.  Here a clip is served by four workgroups (placed on one XCD: workgroup b
// serves clip (b % 8) + 8 (b / 32) as stage (b / 8) % 4); a stage keeps the matrices of its two or three layers in registers for the
// whole launch (48 floats per thread and layer: thread (o, kq) holds a quarter of output row o) and runs them back to back out of LDS
// (a layer = 32 + 16 FMAs per thread, two quad reductions, the gate across two neighbouring quads by DPP, two workgroup barriers;
// four barriers, single FMA chains or no barrier at the end of a step change the step by < 0.3 us: it is the four crossings, ~1.2 us
// each with the poll, and the head that make up most of its 11.5 us).  A step crosses the chip four times instead of
// eleven: the layer input and the skip sum travel to the next stage as 128 data-tagged 8-byte granules {step + 1, value}, the head's
// class goes back to stage 0 the same way.  Where producer and consumer sit on one XCD (every workgroup registers its XCC id at the
// start and reads its successor's) the granules are plain stores that stay in that XCD's L2; one wave of the consumer polls all 128
// with one 16-byte L1-bypassing load per lane.  9.2 -> 8.4 us per step against write-through hand-overs.  The delayed taps x_l[t - d_l] are read from the launch path's history rings in global
// memory (L2), requested at the start of a stage's visit, and every layer input is written there - so the warm-up is the same prefill
// as for the other persistent kernels, scattered into those rings.  scripts/probes/wn_layer_pipe.hip is the stand-alone form.
#include "wavenet_lpipe.h"
#include "sampler256.h"

namespace mmk {

namespace {

typedef unsigned long long u64;
typedef float f32x4_lp __attribute__((ext_vector_type(4)));
constexpr int kC = 64, kH1 = 128, kQ = 256;
constexpr int kLpThreads = 512;
constexpr unsigned kLpSpinLimit = 1u << 22;

// element (row n, column k) of a matrix packed for the fused-linear kernel: Wp[tile][chunk][lane = 16 (k % 16 / 4) + n % 16][k % 4]
__device__ __forceinline__ float packed_at(const float* wp, int k_chunks, int n, int k) {
  return wp[((((int64_t)(n >> 4) * k_chunks + (k >> 4)) * 64) + ((k & 15) >> 2) * 16 + (n & 15)) * 4 + (k & 3)];
}

__device__ __forceinline__ float quad_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
  return v;
}

__device__ __forceinline__ u64 poll(const u64* p, unsigned epoch, int32_t* err) {
  u64 g = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  unsigned spins = 0;
  while ((unsigned)(g >> 32) != epoch) {
    // ~1 s: the other stages are not running beside this one (or another wait has already failed: do not pile up)
    if (++spins > kLpSpinLimit || (MMK_WAIT_ERR_LOOK && (spins & 1023u) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
      atomicExch(err, 3);
      break;
    }
    g = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return g;
}

// (forced inline: left as a function, a stage takes the argument block by address and the compiler copies all of it to scratch - every
//  pointer and count read from it afterwards then sits in vector registers the matrices need)
template <int NL, bool HEAD, bool FIRST, bool COND>
__device__ __forceinline__ void run_stage(const WnLpipeArgs& a, int clip, int stage, int l0, float* embs) {
  __shared__ __attribute__((aligned(16))) float xs[2][kC], taps[3][kC], zs[kC], sk[kC], hid[kH1], lg[kQ + 4], cnds[COND ? 3 : 1][2 * kC];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int o = tid >> 2, kq = tid & 3;
  const bool g_quad = (o & 1) != 0;
  const float gate_scale = g_quad ? -1.4426950408889634f : -2.8853900817779268f;
  const float gate_k = g_quad ? 1.f : 2.f, gate_shift = g_quad ? 0.f : -1.f;
  // ---- this stage's matrices -> registers (once per launch) -------------------------------------------------------------------
  float wc[NL][32], wr[NL][16], bc[NL], br[NL];
  int dil[NL], has_res[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const WnLayerTab lt = a.layers[l0 + i];
    dil[i] = __builtin_amdgcn_readfirstlane(lt.dil);            // (wave-uniform: scalar registers - the vector ones are all taken by the matrices)
    has_res[i] = __builtin_amdgcn_readfirstlane(lt.has_res);
    // (f, g) rows in packed (interleaved) order: quad 2 u holds f of unit u, quad 2 u + 1 its g - four lanes apart, so the gate needs
    // no LDS round trip; K = [x(t - d) | x(t)]
    const int ra = o;
    // wc[.][0 .. 15]: this lane's sixteenth-to-quarter of the TAP columns, wc[.][16 .. 31]: the same columns of x(t) - the tap half of
    // every layer's product is formed before the step's input arrives
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      wc[i][k] = packed_at(lt.A_wp, a.kcA, ra, kq * 16 + k);
      wc[i][16 + k] = packed_at(lt.A_wp, a.kcA, ra, kC + kq * 16 + k);
    }
    bc[i] = lt.A_bias ? lt.A_bias[ra] : 0.f;
    // [res ; skip] rows: without residual rows (the last layer) the skip rows come first
    const int rb = lt.has_res ? o : o - kC;
#pragma unroll
    for (int k = 0; k < 16; ++k) wr[i][k] = rb >= 0 ? packed_at(lt.B_wp, 4, rb, kq * 16 + k) : 0.f;
    br[i] = (rb >= 0 && lt.B_bias) ? lt.B_bias[rb] : 0.f;
  }
  float w0[HEAD ? 16 : 1], w2[HEAD ? 64 : 1], wt[HEAD ? 2 : 1], b0 = 0.f, b2 = 0.f, bt = 0.f;
  if constexpr (HEAD) {
#pragma unroll
    for (int k = 0; k < 16; ++k) w0[k] = packed_at(a.fc0_wp, 4, o, kq * 16 + k);
    b0 = a.fc0_bias[o];
#pragma unroll
    for (int k = 0; k < 64; ++k) w2[k] = packed_at(a.fc2_wp, 8, tid >> 1, (tid & 1) * 64 + k);
    b2 = a.fc2_bias[tid >> 1];
    wt[0] = wt[1] = 0.f;
    if (a.learn_temp) {   // row 256: the temperature column, on wave 0 (two inputs per lane)
      wt[0] = packed_at(a.fc2_wp, 8, kQ, lane);
      wt[1] = packed_at(a.fc2_wp, 8, kQ, 64 + lane);
      bt = a.fc2_bias[kQ];
    }
  }
  // (IN their registers before the step loop: a load the compiler still counts as pending at the loop's entry makes it wait inside
  //  every step, and such a wait also covers the step's own written-through stores - see wavenet_spipe.hip)
#pragma unroll
  for (int i = 0; i < NL; ++i) {
#pragma unroll
    for (int k = 0; k < 32; ++k) asm volatile("" : "+v"(wc[i][k]));
#pragma unroll
    for (int k = 0; k < 16; ++k) asm volatile("" : "+v"(wr[i][k]));
    asm volatile("" : "+v"(bc[i]), "+v"(br[i]));
  }
  if constexpr (HEAD) {
#pragma unroll
    for (int k = 0; k < 16; ++k) asm volatile("" : "+v"(w0[k]));
#pragma unroll
    for (int k = 0; k < 64; ++k) asm volatile("" : "+v"(w2[k]));
    asm volatile("" : "+v"(wt[0]), "+v"(wt[1]), "+v"(b0), "+v"(b2), "+v"(bt));
  }
  if constexpr (FIRST) {
    for (int i = tid; i < kQ * kC; i += kLpThreads) embs[i] = a.emb[i];
  }
  // Is the stage I hand over to on my XCD?  Then my granules may stay in the XCD's L2 (plain stores; its CUs' sc1 loads find them there)
  // instead of being written through to memory.  Every workgroup registers its XCC id, then reads its successor's.
  __shared__ int s_local;
  if (tid == 0) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    id = (id & 0xf) + 1;
    __hip_atomic_store(a.xcc_ids + clip * kLpStages + stage, id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned* nxt = a.xcc_ids + clip * kLpStages + (stage + 1) % kLpStages;
    unsigned other = 0, spins = 0;
    while ((other = __hip_atomic_load(nxt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0) {
      if (++spins > kLpSpinLimit) { atomicExch(a.err_flag, 3); break; }
    }
    s_local = (other == id && a.xcd_local) ? 1 : 0;
  }
  __syncthreads();
  const bool local = s_local != 0;
  const u64* in_g = a.xg + ((int64_t)stage * a.Bmax + clip) * 128;
  u64* out_g = a.xg + ((int64_t)(stage + 1) * a.Bmax + clip) * 128;
  const int64_t slot_stride = (int64_t)a.Bmax * kC;
  float cnd_next = 0.f;
  if constexpr (COND) {
    if (wave < 2 * NL) cnd_next = a.condall[((int64_t)clip * a.cond_steps) * ((int64_t)a.L * 2 * kC) + (int64_t)(l0 + (wave >> 1)) * 2 * kC + (tid & 127)];
  }
  for (int s = 0; s < (int)a.n_steps; ++s) {
    const int64_t tau = a.t0 - 1 + s;                    // the input position of this step; the class it produces goes to tau + 1
    // ---- the delayed taps of my layers (addresses known) ------------------------------------------------------------------------
    float tap = 0.f;
    if (wave < NL) {
      const int i = wave;                                 // (wave-uniform: the ring's pointer and size are scalar loads from the kernel's arguments)
      const int d = dil[i], rm = a.ring[l0 + i] - 1;
      const float* hp = a.hist[l0 + i];
      const int64_t tp = tau - d;
      // (past this CU's L1: the slot was read a ring ago and rewritten since)
      tap = tp >= 0 ? __hip_atomic_load(hp + (tp & rm) * slot_stride + (int64_t)clip * kC + lane, __ATOMIC_RELAXED,
                                        __HIP_MEMORY_SCOPE_AGENT)
                    : 0.f;
    }
    // conv_1x1(c[tau]) of every gate row of the stage's layers (:141-147): staged in LDS with the taps; the row of the NEXT step is asked for
    // here and rides in one register through this step (read at the step's start it would come from HBM in front of the barrier)
    if constexpr (COND) {
      if (wave < 2 * NL) {
        cnds[wave >> 1][tid & 127] = cnd_next;
        const int64_t sn = s + 1 < (int)a.n_steps ? s + 1 : s;
        cnd_next = a.condall[((int64_t)clip * a.cond_steps + sn) * ((int64_t)a.L * 2 * kC) + (int64_t)(l0 + (wave >> 1)) * 2 * kC + (tid & 127)];
      }
    }
    if (wave < NL) taps[wave][lane] = tap;
    __syncthreads();
    float ptap[NL];                                       // W_tap x(t - d) of my rows, off the step's chain
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      float p4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 16; ++k) p4[k & 3] = fmaf(wc[i][k], taps[i][kq * 16 + k], p4[k & 3]);
      ptap[i] = (p4[0] + p4[1]) + (p4[2] + p4[3]);
      if constexpr (COND) ptap[i] += kq == 0 ? cnds[i][o] : 0.f;      // (the quad's sum takes the term once)
    }
    // ---- this step's input -----------------------------------------------------------------------------------------------------------
    if constexpr (FIRST) {
      if (wave == 0) {
        int64_t cls;
        if (s == 0) cls = a.idx[(int64_t)clip * a.idx_rs + tau];
        else cls = (int64_t)(unsigned)poll(a.cg + (int64_t)clip * 16, (unsigned)s, a.err_flag);
        cls = cls < 0 ? 0 : (cls >= kQ ? kQ - 1 : cls);
        xs[0][lane] = embs[cls * kC + lane];
        sk[lane] = 0.f;
      }
    } else if (wave == 0) {
      // the 128 granules (x | skip) in one 16-byte L1-bypassing load per lane; a granule is one aligned 8-byte store inside it, so
      // it is seen entirely old or entirely new and its tag is checked either way
      typedef unsigned u32x4_lp __attribute__((ext_vector_type(4)));
      const u64* gp = in_g + 2 * lane;
      u32x4_lp v;
      unsigned spins = 0;
      for (;;) {
        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(gp) : "memory");
        if (v[1] == (unsigned)(s + 1) && v[3] == (unsigned)(s + 1)) break;
        if (++spins > kLpSpinLimit || (MMK_WAIT_ERR_LOOK && (spins & 1023u) == 0 && __hip_atomic_load(a.err_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
          atomicExch(a.err_flag, 3);
          break;
        }
      }
      float* dst = lane < 32 ? xs[0] + 2 * lane : sk + 2 * (lane - 32);
      dst[0] = __uint_as_float(v[0]);
      dst[1] = __uint_as_float(v[2]);
    }
    __syncthreads();
    // two barriers per layer: the layer input ping-pongs between two LDS rows, a skip element belongs to one thread for the whole step
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const float* xc = xs[i & 1];
      float* xn = xs[(i + 1) & 1];
      // the layer's input at tau goes to its ring for later taps (and for the launch path, should the batch be redone there)
      if (tid < kC) a.hist[l0 + i][(tau & (a.ring[l0 + i] - 1)) * slot_stride + (int64_t)clip * kC + tid] = xc[tid];

      float acc4[4] = {ptap[i], 0.f, 0.f, 0.f};                              // four chains of 4
#pragma unroll
      for (int k = 0; k < 16; ++k) acc4[k & 3] = fmaf(wc[i][16 + k], xc[kq * 16 + k], acc4[k & 3]);
      const float acc = quad_sum((acc4[0] + acc4[1]) + (acc4[2] + acc4[3])) + bc[i];
      // tanh(f) sigmoid(g) (wavenet_v2.py:151), both halves at once on their own quads with the hardware exp2 / rcp, as in the other
      // step kernels: sigmoid(x) = 1 / (1 + 2^(-x log2 e)), tanh(x) = 2 sigmoid(2 x) - 1; the g quad sits four lanes up (row_shl:4)
      const float act = fmaf(mmk_rcp(1.0f + __builtin_amdgcn_exp2f(acc * gate_scale)), gate_k, gate_shift);
      const float other = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(act), 0x104, 0xf, 0xf, false));
      if ((tid & 7) == 0) zs[tid >> 3] = act * other;
      __syncthreads();
      float a4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 16; ++k) a4[k & 3] = fmaf(wr[i][k], zs[kq * 16 + k], a4[k & 3]);
      const float a2 = quad_sum((a4[0] + a4[1]) + (a4[2] + a4[3]));
      if (kq == 0) {
        if (has_res[i]) {
          if (o < kC) xn[o] = xc[o] + (a2 + br[i]);                         // :165-170
          else sk[o - kC] += a2 + br[i];                                    // :172-176
        } else if (o >= kC) {
          sk[o - kC] += a2 + br[i];
        } else {
          xn[o] = xc[o];                                                    // (the last layer: its residual sum is never used)
        }
      }
      __syncthreads();
    }
    const float* xfin = xs[NL & 1];
    if constexpr (!HEAD) {
      if (tid < 2 * kC) {
        const float v = tid < kC ? xfin[tid] : sk[tid - kC];
        const u64 gv = ((u64)(unsigned)(s + 1) << 32) | __float_as_uint(v);
        if (local) __hip_atomic_store(out_g + tid, gv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // stays in this XCD's L2
        else __hip_atomic_store(out_g + tid, gv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else {
      // ---- head: Linear(64 -> 128), Mish, Linear(128 -> 256 [+ 1]), [temperature], argmax / inverse-CDF draw -----------------
      float h4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 16; ++k) h4[k & 3] = fmaf(w0[k], sk[kq * 16 + k], h4[k & 3]);
      const float h = quad_sum((h4[0] + h4[1]) + (h4[2] + h4[3]));
      if (kq == 0) hid[o] = mish_fast(h + b0);
      __syncthreads();
      float q4[4] = {0.f, 0.f, 0.f, 0.f};
      const float* hs = hid + (tid & 1) * 64;
#pragma unroll
      for (int k = 0; k < 64; ++k) q4[k & 3] = fmaf(w2[k], hs[k], q4[k & 3]);
      float q = (q4[0] + q4[1]) + (q4[2] + q4[3]);
      q += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(q), 0xB1, 0xf, 0xf, false));
      if ((tid & 1) == 0) lg[tid >> 1] = q + b2;
      if (wave == 0 && a.learn_temp) {
        float tv = fmaf(wt[0], hid[lane], wt[1] * hid[64 + lane]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) tv += __shfl_xor(tv, off);
        if (lane == 0) lg[kQ] = tv + bt;
      }
      __syncthreads();
      if (wave == 0) {
        float denom = 1.f;
        if (a.learn_temp && a.temperature != nullptr) denom = fmaxf(sigmoidf_(lg[kQ]), a.min_temp);       // mlp.py:60-62 (the greedy pick divides only when it has to)
        if (a.logits_out && s + 1 == (int)a.n_steps)
          for (int c = lane; c < kQ + (a.learn_temp ? 1 : 0); c += 64) a.logits_out[(int64_t)clip * a.logits_ld + c] = lg[c];
        int result;
        if (a.temperature == nullptr) {
          result = greedy_256(lg, a.learn_temp != 0, lg[kQ], a.min_temp, lane);      // (no division where the raw order decides: sampler256.h)
        } else {
          result = sample_256(lg, a.learn_temp != 0, denom, a.temperature[clip], a.uniforms[(int64_t)clip * a.uni_ld + s], lane);
        }
        if (lane == 0) {
          a.idx[(int64_t)clip * a.idx_rs + tau + 1] = result;
          const u64 gv = ((u64)(unsigned)(s + 1) << 32) | (unsigned)result;
          if (local) __hip_atomic_store(a.cg + (int64_t)clip * 16, gv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          else __hip_atomic_store(a.cg + (int64_t)clip * 16, gv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    __syncthreads();
  }
}

template <bool COND>
__device__ __forceinline__ void stage_of(const WnLpipeArgs& a, int clip, int stage, int l0, int nl, float* embs) {
  if (stage == 0) {
    if (nl == 3) run_stage<3, false, true, COND>(a, clip, stage, l0, embs);
    else if (nl == 2) run_stage<2, false, true, COND>(a, clip, stage, l0, embs);
    else run_stage<1, false, true, COND>(a, clip, stage, l0, embs);
  } else if (stage < kLpStages - 1) {
    if (nl == 3) run_stage<3, false, false, COND>(a, clip, stage, l0, embs);
    else if (nl == 2) run_stage<2, false, false, COND>(a, clip, stage, l0, embs);
    else run_stage<1, false, false, COND>(a, clip, stage, l0, embs);
  } else {
    if (nl == 2) run_stage<2, true, false, COND>(a, clip, stage, l0, embs);
    else run_stage<1, true, false, COND>(a, clip, stage, l0, embs);
  }
}

// (two kernels: the conditioned stages carry a few registers more through the step - 15 spilled against 3 - and the unconditioned
//  BASELINE config 2 keeps the allocation it had)
template <bool COND>
__global__ __launch_bounds__(kLpThreads) void wavenet_lpipe_kernel(const WnLpipeArgs a) {
  __shared__ float embs[kQ * kC];                    // stage 0's copy of the embedding table (64 KiB)
  const int b = blockIdx.x;
  const int clip = (b & 7) + 8 * (b >> 5), stage = (b >> 3) & 3;
  if (clip >= a.B) return;
  // (the stage's layer range by selects on constant indices, in scalar registers)
  int l0 = a.first[0], l1 = a.first[1];
  if (stage == 1) { l0 = a.first[1]; l1 = a.first[2]; }
  if (stage == 2) { l0 = a.first[2]; l1 = a.first[3]; }
  if (stage == 3) { l0 = a.first[3]; l1 = a.first[4]; }
  l0 = __builtin_amdgcn_readfirstlane(l0);
  stage_of<COND>(a, clip, stage, l0, __builtin_amdgcn_readfirstlane(l1 - l0), embs);
}

__global__ __launch_bounds__(256) void wn_lpipe_scatter_kernel(const float* __restrict__ h, int64_t h_batch, int64_t t_begin, int64_t t_lo, int n_pos,
                                                               int C, int B, int Bmax, float* __restrict__ ring, int ring_slots) {
  const int64_t total = (int64_t)n_pos * B * C;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(q % C);
    const int b = (int)((q / C) % B);
    const int64_t t = t_lo + q / ((int64_t)C * B);
    ring[((t & (ring_slots - 1)) * Bmax + b) * C + c] = h[(int64_t)b * h_batch + (t - t_begin) * C + c];
  }
}

}  // namespace

void wn_lpipe_split(int L, int32_t (&first)[kLpStages + 1]) {
  // the head's stage takes at most 2 layers, the others at most 3; later stages get the smaller shares
  int n[kLpStages];
  n[kLpStages - 1] = L >= 8 ? 2 : 1;
  int rest = L - n[kLpStages - 1];
  for (int s = 0; s < kLpStages - 1; ++s) {
    n[s] = (rest + (kLpStages - 2 - s)) / (kLpStages - 1 - s);
    rest -= n[s];
  }
  first[0] = 0;
  for (int s = 0; s < kLpStages; ++s) first[s + 1] = first[s] + n[s];
}

bool wn_lpipe_supported(int C, int S, int H1, int n_classes, int L, int n_cond, int batch) {
  // (a head of fewer hidden units or classes runs as the 128 x 256 one: the plan pads its matrices - zero rows / columns, -inf bias for classes that do not exist)
  return C == kC && S == kC && H1 >= 16 && H1 <= kH1 && H1 % 16 == 0 && n_classes >= 2 && n_classes <= kQ && n_cond <= 2 && L >= kLpStages && L <= kLpMaxLayers &&
         batch >= 1 && batch <= 64;
}

int launch_wavenet_lpipe(const WnLpipeArgs& a, hipStream_t stream) {
  if (a.n_steps <= 0 || a.B <= 0) return MMK_OK;
  if (a.B > 64 || a.L > kLpMaxLayers) return fail(MMK_ERR_UNSUPPORTED, "wavenet layer pipeline: %d clips, %d layers", a.B, a.L);
  for (int s = 0; s < kLpStages; ++s) {
    const int nl = a.first[s + 1] - a.first[s];
    if (nl < 1 || nl > 3 || (s == kLpStages - 1 && nl > 2)) return fail(MMK_ERR_UNSUPPORTED, "wavenet layer pipeline: stage %d would own %d layers", s, nl);
  }
  const int grid = 32 * ((a.B + 7) / 8);
  if (a.condall != nullptr) hipLaunchKernelGGL(wavenet_lpipe_kernel<true>, dim3(grid), dim3(kLpThreads), 0, stream, a);
  else hipLaunchKernelGGL(wavenet_lpipe_kernel<false>, dim3(grid), dim3(kLpThreads), 0, stream, a);
  MMK_HIP(hipGetLastError());
  return MMK_OK;
}

int launch_wn_lpipe_scatter(const float* h, int64_t h_batch, int64_t t_begin, int64_t t_lo, int n_pos, int C, int B, int Bmax, float* ring,
                            int ring_slots, hipStream_t stream) {
  if (n_pos <= 0) return MMK_OK;
  const int64_t total = (int64_t)n_pos * B * C;
  hipLaunchKernelGGL(wn_lpipe_scatter_kernel, dim3((unsigned)((total + 255) / 256 > 1024 ? 1024 : (total + 255) / 256)), dim3(256), 0, stream, h, h_batch,
                     t_begin, t_lo, n_pos, C, B, Bmax, ring, ring_slots);
  MMK_HIP(hipGetLastError());
  return MMK_OK;
}

}  // namespace mmk

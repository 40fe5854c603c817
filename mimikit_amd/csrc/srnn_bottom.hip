// SampleRNN bottom tier, several consecutive time steps in ONE launch (gfx950).
//
// Reference: SampleRNN.generate_step, sample_rnn_v2.py:252-260 - per step
//     x = FramedConv1d(linearize(idx[t-fs:t])) + outputs[-1][:, t % fs[-2]]         (K = fs, tiny)
//     logits = MLP(x): fc0 (H -> Hm) + Mish, fc2 (Hm -> q + temperature column)        (modules/mlp.py)
//     idx[t] = argmax | inverse-CDF sample                                              (modules/targets.py)
// Between two updates of the tier above (every fs[-2] steps) nothing but the clip's own previous samples feeds
// these steps, and the whole MLP is small (fc0 256 KB, fc2 136 KB at H = 512, Hm = 128): one workgroup keeps it
// on chip - fc0 fragments in registers, fc2 in LDS - and walks its 4 clips through the steps without leaving
// the CU.  As separate launches a step was four kernels (framed conv, fc0, fc2, sampler) at >= 5 us each.
//
// Matrix products use v_mfma_f32_4x4x1_16b_f32 (16 blocks = 4 column groups x 4 K sub-slices, rows = the 4
// clips; layout probed with scripts/probes/mfma4x4.hip) on the same packed weights as the launch path
// (Wp[tile][chunk][lane][4], lane = 16 q + n holds W[16 tile + n][16 chunk + 4 q ..]).
#include <stdlib.h>

#include "mmk_common.h"
#include "srnn_bottom.h"
#include "plan_util.h"
#include "sampler256.h"

namespace mmk {

// eight partial sums per lane, 16 lanes (one DPP row) that each hold a different K slice: lanes 2 c, 2 c + 1 of the row end with column c's
// total (own + mirror partner, + half-mirror partner, + the lane two further, + the neighbour: a fixed order) - as in wavenet_spipe.hip's head
__device__ __forceinline__ float bot_reduce_scatter8(const float (&v)[8], int ks) {
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


typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) f32x4* gf32x4_ptr;

constexpr int kBotThreads = 512;
constexpr int kBotClips = 4;

// sum over the 4 K sub-slices (lanes n, n+16, n+32, n+48), fixed order ((0+1)+(2+3))
__device__ __forceinline__ f32x4 reduce_ks(f32x4 v) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[i] += __shfl_xor(v[i], 16);
    v[i] += __shfl_xor(v[i], 32);
  }
  return v;
}

template <int NF>   // NF = H / 16: f32x4 fragments of fc0 per lane (its K sub-slice of H / 4)
__global__ __launch_bounds__(kBotThreads) void srnn_bottom_kernel(const SrnnBottomArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr int H = NF * 16;
  constexpr int ldx = H + 4;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Hm = a.Hm, ldh = Hm + 4;
  const int n_out = a.n_out, n_tiles2 = (n_out + 15) / 16, ldl = n_tiles2 * 16 + 4;
  const int kc2 = Hm / 16;
  const int m_first = blockIdx.x * kBotClips;
  const int mg = min(kBotClips, a.B - m_first);
  const int64_t t0 = *a.tau_ptr + a.tau_off;
  // diagnostic (MMK_SRNN_STAMPS=1): 100 MHz wall-clock totals per phase of thread 0 of workgroup 0
  const bool stamping = a.stamps != nullptr && blockIdx.x == 0 && tid == 0;
  unsigned long long st_prev = stamping ? wall_clock64() : 0, st_acc[5] = {0, 0, 0, 0, 0};
  const unsigned long long clk0 = stamping ? clock64() : 0, wall0 = st_prev;
  auto stamp = [&](int slot) {
    if (stamping) {
      const unsigned long long now = wall_clock64();
      st_acc[slot] += now - st_prev;
      st_prev = now;
    }
  };

  char* sp = smem_raw;
  float* xs = (float*)sp;   sp += kBotClips * ldx * 4;
  float* hid = (float*)sp;  sp += kBotClips * ldh * 4;
  float* lbuf = (float*)sp; sp += kBotClips * ldl * 4;
  int* s_win = (int*)sp;    sp += kBotClips * 16 * 4;      // the last fs (<= 16) classes of each clip
  float* b2s = (float*)sp;  sp += n_tiles2 * 16 * 4;       // fc2 bias, zero padded
  f32x4* w2s = (f32x4*)sp;                                  // fc2, packed order: [tile][chunk][64]

  // ---- once per launch: fc2 -> LDS, this wave's fc0 tile -> registers, class window -> LDS -----------------
  {
    // all loads of a batch in flight before the first LDS store (one round trip per batch, not per element)
    gf32x4_ptr src = (gf32x4_ptr)(uintptr_t)a.fc2_wp;
    const int n4 = n_tiles2 * kc2 * 64;
    constexpr int kBatch = 10;
    for (int base = 0; base < n4; base += kBatch * kBotThreads) {
      f32x4 tmp[kBatch];
#pragma unroll
      for (int k = 0; k < kBatch; ++k) {
        const int i = base + k * kBotThreads + tid;
        tmp[k] = src[i < n4 ? i : n4 - 1];
      }
#pragma unroll
      for (int k = 0; k < kBatch; ++k) {
        const int i = base + k * kBotThreads + tid;
        if (i < n4) w2s[i] = tmp[k];
      }
    }
  }
  const int ks = lane >> 4, n = lane & 15;
  f32x4 w0[NF];
  const bool has_tile = wave * 16 < Hm;      // fc0 column tile of this wave (Hm <= 128: one per wave)
  const float fc0_b = has_tile ? a.fc0_bias[wave * 16 + n] : 0.f;
  {
    // the lane's K sub-slice [ks H/4, (ks+1) H/4): fragment f = 4 consecutive k = element (q, n) of chunk c
    gf32x4_ptr src = (gf32x4_ptr)(uintptr_t)a.fc0_wp + (int64_t)(has_tile ? wave : 0) * (H / 16) * 64;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const int k = ks * (H / 4) + 4 * f;
      w0[f] = src[(k / 16) * 64 + ((k % 16) / 4) * 16 + n];
    }
  }
  if (tid < kBotClips * 16) {
    const int m = tid >> 4, i = tid & 15;
    int v = 0;
    if (m < mg && i < a.fs) v = (int)a.idx[(int64_t)(m_first + m) * a.idx_rs + t0 - a.fs + i];
    s_win[tid] = v;
  }
  for (int i = tid; i < kBotClips * ldh; i += kBotThreads) hid[i] = 0.f;
  for (int i = tid; i < n_tiles2 * 16; i += kBotThreads) b2s[i] = i < n_out ? a.fc2_bias[i] : 0.f;
  // x phase: thread handles elements e = tid + k 512 -> (clip e / H, column e % H).  Its conv bias stays in a
  // register; the tier output above is fetched one step ahead (nothing global sits on the per-step chain)
  constexpr int kEpt = kBotClips * H / kBotThreads;          // 1, 2 or 4
  int xm[kEpt], xc[kEpt];
  float xb[kEpt], up_next[kEpt];
  const int u0 = (int)(t0 % a.up_slots);                      // outputs[-1][:, (t % fs[-2]) - fs[-2]]   (:257)
  auto upper_at = [&](int k, int step) -> float {
    const int mm = xm[k] < mg ? xm[k] : 0;
    return a.upper[((int64_t)(m_first + mm) * a.up_slots + (u0 + step) % a.up_slots) * H + xc[k]];
  };
#pragma unroll
  for (int k = 0; k < kEpt; ++k) {
    const int e = tid + k * kBotThreads;
    xm[k] = e / H;
    xc[k] = e - xm[k] * H;
    xb[k] = a.bb[xc[k]];
    up_next[k] = upper_at(k, 0);
  }
  const float* wb_col[kEpt];
#pragma unroll
  for (int k = 0; k < kEpt; ++k) wb_col[k] = a.wb + xc[k] * a.fs;
  float wb0[kEpt];                                            // fs == 1 (the common case): the weight itself
#pragma unroll
  for (int k = 0; k < kEpt; ++k) wb0[k] = wb_col[k][0];
  __syncthreads();
  stamp(0);   // prologue: weights to registers / LDS

  for (int s = 0; s < a.n_steps; ++s) {
    const int64_t t = t0 + s;
    // ---- x = conv(linearize(window)) + bias + upper tier output -----------------------------------------
#pragma unroll
    for (int k = 0; k < kEpt; ++k) {
      const int m = xm[k];
      float acc = 0.f;
      if (a.fs == 1) {
        acc = fmaf((((float)s_win[m * 16] / a.class_size) - .5f) * 2.f, wb0[k], 0.f);   // Linearizer, modules/io.py:106-112
      } else {
        for (int i = 0; i < a.fs; ++i)
          acc = fmaf((((float)s_win[m * 16 + i] / a.class_size) - .5f) * 2.f, wb_col[k][i], acc);
      }
      xs[m * ldx + xc[k]] = m < mg ? (acc + xb[k]) + up_next[k] : 0.f;
    }
    if (s + 1 < a.n_steps) {
#pragma unroll
      for (int k = 0; k < kEpt; ++k) up_next[k] = upper_at(k, s + 1);
    }
    __syncthreads();
    stamp(1);   // x
    // ---- fc0 + Mish: wave w owns columns 16 w .. 16 w + 15 -----------------------------------------------
    if (has_tile) {
      const float* x = xs + (lane & 3) * ldx + ks * (H / 4);
      // two independent accumulation chains (even / odd fragments): a dependent 4x4 MFMA waits for its predecessor;
      // operands are read from LDS eight fragments at a time (all NF at once would not fit beside the weights)
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int f0 = 0; f0 < NF; f0 += 8) {
        f32x4 xv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) xv[j] = *reinterpret_cast<const f32x4*>(x + 4 * (f0 + j));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            acc = __builtin_amdgcn_mfma_f32_4x4x1f32(xv[j][i], w0[f0 + j][i], acc, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(xv[j + 1][i], w0[f0 + j + 1][i], acc1, 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] += acc1[i];
      acc = reduce_ks(acc);                                   // every lane now holds the totals of its column
      // lane (ks, n) finishes clip ks of column n: one Mish per lane instead of four on a quarter of the lanes
      const float mine = ks == 0 ? acc[0] : (ks == 1 ? acc[1] : (ks == 2 ? acc[2] : acc[3]));
      hid[ks * ldh + wave * 16 + n] = mish_fast(mine + fc0_b);   // MLPIO activation
    }
    __syncthreads();
    stamp(2);   // fc0 + Mish
    // ---- fc2: tiles w, w + 8, ... from LDS ----------------------------------------------------------------
    for (int tile = wave; tile < n_tiles2; tile += kBotThreads / 64) {
      const float* x = hid + (lane & 3) * ldh + ks * (Hm / 4);
      const int nf2 = Hm / 16;               // fragments of the lane's sub-slice of Hm / 4 (<= 8)
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int f0 = 0; f0 < 8; f0 += 4) {     // four fragments' LDS reads in flight, then their MFMAs
        if (f0 < nf2) {
          f32x4 wv[4], xv[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int f = f0 + j < nf2 ? f0 + j : f0;
            const int k = ks * (Hm / 4) + 4 * f;
            wv[j] = w2s[(tile * kc2 + k / 16) * 64 + ((k % 16) / 4) * 16 + n];
            xv[j] = *reinterpret_cast<const f32x4*>(x + 4 * f);
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (f0 + j < nf2) {
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                if (j & 1) acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(xv[j][i], wv[j][i], acc1, 0, 0, 0);
                else acc = __builtin_amdgcn_mfma_f32_4x4x1f32(xv[j][i], wv[j][i], acc, 0, 0, 0);
              }
            }
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] += acc1[i];
      acc = reduce_ks(acc);
      if (ks == 0) {
        const int col = tile * 16 + n;
        const float bias = b2s[col];
#pragma unroll
        for (int m = 0; m < kBotClips; ++m) lbuf[m * ldl + col] = acc[m] + bias;
      }
    }
    __syncthreads();
    stamp(3);   // fc2
    // ---- temperature column + argmax / inverse-CDF sample: one clip per wave ------------------------------
    if (wave < mg) {
      const int m = wave, clip = m_first + m;
      const float* lg = lbuf + m * ldl;
      const int nc = a.Q;
      const int per = (nc + 63) / 64;
      if (a.logits_out && s + 1 == a.n_steps)
        for (int c = lane; c < n_out; c += 64) a.logits_out[(int64_t)clip * a.logits_ld + c] = lg[c];
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
#define MMK_DPP_STEP(CTRL)                                                                                         \
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
      // the new class joins the window and the caller's tensor
      if (lane < a.fs) {
        const int keep = lane + 1 < a.fs ? s_win[m * 16 + lane + 1] : result;
        s_win[m * 16 + lane] = keep;     // wave-synchronous shift: every lane read before any lane writes
      }
      if (lane == 0) a.idx[(int64_t)clip * a.idx_rs + t] = result;
    }
    __syncthreads();
    stamp(4);   // sampler
  }
  if (stamping) {
    for (int i = 0; i < 5; ++i) a.stamps[i] += st_acc[i];   // launches of one stream are serial
    a.stamps[7] += 1;
    a.stamps[5] += clock64() - clk0;
    a.stamps[6] += wall_clock64() - wall0;
  }
}

// ---- one clip per workgroup, vector-ALU products ---------------------------------------------------------
// The kernel above gives a workgroup four clips because v_mfma_f32_4x4x1 wants four rows; its products are then bound by
// what ONE CU multiplies: 4 clips x 128 x 512 MACs of fc0 alone are ~0.9 us at the CU's fp32 rate (128 MAC per cycle - matrix
// and vector pipes peak alike in fp32), and 16 workgroups leave 240 CUs idle.  With B <= a few hundred clips it pays to spend
// a CU per clip instead: 65 k MACs of fc0 as packed fp32 FMAs (each thread a quarter of one hidden unit's dot product, its
// weights in registers), fc2 the same way (half a class per thread), one wave samples.  Same per-step structure, same
// fp32 arithmetic in a different association (pinned by the same goldens / oracle tests).
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NF, bool COMPOSED>   // NF = H / 16; COMPOSED: the pre-multiplied association (a.a_comp / a.b_comp), see the step loop
__global__ __launch_bounds__(kBotThreads) void srnn_bottom1_kernel(const SrnnBottomArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr int H = NF * 16;
  constexpr int kHmMax = 128;
  // Thread layout of both products: lane = 16 cgl + ks.  The 16 lanes of a DPP row split K in 16 slices (partial sums meet
  // through row reductions), the 4 rows of a wave and the 8 waves give 32 column groups.  fc0: slice of H / 16 inputs, 4 hidden
  // units per thread (Hm <= 128); fc2: slice of Hm / 16 hidden units, 8 output columns per thread (the first 256 outputs).
  // Each thread reads only ITS slice of the input vector from LDS (the vector is read 32 times per step, not 512 times).
  constexpr int KS0 = H / 16;                 // fc0 inputs per thread: H/16 (8, 16 or 32 floats)
  constexpr int F0 = KS0 / 4;                 // ... as f32x4 fragments, per column
  constexpr int kPad0 = KS0 + 4;              // LDS stride of a slice: 16 lanes x 16 bytes land in 16 different bank groups
  constexpr int KS2 = kHmMax / 16;            // fc2 inputs per thread (8 floats)
  constexpr int kPad2 = KS2 + 4;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Hm = a.Hm;
  const int n_out = a.n_out;
  const int clip = blockIdx.x;
  const int64_t t0 = *a.tau_ptr + a.tau_off;
  const bool stamping = a.stamps != nullptr && blockIdx.x == 0 && tid == 0;
  unsigned long long st_prev = stamping ? wall_clock64() : 0, st_acc[5] = {0, 0, 0, 0, 0};
  const unsigned long long clk0 = stamping ? clock64() : 0, wall0 = st_prev;
  auto stamp = [&](int slot) {
    if (stamping) {
      const unsigned long long now = wall_clock64();
      st_acc[slot] += now - st_prev;
      st_prev = now;
    }
  };

  char* sp = smem_raw;
  float* xs = (float*)sp;   sp += 16 * kPad0 * 4;           // x, slice-padded: element k lives at (k / KS0) kPad0 + k % KS0
  float* hid = (float*)sp;  sp += 16 * kPad2 * 4;           // hidden units, slice-padded likewise (KS2)
  float* lbuf = (float*)sp; sp += 1024 * 4;                 // logits (n_out <= 1024)
  int* s_win = (int*)sp;    sp += 16 * 4;
  float* wx = (float*)sp;                                   // fc2 rows past the first 256 outputs (the temperature column): (n_out - 256, Hm)

  // ---- once per launch: this thread's weights -> registers ------------------------------------------------------
  const int ks = lane & 15, cg = wave * 4 + (lane >> 4);    // K slice, column group (0 .. 31)
  f32x4 w0[4][F0];                                          // hidden units cg * 4 + j, inputs ks * KS0 ..
  // (from the row-major matrices: a thread's slice of a row is KS0 * 4 contiguous bytes - whole cache lines; the packed
  //  MFMA order would scatter every 16 bytes of it over a different line)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int u = cg * 4 + j;
    const bool has = u < Hm;
    gf32x4_ptr src = (gf32x4_ptr)(uintptr_t)(a.fc0_raw + (int64_t)(has ? u : 0) * H + ks * KS0);
#pragma unroll
    for (int f = 0; f < F0; ++f) {
      const f32x4 v = src[f];
      w0[j][f] = has ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  f32x4 w2[8][2];                                           // output columns cg * 8 + j (< 256), hidden units ks * 8 ..
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
  // the lane that ends up with a column's total after the row reduction keeps its bias: lane ks == j of the row for fc0
  // (j < 4), ks == j for fc2 (j < 8)
  const float fc0_b = (ks < 4 && cg * 4 + ks < Hm) ? a.fc0_bias[cg * 4 + ks] : 0.f;
  // (the row's reduce-scatter below leaves class cg 8 + c in lanes 2 c, 2 c + 1: the even one writes it)
  const float fc2_b = ((ks & 1) == 0 && cg * 8 + (ks >> 1) < n_out) ? a.fc2_bias[cg * 8 + (ks >> 1)] : 0.f;
  // composed mode (srnn_plan.hip: W0 wb_i and W0 bb + b0 pre-multiplied at commit): this lane's unit
  constexpr bool composed = COMPOSED;
  const float a_c0 = (composed && ks < 4 && cg * 4 + ks < Hm) ? a.a_comp[cg * 4 + ks] : 0.f;
  const float b_c = (composed && ks < 4 && cg * 4 + ks < Hm) ? a.b_comp[cg * 4 + ks] : 0.f;
  // (the MLP slices IN their registers before the step loop: a load the compiler still counts as pending at the loop's entry makes
  //  it wait inside every step, and such a wait also covers the step's own stores - see wavenet_spipe.hip)
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int f = 0; f < F0; ++f) asm volatile("" : "+v"(w0[j][f]));
#pragma unroll
  for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(w2[j][0]), "+v"(w2[j][1]));
  const int n_extra = n_out > 256 ? n_out - 256 : 0;
  for (int i = tid; i < n_extra * Hm; i += kBotThreads) wx[i] = a.fc2_raw[(int64_t)256 * Hm + i];
  for (int i = tid; i < 16 * kPad2; i += kBotThreads) hid[i] = 0.f;
  if (tid < 16) s_win[tid] = tid < a.fs ? (int)a.idx[(int64_t)clip * a.idx_rs + t0 - a.fs + tid] : 0;
  // x phase: thread = column (H <= 512)
  const int xc = tid < H ? tid : 0;
  const int xs_at = (xc / KS0) * kPad0 + xc % KS0;
  const float xb = a.bb[xc];
  const float* wb_col = a.wb + xc * a.fs;
  const float wb0 = wb_col[0];
  const int u0 = (int)(t0 % a.up_slots);                      // outputs[-1][:, (t % fs[-2]) - fs[-2]]   (:257)
  auto upper_at = [&](int step) -> float { return a.upper[((int64_t)clip * a.up_slots + (u0 + step) % a.up_slots) * H + xc]; };
  float up_next = upper_at(0);
  // sum over the 16 lanes of a DPP row, every lane ends with the total (fixed order)
  auto row_sum = [](float v) -> float {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, false));   // row_half_mirror
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, false));   // row_mirror
    return v;
  };
  __syncthreads();
  stamp(0);
  int first_cls = s_win[0];

  // W0 . xs for this lane's K slice of its four hidden units, slices summed across the DPP row (no bias, no activation)
  auto fc0_product = [&]() -> float {
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
    return ks == 0 ? tot[0] : (ks == 1 ? tot[1] : (ks == 2 ? tot[2] : tot[3]));   // lane ks < 4 of a row: the total of unit cg * 4 + ks
  };
  auto fc2_phase = [&]() {
    const f32x4* h4 = reinterpret_cast<const f32x4*>(hid + ks * kPad2);
    const f32x4 h0 = h4[0], h1 = h4[1];
    // eight partial sums per lane, added up across the 16 lanes of the row as ONE reduce-scatter (seven DPP steps; eight row sums of
    // four dependent steps each before: the sampler of every step waits for this)
    float part[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      f32x4 acc = h0 * w2[j][0];
      acc += h1 * w2[j][1];
      part[j] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    }
    const float mine = bot_reduce_scatter8(part, ks);            // (class cg 8 + ks / 2, in two lanes)
    const int c = cg * 8 + (ks >> 1);
    if ((ks & 1) == 0 && c < n_out) lbuf[c] = mine + fc2_b;
    for (int r = wave; r < n_extra; r += kBotThreads / 64) {      // rows past 256: one wave each, lanes over k
      float p = 0.f;
      for (int k = lane; k < Hm; k += 64) p = fmaf(hid[(k / KS2) * kPad2 + k % KS2], wx[r * Hm + k], p);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) p += __shfl_xor(p, o);
      if (lane == 0) lbuf[256 + r] = p + a.fc2_bias[256 + r];
    }
  };
  // temperature column + argmax / inverse-CDF sample: wave 0
  // Greedy decode of 256 classes with a frame of one sample (cfg 3's bottom tier): EVERY wave picks the class itself from the logits in LDS
  // (the same deterministic pick eight times) and keeps it in a register - the barrier behind the sampler, which only handed wave 0's class
  // to the other waves, is gone (three barriers per step -> two)
  const bool every_wave_picks = a.temperature == nullptr && a.fs == 1 && a.Q == 256;
  int cur_cls = first_cls;
  auto sampler_phase = [&](int s, int64_t t) {
  if (every_wave_picks) {
    const int result = greedy_256(lbuf, a.learn_temp != 0, lbuf[256], a.min_temp, lane);
    cur_cls = result;
    if (wave == 0) {
      if (a.logits_out && s + 1 == a.n_steps)
        for (int c = lane; c < n_out; c += 64) a.logits_out[(int64_t)clip * a.logits_ld + c] = lbuf[c];
      if (lane == 0) {
        s_win[0] = result;
        a.idx[(int64_t)clip * a.idx_rs + t] = result;
      }
    }
  } else if (wave == 0) {
    const float* lg = lbuf;
    const int nc = a.Q;
    const int per = (nc + 63) / 64;
    if (a.logits_out && s + 1 == a.n_steps)
      for (int c = lane; c < n_out; c += 64) a.logits_out[(int64_t)clip * a.logits_ld + c] = lg[c];
    float denom = 1.f;
    if (a.learn_temp && (a.temperature != nullptr || nc != 256)) denom = fmaxf(sigmoidf_(lg[nc]), a.min_temp);   // mlp.py:60-62 (the greedy pick of 256 classes divides only when it has to)
    int result;
    if (a.temperature == nullptr) {
      float best = -INFINITY;
      int bi = 0x7fffffff;
      if (nc == 256) {
        bi = greedy_256(lg, a.learn_temp != 0, lg[nc], a.min_temp, lane);      // (no division where the raw order decides: sampler256.h)
        best = 0.f;
      } else {
        for (int q = 0; q < per; ++q) {
          const int c = lane * per + q;
          if (c < nc) {
            const float v = a.learn_temp ? lg[c] / denom : lg[c];
            if (v > best || bi == 0x7fffffff) { best = v; bi = c; }
          }
        }
      }
      if (nc != 256) bi = wave_argmax_first(best, bi);        // first maximum wins (torch.argmax)
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
    if (lane < a.fs) {
      const int keep = lane + 1 < a.fs ? s_win[lane + 1] : result;
      s_win[lane] = keep;     // wave-synchronous shift: every lane read before any lane writes
    }
    if (lane == 0) a.idx[(int64_t)clip * a.idx_rs + t] = result;
  }
  };
  const int hid_u = cg * 4 + ks;                              // the hidden unit lane ks < 4 of a row finishes
  const int hid_at = (hid_u / KS2) * kPad2 + hid_u % KS2;
  if constexpr (!composed) {
  for (int s = 0; s < a.n_steps; ++s) {
    const int64_t t = t0 + s;
    // ---- x = conv(linearize(window)) + bias + upper tier output -----------------------------------------
    if (tid < H) {
      float acc = 0.f;
      if (a.fs == 1) {
        acc = fmaf((((float)(every_wave_picks ? cur_cls : s_win[0]) / a.class_size) - .5f) * 2.f, wb0, 0.f);   // Linearizer, modules/io.py:106-112
      } else {
        for (int i = 0; i < a.fs; ++i) acc = fmaf((((float)s_win[i] / a.class_size) - .5f) * 2.f, wb_col[i], acc);
      }
      xs[xs_at] = (acc + xb) + up_next;
    }
    if (s + 1 < a.n_steps) up_next = upper_at(s + 1);
    __syncthreads();
    stamp(1);
    // ---- fc0 + Mish ---------------------------------------------------------------------------------------------
    {
      const float mine = fc0_product();
      if (ks < 4 && hid_u < Hm) hid[hid_at] = mish_fast(mine + fc0_b);   // one Mish per lane, four lanes per row
    }
    __syncthreads();
    stamp(2);
    // ---- fc2 ---------------------------------------------------------------------------------------------------------
    fc2_phase();
    __syncthreads();
    stamp(3);
    sampler_phase(s, t);
    if (!every_wave_picks) __syncthreads();
    stamp(4);
  }
  } else {
    // Composed mode: fc0(x) = W0 up + sum_i lin_i (W0 wb_i) + (W0 bb + b0).  W0 up - the only real product - does not depend on
    // the newest class: for every step but the first of a frame it is multiplied one step ahead, in the same phase as fc2 (two
    // latency-bound products side by side), and the x phase with its barrier is gone.  Three barriers per step instead of four.
    float p_cur = 0.f;                                        // W0 up of the current step (lanes ks < 4)
    bool have_p = false;
    for (int s = 0; s < a.n_steps; ++s) {
      const int64_t t = t0 + s;
      if (!have_p) {                                          // first step of the launch / of a frame: its row, then its product
        if (tid < H) xs[xs_at] = upper_at(s);
        __syncthreads();
        p_cur = fc0_product();
        __syncthreads();                                      // xs is rewritten below
      }
      stamp(1);
      // ---- hidden units of step s; the next step's row -> LDS -------------------------------------------------------
      const bool ahead = s + 1 < a.n_steps;
      if (ks < 4 && hid_u < Hm) {
        float pre = p_cur;
        if (a.fs == 1) {
          pre = fmaf((((float)(every_wave_picks ? cur_cls : s_win[0]) / a.class_size) - .5f) * 2.f, a_c0, pre);
        } else {
          for (int i = 0; i < a.fs; ++i) pre = fmaf((((float)s_win[i] / a.class_size) - .5f) * 2.f, a.a_comp[i * Hm + hid_u], pre);
        }
        hid[hid_at] = mish_fast(pre + b_c);
      }
      if (ahead && tid < H) xs[xs_at] = upper_at(s + 1);
      __syncthreads();
      stamp(2);
      // ---- fc2 of step s next to W0 up of step s + 1 --------------------------------------------------------------
      fc2_phase();
      float p_next = 0.f;
      if (ahead) p_next = fc0_product();
      __syncthreads();
      stamp(3);
      sampler_phase(s, t);
      if (!every_wave_picks) __syncthreads();
      stamp(4);
      have_p = ahead;
      p_cur = p_next;
    }
  }
  if (stamping) {
    for (int i = 0; i < 5; ++i) a.stamps[i] += st_acc[i];
    a.stamps[7] += 1;
    a.stamps[5] += clock64() - clk0;
    a.stamps[6] += wall_clock64() - wall0;
  }
}

static size_t srnn_bottom1_lds_bytes(const SrnnBottomArgs& a) {
  const int n_extra = a.n_out > 256 ? a.n_out - 256 : 0;
  return (size_t)16 * (a.H / 16 + 4) * 4 + (size_t)16 * 12 * 4 + 1024 * 4 + 16 * 4 + 16 + (size_t)n_extra * a.Hm * 4 + 64;
}

// one clip per workgroup pays while the clips fit the chip a few times over, and needs the first 256 outputs to cover the
// classes (threads in pairs) and the whole MLP in a workgroup's registers
static bool srnn_bottom1_applies(const SrnnBottomArgs& a) {
  static const bool off = [] { const char* e = diag_only("MMK_SRNN_BOTTOM_MFMA"); return e && e[0] == '1'; }();
  return !off && a.fc0_raw && a.fc2_raw && a.B <= 1024 && a.n_out <= 1024 && a.Q <= 256 && a.Hm % 16 == 0 && a.Hm <= 128 && srnn_bottom1_lds_bytes(a) <= 64 * 1024;
}

size_t srnn_bottom_lds_bytes(const SrnnBottomArgs& a) {
  const int n_tiles2 = (a.n_out + 15) / 16;
  return (size_t)kBotClips * (a.H + 4) * 4 + (size_t)kBotClips * (a.Hm + 4) * 4 + (size_t)kBotClips * (n_tiles2 * 16 + 4) * 4 +
         kBotClips * 16 * 4 + (size_t)n_tiles2 * 16 * 4 + (size_t)n_tiles2 * (a.Hm / 16) * 1024;
}

bool srnn_bottom_supported(int H, int Hm, int n_out, int fs) {
  if (!(H == 128 || H == 256 || H == 512)) return false;
  if (Hm < 16 || Hm > 128 || Hm % 16) return false;
  if (fs < 1 || fs > 16) return false;
  SrnnBottomArgs a = {};
  a.H = H; a.Hm = Hm; a.n_out = n_out;
  return srnn_bottom_lds_bytes(a) <= 160 * 1024;
}

int launch_srnn_bottom(const SrnnBottomArgs& a, hipStream_t stream) {
  if (!srnn_bottom_supported(a.H, a.Hm, a.n_out, a.fs)) return fail(MMK_ERR_UNSUPPORTED, "srnn bottom kernel: geometry H=%d Hm=%d", a.H, a.Hm);
  if (srnn_bottom1_applies(a)) {
    const size_t lds1 = srnn_bottom1_lds_bytes(a);
    dim3 grid1(a.B), block1(kBotThreads);
    switch (a.H) {
      case 128:
        if (a.a_comp) hipLaunchKernelGGL((srnn_bottom1_kernel<8, true>), grid1, block1, lds1, stream, a);
        else hipLaunchKernelGGL((srnn_bottom1_kernel<8, false>), grid1, block1, lds1, stream, a);
        break;
      case 256:
        if (a.a_comp) hipLaunchKernelGGL((srnn_bottom1_kernel<16, true>), grid1, block1, lds1, stream, a);
        else hipLaunchKernelGGL((srnn_bottom1_kernel<16, false>), grid1, block1, lds1, stream, a);
        break;
      default:
        if (a.a_comp) hipLaunchKernelGGL((srnn_bottom1_kernel<32, true>), grid1, block1, lds1, stream, a);
        else hipLaunchKernelGGL((srnn_bottom1_kernel<32, false>), grid1, block1, lds1, stream, a);
        break;
    }
    MMK_HIP(hipGetLastError());
    return MMK_OK;
  }
  const size_t lds = srnn_bottom_lds_bytes(a);
  dim3 grid((a.B + kBotClips - 1) / kBotClips), block(kBotThreads);
  switch (a.H) {
    case 128: hipLaunchKernelGGL((srnn_bottom_kernel<8>), grid, block, lds, stream, a); break;
    case 256: hipLaunchKernelGGL((srnn_bottom_kernel<16>), grid, block, lds, stream, a); break;
    default: hipLaunchKernelGGL((srnn_bottom_kernel<32>), grid, block, lds, stream, a); break;
  }
  MMK_HIP(hipGetLastError());
  return MMK_OK;
}

}  // namespace mmk

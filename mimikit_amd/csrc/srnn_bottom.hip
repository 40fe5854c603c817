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

namespace mmk {

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

template <int NF>   // NF = H / 16: f32x4 fragments of a thread's quarter of an fc0 row
__global__ __launch_bounds__(kBotThreads) void srnn_bottom1_kernel(const SrnnBottomArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr int H = NF * 16;
  constexpr int kHmMax = 128;                 // hidden units of the MLP: four threads each
  constexpr int kF2 = kHmMax / 8;             // f32x4 fragments of a thread's half of an fc2 row
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
  float* xs = (float*)sp;   sp += H * 4;
  float* hid = (float*)sp;  sp += kHmMax * 4;
  float* lbuf = (float*)sp; sp += 1024 * 4;                 // logits (n_out <= 1024)
  int* s_win = (int*)sp;    sp += 16 * 4;
  float* wx = (float*)sp;                                   // fc2 rows past the first 256 outputs (the temperature column): (n_out - 256, Hm)

  // ---- once per launch: this thread's weights -> registers ------------------------------------------------------
  // fc0: hidden unit u = tid / 4, quarter ks = tid % 4 of its row: k in [ks H/4, (ks+1) H/4)
  const int u = tid >> 2, ks = tid & 3;
  const bool has_u = u < Hm;
  f32x4 w0[NF];
  {
    const int tile = (has_u ? u : 0) >> 4, n = (has_u ? u : 0) & 15;
    gf32x4_ptr src = (gf32x4_ptr)(uintptr_t)a.fc0_wp + (int64_t)tile * (H / 16) * 64;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const int k = ks * (H / 4) + 4 * f;
      w0[f] = src[(k / 16) * 64 + ((k % 16) / 4) * 16 + n];
    }
  }
  const float fc0_b = has_u ? a.fc0_bias[u] : 0.f;
  // fc2: output column c = tid / 2 (< 256), half kh = tid % 2 of its row: k in [kh Hm/2, (kh+1) Hm/2)
  const int c2 = tid >> 1, kh = tid & 1;
  const bool has_c = c2 < n_out;
  const int kc2 = Hm / 16;
  f32x4 w2[kF2];
  {
    const int tile = (has_c ? c2 : 0) >> 4, n = (has_c ? c2 : 0) & 15;
    gf32x4_ptr src = (gf32x4_ptr)(uintptr_t)a.fc2_wp + (int64_t)tile * kc2 * 64;
#pragma unroll
    for (int f = 0; f < kF2; ++f) {
      const int k = kh * (Hm / 2) + 4 * f;
      const bool in = 4 * f < Hm / 2;
      const int kk = in ? k : 0;
      const f32x4 v = src[(kk / 16) * 64 + ((kk % 16) / 4) * 16 + n];
      w2[f] = in ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  const float fc2_b = has_c ? a.fc2_bias[c2] : 0.f;
  // outputs 256 .. n_out-1 (with 256 classes: the temperature column): their rows unpacked into LDS, one wave per row
  const int n_extra = n_out > 256 ? n_out - 256 : 0;
  for (int i = tid; i < n_extra * Hm; i += kBotThreads) {
    const int r = 256 + i / Hm, k = i % Hm;
    const int tile = r >> 4, n = r & 15;
    wx[i] = a.fc2_wp[(((int64_t)tile * kc2 + k / 16) * 64 + ((k % 16) / 4) * 16 + n) * 4 + (k & 3)];
  }
  if (tid < 16) s_win[tid] = tid < a.fs ? (int)a.idx[(int64_t)clip * a.idx_rs + t0 - a.fs + tid] : 0;
  // x phase: thread = column (H <= 512)
  const int xc = tid < H ? tid : 0;
  const float xb = a.bb[xc];
  const float* wb_col = a.wb + xc * a.fs;
  const float wb0 = wb_col[0];
  const int u0 = (int)(t0 % a.up_slots);                      // outputs[-1][:, (t % fs[-2]) - fs[-2]]   (:257)
  auto upper_at = [&](int step) -> float { return a.upper[((int64_t)clip * a.up_slots + (u0 + step) % a.up_slots) * H + xc]; };
  float up_next = upper_at(0);
  __syncthreads();
  stamp(0);

  for (int s = 0; s < a.n_steps; ++s) {
    const int64_t t = t0 + s;
    // ---- x = conv(linearize(window)) + bias + upper tier output -----------------------------------------
    if (tid < H) {
      float acc = 0.f;
      if (a.fs == 1) {
        acc = fmaf((((float)s_win[0] / a.class_size) - .5f) * 2.f, wb0, 0.f);   // Linearizer, modules/io.py:106-112
      } else {
        for (int i = 0; i < a.fs; ++i) acc = fmaf((((float)s_win[i] / a.class_size) - .5f) * 2.f, wb_col[i], acc);
      }
      xs[tid] = (acc + xb) + up_next;
    }
    if (s + 1 < a.n_steps) up_next = upper_at(s + 1);
    __syncthreads();
    stamp(1);
    // ---- fc0 + Mish ---------------------------------------------------------------------------------------------
    {
      const f32x4* x4 = reinterpret_cast<const f32x4*>(xs + ks * (H / 4));
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int f = 0; f < NF; f += 2) {      // the 4 threads of a unit read 4 different quarters, the 16 units of a wave the same ones: broadcasts
        const f32x4 xa = x4[f], xb4 = x4[f + 1];
        acc += xa * w0[f];
        acc1 += xb4 * w0[f + 1];
      }
      acc += acc1;
      float v = (acc[0] + acc[1]) + (acc[2] + acc[3]);
      // the four quarters of a unit sit in four adjacent lanes: quad reductions through DPP
      v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
      v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
      if (ks == 0 && has_u) hid[u] = mish_fast(v + fc0_b);
    }
    __syncthreads();
    stamp(2);
    // ---- fc2 ---------------------------------------------------------------------------------------------------------
    {
      const f32x4* h4 = reinterpret_cast<const f32x4*>(hid + kh * (Hm / 2));
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int f = 0; f < kF2; ++f)
        if (4 * f < Hm / 2) acc += h4[f] * w2[f];
      float v = (acc[0] + acc[1]) + (acc[2] + acc[3]);
      v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));   // the row's other half
      if (kh == 0 && has_c) lbuf[c2] = v + fc2_b;
      for (int r = wave; r < n_extra; r += kBotThreads / 64) {      // rows past 256: one wave each, lanes over k
        float p = 0.f;
        for (int k = lane; k < Hm; k += 64) p = fmaf(hid[k], wx[r * Hm + k], p);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) p += __shfl_xor(p, o);
        if (lane == 0) lbuf[256 + r] = p + a.fc2_bias[256 + r];
      }
    }
    __syncthreads();
    stamp(3);
    // ---- temperature column + argmax / inverse-CDF sample: wave 0 -----------------------------------------------
    if (wave == 0) {
      const float* lg = lbuf;
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
        if (nc == 256) {
          const f32x4 v4 = *reinterpret_cast<const f32x4*>(lg + lane * 4);
          float vv[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) vv[q] = a.learn_temp ? v4[q] / denom : v4[q];
          best = vv[0]; bi = lane * 4;
#pragma unroll
          for (int q = 1; q < 4; ++q)
            if (vv[q] > best) { best = vv[q]; bi = lane * 4 + q; }
        } else {
          for (int q = 0; q < per; ++q) {
            const int c = lane * per + q;
            if (c < nc) {
              const float v = a.learn_temp ? lg[c] / denom : lg[c];
              if (v > best || bi == 0x7fffffff) { best = v; bi = c; }
            }
          }
        }
        auto take = [&](float ob, int oi) {     // first maximum wins (torch.argmax)
          if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        };
#define MMK_DPP_STEP(CTRL)                                                                                         \
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
    __syncthreads();
    stamp(4);
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
  return (size_t)a.H * 4 + 128 * 4 + 1024 * 4 + 16 * 4 + (size_t)n_extra * a.Hm * 4 + 64;
}

// one clip per workgroup pays while the clips fit the chip a few times over, and needs the first 256 outputs to cover the
// classes (threads in pairs) and the whole MLP in a workgroup's registers
static bool srnn_bottom1_applies(const SrnnBottomArgs& a) {
  static const bool off = [] { const char* e = getenv("MMK_SRNN_BOTTOM_MFMA"); return e && e[0] == '1'; }();
  return !off && a.B <= 1024 && a.n_out <= 1024 && a.Q <= 256 && a.Hm % 8 == 0 && a.Hm <= 128 && srnn_bottom1_lds_bytes(a) <= 64 * 1024;
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
      case 128: hipLaunchKernelGGL((srnn_bottom1_kernel<8>), grid1, block1, lds1, stream, a); break;
      case 256: hipLaunchKernelGGL((srnn_bottom1_kernel<16>), grid1, block1, lds1, stream, a); break;
      default: hipLaunchKernelGGL((srnn_bottom1_kernel<32>), grid1, block1, lds1, stream, a); break;
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

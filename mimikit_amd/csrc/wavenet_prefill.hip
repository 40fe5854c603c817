// WaveNet warm-up as a prefill: the prompt goes through the layers as GEMMs over ALL its positions at once
// (gfx950, v_mfma_f32_16x16x4_f32) instead of one position at a time through the persistent step kernel.
//
// Teacher-forced positions do not depend on each other, only layer l+1 on layer l (wavenet_v2.py:140-160):
//     z   = [ h_l[t-d] | h_l[t] | c[t] ] . A_l^T + b          A_l = the packed gate matrix of the step path
//     y   = tanh(z_f) * sigmoid(z_g)
//     h_{l+1}[t] = h_l[t] + W_res y + b_res
// With the staircase of the step path (a layer only runs where its output is still needed when generation
// starts) that is ~2.2 TFLOP for the 3070-sample prompt of cfg 4: 170 ms as 3070 chained steps, ~25 ms as GEMMs.
// Afterwards the tails of every h_l are copied into the private history rings of the step kernel.
//
// One kernel, two epilogues.  A workgroup computes 32 rows (positions of one clip) x 128 columns; the rows of the
// up-to-three K segments are staged once in LDS, wave w owns column tile w and two 16-row accumulators, weight
// fragments are streamed (next chunk's load in flight while the current one is multiplied).
#include "wavenet_prefill.h"

namespace mmk {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) f32x4* gf32x4_ptr;

constexpr int kPfThreads = 512;
constexpr int kPfBM = 64, kPfBN = 128;      // (round 4: 64 rows, staged one K segment at a time - see the kernel)

template <int EPI>   // 0: gate (bias, tanh * sigmoid of adjacent columns) -> y ; 1: residual (bias + res_in) -> out
__global__ __launch_bounds__(kPfThreads) void wn_prefill_kernel(const WnPrefillArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* as = reinterpret_cast<float*>(smem_raw);
  int kmax = 16;
  for (int s = 0; s < a.nseg; ++s) kmax = a.seg_k[s] > kmax ? a.seg_k[s] : kmax;
  const int ldk = kmax + 4;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m_first = blockIdx.y * kPfBM;
  const int tile = blockIdx.x * (kPfBN / 16) + wave;
  const int b = blockIdx.z;
  const bool live = tile < a.n_tiles;
  // Round 4: 64 rows per workgroup, ONE K segment in LDS at a time (66 KB: two workgroups per CU, one staging while the other multiplies).
  // The first form staged all of K for 32 rows (99 KB, one workgroup per CU): a weight fragment fetched from L2 met 32 rows - 12.8 FLOP per
  // byte of L2 traffic, and nothing ran beside the staging; 60 TFLOP/s on the full-size layers of cfg 4.
  gf32x4_ptr w = (gf32x4_ptr)(uintptr_t)a.wp + (int64_t)(live ? tile : 0) * a.k_chunks * 64 + lane;
  const float* x = as + (lane & 15) * ldk + 4 * (lane >> 4);
  f32x4 acc[4] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  int c0 = 0;
  for (int s = 0; s < a.nseg; ++s) {
    const int K = a.seg_k[s];                                  // multiple of 16
    const float* base = a.seg[s] + (int64_t)b * a.seg_batch[s];
    const int k4 = K / 4;
    if (s > 0) __syncthreads();                                // (every wave is through with the segment before)
    for (int q = tid; q < kPfBM * k4; q += kPfThreads) {
      const int m = q / k4, c = (q - m * k4) * 4;
      f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
      if (m_first + m < a.M) v = *reinterpret_cast<const f32x4*>(base + (int64_t)(m_first + m) * a.seg_ld[s] + c);
      *reinterpret_cast<f32x4*>(as + m * ldk + c) = v;
    }
    __syncthreads();
    if (live) {
      const int kc = K / 16;
      f32x4 wv = w[(int64_t)c0 * 64];
      for (int c = 0; c < kc; ++c) {
        const f32x4 wn = w[(int64_t)(c0 + (c + 1 < kc ? c + 1 : c)) * 64];   // next fragment in flight
        f32x4 xr[4];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) xr[rt] = *reinterpret_cast<const f32x4*>(x + rt * 16 * ldk + c * 16);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
          for (int rt = 0; rt < 4; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xr[rt][i], wv[i], acc[rt], 0, 0, 0);
        }
        wv = wn;
      }
    }
    c0 += K / 16;
  }
  if (!live) return;
  // ---- D: column lane & 15, rows 4 (lane >> 4) + r of each 16-row tile ------------------------------------------
  const int n = lane & 15, col = tile * 16 + n;
  const float bias = (a.bias && col < a.N) ? a.bias[col] : 0.f;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m_first + mt * 16 + 4 * (lane >> 4) + r;
      const float v = acc[mt][r] + bias;
      if (EPI == 0) {
        // even packed columns hold f, odd columns g of the same channel (wavenet_v2.py:151): tanh(f) * sigmoid(g)
        const float act = (n & 1) ? sigmoidf_(v) : tanhf(v);
        const float other = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(act), 0x101, 0xf, 0xf, false));   // lane + 1
        if (!(n & 1) && m < a.M && col < a.N)
          a.out[(int64_t)b * a.out_batch + (int64_t)m * a.out_ld + tile * 8 + (n >> 1)] = act * other;
      } else {
        if (m < a.M && col < a.N) {
          const int64_t o = (int64_t)b * a.out_batch + (int64_t)m * a.out_ld + col;
          a.out[o] = a.res_in[(int64_t)b * a.res_batch + (int64_t)m * a.res_ld + col] + v;
        }
      }
    }
  }
}

int launch_wn_prefill(const WnPrefillArgs& a, int epilogue, int batch, hipStream_t stream) {
  if (a.M <= 0 || batch <= 0 || a.n_tiles <= 0) return MMK_OK;
  int kmax = 16;
  for (int s = 0; s < a.nseg; ++s) kmax = a.seg_k[s] > kmax ? a.seg_k[s] : kmax;
  const size_t lds = (size_t)kPfBM * (kmax + 4) * sizeof(float);
  if (lds > 160 * 1024) return fail(MMK_ERR_UNSUPPORTED, "wavenet prefill: K does not fit the LDS stage");
  dim3 grid((a.n_tiles + kPfBN / 16 - 1) / (kPfBN / 16), (a.M + kPfBM - 1) / kPfBM, batch), block(kPfThreads);
  if (epilogue == 0) hipLaunchKernelGGL((wn_prefill_kernel<0>), grid, block, lds, stream, a);
  else hipLaunchKernelGGL((wn_prefill_kernel<1>), grid, block, lds, stream, a);
  MMK_HIP(hipGetLastError());
  return MMK_OK;
}

// h0[b][p][:] = embedding row of idx[b][t_begin + p]   (torch raises on an out-of-range class: NaN row here)
__global__ void wn_prefill_embed_kernel(const int64_t* __restrict__ idx, int64_t idx_rs, int64_t t_begin, const float* __restrict__ emb,
                                        int q_levels, int C, int n_pos, float* __restrict__ out, int64_t out_batch) {
  const int b = blockIdx.y;
  const int c4 = C / 4;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < (int64_t)n_pos * c4; e += (int64_t)gridDim.x * blockDim.x) {
    const int p = (int)(e / c4), c = (int)(e - (int64_t)p * c4) * 4;
    const int64_t cls = idx[(int64_t)b * idx_rs + t_begin + p];
    const float nanv = __builtin_nanf("");
    f32x4 v = f32x4{nanv, nanv, nanv, nanv};
    if (cls >= 0 && cls < q_levels) v = *reinterpret_cast<const f32x4*>(emb + cls * C + c);
    *reinterpret_cast<f32x4*>(out + (int64_t)b * out_batch + (int64_t)p * C + c) = v;
  }
}
int launch_wn_prefill_embed(const int64_t* idx, int64_t idx_rs, int64_t t_begin, const float* emb, int q_levels, int C, int n_pos,
                            float* out, int64_t out_batch, int batch, hipStream_t stream) {
  if (n_pos <= 0 || batch <= 0) return MMK_OK;
  const int64_t total = (int64_t)n_pos * (C / 4);
  dim3 grid((unsigned)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256), batch);
  hipLaunchKernelGGL(wn_prefill_embed_kernel, grid, dim3(256), 0, stream, idx, idx_rs, t_begin, emb, q_levels, C, n_pos, out, out_batch);
  MMK_HIP(hipGetLastError());
  return MMK_OK;
}

// positions [t_lo, t_end) of one layer's input -> slot t & mask of that layer's ring in EVERY workgroup's private block:
// ring[(g Gn + j)][slot][m][:] = h[g Mg + m][t - t_begin][:]
__global__ void wn_prefill_scatter_kernel(const float* __restrict__ h, int64_t h_batch, int64_t t_begin, int64_t t_lo, int n_pos, int C,
                                          int B, int Mg, int Gn, float* __restrict__ rings, int64_t ring_floats_per_wg,
                                          int64_t ring_offset, int ring_mask) {
  const int wg = blockIdx.y;                 // g * Gn + j
  const int g = wg / Gn;
  const int c4 = C / 4;
  const int64_t per_pos = (int64_t)Mg * c4;
  float* dst0 = rings + (int64_t)wg * ring_floats_per_wg + ring_offset;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n_pos * per_pos; e += (int64_t)gridDim.x * blockDim.x) {
    const int p = (int)(e / per_pos);
    const int rem = (int)(e - p * per_pos);
    const int m = rem / c4, c = (rem - m * c4) * 4;
    const int clip = g * Mg + m;
    if (clip >= B) continue;
    const int64_t t = t_lo + p;
    const f32x4 v = *reinterpret_cast<const f32x4*>(h + (int64_t)clip * h_batch + (t - t_begin) * C + c);
    *reinterpret_cast<f32x4*>(dst0 + ((t & ring_mask) * (int64_t)Mg + m) * C + c) = v;
  }
}
int launch_wn_prefill_scatter(const float* h, int64_t h_batch, int64_t t_begin, int64_t t_lo, int n_pos, int C, int B, int Mg, int Gc,
                              int Gn, float* rings, int64_t ring_floats_per_wg, int64_t ring_offset, int ring_mask,
                              hipStream_t stream) {
  if (n_pos <= 0) return MMK_OK;
  const int64_t total = (int64_t)n_pos * Mg * (C / 4);
  dim3 grid((unsigned)((total + 255) / 256 > 64 ? 64 : (total + 255) / 256), Gc * Gn);
  hipLaunchKernelGGL(wn_prefill_scatter_kernel, grid, dim3(256), 0, stream, h, h_batch, t_begin, t_lo, n_pos, C, B, Mg, Gn, rings,
                     ring_floats_per_wg, ring_offset, ring_mask);
  MMK_HIP(hipGetLastError());
  return MMK_OK;
}

}  // namespace mmk

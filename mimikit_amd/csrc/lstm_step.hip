// One LSTM time step (recurrent half) in ONE launch, both directions of a bi-LSTM at once (gfx950).
//
// Reference: torch.nn.LSTM inside the Seq2Seq encoder / decoder (s2s_lstm_v2.py:90-171), per time step
//     gates = (W_ih x_t + b)  [precomputed for all frames by one GEMM]  +  W_hh h_{t-1}
//     i, f, o = sigmoid(.), g = tanh(.);  c' = f c + i g;  h' = o tanh(c')                    (ATen LSTMCell)
// As separate launches a step was a (M x 4H x H) fused-linear launch plus a cell launch per direction (22 + 5 us
// each at H = 1024).  Here a workgroup owns 16 hidden units x 16 rows of one direction: it stages its rows of
// h_{t-1} in LDS, streams the four 16-row tiles of W_hh that belong to its units (K split over 8 waves,
// v_mfma_f32_16x16x4_f32), reduces in LDS and runs the cell for its 256 (row, unit) pairs.  The new state goes to
// a second buffer (other workgroups still read the old one); the host swaps the two.
#include "lstm_step.h"

namespace mmk {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) f32x4* gf32x4_ptr;

constexpr int kLsThreads = 512;
constexpr int kLsWaves = kLsThreads / 64;

template <int CPW>   // K-chunks per wave: H = 128 CPW
__global__ __launch_bounds__(kLsThreads) void lstm_step_kernel(const LstmStepArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr int KC = CPW * kLsWaves;
  constexpr int H = KC * 16;
  constexpr int ldx = H + 4;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ub = blockIdx.x;                       // block of 16 hidden units
  const int m_first = blockIdx.y * 16;
  const int mg = min(16, a.M - m_first);
  const LstmStepDir d = a.dir[blockIdx.z];

  float* hs = reinterpret_cast<float*>(smem_raw);                       // old state rows [16][ldx]
  f32x4* red = reinterpret_cast<f32x4*>(smem_raw + 16 * ldx * 4);       // split-K partials [gate][wave][lane]

  // ---- weights first: 4 gate tiles x CPW chunks of this wave ------------------------------------------------------
  f32x4 w[4][CPW];
  {
    const int c0 = wave * CPW;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      gf32x4_ptr src = (gf32x4_ptr)(uintptr_t)d.whh_wp + ((int64_t)(g * KC + ub) * KC + c0) * 64 + lane;
#pragma unroll
      for (int u = 0; u < CPW; ++u) w[g][u] = src[u * 64];
    }
  }
  // ---- the additive gate terms and the old cell state of this thread's (row, unit) pair ------------------------------
  const int e_m = tid >> 4, e_n = tid & 15;
  const int unit = ub * 16 + e_n;
  const bool cell = tid < 256 && e_m < mg;
  float ga[4] = {0.f, 0.f, 0.f, 0.f}, c_old = 0.f;
  {
    const int mm = m_first + (e_m < mg ? e_m : 0);   // unconditional loads from clamped addresses
    const float* g0 = d.gadd + (int64_t)mm * a.gadd_ld + unit;
#pragma unroll
    for (int g = 0; g < 4; ++g) ga[g] = g0[g * H];
    c_old = d.c[(int64_t)mm * H + unit];
  }
  // ---- old state rows -> LDS ---------------------------------------------------------------------------------------
  for (int q = tid; q < 16 * (H / 4); q += kLsThreads) {
    const int m = q / (H / 4), c = (q - m * (H / 4)) * 4;
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (m < mg) v = *reinterpret_cast<const f32x4*>(d.h_in + (int64_t)(m_first + m) * H + c);
    *reinterpret_cast<f32x4*>(hs + m * ldx + c) = v;
  }
  __syncthreads();
  // ---- four 16 x 16 tiles, this wave's K range -----------------------------------------------------------------------
  {
    const float* hr = hs + (lane & 15) * ldx + wave * CPW * 16 + 4 * (lane >> 4);
    f32x4 acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < CPW; ++u) {
      const f32x4 hv = *reinterpret_cast<const f32x4*>(hr + u * 16);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[i], w[g][u][i], acc[g], 0, 0, 0);
      }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) red[(g * kLsWaves + wave) * 64 + lane] = acc[g];
  }
  __syncthreads();
  // ---- cell: one (row, unit) pair per thread ----------------------------------------------------------------------------
  if (tid < 256) {
    const int frag = ((e_m >> 2) * 16 + e_n) * 4 + (e_m & 3);      // (row m, col n) of a 16x16 accumulator image
    float s[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float* f = reinterpret_cast<const float*>(red + g * kLsWaves * 64) + frag;
      float v = 0.f;
#pragma unroll
      for (int wv = 0; wv < kLsWaves; ++wv) v += f[wv * 256];
      s[g] = v + ga[g];
    }
    if (cell) {
      const float ig = sigmoidf_(s[0]), fg = sigmoidf_(s[1]), cg = tanhf(s[2]), og = sigmoidf_(s[3]);
      const float cn = fg * c_old + ig * cg;
      const float hn = og * tanhf(cn);
      const int64_t row = m_first + e_m;
      d.c[row * H + unit] = cn;
      d.h_out[row * H + unit] = hn;
      if (d.y) d.y[row * a.y_ld + unit] = hn;
    }
  }
}

size_t lstm_step_lds_bytes(int H) { return (size_t)16 * (H + 4) * 4 + (size_t)4 * kLsWaves * 64 * 16; }

bool lstm_step_supported(int H) { return H == 128 || H == 256 || H == 512 || H == 1024; }

int launch_lstm_step(const LstmStepArgs& a, hipStream_t stream) {
  if (!lstm_step_supported(a.H)) return fail(MMK_ERR_UNSUPPORTED, "lstm step kernel: H=%d", a.H);
  const size_t lds = lstm_step_lds_bytes(a.H);
  dim3 grid(a.H / 16, (a.M + 15) / 16, a.n_dir), block(kLsThreads);
  switch (a.H) {
    case 128: hipLaunchKernelGGL((lstm_step_kernel<1>), grid, block, lds, stream, a); break;
    case 256: hipLaunchKernelGGL((lstm_step_kernel<2>), grid, block, lds, stream, a); break;
    case 512: hipLaunchKernelGGL((lstm_step_kernel<4>), grid, block, lds, stream, a); break;
    default: hipLaunchKernelGGL((lstm_step_kernel<8>), grid, block, lds, stream, a); break;
  }
  MMK_HIP(hipGetLastError());
  return MMK_OK;
}

}  // namespace mmk

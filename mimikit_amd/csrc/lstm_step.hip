// One LSTM time step (recurrent half) in ONE launch, both directions of a bi-LSTM at once (gfx950).
//
// Reference: torch.nn.LSTM inside the Seq2Seq encoder / decoder (s2s_lstm_v2.py:90-171), per time step
//     gates = (W_ih x_t + b)  [precomputed for all frames by one GEMM]  +  W_hh h_{t-1}
//     i, f, o = sigmoid(.), g = tanh(.);  c' = f c + i g;  h' = o tanh(c')                    (ATen LSTMCell)
// As separate launches a step was a (M x 4H x H) fused-linear launch plus a cell launch per direction (22 + 5 us
// each at H = 1024).  Here a workgroup owns 16 hidden units x 32 rows (16 for a batch of <= 16) of one direction:
// every wave takes 1/8 of K, reads its slice of h_{t-1} straight from global memory in MFMA operand order (the state is
// 256 KB, L2 resident) and the four 16-row tiles of W_hh that belong to the units (v_mfma_f32_16x16x4_f32), the
// partial sums meet in LDS and each thread runs the cell for one (row, unit) pair.  64 rows x H = 1024 gives 256
// workgroups, one per CU, 2 waves per SIMD; the first version (16 rows per workgroup, h staged in 64 KB of LDS) needed
// two rounds of 512 single-resident workgroups and re-read the weights four times: 22 us against 6.8 us of MFMA work.
// The new state goes to a second buffer (other workgroups still read the old one); the host swaps the two.
//
// What the weight stream costs (scripts/probes/l2_persist.hip, loads only, per launch of a back-to-back train): an XCD's L2 keeps
// its lines from launch to launch - a 32 MB set read by the same workgroups comes back at > 11 TB/s (2.9 us with 2.5 us of launch),
// a 40 MB one at 6 TB/s.  The two W_hh of H = 1024 are 33.5 MB = 4.19 MB per XCD beside the state and the gate terms: this kernel's
// pattern (pairs of workgroups on 256 KiB of weights, 128 KiB of state each) takes 9.4 us there, ~100 MB through the L1s per step.
// Measured and dropped in round 2: (1) the `nt` bit on the weight loads of a wave's last 1 - 4 chunks, so that the rest stays
// resident: 456 -> 460 / 465 / 468 / 475 us per cfg-5 generate step (an nt line read by a pair of workgroups is fetched twice);
// (2) a workgroup per 8 units and all 64 rows, W_hh packed as (i, f) / (g, o) tiles - every weight line has one reader, 7.1 us
// in the probe: 483 us per generate step against 458 (549 without the bubbles): four state fragments per chunk instead of two put
// 64 MB of reads on the same 512 KB of L2 lines.
#include <type_traits>

#include "lstm_step.h"

namespace mmk {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) f32x4* gf32x4_ptr;

constexpr int kLsThreads = 512;
constexpr int kLsWaves = kLsThreads / 64;

// One chunk of the K pipeline: the wait is tied to the chunk's registers ("+v"), so nothing that uses them can be scheduled
// ahead of it; N = loads that may still be in flight (those of the later chunks).
template <int N>
__device__ __forceinline__ void wait_chunk(f32x4& w0, f32x4& w1, f32x4& w2, f32x4& w3, f32x4& h0, f32x4& h1) {
  asm volatile("s_waitcnt vmcnt(%6)" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(h0), "+v"(h1) : "n"(N));
}

template <int CPW, int RB, bool ZERO>   // K-chunks per wave: H = 128 CPW; 16-row blocks per workgroup; ZERO: h = c = 0 (no product)
__global__ __launch_bounds__(kLsThreads) void lstm_step_kernel(const LstmStepArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr int KC = CPW * kLsWaves;
  constexpr int H = KC * 16;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ub = blockIdx.x;                       // block of 16 hidden units
  const int m_first = blockIdx.y * (16 * RB);
  const int mg = min(16 * RB, a.M - m_first);
  const LstmStepDir d = a.dir[blockIdx.z];
  f32x4* red = reinterpret_cast<f32x4*>(smem_raw);                      // split-K partials [row block][gate][wave][lane]

  constexpr bool zero = ZERO;                    // fresh state: the gates are the precomputed input terms alone
  // ---- the additive gate terms and the old cell state of this thread's (row, unit) pair (requested first: they are the
  //      oldest entries of the memory pipe and never sit behind the weight stream) -----------------------------------------
  const int e_m = tid >> 4, e_n = tid & 15;                             // rows 0..31
  const int unit = ub * 16 + e_n;
  const bool cell = e_m < mg;
  float ga[4] = {0.f, 0.f, 0.f, 0.f}, c_old = 0.f;
  if (e_m < 16 * RB) {
    const int mm = m_first + (e_m < mg ? e_m : 0);   // unconditional loads from clamped addresses
    const float* g0 = d.gadd + (int64_t)mm * a.gadd_ld + unit;
#pragma unroll
    for (int g = 0; g < 4; ++g) ga[g] = g0[g * H];
    c_old = zero ? 0.f : d.c[(int64_t)mm * H + unit];
  }
  if constexpr (!zero) {
    // ---- this wave's K range: 4 gate tiles of W_hh and RB row blocks of the old state, chunk by chunk.  The loads are asm
    //      statements so that they are issued in chunk order (left to itself the compiler sorts them by address register and
    //      ends up waiting for nearly all of them before the first MFMA: 11.4 us of stream + 8.2 us of MFMAs back to back,
    //      measured); every chunk's MFMAs wait for their own six loads only --------------------------------------------------
    f32x4 w[CPW][4], hv[CPW][2];
    const int c0 = wave * CPW;
    const f32x4* wsrc = reinterpret_cast<const f32x4*>(d.whh_wp) + ((int64_t)ub * KC + c0) * 64 + lane;
    const int64_t gate_stride = (int64_t)KC * KC * 64;                   // f32x4 elements between the gates' tile rows
    const float* hsrc[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      const int m = (rb < RB ? rb : 0) * 16 + (lane & 15);
      hsrc[rb] = d.h_in + (int64_t)(m_first + (m < mg ? m : 0)) * H + c0 * 16 + 4 * (lane >> 4);   // clamped, unconditional
    }
#pragma unroll
    for (int u = 0; u < CPW; ++u) {
#pragma unroll
      for (int g = 0; g < 4; ++g) asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(w[u][g]) : "v"(wsrc + g * gate_stride + u * 64) : "memory");
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(hv[u][rb]) : "v"(hsrc[rb] + u * 16) : "memory");
    }
    f32x4 acc[RB][4];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[rb][g] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto chunk = [&](auto uc) {
      constexpr int u = decltype(uc)::value;
      if constexpr (u < CPW) {
        wait_chunk<6 * (CPW - 1 - u)>(w[u][0], w[u][1], w[u][2], w[u][3], hv[u][0], hv[u][1]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
          for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              acc[rb][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[u][rb][i], w[u][g][i], acc[rb][g], 0, 0, 0);
              if (g & 1) {
                // a bubble of 8 cycles behind every second MFMA: returning loads are written to the registers through a path
                // that back-to-back MFMAs keep busy - without bubbles the later chunks arrive at half the rate (22.7 us per
                // step instead of 17.6; the stream alone takes 11.2 us, the MFMAs alone 8.2 us, measured)
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_nop 7");
                __builtin_amdgcn_sched_barrier(0);
              }
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);     // the next chunk's wait stays behind these MFMAs
      }
    };
    chunk(std::integral_constant<int, 0>{}); chunk(std::integral_constant<int, 1>{}); chunk(std::integral_constant<int, 2>{});
    chunk(std::integral_constant<int, 3>{}); chunk(std::integral_constant<int, 4>{}); chunk(std::integral_constant<int, 5>{});
    chunk(std::integral_constant<int, 6>{}); chunk(std::integral_constant<int, 7>{});
    static_assert(CPW <= 8, "H <= 1024");
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int g = 0; g < 4; ++g) red[((rb * 4 + g) * kLsWaves + wave) * 64 + lane] = acc[rb][g];
  }
  __syncthreads();
  // ---- cell: one (row, unit) pair per thread ----------------------------------------------------------------------------
  if (e_m < 16 * RB) {
    const int rb = e_m >> 4, r = e_m & 15;
    const int frag = ((r >> 2) * 16 + e_n) * 4 + (r & 3);          // (row r, col n) of a 16x16 accumulator image
    float s[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float* f = reinterpret_cast<const float*>(red + (rb * 4 + g) * kLsWaves * 64) + frag;
      float v = 0.f;
      if constexpr (!zero) {
#pragma unroll
        for (int wv = 0; wv < kLsWaves; ++wv) v += f[wv * 256];
      }
      s[g] = v + ga[g];
    }
    if (cell) {
      const float ig = sigmoid_fast(s[0]), fg = sigmoid_fast(s[1]), cg = tanh_fast(s[2]), og = sigmoid_fast(s[3]);
      const float cn = fg * c_old + ig * cg;
      const float hn = og * tanh_fast(cn);
      const int64_t row = m_first + e_m;
      d.c[row * H + unit] = cn;
      d.h_out[row * H + unit] = hn;
      if (d.y) d.y[row * a.y_ld + unit] = hn;
    }
  }
}

size_t lstm_step_lds_bytes(int rb) { return (size_t)rb * 4 * kLsWaves * 64 * 16; }

bool lstm_step_supported(int H) { return H == 128 || H == 256 || H == 512 || H == 1024; }

int launch_lstm_step(const LstmStepArgs& a, hipStream_t stream) {
  if (!lstm_step_supported(a.H)) return fail(MMK_ERR_UNSUPPORTED, "lstm step kernel: H=%d", a.H);
  const int rb = a.M > 16 ? 2 : 1;
  const size_t lds = lstm_step_lds_bytes(rb);
  dim3 grid(a.H / 16, (a.M + 16 * rb - 1) / (16 * rb), a.n_dir), block(kLsThreads);
#define MMK_LS(CPW_)                                                                                     \
  if (a.zero_state) {                                                                                    \
    if (rb == 2) hipLaunchKernelGGL((lstm_step_kernel<CPW_, 2, true>), grid, block, lds, stream, a);      \
    else hipLaunchKernelGGL((lstm_step_kernel<CPW_, 1, true>), grid, block, lds, stream, a);              \
  }   else if (rb == 2) hipLaunchKernelGGL((lstm_step_kernel<CPW_, 2, false>), grid, block, lds, stream, a); \
  else hipLaunchKernelGGL((lstm_step_kernel<CPW_, 1, false>), grid, block, lds, stream, a)
  switch (a.H) {
    case 128: MMK_LS(1); break;
    case 256: MMK_LS(2); break;
    case 512: MMK_LS(4); break;
    default: MMK_LS(8); break;
  }
#undef MMK_LS
  MMK_HIP(hipGetLastError());
  return MMK_OK;
}

}  // namespace mmk

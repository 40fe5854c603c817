// C[M <= 64, N] = act(A[M, K] . W^T + bias) on a packed weight matrix, for the few-rows products whose weights are large
// (the Seq2Seq decoder's up-sampler dec.fc: 64 x 8192 x 1024, 32 MB of weights per call).
//
// Same shape of work as an LSTM time step (lstm_step.hip): a workgroup owns two 16-column tiles x all rows (up to four blocks of
// 16), every wave takes 1 / 8 of K, requests its weight fragments and its slice of A (straight from global memory in MFMA
// operand order) in one burst, multiplies, and the partial sums meet in LDS.  N / 32 workgroups: 256 for dec.fc, one per CU;
// the row-tile kernel of linear.hip ran that product at 1.2 TB/s of weights (27.7 us), this one streams them like the LSTM step
// does.
#include "mmk_common.h"

namespace mmk {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) f32x4* gf32x4_ptr;

constexpr int kSkThreads = 512;
constexpr int kSkWaves = kSkThreads / 64;

template <int CPW, int RB>   // K = 128 CPW; 16-row blocks
__global__ __launch_bounds__(kSkThreads) void skinny_linear_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ Wp,
                                                                   const float* __restrict__ bias, float* __restrict__ C, int64_t ldc, int M,
                                                                   int n_tiles, int N, int act) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr int KC = CPW * kSkWaves;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile0 = blockIdx.x * 2;
  f32x4* red = reinterpret_cast<f32x4*>(smem_raw);                      // [row block][tile][wave][lane]
  const int c0 = wave * CPW;
  f32x4 w[CPW][2], xv[CPW][RB];
  {
    const int t1 = tile0 + 1 < n_tiles ? tile0 + 1 : tile0;
    gf32x4_ptr w0 = (gf32x4_ptr)(uintptr_t)Wp + ((int64_t)tile0 * KC + c0) * 64 + lane;
    gf32x4_ptr w1 = (gf32x4_ptr)(uintptr_t)Wp + ((int64_t)t1 * KC + c0) * 64 + lane;
    const float* xs[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const int m = rb * 16 + (lane & 15);
      xs[rb] = A + (int64_t)(m < M ? m : M - 1) * lda + c0 * 16 + 4 * (lane >> 4);   // clamped, unconditional
    }
#pragma unroll
    for (int u = 0; u < CPW; ++u) {
      w[u][0] = w0[u * 64];
      w[u][1] = w1[u * 64];
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) xv[u][rb] = *reinterpret_cast<const f32x4*>(xs[rb] + u * 16);
    }
  }
  __builtin_amdgcn_sched_barrier(0);          // every request is out before the first MFMA (the compiler would otherwise trickle them)
  f32x4 acc[RB][2];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) acc[rb][0] = acc[rb][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < CPW; ++u) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        acc[rb][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[u][rb][i], w[u][0][i], acc[rb][0], 0, 0, 0);
        acc[rb][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[u][rb][i], w[u][1][i], acc[rb][1], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int t = 0; t < 2; ++t) red[((rb * 2 + t) * kSkWaves + wave) * 64 + lane] = acc[rb][t];
  __syncthreads();
  for (int e = tid; e < RB * 2 * 256; e += kSkThreads) {                // (row block, tile, row, column)
    const int rt = e >> 8, r = (e >> 4) & 15, n = e & 15;
    const int rb = rt >> 1, t = rt & 1;
    const int frag = ((r >> 2) * 16 + n) * 4 + (r & 3);                 // (row r, col n) of a 16x16 accumulator image
    const float* f = reinterpret_cast<const float*>(red + rt * kSkWaves * 64) + frag;
    float v = 0.f;
#pragma unroll
    for (int wv = 0; wv < kSkWaves; ++wv) v += f[wv * 256];
    const int m = rb * 16 + r, col = (tile0 + t) * 16 + n;
    if (m < M && tile0 + t < n_tiles && col < N) C[(int64_t)m * ldc + col] = apply_act(v + (bias ? bias[col] : 0.f), act);
  }
}

bool skinny_linear_supported(const float* A, int64_t lda, int M, int K, int k_chunks) {
  return M >= 1 && M <= 64 && K % 128 == 0 && K >= 128 && K <= 1024 && k_chunks * 16 == K && (lda % 4) == 0 &&
         (reinterpret_cast<uintptr_t>(A) & 15) == 0;
}

int launch_skinny_linear(const float* A, int64_t lda, const float* Wp, const float* bias, int n_tiles, int k_chunks, int N, int K, float* C,
                         int64_t ldc, int M, int act, hipStream_t stream) {
  if (!skinny_linear_supported(A, lda, M, K, k_chunks)) return fail(MMK_ERR_UNSUPPORTED, "skinny linear: M=%d K=%d", M, K);
  const int rb = (M + 15) / 16;
  dim3 grid((n_tiles + 1) / 2), block(kSkThreads);
  const size_t lds = (size_t)rb * 2 * kSkWaves * 64 * 16;
#define MMK_SK2(CPW_, RB_) hipLaunchKernelGGL((skinny_linear_kernel<CPW_, RB_>), grid, block, lds, stream, A, lda, Wp, bias, C, ldc, M, n_tiles, N, act)
#define MMK_SK(CPW_)                 \
  switch (rb) {                      \
    case 1: MMK_SK2(CPW_, 1); break; \
    case 2: MMK_SK2(CPW_, 2); break; \
    case 3: MMK_SK2(CPW_, 3); break; \
    default: MMK_SK2(CPW_, 4); break; \
  }
  switch (K / 128) {
    case 1: MMK_SK(1); break;
    case 2: MMK_SK(2); break;
    case 3: MMK_SK(3); break;
    case 4: MMK_SK(4); break;
    case 5: MMK_SK(5); break;
    case 6: MMK_SK(6); break;
    case 7: MMK_SK(7); break;
    default: MMK_SK(8); break;
  }
#undef MMK_SK
#undef MMK_SK2
  MMK_HIP(hipGetLastError());
  return MMK_OK;
}

}  // namespace mmk

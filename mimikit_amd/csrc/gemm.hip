// Plain fp32 GEMM on packed weights:  C[M, N] = A[M, K] . W^T   (gfx950, v_mfma_f32_16x16x4_f32).
//
// Used for the one GEMM-shaped op of the WaveNet path: the conditioning products of all layers for a block of
// positions (M = positions of a clip, N = L x 2C, K = cond channels; 128 GFLOP per 1024-position block on cfg 4),
// which the generic fused-linear kernel (built for M <= 64) ran at 22 TFLOP/s.
//   * W is the packed matrix of linear.hip: Wp[n_tile][k_chunk][lane][4], lane = 16 q + n holds W[16 tile + n][16 chunk + 4 q ..]
//   * a workgroup computes 64 rows x 128 columns: the 64 x K block of A is staged once in LDS, wave w owns column
//     tile w and four 16-row accumulators; per K-chunk a wave issues one 1-KiB fragment load (next chunk's, while the
//     current one is multiplied), four LDS reads and 16 MFMAs - the matrix pipe is the bound, not memory.
#include "mmk_common.h"

namespace mmk {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) f32x4* gf32x4_ptr;

constexpr int kGemmThreads = 512;
constexpr int kGemmBM = 64, kGemmBN = 128;

__global__ __launch_bounds__(kGemmThreads) void gemm_f32_kernel(const float* __restrict__ A, int64_t lda, int64_t a_batch,
                                                                const float* __restrict__ Wp, float* __restrict__ C,
                                                                int64_t ldc, int64_t c_batch, int M, int n_tiles, int N,
                                                                int K, int k_chunks) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* as = reinterpret_cast<float*>(smem_raw);
  const int ldk = k_chunks * 16 + 4;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m_first = blockIdx.y * kGemmBM;
  const int tile = blockIdx.x * (kGemmBN / 16) + wave;          // this wave's column tile
  A += (int64_t)blockIdx.z * a_batch;
  C += (int64_t)blockIdx.z * c_batch;
  // ---- A block -> LDS (zero padded rows / columns) ---------------------------------------------------------
  const int k4 = k_chunks * 4;                                  // float4 pieces per row
  for (int q = tid; q < kGemmBM * k4; q += kGemmThreads) {
    const int m = q / k4, c = (q - m * k4) * 4;
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (m_first + m < M) {
      const float* src = A + (int64_t)(m_first + m) * lda + c;
      if (c + 3 < K) v = *reinterpret_cast<const f32x4*>(src);
      else
        for (int i = 0; i < 4; ++i) if (c + i < K) v[i] = src[i];
    }
    *reinterpret_cast<f32x4*>(as + m * ldk + c) = v;
  }
  __syncthreads();
  const int wtile = tile < n_tiles ? tile : n_tiles - 1;       // surplus waves recompute the last tile (never stored)
  gf32x4_ptr w = (gf32x4_ptr)(uintptr_t)Wp + (int64_t)wtile * k_chunks * 64 + lane;
  const float* x = as + (lane & 15) * ldk + 4 * (lane >> 4);
  f32x4 acc[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 wv = w[0];
  for (int c = 0; c < k_chunks; ++c) {
    const f32x4 wn = w[(int64_t)(c + 1 < k_chunks ? c + 1 : c) * 64];   // next fragment in flight
    f32x4 xv[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) xv[mt] = *reinterpret_cast<const f32x4*>(x + mt * 16 * ldk + c * 16);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[mt][i], wv[i], acc[mt], 0, 0, 0);
    }
    wv = wn;
  }
  // ---- D (column lane & 15, rows 4 (lane >> 4) + r) -> LDS -> rows of 512 contiguous bytes ------------------------
  __syncthreads();                                              // every wave is done with the A block
  constexpr int ldo = kGemmBN + 4;
  float* os = as;                                               // 64 x 132 floats <= the A stage (K >= 128) or its own 33 KiB
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) os[(mt * 16 + 4 * (lane >> 4) + r) * ldo + wave * 16 + (lane & 15)] = acc[mt][r];
  __syncthreads();
  const int col0 = blockIdx.x * kGemmBN;
  for (int q = tid; q < kGemmBM * (kGemmBN / 4); q += kGemmThreads) {
    const int m = q / (kGemmBN / 4), c = (q - m * (kGemmBN / 4)) * 4;
    if (m_first + m >= M) continue;
    float* dst = C + (int64_t)(m_first + m) * ldc + col0 + c;
    const f32x4 v = *reinterpret_cast<const f32x4*>(os + m * ldo + c);
    if (col0 + c + 3 < N && (ldc & 3) == 0) *reinterpret_cast<f32x4*>(dst) = v;
    else
      for (int i = 0; i < 4; ++i) if (col0 + c + i < N) dst[i] = v[i];
  }
}

// C[b][M, N] = A[b][M, K] . W^T for b < batch (A, C advance by a_batch / c_batch floats per b)
int launch_gemm_f32(const float* A, int64_t lda, int64_t a_batch, const float* Wp, int n_tiles, int k_chunks, int N, int K,
                    float* C, int64_t ldc, int64_t c_batch, int M, int batch, hipStream_t stream) {
  if (M <= 0 || batch <= 0 || n_tiles <= 0) return MMK_OK;
  size_t lds = (size_t)kGemmBM * (k_chunks * 16 + 4) * sizeof(float);
  const size_t lds_out = (size_t)kGemmBM * (kGemmBN + 4) * sizeof(float);
  if (lds < lds_out) lds = lds_out;
  if (lds > 160 * 1024) return fail(MMK_ERR_UNSUPPORTED, "gemm: K=%d does not fit the LDS stage", K);
  if ((lda % 4) != 0 || (reinterpret_cast<uintptr_t>(A) & 15) != 0) return fail(MMK_ERR_UNSUPPORTED, "gemm: A must be 16-byte aligned with lda %% 4 == 0");
  dim3 grid((n_tiles + kGemmBN / 16 - 1) / (kGemmBN / 16), (M + kGemmBM - 1) / kGemmBM, batch), block(kGemmThreads);
  hipLaunchKernelGGL(gemm_f32_kernel, grid, block, lds, stream, A, lda, a_batch, Wp, C, ldc, c_batch, M, n_tiles, N, K, k_chunks);
  MMK_HIP(hipGetLastError());
  return MMK_OK;
}

}  // namespace mmk

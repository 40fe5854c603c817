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
#include "plan_util.h"

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

// ---- K-pipelined variant with an epilogue:  C[M, N] = act(A[M, K] . W^T + bias) ----------------------------------------------
// For the GEMM-shaped ops of the Seq2Seq step (W_ih x over all hop frames of all clips: 512 x 4096 x 1024, and the
// output projection), which the row-tile kernel of linear.hip (built for M <= 64) ran at 42 TFLOP/s.
//   * 64 rows x 64 columns per workgroup, 4 waves as 2 x 2, a wave owns 32 x 32 = 2 x 2 MFMA tiles; 512 x 4096 gives
//     512 workgroups: two per CU, which run out of phase and fill each other's barrier / LDS waits (one 128 x 64
//     workgroup of 8 waves per CU keeps all its waves in lock step: 57 us against 52 us)
//   * A goes through LDS in 64-k slabs, double buffered; the next slab travels global -> registers while the current
//     one is multiplied (2 x 64 MFMAs per SIMD and stage = 1.7 us, longer than the loads take)
//   * W fragments (1 KiB, packed order) come straight from global / L2 into registers, one stage ahead
//   * per K-chunk a wave issues 2 LDS reads and 16 MFMAs: the matrix pipe is the bound
//   * measured and dropped: the operands of TWO stages in flight (unconditional, masked loads so that the compiler's waits are
//     counted ones, 182 VGPRs, no spills): 459.6 -> 458.8 us per cfg-5 generate step, i.e. nothing - a stage does not end in a
//     wait for its loads when the grid fills the chip
//   * the W fragments go through LDS as well (WLDS): each of the workgroup's four column tiles is fetched by one wave instead of two,
//     a third less through the L1s - 462 -> 452 us per cfg-5 generate step (the 512 x 4096 x 516 products 34 -> 30 us; the
//     512 x 4096 x 1024 ones stay at 50 us: they are neither L2-bound nor, at 146 - 155 TFLOP/s of sustained MFMA rate on this chip
//     (scripts/probes/mfma_clock.hip), clock-bound; starting the CUs' second residents a part of a stage late changed nothing either)
//   * where the 50 us of a 512 x 4096 x 1024 product are (timing builds, rocprofv3 durations): without the global loads after the first
//     stage 45 us, without loads and LDS stores 41 us, without the MFMAs 25 us; the fp32 MFMA rate of this chip is 133 - 140 TFLOP/s
//     with one wave per SIMD, 146 - 153 with two, 154 with four (the clock follows occupancy and kernel length: 2.12 - 2.39 GHz), i.e.
//     ~35 us for this structure at its best.  No effect: starting one of a CU's two workgroups late (by grid half, by hardware wave
//     slot parity), the LDS reads of chunk c + 1 in front of the MFMAs of chunk c.  One workgroup of 4 waves per CU on a
//     64 x 128 tile (no SIMD shared between workgroups with their own barriers, 32 MFMAs per 6 LDS reads, masked unconditional
//     loads, 148 VGPRs) was built and measured too: 63 us - with one wave per SIMD every wait is the SIMD's.  Stages of 32 k (twice the barriers) take
//     the same 50 us, so do 1024 workgroups (split K, four per CU): neither the barriers nor the number of waves that can hide a
//     wait is the limit.  s_memtime / s_memrealtime around the K loop of a workgroup: 43.4 us at 2.05 - 2.19 GHz = 5700 clocks per
//     stage against the 4096 its SIMD's 128 MFMAs take (72 % busy inside the loop); 6.6 us of a launch are outside the loop.
//   * a launch with few tiles (the output projection: 72 workgroups, 16 dependent stages of ~2 us each when a workgroup has a CU to
//     itself) splits K over blockIdx.z and a second launch adds the partial sums in split order: 33 -> ~21 us for both launches
constexpr int kTgThreads = 256;             // 4 waves as 2 x 2; two workgroups per CU run out of phase and fill each other's barrier / LDS waits
constexpr int kTgBM = 64, kTgBN = 64;
constexpr int kTgCh = 4;                       // K-chunks (of 16) per pipeline stage: 1.7 us of MFMAs hide the next stage's loads
constexpr int kTgLd = kTgCh * 16 + 4;          // LDS row stride: the 16 lanes of a quarter wave hit 64 different banks

template <bool WLDS>   // WLDS: the W fragments go through LDS too (each of a workgroup's four column tiles is fetched by ONE wave)
__global__ __launch_bounds__(kTgThreads, 2) void gemm_bias_act_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ Wp,
                                                                 const float* __restrict__ bias, float* __restrict__ C, int64_t ldc,
                                                                 int M, int n_tiles, int N, int K, int k_chunks, int act, GemmRowMap rm,
                                                                 int k_split, float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* as = reinterpret_cast<float*>(smem_raw);              // [2][kTgBM][kTgLd]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;                       // 32 rows x 32 columns per wave
  const int m_first = blockIdx.y * kTgBM;
  const int tile0 = blockIdx.x * (kTgBN / 16) + wn * 2;         // this wave's two column tiles
  // A slab loader: four lanes move one row's 16 floats of each of the stage's chunks.  Rows 4 apart sit in neighbouring lane quads:
  // at a row stride of 68 floats the sixteen lanes of a quarter wave then write 64 different banks (consecutive rows in consecutive
  // quads collide four ways: SQ_LDS_BANK_CONFLICT 13.7 M -> 6.8 M cycles over the GEMM launches of a cfg-5 pass, LDS-active 41 -> 34 M;
  // the launches take the same time - LDS is not what they wait for)
  const int a_row = 16 * wave + (lane >> 4) + 4 * ((lane >> 2) & 3), a_col = (lane & 3) * 4;
  const int r0 = m_first + a_row;
  const float* a0 = A + (int64_t)(r0 < M ? r0 : M - 1) * lda + a_col;     // clamped rows are never stored
  f32x4 va[kTgCh];
  auto load_a = [&](int stage) {
#pragma unroll
    for (int cc = 0; cc < kTgCh; ++cc) {
      const int c = stage * kTgCh + cc;
      const int k = c * 16 + a_col;
      if (k + 3 < K) {
        va[cc] = *reinterpret_cast<const f32x4*>(a0 + c * 16);
      } else {                                                 // the ragged end of K (and chunks beyond it): zeros
        for (int i = 0; i < 4; ++i) va[cc][i] = k + i < K ? a0[c * 16 + i] : 0.f;
      }
    }
  };
  auto store_a = [&](int buf) {
    float* dst = as + buf * (kTgBM * kTgLd);
#pragma unroll
    for (int cc = 0; cc < kTgCh; ++cc) {
      *reinterpret_cast<f32x4*>(dst + a_row * kTgLd + cc * 16 + a_col) = va[cc];
    }
  };
  const int t0 = tile0 < n_tiles ? tile0 : n_tiles - 1, t1 = tile0 + 1 < n_tiles ? tile0 + 1 : n_tiles - 1;
  gf32x4_ptr w0 = (gf32x4_ptr)(uintptr_t)Wp + (int64_t)t0 * k_chunks * 64 + lane;
  gf32x4_ptr w1 = (gf32x4_ptr)(uintptr_t)Wp + (int64_t)t1 * k_chunks * 64 + lane;
  f32x4 wc[WLDS ? 1 : kTgCh][2], wx[WLDS ? 1 : kTgCh][2];
  auto load_w = [&](int stage, f32x4 (&w)[WLDS ? 1 : kTgCh][2]) {
    if constexpr (!WLDS) {
#pragma unroll
      for (int cc = 0; cc < kTgCh; ++cc) {
        const int c = stage * kTgCh + cc;
        const int cl = c < k_chunks ? c : k_chunks - 1;           // beyond K the A slab is zero: any fragment will do
        w[cc][0] = w0[(int64_t)cl * 64];
        w[cc][1] = w1[(int64_t)cl * 64];
      }
    }
  };
  // WLDS: wave (wm, wn) fetches column tile 2 wn + wm of the workgroup's four; every wave reads its two tiles back from LDS
  f32x4* ws = reinterpret_cast<f32x4*>(as + 2 * kTgBM * kTgLd);      // [2][4 tiles][kTgCh][64 lanes]
  const int tl = wn * 2 + wm;
  const int tg = blockIdx.x * (kTgBN / 16) + tl;
  gf32x4_ptr wl = (gf32x4_ptr)(uintptr_t)Wp + (int64_t)(tg < n_tiles ? tg : n_tiles - 1) * k_chunks * 64 + lane;
  f32x4 vw[kTgCh];
  auto load_wl = [&](int stage) {
#pragma unroll
    for (int cc = 0; cc < kTgCh; ++cc) {
      const int c = stage * kTgCh + cc;
      vw[cc] = wl[(int64_t)(c < k_chunks ? c : k_chunks - 1) * 64];
    }
  };
  auto store_wl = [&](int buf) {
#pragma unroll
    for (int cc = 0; cc < kTgCh; ++cc) ws[((buf * 4 + tl) * kTgCh + cc) * 64 + lane] = vw[cc];
  };
  f32x4 acc[2][2];
#pragma unroll
  for (int r = 0; r < 2; ++r) acc[r][0] = acc[r][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  // k_split > 1: workgroup z of a tile takes the stages [z S / k_split, (z + 1) S / k_split) and leaves its raw sums in `partial`
  const int all_stages = (k_chunks + kTgCh - 1) / kTgCh;
  const int st_first = (int)(((int64_t)blockIdx.z * all_stages) / k_split);
  const int n_stages = (int)(((int64_t)(blockIdx.z + 1) * all_stages) / k_split);
  load_a(st_first);
  if constexpr (WLDS) load_wl(st_first);
  else load_w(st_first, wc);
  store_a(st_first & 1);
  if constexpr (WLDS) store_wl(st_first & 1);
  __syncthreads();
  const int x_off = (wm * 32 + (lane & 15)) * kTgLd + 4 * (lane >> 4);
  for (int st = st_first; st < n_stages; ++st) {
    const bool more = st + 1 < n_stages;
    if (more) {                                                // next stage's slab and fragments in flight
      load_a(st + 1);
      if constexpr (WLDS) load_wl(st + 1);
      else load_w(st + 1, wx);
    }
    const float* x = as + (st & 1) * (kTgBM * kTgLd) + x_off;
    const f32x4* wsb = ws + (((st & 1) * 4 + wn * 2) * kTgCh) * 64 + lane;
#pragma unroll
    for (int cc = 0; cc < kTgCh; ++cc) {
      f32x4 xv[2], wf[2];
#pragma unroll
      for (int r = 0; r < 2; ++r) xv[r] = *reinterpret_cast<const f32x4*>(x + r * 16 * kTgLd + cc * 16);
      if constexpr (WLDS) {
        wf[0] = wsb[cc * 64];
        wf[1] = wsb[(kTgCh + cc) * 64];
      } else {
        wf[0] = wc[cc][0];
        wf[1] = wc[cc][1];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          acc[r][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[r][i], wf[0][i], acc[r][0], 0, 0, 0);
          acc[r][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[r][i], wf[1][i], acc[r][1], 0, 0, 0);
        }
      }
    }
    if (more) {
      // the other buffer was last read in stage st - 1, which every wave left through the barrier below
      store_a((st + 1) & 1);
      if constexpr (WLDS) {
        store_wl((st + 1) & 1);
      } else {
#pragma unroll
        for (int cc = 0; cc < kTgCh; ++cc) wc[cc][0] = wx[cc][0], wc[cc][1] = wx[cc][1];
      }
    }
    __syncthreads();
  }
  // ---- D: column lane & 15, rows 4 (lane >> 4) + j of each 16-row block ------------------------------------------------------
  if (k_split > 1) {
    const int np = n_tiles * 16;
    float* dst = partial + (int64_t)blockIdx.z * M * np;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int tile = tile0 + t;
      if (tile >= n_tiles) continue;
      const int col = tile * 16 + (lane & 15);
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int m = m_first + wm * 32 + r * 16 + 4 * (lane >> 4) + j;
          if (m < M) dst[(int64_t)m * np + col] = acc[r][t][j];
        }
    }
    return;
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int tile = tile0 + t;
    const int col = tile * 16 + (lane & 15);
    if (tile >= n_tiles || col >= N) continue;
    const float b = bias ? bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int m = m_first + wm * 32 + r * 16 + 4 * (lane >> 4) + j;
        if (m < M) {
          if (rm.group <= 0) {
            C[(int64_t)m * ldc + col] = apply_act(acc[r][t][j] + b, act);
          } else {   // rows (g, i) = (m / group, m % group) scattered to C + g group_stride + i row_stride; rows i >= kept dropped
            const int g = m / rm.group, i = m - g * rm.group;
            if (i < rm.kept) C[(int64_t)g * rm.group_stride + (int64_t)i * rm.row_stride + col] = apply_act(acc[r][t][j] + b, act);
          }
        }
      }
    }
  }
}

// the partial sums of a split-K launch, added in split order (the same order on every run), then bias / activation / row map
__global__ __launch_bounds__(256) void gemm_split_reduce_kernel(const float* __restrict__ partial, int k_split, int M, int np, int N,
                                                                const float* __restrict__ bias, float* __restrict__ C, int64_t ldc, int act,
                                                                GemmRowMap rm) {
  const int m = blockIdx.x;
  const int g = rm.group > 0 ? m / rm.group : 0, i = rm.group > 0 ? m - g * rm.group : 0;
  if (rm.group > 0 && i >= rm.kept) return;
  float* dst = rm.group > 0 ? C + (int64_t)g * rm.group_stride + (int64_t)i * rm.row_stride : C + (int64_t)m * ldc;
  for (int col = threadIdx.x; col < N; col += blockDim.x) {
    float v = partial[(int64_t)m * np + col];
    for (int z = 1; z < k_split; ++z) v += partial[((int64_t)z * M + m) * np + col];
    dst[col] = apply_act(v + (bias ? bias[col] : 0.f), act);
  }
}

bool gemm_bias_act_supported(const float* A, int64_t lda, int M, int K) {
  // (M = 64 - dec.fc of the Seq2Seq decoder, 64 x 8192 x 1024 - was measured on this kernel too: slower than the row-tile
  //  kernel's 27.7 us, cfg 5 218 instead of 227 M samples/s; MMK_GEMM_MIN_M moves the threshold)
  static const int min_m = [] { const char* e = diag_only("MMK_GEMM_MIN_M"); return e ? atoi(e) : 128; }();
  return M >= min_m && K >= 16 && (lda % 4) == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0;
}

// How many ways a launch with few tiles splits K: a tile's K loop is a chain of stages of ~2 us each (a slab's way from L2 is
// longer than its MFMAs), so a grid that leaves CUs idle is cut along K until it fills them (512 workgroups = two per CU)
int gemm_bias_act_k_split(int M, int n_tiles, int k_chunks, int forced = 0) {
  const int wgs = ((n_tiles + kTgBN / 16 - 1) / (kTgBN / 16)) * ((M + kTgBM - 1) / kTgBM);
  const int stages = (k_chunks + kTgCh - 1) / kTgCh;
  int ks = 1;
  while (ks * 2 <= stages && wgs * ks * 2 <= 512 && ks < 8) ks *= 2;
  if (forced > 0) ks = forced < stages ? forced : stages;
  return ks;
}

int64_t gemm_bias_act_partial_floats(int M, int n_tiles, int k_chunks) {
  const int ks = gemm_bias_act_k_split(M, n_tiles, k_chunks);
  return ks > 1 ? (int64_t)ks * M * n_tiles * 16 : 0;
}

int launch_gemm_bias_act(const float* A, int64_t lda, const float* Wp, const float* bias, int n_tiles, int k_chunks, int N, int K, float* C,
                         int64_t ldc, int M, int act, hipStream_t stream, GemmRowMap rm, float* partial, int64_t partial_floats, int forced_k_split) {
  if (M <= 0 || n_tiles <= 0) return MMK_OK;
  if (!gemm_bias_act_supported(A, lda, M, K)) return fail(MMK_ERR_UNSUPPORTED, "gemm_bias_act: needs M >= 128 and a 16-byte aligned A");
  int ks = partial ? gemm_bias_act_k_split(M, n_tiles, k_chunks, forced_k_split) : 1;
  if (ks > 1 && (int64_t)ks * M * n_tiles * 16 > partial_floats) ks = 1;
  dim3 grid((n_tiles + kTgBN / 16 - 1) / (kTgBN / 16), (M + kTgBM - 1) / kTgBM, ks), block(kTgThreads);
  const size_t lds = (size_t)2 * kTgBM * kTgLd * sizeof(float);
  const char* we = diag_only("MMK_GEMM_WLDS");
  if (!(we && we[0] == '0'))
    hipLaunchKernelGGL(gemm_bias_act_kernel<true>, grid, block, lds + (size_t)2 * 4 * kTgCh * 64 * 16, stream, A, lda, Wp, bias, C, ldc, M, n_tiles, N, K,
                       k_chunks, act, rm, ks, partial);
  else
    hipLaunchKernelGGL(gemm_bias_act_kernel<false>, grid, block, lds, stream, A, lda, Wp, bias, C, ldc, M, n_tiles, N, K, k_chunks, act, rm, ks, partial);
  MMK_HIP(hipGetLastError());
  if (ks > 1) {
    hipLaunchKernelGGL(gemm_split_reduce_kernel, dim3(M), dim3(256), 0, stream, partial, ks, M, n_tiles * 16, N, bias, C, ldc, act, rm);
    MMK_HIP(hipGetLastError());
  }
  return MMK_OK;
}

}  // namespace mmk

// Probe: how long does ONE iteration of a WaveNet cfg-4 stage take when the visit loop is plain code (poll, LDS, FMAs from registers,
// lane reductions, gate, publish) instead of the pipelined kernel's MFMA blocks, loader waves and request bookkeeping?
//   hipcc --offload-arch=gfx950 -O3 -o wn_lean_stage wn_lean_stage.hip && ./wn_lean_stage
// The stage of wavenet_pipe.hip: 32 workgroups on one XCD, each owning 16 rows of the stage's 4 iterations (C = 256 channels, a clip
// group of 4).  Per iteration a workgroup gathers the 2 x 256 values x 4 clips its 31 neighbours and itself published for the
// previous iteration (2048 data-tagged granules, two generations), multiplies its 16 rows against [tap | y | h] (K = 768: 24 inputs x
// 4 clips per thread, weights in registers), reduces over the 32 lanes of a row, gates row pairs and publishes 16 x 4 values.
// Synthetic numbers, no result check: it is the time per iteration that is asked.  256 workgroups are launched and those that do not
// land on XCD 0 (HW_REG_XCC_ID) leave at once... they cannot be told apart before launch, so every eighth workgroup (round-robin
// placement) is taken and the XCC id is verified.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef unsigned long long u64;
#ifdef FAT   // one layer per 8 workgroups: 64 rows each, 8 lanes per row, an 8-way all-gather
constexpr int C = 256, MG = 4, NWG = 8, ROWS = 64, NIT = 1, K = 3 * C, LPR = 8;
#else
constexpr int C = 256, MG = 4, NWG = 32, ROWS = 16, NIT = 4, K = 3 * C, LPR = 32;
#endif
constexpr int KPT = K / LPR;     // inputs per thread
constexpr int kThreads = 512;

struct Params {
  const float* w;      // [NWG][NIT][ROWS][K]
  float* ring;         // [NIT][64][MG][C] taps (read only here)
  u64* gran;           // [2 generations][2 vectors][MG][C]
  float* sink;
  unsigned* err;
  unsigned* xcc;       // [NWG]
  int n_steps;
};

__device__ __forceinline__ float row32_sum(float v) {      // sum over the 32 lanes of a half wave, in every lane of it
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, false));
  if (LPR == 32) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, false));
    v += __shfl_xor(v, 16);
  }
  return v;
}

__global__ __launch_bounds__(kThreads) void lean_stage_kernel(const Params p) {
  if (blockIdx.x % 8 != 0 || blockIdx.x / 8 >= NWG) return;
  const int j = blockIdx.x / 8;                      // tile owner
  __shared__ __attribute__((aligned(16))) float xin[MG][K];
  const int tid = threadIdx.x;
  const int r = tid / LPR, kq = tid % LPR;           // row, K slice
  if (tid == 0) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    p.xcc[j] = id & 0xf;
  }
  float w[NIT][KPT];
#pragma unroll
  for (int it = 0; it < NIT; ++it)
#pragma unroll
    for (int k = 0; k < KPT; ++k) w[it][k] = p.w[(((size_t)j * NIT + it) * ROWS + r) * K + kq * KPT + k];
  const float gscale = (r & 1) ? -1.4426950408889634f : -2.8853900817779268f, gk = (r & 1) ? 1.f : 2.f, gs = (r & 1) ? 0.f : -1.f;
  // everybody publishes generation-0 values for "iteration -1"
  unsigned epoch = 1;
  if (tid < ROWS * MG) {
    const int c = tid & 3, rr = tid >> 2;
    __hip_atomic_store(p.gran + ((size_t)(0 * 2 + (rr / (ROWS / 2))) * MG + c) * C + j * (ROWS / 2) + (rr % (ROWS / 2)), ((u64)epoch << 32) | __float_as_uint(0.01f * rr), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  }
  float acc_sink = 0.f;
  for (int s = 0; s < p.n_steps; ++s) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int gen = (epoch - 1) & 1;
      // taps of this iteration: plain loads (local ring), 2 values per thread
      const float t0 = p.ring[(((size_t)it * 64 + (s & 63)) * MG) * C + tid], t1 = p.ring[(((size_t)it * 64 + (s & 63)) * MG) * C + 512 + tid];
      // gather y | h of the previous iteration: 2048 granules, 4 per thread
      float v[4];
#ifndef PARPOLL
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const u64* g = p.gran + (size_t)gen * 2 * MG * C + q * 512 + tid;
        u64 x = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while ((unsigned)(x >> 32) != epoch) {
          if (++spins > (1u << 22)) { atomicExch(p.err, 1u); break; }
          x = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        v[q] = __uint_as_float((unsigned)x);
      }
#else
      {
        // all four requests out at once; only the stale ones are asked for again
        const u64* g = p.gran + (size_t)gen * 2 * MG * C + tid;
        u64 x[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) x[q] = __hip_atomic_load(g + q * 512, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        for (;;) {
          bool all = true;
#pragma unroll
          for (int q = 0; q < 4; ++q) all = all && (unsigned)(x[q] >> 32) == epoch;
          if (all) break;
          if (++spins > (1u << 22)) { atomicExch(p.err, 1u); break; }
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if ((unsigned)(x[q] >> 32) != epoch) x[q] = __hip_atomic_load(g + q * 512, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = __uint_as_float((unsigned)x[q]);
      }
#endif
      // LDS layout xin[clip][tap 256 | y 256 | h 256]; gathered index q * 512 + tid = (vector, clip, channel)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int flat = q * 512 + tid, vec = flat / (MG * C), c = (flat / C) % MG, ch = flat % C;
        xin[c][C + vec * C + ch] = v[q];
      }
      xin[(tid >> 8)][tid & 255] = t0;
      xin[2 + (tid >> 8)][tid & 255] = t1;
      __syncthreads();
      float acc[MG] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < KPT; ++k)
#pragma unroll
        for (int c = 0; c < MG; ++c) acc[c] = fmaf(w[it][k], xin[c][kq * KPT + k], acc[c]);
#pragma unroll
      for (int c = 0; c < MG; ++c) acc[c] = row32_sum(acc[c]);
      ++epoch;
      // gate: rows (2 p, 2 p + 1) are one unit's (f, g): the g row sits one half wave up
      float out[MG];
#pragma unroll
      for (int c = 0; c < MG; ++c) {
        const float act = fmaf(__frcp_rn(1.0f + __builtin_amdgcn_exp2f(acc[c] * gscale)), gk, gs);
        const float other = LPR == 32 ? __shfl_xor(act, 32) : __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(act), 0x108, 0xf, 0xf, false));
        out[c] = (r & 1) ? act : act * other;
      }
      if (kq < MG) {                                  // lanes 0 .. 3 of every row publish that row's value for clip kq
        const float val = kq == 0 ? out[0] : (kq == 1 ? out[1] : (kq == 2 ? out[2] : out[3]));
        __hip_atomic_store(p.gran + ((size_t)(((epoch - 1) & 1) * 2 + (r / (ROWS / 2))) * MG + kq) * C + j * (ROWS / 2) + (r % (ROWS / 2)), ((u64)epoch << 32) | __float_as_uint(val),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      acc_sink += out[0];
      __syncthreads();
    }
  }
  if (acc_sink == 123.456f) p.sink[tid] = acc_sink;
}

int main() {
  const int n_steps = 2000;
  Params p = {};
  std::vector<float> w((size_t)NWG * NIT * ROWS * K);
  srand(3);
  for (auto& x : w) x = (rand() / (float)RAND_MAX - 0.5f) * 0.1f;
  float* dw; hipMalloc(&dw, w.size() * 4); hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice);
  p.w = dw;
  hipMalloc(&p.ring, (size_t)NIT * 64 * MG * C * 4); hipMemset(p.ring, 0, (size_t)NIT * 64 * MG * C * 4);
  hipMalloc(&p.gran, (size_t)2 * 2 * MG * C * 8); hipMemset(p.gran, 0, (size_t)2 * 2 * MG * C * 8);
  hipMalloc(&p.sink, 4096); hipMalloc(&p.err, 4); hipMemset(p.err, 0, 4);
  hipMalloc(&p.xcc, NWG * 4); hipMemset(p.xcc, 0xff, NWG * 4);
  p.n_steps = n_steps;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a);
  hipLaunchKernelGGL(lean_stage_kernel, dim3(256), dim3(kThreads), 0, 0, p);
  hipEventRecord(b);
  hipError_t rc = hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, a, b);
  unsigned err = 0, xcc[NWG];
  hipMemcpy(&err, p.err, 4, hipMemcpyDeviceToHost);
  hipMemcpy(xcc, p.xcc, sizeof(xcc), hipMemcpyDeviceToHost);
  bool same = true;
  for (int i = 1; i < NWG; ++i) same = same && xcc[i] == xcc[0];
  printf("launch: %s, timeouts: %u, the 32 workgroups on one XCD: %s (XCC %u)\n", hipGetErrorString(rc), err, same ? "yes" : "NO", xcc[0]);
  printf("%d workgroups x %d rows: %d steps x %d iterations: %.1f us total, %.2f us per iteration (pipelined kernel: 2.2 - 2.4 us)\n", NWG, ROWS, n_steps, NIT, ms * 1e3,
         ms * 1e3 / (n_steps * NIT));
  return 0;
}

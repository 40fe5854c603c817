// Probe: the shader clock under a chip-wide stream of v_mfma_f32_16x16x4_f32 (random operands), and the rate that gives.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_clock mfma_clock.hip && ./mfma_clock
// s_memtime counts shader clocks, s_memrealtime a constant 100 MHz: their ratio over a long MFMA loop is the clock the matrix pipes
// ran at; the loop issues N MFMAs per wave on 4 independent accumulators (no dependency stalls), 2 and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void mfma_loop(const float* __restrict__ src, int iters, unsigned long long* out, float* sink) {
  const int lane = threadIdx.x & 63;
  float a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = src[(threadIdx.x * 4 + i) & 4095]; b[i] = src[(threadIdx.x * 4 + i + 1777) & 4095]; }
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned long long t0 = __builtin_readcyclecounter();     // s_memtime
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(u + i) & 3], b[(u * 3 + i) & 3], acc[i], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = r1 - r0; }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 123.456f) sink[lane] = s;
}

int main() {
  float* src; unsigned long long* out; float* sink;
  hipMalloc(&src, 4096 * 4); hipMalloc(&out, 4096 * 16); hipMalloc(&sink, 1024);
  std::vector<float> h(4096);
  srand(1);
  for (auto& v : h) v = (rand() / (float)RAND_MAX - 0.5f) * 1e-3f;
  hipMemcpy(src, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  for (int wgs_per_cu = 1; wgs_per_cu <= 4; wgs_per_cu *= 2) {
    for (int iters : {2000, 20000}) {
      const int grid = 256 * wgs_per_cu;
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      hipLaunchKernelGGL(mfma_loop, dim3(grid), dim3(256), 0, 0, src, 100, out, sink);
      hipDeviceSynchronize();
      hipEventRecord(a);
      hipLaunchKernelGGL(mfma_loop, dim3(grid), dim3(256), 0, 0, src, iters, out, sink);
      hipEventRecord(b);
      hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, a, b);
      std::vector<unsigned long long> o(2 * grid);
      hipMemcpy(o.data(), out, o.size() * 8, hipMemcpyDeviceToHost);
      double clk = 0, rt = 0;
      for (int i = 0; i < grid; ++i) { clk += o[2 * i]; rt += o[2 * i + 1]; }
      const double ghz = clk / rt * 0.1;                       // 100 MHz real-time counter
      const double mfmas = (double)grid * 4 * iters * 32;
      const double tflops = mfmas * 2048 / (ms * 1e-3) / 1e12;
      printf("%d workgroups of 4 waves per CU, %6d x 32 MFMAs per wave: %8.1f us, shader clock %.2f GHz, %.1f TFLOP/s (%.1f clocks per MFMA and SIMD)\n",
             wgs_per_cu, iters, ms * 1e3, ghz, tflops, clk / grid / ((double)iters * 32 * wgs_per_cu));
    }
  }
  return 0;
}

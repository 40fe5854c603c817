// Probe: what the memory system gives a streaming kernel by its read : write mix (the mu-law kernel reads 4 B and writes 8 B per sample).
//   hipcc --offload-arch=gfx950 -O3 -o rw_mix rw_mix.hip && ./rw_mix
// read-only sum, fill, copy (1:1), 1:2 (fp32 in, int64 out) with the lane's two 16-byte stores 32 bytes apart (the product kernel's pattern)
// and with both store instructions covering 1 KB of consecutive lanes each (the pair exchanged across lanes first).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef long long i64x2 __attribute__((ext_vector_type(2)));
constexpr int kU = 4;

__global__ __launch_bounds__(256) void k_read(const f32x4* __restrict__ x, int64_t n4, float* sink) {
  f32x4 acc = {0, 0, 0, 0};
  const int64_t tile = 256 * kU;
  for (int64_t base = (int64_t)blockIdx.x * tile; base < n4; base += (int64_t)gridDim.x * tile) {
    f32x4 v[kU];
#pragma unroll
    for (int k = 0; k < kU; ++k) { const int64_t i = base + k * 256 + threadIdx.x; v[k] = i < n4 ? x[i] : f32x4{0, 0, 0, 0}; }
#pragma unroll
    for (int k = 0; k < kU; ++k) acc += v[k];
  }
  if (acc[0] == 1.2345f) sink[0] = acc[1];
}
__global__ __launch_bounds__(256) void k_fill(f32x4* __restrict__ y, int64_t n4) {
  const int64_t tile = 256 * kU;
  for (int64_t base = (int64_t)blockIdx.x * tile; base < n4; base += (int64_t)gridDim.x * tile) {
#pragma unroll
    for (int k = 0; k < kU; ++k) { const int64_t i = base + k * 256 + threadIdx.x; if (i < n4) y[i] = f32x4{1.f, 2.f, 3.f, (float)i}; }
  }
}
__global__ __launch_bounds__(256) void k_copy(const f32x4* __restrict__ x, f32x4* __restrict__ y, int64_t n4) {
  const int64_t tile = 256 * kU;
  for (int64_t base = (int64_t)blockIdx.x * tile; base < n4; base += (int64_t)gridDim.x * tile) {
    f32x4 v[kU];
#pragma unroll
    for (int k = 0; k < kU; ++k) { const int64_t i = base + k * 256 + threadIdx.x; v[k] = i < n4 ? x[i] : f32x4{0, 0, 0, 0}; }
#pragma unroll
    for (int k = 0; k < kU; ++k) { const int64_t i = base + k * 256 + threadIdx.x; if (i < n4) y[i] = v[k]; }
  }
}
template <int MODE>   // 0: lane's two stores 32 B apart; 1: exchanged, each store instruction 1 KB contiguous
__global__ __launch_bounds__(256) void k_r1w2(const f32x4* __restrict__ x, i64x2* __restrict__ y, int64_t n4) {
  const int64_t tile = 256 * kU;
  const int lane = threadIdx.x & 63;
  for (int64_t base = (int64_t)blockIdx.x * tile; base < n4; base += (int64_t)gridDim.x * tile) {
    f32x4 v[kU];
#pragma unroll
    for (int k = 0; k < kU; ++k) { const int64_t i = base + k * 256 + threadIdx.x; v[k] = i < n4 ? x[i] : f32x4{0, 0, 0, 0}; }
#pragma unroll
    for (int k = 0; k < kU; ++k) {
      const int64_t i = base + k * 256 + threadIdx.x;
      int c[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) c[q] = (int)(v[k][q] * 127.f + 128.f) & 255;
      if (MODE == 0) {
        if (i < n4) { y[2 * i] = i64x2{c[0], c[1]}; y[2 * i + 1] = i64x2{c[2], c[3]}; }
      } else {
        const int packed = c[0] | (c[1] << 8) | (c[2] << 16) | (c[3] << 24);
        // store A: lane l writes samples 2l, 2l+1 of the wave's 256 -> from lane l/2, bytes (l&1)*2..; store B: samples 128+2l.. -> lane 32 + l/2
        const int pa = __builtin_amdgcn_ds_bpermute((lane >> 1) << 2, packed), pb = __builtin_amdgcn_ds_bpermute((32 + (lane >> 1)) << 2, packed);
        const int sh = (lane & 1) * 16;
        const int64_t w0 = (base + k * 256 + (threadIdx.x & ~63)) * 2;      // first i64x2 of the wave's 2 KB
        if (i < n4) {      // (whole waves in range in this probe)
          y[w0 + lane] = i64x2{(pa >> sh) & 255, (pa >> (sh + 8)) & 255};
          y[w0 + 64 + lane] = i64x2{(pb >> sh) & 255, (pb >> (sh + 8)) & 255};
        }
      }
    }
  }
}

template <typename F>
static double time_us(F launch, int reps) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  launch(); hipDeviceSynchronize();
  hipEventRecord(a);
  for (int r = 0; r < reps; ++r) launch();
  hipEventRecord(b); hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, a, b);
  return 1e3 * ms / reps;
}

int main() {
  const int64_t n = 64ll * 960000, n4 = n / 4;
  float *x, *sink; long long* y;
  hipMalloc(&x, n * 4); hipMalloc(&y, n * 8); hipMalloc(&sink, 64);
  hipMemset(x, 0x3c, n * 4);
  for (int blocks : {1024, 2048, 4096, 8192}) {
    const dim3 g(blocks), b(256);
    const double tr = time_us([&] { hipLaunchKernelGGL(k_read, g, b, 0, 0, (const f32x4*)x, n4, sink); }, 20);
    const double tf = time_us([&] { hipLaunchKernelGGL(k_fill, g, b, 0, 0, (f32x4*)y, n4 * 2); }, 20);
    const double tc = time_us([&] { hipLaunchKernelGGL(k_copy, g, b, 0, 0, (const f32x4*)y, (f32x4*)y + n4, n4); }, 20);
    const double t0 = time_us([&] { hipLaunchKernelGGL((k_r1w2<0>), g, b, 0, 0, (const f32x4*)x, (i64x2*)y, n4); }, 20);
    const double t1 = time_us([&] { hipLaunchKernelGGL((k_r1w2<1>), g, b, 0, 0, (const f32x4*)x, (i64x2*)y, n4); }, 20);
    printf("blocks %5d: read %.2f TB/s (%.1f us)  fill %.2f TB/s (%.1f us)  copy %.2f TB/s (%.1f us)  1:2 strided stores %.2f TB/s (%.1f us)  1:2 contiguous stores %.2f TB/s (%.1f us)\n",
           blocks, n * 4 / tr / 1e6, tr, n * 8 / tf / 1e6, tf, n * 8 / tc / 1e6, tc, n * 12 / t0 / 1e6, t0, n * 12 / t1 / 1e6, t1);
  }
  return 0;
}

// Probe: vector-memory instruction width.  The same bytes read (and written) as 4-, 8- and 16-byte accesses per lane, every instruction of a wave
// covering a contiguous block: how much of the streaming rate is left when a kernel's layout forces narrow accesses (the STFT kernel loads its frames
// and stores its 513-bin rows as one dword per lane).
//   hipcc --offload-arch=gfx950 -O3 -o vmem_width vmem_width.hip && ./vmem_width
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <typename T, int U>
__global__ __launch_bounds__(256) void k_read(const T* __restrict__ x, int64_t n, float* sink) {
  float acc = 0.f;
  const int64_t tile = 256 * U;
  for (int64_t base = (int64_t)blockIdx.x * tile; base < n; base += (int64_t)gridDim.x * tile) {
    T v[U];
#pragma unroll
    for (int k = 0; k < U; ++k) { const int64_t i = base + k * 256 + threadIdx.x; v[k] = x[i < n ? i : 0]; }
#pragma unroll
    for (int k = 0; k < U; ++k) acc += ((const float*)&v[k])[0];
  }
  if (acc == 1.2345f) sink[0] = acc;
}
template <typename T, int U>
__global__ __launch_bounds__(256) void k_write(T* __restrict__ y, int64_t n) {
  const int64_t tile = 256 * U;
  T v;
  for (int q = 0; q < (int)(sizeof(T) / 4); ++q) ((float*)&v)[q] = (float)threadIdx.x;
  for (int64_t base = (int64_t)blockIdx.x * tile; base < n; base += (int64_t)gridDim.x * tile) {
#pragma unroll
    for (int k = 0; k < U; ++k) { const int64_t i = base + k * 256 + threadIdx.x; if (i < n) y[i] = v; }
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
  const int64_t bytes = 512ll << 20;
  void *x, *y; float* sink;
  hipMalloc(&x, bytes); hipMalloc(&y, bytes); hipMalloc(&sink, 64);
  hipMemset(x, 0x3c, bytes);
  const dim3 g(4096), b(256);
  const double r4 = time_us([&] { hipLaunchKernelGGL((k_read<float, 16>), g, b, 0, 0, (const float*)x, bytes / 4, sink); }, 10);
  const double r8 = time_us([&] { hipLaunchKernelGGL((k_read<float2, 8>), g, b, 0, 0, (const float2*)x, bytes / 8, sink); }, 10);
  const double r16 = time_us([&] { hipLaunchKernelGGL((k_read<float4, 4>), g, b, 0, 0, (const float4*)x, bytes / 16, sink); }, 10);
  const double w4 = time_us([&] { hipLaunchKernelGGL((k_write<float, 16>), g, b, 0, 0, (float*)y, bytes / 4); }, 10);
  const double w8 = time_us([&] { hipLaunchKernelGGL((k_write<float2, 8>), g, b, 0, 0, (float2*)y, bytes / 8); }, 10);
  const double w16 = time_us([&] { hipLaunchKernelGGL((k_write<float4, 4>), g, b, 0, 0, (float4*)y, bytes / 16); }, 10);
  printf("read  4 B/lane %.2f TB/s   8 B/lane %.2f TB/s   16 B/lane %.2f TB/s\n", bytes / r4 / 1e6, bytes / r8 / 1e6, bytes / r16 / 1e6);
  printf("write 4 B/lane %.2f TB/s   8 B/lane %.2f TB/s   16 B/lane %.2f TB/s\n", bytes / w4 / 1e6, bytes / w8 / 1e6, bytes / w16 / 1e6);
  return 0;
}

// Probe: how fast can every CU pull "its" slice of a weight set that ALL eight XCDs read once per pass (the access pattern of
// the persistent WaveNet kernels: 32 tile owners per XCD, the same 61 MB on every XCD, pass after pass)?
//   hipcc --offload-arch=gfx950 -O3 -o stream_bw stream_bw.hip && ./stream_bw
// Prints us per pass and GB/s per CU for several in-flight depths and set sizes.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// grid = 256 workgroups x 512 threads.  Workgroup b: owner j = b / 8 of XCD b % 8 (round-robin placement).  A "tile" is
// tile_kb KiB; per pass the workgroup reads n_tiles tiles: tile t of owner j lives at base + (t * 32 + j) * tile bytes.
// DEPTH tiles are requested before the first is consumed.
template <int F4_PER_THREAD, int DEPTH>
__global__ __launch_bounds__(512) void stream_kernel(const f32x4* __restrict__ base, int n_tiles, int passes, float* sink) {
  const int j = blockIdx.x / 8;
  const int tid = threadIdx.x;
  const size_t tile_f4 = (size_t)F4_PER_THREAD * 512;
  f32x4 buf[DEPTH][F4_PER_THREAD];
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  const int total = n_tiles * passes;
  auto issue = [&](int slot, int t) {
    const f32x4* p = base + ((size_t)(t % n_tiles) * 32 + j) * tile_f4 + tid;
#pragma unroll
    for (int u = 0; u < F4_PER_THREAD; ++u) buf[slot][u] = p[(size_t)u * 512];
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) issue(d, d);
  for (int t = 0; t < total; t += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
      for (int u = 0; u < F4_PER_THREAD; ++u) acc += buf[d][u];
      issue(d, t + d + DEPTH);
    }
  }
  if (acc[0] == 123.456f) sink[0] = acc[1];
}

template <int F, int D>
static void run(const f32x4* base, int n_tiles, int passes, float* sink, const char* what) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((stream_kernel<F, D>), dim3(256), dim3(512), 0, 0, base, n_tiles, 2, sink);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL((stream_kernel<F, D>), dim3(256), dim3(512), 0, 0, base, n_tiles, passes, sink);
  hipEventRecord(b);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double tile_kb = F * 512 * 16 / 1024.0;
  const double us_pass = 1e3 * ms / passes;
  const double per_cu = n_tiles * tile_kb * 1024.0 / (us_pass * 1e-6) / 1e9;
  printf("%-28s tile %5.0f KiB  depth %d (%4.0f KiB in flight)  set %6.1f MB: %8.2f us/pass  %6.1f GB/s per CU  %5.2f TB/s chip\n", what, tile_kb, D,
         tile_kb * D, n_tiles * 32 * tile_kb / 1024.0, us_pass, per_cu, per_cu * 256 / 1e3);
}

int main() {
  const size_t bytes = (size_t)512 << 20;
  f32x4* buf;
  float* sink;
  hipMalloc(&buf, bytes);
  hipMalloc(&sink, 64);
  hipMemset(buf, 0, bytes);
  // 31 tiles of 64 KiB x 32 owners = 62 MB (the chain kernel's set); 30 x 48 KiB = 45 MB (wavenet_persist's)
  run<8, 1>(buf, 31, 200, sink, "62 MB set, all XCDs");
  run<8, 2>(buf, 31, 200, sink, "62 MB set, all XCDs");
  run<8, 3>(buf, 31, 200, sink, "62 MB set, all XCDs");
  run<4, 2>(buf, 62, 200, sink, "62 MB set, all XCDs");
  run<4, 4>(buf, 62, 200, sink, "62 MB set, all XCDs");
  run<4, 6>(buf, 62, 200, sink, "62 MB set, all XCDs");
  run<2, 8>(buf, 124, 200, sink, "62 MB set, all XCDs");
  run<8, 2>(buf, 1, 2000, sink, "2 MB set (L2 resident)");
  run<8, 2>(buf, 100, 100, sink, "200 MB set");
  run<8, 2>(buf, 200, 50, sink, "400 MB set (HBM)");
  hipFree(buf);
  hipFree(sink);
  return 0;
}

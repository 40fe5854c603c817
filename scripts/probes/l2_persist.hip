// Probe: does an XCD's L2 keep read-only lines from one kernel launch to the next?
//   hipcc --offload-arch=gfx950 -O3 -o l2_persist l2_persist.hip && ./l2_persist
// 256 workgroups x 512 threads; workgroup b reads ITS slice (set / 256 bytes, the same slice in every launch; placement is
// round-robin, so an XCD sees the same 1/8 of the set every time).  All of a slice's loads are issued before the first is consumed
// (the burst of lstm_step_kernel).  20 launches back to back, per-launch time from events around the whole train; the same train
// again with `nt` loads for the part of each slice beyond `keep` bytes (those lines should not displace the kept ones).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int F4, bool NT_TAIL>
__global__ __launch_bounds__(512) void read_kernel(const f32x4* __restrict__ base, int keep_f4, float* sink) {
  const f32x4* p = base + (size_t)blockIdx.x * F4 * 512 + threadIdx.x;
  f32x4 v[F4];
#pragma unroll
  for (int u = 0; u < F4; ++u) {
    const f32x4* q = p + (size_t)u * 512;
    if (NT_TAIL) {     // (__builtin_nontemporal_load does not set the nt bit on these loads: asm)
      if (u >= keep_f4) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v[u]) : "v"(q) : "memory");
      else asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[u]) : "v"(q) : "memory");
    } else {
      v[u] = *q;
    }
  }
  if (NT_TAIL) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < F4; ++u) asm volatile("" : "+v"(v[u]));
  }
  f32x4 acc = v[0];
#pragma unroll
  for (int u = 1; u < F4; ++u) acc += v[u];
  if (acc[0] == 123.456f) sink[blockIdx.x] = acc[1];
}

// the LSTM step's pattern: workgroups b and b + 64 (same XCD) read the SAME slice of F4 x 8 KiB, 128 slices in all
template <int F4>
__global__ __launch_bounds__(512) void read_shared_kernel(const f32x4* __restrict__ base, float* sink) {
  const int slice = (blockIdx.x % 64) + 64 * (blockIdx.x / 128);
  const f32x4* p = base + (size_t)slice * F4 * 512 + threadIdx.x;
  f32x4 v[F4];
#pragma unroll
  for (int u = 0; u < F4; ++u) v[u] = p[(size_t)u * 512];
  f32x4 acc = v[0];
#pragma unroll
  for (int u = 1; u < F4; ++u) acc += v[u];
  if (acc[0] == 123.456f) sink[blockIdx.x] = acc[1];
}

template <int F4>
static void shared_train(const f32x4* base, float* sink) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((read_shared_kernel<F4>), dim3(256), dim3(512), 0, 0, base, sink);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < 40; ++i) hipLaunchKernelGGL((read_shared_kernel<F4>), dim3(256), dim3(512), 0, 0, base, sink);
  hipEventRecord(b);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double mb = F4 * 512 * 16 * 128 / 1048576.0;
  printf("pairs of workgroups on one slice, %3d KiB per workgroup, %5.2f MB unique (%4.2f MB per XCD): %6.2f us per launch\n", F4 * 8, mb, mb / 8,
         1e3 * ms / 40);
}

// the proposed LSTM step: every workgroup reads its OWN 128 KiB of weights (the last NT_F4 of 16 x 8 KiB with nt), the 256 KiB
// state of its direction (shared by 128 workgroups, all XCDs) and 16 KiB of gate terms that move from launch to launch
template <int NT_F4>
__global__ __launch_bounds__(512) void read_step_kernel(const f32x4* __restrict__ wbase, const f32x4* __restrict__ hbase,
                                                        const f32x4* __restrict__ gbase, float* sink) {
  const f32x4* p = wbase + (size_t)blockIdx.x * 16 * 512 + threadIdx.x;
  const f32x4* h = hbase + (size_t)(blockIdx.x / 128) * 32 * 512 + threadIdx.x;
  const f32x4* g = gbase + (size_t)blockIdx.x * 2 * 512 + threadIdx.x;
  f32x4 v[16], hv[32], gv[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(gv[u]) : "v"(g + (size_t)u * 512) : "memory");
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    if (u >= 16 - NT_F4) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v[u]) : "v"(p + (size_t)u * 512) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[u]) : "v"(p + (size_t)u * 512) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(hv[2 * u]) : "v"(h + (size_t)(2 * u) * 512) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(hv[2 * u + 1]) : "v"(h + (size_t)(2 * u + 1) * 512) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // (every destination stays live across the wait: the compiler takes an asm's output as ready and would reuse the registers)
#pragma unroll
  for (int u = 0; u < 16; ++u) asm volatile("" : "+v"(v[u]), "+v"(hv[2 * u]), "+v"(hv[2 * u + 1]));
  asm volatile("" : "+v"(gv[0]), "+v"(gv[1]));
  f32x4 acc = gv[0] + gv[1];
#pragma unroll
  for (int u = 0; u < 16; ++u) acc += v[u] + hv[2 * u] + hv[2 * u + 1];
  if (acc[0] == 123.456f) sink[blockIdx.x] = acc[1];
}

// today's LSTM step: workgroups b and b + 64 share 256 KiB of weights, each reads the 128 KiB state of its 32 rows (shared by the 64
// workgroups of its direction and row block) and 16 KiB of moving gate terms
template <int ROT, bool WITH_H>
__global__ __launch_bounds__(512) void read_step_now_kernel(const f32x4* __restrict__ wbase, const f32x4* __restrict__ hbase,
                                                            const f32x4* __restrict__ gbase, float* sink) {
  const int slice = (blockIdx.x % 64) + 64 * (blockIdx.x / 128);
  const f32x4* p = wbase + (size_t)slice * 32 * 512 + threadIdx.x;
  const f32x4* h = hbase + (size_t)(blockIdx.x / 64) * 16 * 512 + threadIdx.x;
  const int rot = ROT ? (blockIdx.x * ROT) & 15 : 0;       // ROT: every workgroup starts its sweep over the shared state elsewhere
  const f32x4* g = gbase + (size_t)blockIdx.x * 2 * 512 + threadIdx.x;
  f32x4 v[32], hv[16], gv[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(gv[u]) : "v"(g + (size_t)u * 512) : "memory");
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[2 * u]) : "v"(p + (size_t)(2 * u) * 512) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[2 * u + 1]) : "v"(p + (size_t)(2 * u + 1) * 512) : "memory");
    if (WITH_H) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(hv[u]) : "v"(h + (size_t)((u + rot) & 15) * 512) : "memory");
    else hv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int u = 0; u < 16; ++u) asm volatile("" : "+v"(v[2 * u]), "+v"(v[2 * u + 1]), "+v"(hv[u]));
  asm volatile("" : "+v"(gv[0]), "+v"(gv[1]));
  f32x4 acc = gv[0] + gv[1];
#pragma unroll
  for (int u = 0; u < 16; ++u) acc += v[2 * u] + v[2 * u + 1] + hv[u];
  if (acc[0] == 123.456f) sink[blockIdx.x] = acc[1];
}

template <int ROT, bool WITH_H>
static void step_now_train(const f32x4* base, float* sink) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const f32x4* hbase = base + ((size_t)64 << 20) / 16;
  const f32x4* gbase = base + ((size_t)96 << 20) / 16;
  auto go = [&](int i) {
    hipLaunchKernelGGL((read_step_now_kernel<ROT, WITH_H>), dim3(256), dim3(512), 0, 0, base, hbase + (size_t)(i % 2) * 64 * 512, gbase + (size_t)(i % 8) * 512 * 512, sink);
  };
  for (int i = 0; i < 4; ++i) go(i);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < 40; ++i) go(i);
  hipEventRecord(b);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  printf("today's step pattern (pairs on 256 KiB of weights; %s; moving gate terms; sweep rotated by %d x workgroup): %6.2f us per launch\n",
         WITH_H ? "128 KiB state per row block" : "no state reads", ROT, 1e3 * ms / 40);
}

template <int NT_F4>
static void step_train(const f32x4* base, float* sink) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const f32x4* hbase = base + ((size_t)64 << 20) / 16;
  const f32x4* gbase = base + ((size_t)96 << 20) / 16;
  auto go = [&](int i) {
    hipLaunchKernelGGL((read_step_kernel<NT_F4>), dim3(256), dim3(512), 0, 0, base, hbase + (size_t)(i % 2) * 64 * 512, gbase + (size_t)(i % 8) * 512 * 512, sink);
  };
  for (int i = 0; i < 4; ++i) go(i);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < 40; ++i) go(i);
  hipEventRecord(b);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  printf("step pattern (own 128 KiB of weights, %2d/16 nt; shared 256 KiB state; moving gate terms): %6.2f us per launch\n", NT_F4, 1e3 * ms / 40);
}

template <int F4, bool NT>
static double train(const f32x4* base, int keep_f4, float* sink, int n) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((read_kernel<F4, NT>), dim3(256), dim3(512), 0, 0, base, keep_f4, sink);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL((read_kernel<F4, NT>), dim3(256), dim3(512), 0, 0, base, keep_f4, sink);
  hipEventRecord(b);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return 1e3 * ms / n;
}

template <int F4>
static void one(const f32x4* base, float* sink) {
  const double mb = F4 * 512 * 16 * 256 / 1048576.0;
  const double plain = train<F4, false>(base, F4, sink, 40);
  printf("set %6.1f MB (%4.2f MB per XCD): %6.2f us per launch", mb, mb / 8, plain);
  for (int keep = F4 / 4; keep < F4; keep += F4 / 4) {
    const double t = train<F4, true>(base, keep, sink, 40);
    printf("  | nt beyond %4.2f MB/XCD: %6.2f", keep * 512 * 16 * 32 / 1048576.0, t);
  }
  printf("\n");
}

int main() {
  f32x4* base;
  float* sink;
  hipMalloc(&base, (size_t)256 << 20);
  hipMalloc(&sink, 4096);
  hipMemset(base, 0, (size_t)256 << 20);
  // an empty-ish launch for the fixed cost
  printf("fixed: %.2f us per launch (512 B per thread)\n", train<2, false>(base, 2, sink, 40));
  one<4>(base, sink);     //   8 MB
  one<8>(base, sink);     //  16 MB
  one<12>(base, sink);    //  24 MB
  one<16>(base, sink);    //  32 MB (the LSTM step's 33.5 MB)
  one<20>(base, sink);    //  40 MB
  one<32>(base, sink);    //  64 MB
  one<17>(base, sink);    //  34 MB: just over
  one<18>(base, sink);    //  36 MB
  step_now_train<0, true>(base, sink);
  step_now_train<1, true>(base, sink);
  step_now_train<3, true>(base, sink);
  step_now_train<0, false>(base, sink);
  step_train<0>(base, sink);
  step_train<2>(base, sink);
  step_train<4>(base, sink);
  step_train<6>(base, sink);
  step_train<8>(base, sink);
  step_train<16>(base, sink);
  shared_train<24>(base, sink);
  shared_train<28>(base, sink);
  shared_train<30>(base, sink);
  shared_train<32>(base, sink);   // the LSTM step: 33.5 MB unique
  shared_train<34>(base, sink);
  shared_train<40>(base, sink);
  return 0;
}

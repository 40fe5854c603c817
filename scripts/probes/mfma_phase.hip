// Probe: cycles of the compute part of one persistent-kernel iteration on ONE workgroup:
//   8 x ds_read_b128 operands -> 32 x v_mfma_f32_4x4x1_16b_f32 (two accumulators) -> DPP reduce -> ds_write_b128 -> barrier
// for 4 / 8 / 12 waves per workgroup, and the MFMA chain alone.   hipcc --offload-arch=gfx950 -O3 -o mfma_phase mfma_phase.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NW, int MODE>
__global__ __launch_bounds__(NW * 64) void k(float* out, long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) float x[4 * 260 * 3];
  __shared__ f32x4 red[NW * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 4 * 260 * 3; i += NW * 64) x[i] = 0.001f * i;
  f32x4 w[8];
  for (int u = 0; u < 8; ++u) w[u] = f32x4{0.01f * (lane + u), 0.02f, 0.03f, 0.04f};
  __syncthreads();
  const int x_off = (lane & 3) * 260 + (wave & 1) * 128 + ((lane >> 2) & 3) * 32 + (wave % 3) * 1040;
  f32x4 tot = {0, 0, 0, 0};
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    f32x4 xv[8];
    if (MODE != 1) {
#pragma unroll
      for (int u = 0; u < 8; ++u) xv[u] = *reinterpret_cast<const f32x4*>(x + x_off + u * 4);
    } else {
#pragma unroll
      for (int u = 0; u < 8; ++u) xv[u] = w[u];
    }
    f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        if (kk & 1) a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(xv[u][kk], w[u][kk], a1, 0, 0, 0);
        else a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(xv[u][kk], w[u][kk], a0, 0, 0, 0);
      }
    f32x4 acc = a0 + a1;
    if (MODE != 1) {
      red[tid] = acc;
      __syncthreads();
      tot += red[(tid + 64) % (NW * 64)];
      if (MODE == 2) __syncthreads();
    } else {
      tot += acc;
    }
  }
  const long long t1 = clock64();
  if (tid == 0) cyc[0] = t1 - t0;
  out[tid] = tot[0] + tot[1];
}

template <int NW, int MODE>
void run(const char* what) {
  float* out; long long* cyc;
  hipMalloc(&out, 4096 * 4); hipMalloc(&cyc, 8);
  const int iters = 20000;
  hipLaunchKernelGGL((k<NW, MODE>), dim3(1), dim3(NW * 64), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  long long h = 0;
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-44s %2d waves: %7.1f cycles / iteration\n", what, NW, (double)h / iters);
  hipFree(out); hipFree(cyc);
}

int main() {
  run<4, 1>("32 MFMA 4x4x1 (2 chains), registers only");
  run<8, 1>("32 MFMA 4x4x1 (2 chains), registers only");
  run<12, 1>("32 MFMA 4x4x1 (2 chains), registers only");
  run<4, 0>("8 ds_read_b128 + 32 MFMA + write + 1 barrier");
  run<8, 0>("8 ds_read_b128 + 32 MFMA + write + 1 barrier");
  run<12, 0>("8 ds_read_b128 + 32 MFMA + write + 1 barrier");
  run<8, 2>("same + second barrier");
  run<12, 2>("same + second barrier");
  return 0;
}

// Probe of the operand layout of v_mfma_f32_4x4x1_16b_f32 on gfx950 (diagnostic, not part of the library).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const float* a, const float* b, float* d) {
  const int lane = threadIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[lane], b[lane], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) d[lane * 4 + r] = acc[r];
}
int main() {
  float ha[64], hb[64], hd[256];
  float *a, *b, *d;
  hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 1024);
  for (int pass = 0; pass < 2; ++pass) {
    for (int i = 0; i < 64; ++i) { ha[i] = pass == 0 ? (float)(i + 1) : 1.f; hb[i] = pass == 0 ? 1.f : (float)(i + 1); }
    hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(a, b, d);
    hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
    printf("pass %d (%s lane feeding D[lane][reg]):\n", pass, pass == 0 ? "A" : "B");
    for (int l = 0; l < 64; ++l) printf("  lane %2d: %3.0f %3.0f %3.0f %3.0f\n", l, hd[l * 4] - 1, hd[l * 4 + 1] - 1, hd[l * 4 + 2] - 1, hd[l * 4 + 3] - 1);
  }
  return 0;
}

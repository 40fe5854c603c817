// Checks sampler256.h's wave helpers against host arithmetic: hipcc --offload-arch=gfx950 -O3 -I mimikit_amd/csrc -I include sampler_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "sampler256.h"

__global__ void probe(const float* lg, const float* u, int* out, float* scan_out, float* max_out, int rows) {
  __shared__ __attribute__((aligned(16))) float lbuf[260];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int r = blockIdx.x; r < rows; r += gridDim.x) {
    const float* row = lg + r * 256;
    if (tid < 256) lbuf[tid] = row[tid];
    if (tid == 256) lbuf[256] = 3.f;
    __syncthreads();
    if (wave == 0) {
      const float denom = fmaxf(1.f / (1.f + expf(-lbuf[256])), 1e-4f);
      out[r] = mmk::sample_256(lbuf, true, denom, 0.9f, u[r], lane);
      const float loc = row[lane];
      scan_out[r * 64 + lane] = mmk::wave_scan_dpp(loc);
      max_out[r * 64 + lane] = mmk::wave_max_dpp(loc);
    }
    __syncthreads();
  }
}

int main() {
  const int rows = 512;
  std::vector<float> lg(rows * 256), u(rows);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) * (1.0f / 16777216.0f); };
  for (auto& v : lg) v = (rnd() - 0.5f) * 8.f;
  for (auto& v : u) v = rnd();
  float *dl, *du, *dscan, *dmax; int* dout;
  hipMalloc(&dl, lg.size() * 4); hipMalloc(&du, rows * 4); hipMalloc(&dout, rows * 4); hipMalloc(&dscan, rows * 64 * 4); hipMalloc(&dmax, rows * 64 * 4);
  hipMemcpy(dl, lg.data(), lg.size() * 4, hipMemcpyHostToDevice); hipMemcpy(du, u.data(), rows * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(64), dim3(512), 0, 0, dl, du, dout, dscan, dmax, rows);
  std::vector<int> out(rows); std::vector<float> scan(rows * 64), mx(rows * 64);
  hipMemcpy(out.data(), dout, rows * 4, hipMemcpyDeviceToHost); hipMemcpy(scan.data(), dscan, rows * 64 * 4, hipMemcpyDeviceToHost);
  hipMemcpy(mx.data(), dmax, rows * 64 * 4, hipMemcpyDeviceToHost);
  int bad_scan = 0, bad_max = 0, bad_pick = 0;
  for (int r = 0; r < rows; ++r) {
    double acc = 0; float m = -1e30f;
    for (int l = 0; l < 64; ++l) m = fmaxf(m, lg[r * 256 + l]);
    for (int l = 0; l < 64; ++l) {
      acc += lg[r * 256 + l];
      if (fabs(scan[r * 64 + l] - acc) > 1e-3) { if (bad_scan < 5) printf("scan row %d lane %d: %f vs %f\n", r, l, scan[r * 64 + l], acc); ++bad_scan; }
      if (mx[r * 64 + l] != m) { if (bad_max < 5) printf("max row %d lane %d: %f vs %f\n", r, l, mx[r * 64 + l], m); ++bad_max; }
    }
    // host inverse CDF
    const double dn = 1.0 / (1.0 + exp(-3.0));
    double mxv = -1e30; for (int c = 0; c < 256; ++c) mxv = fmax(mxv, lg[r * 256 + c] / dn / 0.9);
    std::vector<double> e(256); double tot = 0; for (int c = 0; c < 256; ++c) { e[c] = exp(lg[r * 256 + c] / dn / 0.9 - mxv); tot += e[c]; }
    double run = 0; int pick = -1; for (int c = 0; c < 256; ++c) { run += e[c]; if (run > u[r] * tot) { pick = c; break; } }
    if (pick != out[r]) {
      // tolerate a draw within 1e-5 of a CDF step
      double lo = 0; for (int c = 0; c < out[r]; ++c) lo += e[c];
      const double hi = lo + e[out[r] < 0 ? 0 : out[r]];
      const double tg = u[r] * tot;
      if (!(tg > lo - 1e-5 * tot && tg < hi + 1e-5 * tot)) { if (bad_pick < 8) printf("pick row %d: device %d host %d (u %.6f)\n", r, out[r], pick, u[r]); ++bad_pick; }
    }
  }
  printf("bad scan %d, bad max %d, bad pick %d of %d rows\n", bad_scan, bad_max, bad_pick, rows);
  return (bad_scan || bad_max || bad_pick) ? 1 : 0;
}

// Small device helpers shared by the n_fft = 1024 kernels (istft.hip) and the n_fft = 2048 kernels (spectral2048.hip).
#pragma once
#include "fft1024.h"

namespace mmk {

constexpr int kIstftWaves = 4;

// The tables every n_fft = 1024 / 2048 kernel needs - the twiddles exp(-2 pi i m / 1024), the periodic Hann windows of 1024 and 2048 samples, the
// untangling factors exp(-pi i k / 1024) of the 2048-point transforms - are made ONCE per device by a small kernel (sincospif / cospif: the same
// values the kernels used to compute for themselves) into arrays of the code object, and read from L2 afterwards.  Computed per workgroup they were
// ~3000 vector instructions per wave in front of the ~500 per frame pair of a wave that handles nine pairs: 40 % of the STFT kernel.
struct SpectralTables {
  const cf32* tw1024;        // [1024]
  const float* hann1024;     // [1024]
  const float* hann2048;     // [2048]
  const cf32* w2048;         // [1088]: exp(-pi i k / 1024), k <= 1087 (the bins lane + 64 j of a lane, j <= 16)
};
int spectral_tables(hipStream_t stream, SpectralTables* out);     // (istft.hip)

__device__ __forceinline__ void load_twiddles(cf32* tw, const cf32* __restrict__ table, int tid, int nthreads) {
  for (int m = tid; m < 1024; m += nthreads) tw[m] = table[m];
}

// sin / cos of an angle in radians: three-constant Cody-Waite reduction to |r| <= pi/4 and the cephes single-precision
// kernels (abs error ~1e-7 for |x| < 1e4; no Payne-Hanek path, which costs the library sincosf 300 B of scratch here).
__device__ __forceinline__ void sincos_cw(float x, float* sn, float* cs) {
  const float q = rintf(x * 0.636619772367581343f);
  float r = fmaf(q, -1.5703125f, x);
  r = fmaf(q, -4.837512969970703125e-4f, r);
  r = fmaf(q, -7.54978995489188216e-8f, r);
  const float z = r * r;
  const float sp = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, r, r);
  const float cp = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f), z * z, fmaf(-0.5f, z, 1.0f));
  const int qi = (int)q;
  const float s0 = (qi & 1) ? cp : sp, c0 = (qi & 1) ? sp : cp;
  *sn = (qi & 2) ? -s0 : s0;
  *cs = ((qi + 1) & 2) ? -c0 : c0;
}

// the same through the hardware: v_sin_f32 / v_cos_f32 take revolutions.  The angle is first reduced to |r| <= pi with a two-constant
// Cody-Waite step (2 pi = 6.28125 + 1.9353e-3: q * 6.28125 is exact for |q| < 2^15, so the reduction's error stays at an ulp of r, not of
// x / 2 pi - accumulated or unwrapped phases of thousands of radians keep fp32 accuracy, as torch.exp(1j * angle) does); 7 instructions
// instead of sincos_cw's 22
__device__ __forceinline__ void sincos_hw(float x, float* sn, float* cs) {
  const float q = rintf(x * 0.15915494309189535f);
  float r = fmaf(q, -6.28125f, x);
  r = fmaf(q, -1.9353071795864769e-3f, r);
  const float t = r * 0.15915494309189535f;
  *sn = __builtin_amdgcn_sinf(t);
  *cs = __builtin_amdgcn_cosf(t);
}

template <int MODE>
__device__ __forceinline__ cf32 istft_bin(cf32 c, float m) {
  if (MODE == 1) {                                          // abs * exp(i angle)   (functionals.py:556)
    float sn, cs;
    sincos_hw(c.y, &sn, &cs);      // (sincos_cw, 22 instructions per bin, was a fifth of the ISTFT kernel: 153 -> 127 us for 64 x 862 frames)
    return cf32{c.x * cs, c.x * sn};
  }
  if (MODE == 2) return cf32{m * c.x, m * c.y};
  return c;
}

}  // namespace mmk

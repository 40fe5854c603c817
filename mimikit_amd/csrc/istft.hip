// Inverse STFT, complex STFT and Griffin-Lim (the output side of the Seq2Seq / spectral
// generate path).  n_fft = 1024, periodic Hann, one frame PAIR per wave through the register-resident FFT of fft1024.h.
//
//   ISTFT.torch_func (features/functionals.py:553-564)  = torch.istft(spec^T, n_fft, hop, window=hann)   (center=True)
//   STFT.torch_func  (features/functionals.py:506-523)  = torch.stft(..., return_complex=True) in 'car' / 'pol' / 'angle'
//   GLA.torch_func   (features/functionals.py:634-642)  = torchaudio.transforms.GriffinLim(n_fft, hop, power=1.)
//                                                          (torchaudio 2.0.1 functional.griffinlim: momentum 0.99, 32 its)
//
// All three are HBM-bound: a frame pair is 8 KiB of spectrum in, 8 KiB of windowed frames out (ISTFT), then the overlap-add
// reads every windowed sample once.  Two real inverse transforms share one complex FFT:
//       Z = A + i B  (A, B Hermitian-extended half spectra)   =>   ifft(Z) = a + i b ,   ifft(Z) = conj(fft(conj Z)) / N.
#include "mmk_common.h"
#include "fft1024.h"

namespace mmk {

constexpr int kIstftWaves = 4;

__device__ __forceinline__ void make_twiddles(cf32* tw, int tid, int nthreads) {
  for (int m = tid; m < 1024; m += nthreads) {
    float sn, cs;
    sincospif(-2.0f * (float)m / 1024.0f, &sn, &cs);
    tw[m] = cf32{cs, sn};
  }
}

// ---- spectrum -> windowed frames ------------------------------------------------------------------------------------------
// MODE 0: spec = (batch, frames, 513) complex (re, im);  MODE 1: (mag, angle) pairs;  MODE 2: mag plane x complex plane.
template <int MODE>
__device__ __forceinline__ cf32 load_bin(const float* __restrict__ spec, const float* __restrict__ mag, int64_t frame, int k) {
  const int64_t e = frame * 513 + k;
  const cf32 c = *reinterpret_cast<const cf32*>(spec + 2 * e);
  cf32 z;
  if (MODE == 0) z = c;
  if (MODE == 1) {                                          // mag * exp(i angle)   (functionals.py:556)
    float sn, cs;
    sincosf(c.y, &sn, &cs);
    z = cf32{c.x * cs, c.x * sn};
  }
  if (MODE == 2) {
    const float m = mag[e];
    z = cf32{m * c.x, m * c.y};
  }
  if (k == 0 || k == 512) z.y = 0.f;                        // a real signal's DC / Nyquist bins: the C2R transform ignores them
  return z;
}

template <int MODE>
__global__ __launch_bounds__(64 * kIstftWaves) __attribute__((amdgpu_waves_per_eu(3, 3)))
void istft1024_frames_kernel(const float* __restrict__ spec, const float* __restrict__ mag, int64_t n_frames, int64_t total_pairs,
                             float* __restrict__ frames) {
  constexpr int N = 1024;
  __shared__ cf32 tw[N];
  __shared__ cf32 bufs[kIstftWaves * kFftWaveLds];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  make_twiddles(tw, tid, 64 * kIstftWaves);
  float win[16];                                            // periodic Hann / N at n = lane + 64 r
#pragma unroll
  for (int r = 0; r < 16; ++r) win[r] = (0.5f - 0.5f * cospif(2.0f * (float)(lane + 64 * r) / (float)N)) * (1.0f / (float)N);
  __syncthreads();
  cf32* buf = bufs + wave * kFftWaveLds;
  const int64_t pairs_per_row = (n_frames + 1) >> 1;

  for (int64_t pair = (int64_t)blockIdx.x * kIstftWaves + wave; pair < total_pairs; pair += (int64_t)gridDim.x * kIstftWaves) {
    const int64_t b = pair / pairs_per_row;
    const int64_t f0 = (pair - b * pairs_per_row) * 2;
    const bool has_b = (f0 + 1) < n_frames;
    const int64_t fa = b * n_frames + f0;
    const int64_t fb = has_b ? fa + 1 : fa;                 // clamped: loads stay unconditional
    cf32 v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      // n = lane + 64 r.  n <= 512 reads bin n, n > 512 the conjugate of bin N - n  (lane 0 of r = 8 is bin 512 either way)
      const int k = (r < 8) ? lane + 64 * r : N - (lane + 64 * r);
      const cf32 A = load_bin<MODE>(spec, mag, fa, k);
      cf32 B = load_bin<MODE>(spec, mag, fb, k);
      if (!has_b) B = cf32{0.f, 0.f};
      // Z = A + i B (n <= 512) or conj(A) + i conj(B); the FFT input is conj(Z)
      if (r < 8) v[r] = cf32{A.x - B.y, -(A.y + B.x)};
      else v[r] = cf32{A.x + B.y, -(B.x - A.y)};
    }
    fft1024_wave(v, buf, tw, lane);
    // a[m] = Re(Y[m]) / N , b[m] = -Im(Y[m]) / N ; windowed
    float* oa = frames + fa * N;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const cf32 y = buf[lane + 64 * j];
      oa[lane + 64 * j] = y.x * win[j];
      if (has_b) oa[N + lane + 64 * j] = -y.y * win[j];
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- overlap-add / window envelope (torch.istft: fold, divide by the folded window^2, trim n_fft/2 on both sides) -------
// out[b][n] = sum_f wf[b][f][t - f hop] / sum_f w^2[t - f hop],  t = n + N/2, over the frames that cover t, in frame order.
template <int V>
__global__ __launch_bounds__(256) void istft_ola_kernel(const float* __restrict__ frames, int64_t n_frames, int hop, int64_t n_out,
                                                        int64_t total, float* __restrict__ out) {
  constexpr int N = 1024;
  __shared__ float w2[N];
  for (int m = threadIdx.x; m < N; m += 256) {
    const float w = 0.5f - 0.5f * cospif(2.0f * (float)m / (float)N);
    w2[m] = w * w;
  }
  __syncthreads();
  const int64_t row_v = n_out / V;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t b = e / row_v;
    const int64_t n = (e - b * row_v) * V;
    const int64_t t = n + N / 2;
    int64_t f_hi = (t + V - 1) / hop;
    if (f_hi > n_frames - 1) f_hi = n_frames - 1;
    int64_t f_lo = (t - N + hop) / hop;                     // ceil((t - N + 1) / hop) for t >= N/2 > 0
    if (t - N + 1 <= 0) f_lo = 0;
    float acc[V], env[V];
#pragma unroll
    for (int i = 0; i < V; ++i) acc[i] = 0.f, env[i] = 0.f;
    const float* fr = frames + b * n_frames * N;
    for (int64_t f = f_lo; f <= f_hi; ++f) {
      const int64_t o = t - f * hop;                        // offset of sample n in frame f
      if (V == 4) {                                         // hop % 4 == 0: the four samples share their frames
        if (o < 0 || o >= N) continue;
        const float4 x = *reinterpret_cast<const float4*>(fr + f * N + o);
        acc[0] += x.x, acc[1] += x.y, acc[2] += x.z, acc[3] += x.w;
        env[0] += w2[o], env[1] += w2[o + 1], env[2] += w2[o + 2], env[3] += w2[o + 3];
      } else {
        if (o < 0 || o >= N) continue;
        acc[0] += fr[f * N + o];
        env[0] += w2[o];
      }
    }
    if (V == 4) {
      *reinterpret_cast<float4*>(out + b * n_out + n) = make_float4(acc[0] / env[0], acc[1] / env[1], acc[2] / env[2], acc[3] / env[3]);
    } else {
      out[b * n_out + n] = acc[0] / env[0];
    }
  }
}

// ---- complex STFT (center, constant or reflect padding) with four epilogues ------------------------------------------------
// OUT 0: (re, im)   OUT 1: (|S|, angle S)   OUT 2: angle S   OUT 3: the Griffin-Lim phase update
//   angles = S - m tprev ; angles /= |angles| + 1e-16 ; tprev = S          (torchaudio functional.griffinlim, 2.0.1)
template <int OUT>
__global__ __launch_bounds__(64 * kIstftWaves) __attribute__((amdgpu_waves_per_eu(3, 3)))
void stft1024_complex_kernel(const float* __restrict__ x, int64_t x_row_stride, int64_t n_samples, int hop, int center, int reflect,
                             int64_t n_frames, int64_t total_pairs, float* __restrict__ out, float* __restrict__ tprev, float momentum) {
  constexpr int N = 1024, bins = 513;
  __shared__ cf32 tw[N];
  __shared__ cf32 bufs[kIstftWaves * kFftWaveLds];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  make_twiddles(tw, tid, 64 * kIstftWaves);
  float win[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) win[r] = 0.5f - 0.5f * cospif(2.0f * (float)(lane + 64 * r) / (float)N);
  __syncthreads();
  cf32* buf = bufs + wave * kFftWaveLds;
  const int64_t pairs_per_row = (n_frames + 1) >> 1;
  const int64_t pad = center ? N / 2 : 0;

  for (int64_t pair = (int64_t)blockIdx.x * kIstftWaves + wave; pair < total_pairs; pair += (int64_t)gridDim.x * kIstftWaves) {
    const int64_t b = pair / pairs_per_row;
    const int64_t f0 = (pair - b * pairs_per_row) * 2;
    const bool has_b = (f0 + 1) < n_frames;
    const float* xr = x + b * x_row_stride;
    cf32 v[16];
    const int64_t start = f0 * hop - pad;
    if (start >= 0 && start + hop + N <= n_samples && has_b) {
      const float* pa = xr + start + lane;
      const float* pb = pa + hop;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        v[r].x = pa[64 * r] * win[r];
        v[r].y = pb[64 * r] * win[r];
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t ia = start + lane + 64 * r, ib = ia + hop;
        const bool ina = ia >= 0 && ia < n_samples, inb = ib >= 0 && ib < n_samples;
        int64_t ja = ia, jb = ib;
        if (reflect) {                                      // torch 'reflect': no repeat of the edge sample
          ja = ia < 0 ? -ia : (ia >= n_samples ? 2 * (n_samples - 1) - ia : ia);
          jb = ib < 0 ? -ib : (ib >= n_samples ? 2 * (n_samples - 1) - ib : ib);
        }
        ja = ja < 0 ? 0 : (ja >= n_samples ? n_samples - 1 : ja);
        jb = jb < 0 ? 0 : (jb >= n_samples ? n_samples - 1 : jb);
        const float a = xr[ja], bb = xr[jb];
        v[r].x = (reflect || ina) ? a * win[r] : 0.f;
        v[r].y = (has_b && (reflect || inb)) ? bb * win[r] : 0.f;
      }
    }
    fft1024_wave(v, buf, tw, lane);
    // A[k] = (Z[k] + conj(Z[N-k])) / 2 ,  B[k] = (Z[k] - conj(Z[N-k])) / (2i)
    const int64_t ea = (b * n_frames + f0) * bins;
#pragma unroll
    for (int jj = 0; jj < 9; ++jj) {
      const int k = lane + 64 * jj;
      if (k < bins) {
        const cf32 z = buf[k];
        const cf32 zc = buf[(N - k) & (N - 1)];
        cf32 s[2];
        s[0] = cf32{0.5f * (z.x + zc.x), 0.5f * (z.y - zc.y)};
        s[1] = cf32{0.5f * (z.y + zc.y), -0.5f * (z.x - zc.x)};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          if (q == 1 && !has_b) break;
          const int64_t e = ea + q * bins + k;
          if (OUT == 0) *reinterpret_cast<cf32*>(out + 2 * e) = s[q];
          if (OUT == 1) *reinterpret_cast<cf32*>(out + 2 * e) = cf32{sqrtf(s[q].x * s[q].x + s[q].y * s[q].y), atan2f(s[q].y, s[q].x)};
          if (OUT == 2) out[e] = atan2f(s[q].y, s[q].x);
          if (OUT == 3) {
            const cf32 tp = *reinterpret_cast<const cf32*>(tprev + 2 * e);
            const cf32 g = cf32{s[q].x - momentum * tp.x, s[q].y - momentum * tp.y};
            const float d = sqrtf(g.x * g.x + g.y * g.y) + 1e-16f;
            *reinterpret_cast<cf32*>(out + 2 * e) = cf32{g.x / d, g.y / d};
            *reinterpret_cast<cf32*>(tprev + 2 * e) = s[q];
          }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

__global__ void fill_complex_kernel(float* __restrict__ dst, int64_t n, cf32 v) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
    *reinterpret_cast<cf32*>(dst + 2 * e) = v;
}

static int check_1024(const char* what, int n_fft, int hop) {
  if (n_fft != 1024) return fail(MMK_ERR_UNSUPPORTED, "%s: n_fft must be 1024 in this build, got %d", what, n_fft);
  if (hop <= 0 || hop >= n_fft) return fail(MMK_ERR_INVALID, "%s: hop must be in [1, n_fft), got %d", what, hop);
  return MMK_OK;
}

static unsigned pair_grid(int64_t total_pairs) {
  const int64_t wgs = (total_pairs + kIstftWaves - 1) / kIstftWaves;
  return (unsigned)(wgs < 768 ? wgs : 768);                 // 3 workgroups of 4 waves per CU, all resident
}

static int launch_istft(const float* spec, const float* mag, int mode, int batch, int64_t n_frames, int hop, float* work, float* out,
                        hipStream_t stream) {
  const int64_t total_pairs = (int64_t)batch * ((n_frames + 1) / 2);
  const dim3 grid(pair_grid(total_pairs)), block(64 * kIstftWaves);
  if (mode == 0) hipLaunchKernelGGL((istft1024_frames_kernel<0>), grid, block, 0, stream, spec, mag, n_frames, total_pairs, work);
  else if (mode == 1) hipLaunchKernelGGL((istft1024_frames_kernel<1>), grid, block, 0, stream, spec, mag, n_frames, total_pairs, work);
  else hipLaunchKernelGGL((istft1024_frames_kernel<2>), grid, block, 0, stream, spec, mag, n_frames, total_pairs, work);
  MMK_HIP(hipGetLastError());
  const int64_t n_out = (int64_t)hop * (n_frames - 1);
  if (n_out <= 0) return MMK_OK;
  const bool v4 = (hop % 4) == 0;
  const int64_t total = (int64_t)batch * (v4 ? n_out / 4 : n_out);
  int64_t blocks = (total + 255) / 256;
  blocks = blocks > 4096 ? 4096 : blocks;
  if (v4) hipLaunchKernelGGL((istft_ola_kernel<4>), dim3((unsigned)blocks), dim3(256), 0, stream, work, n_frames, hop, n_out, total, out);
  else hipLaunchKernelGGL((istft_ola_kernel<1>), dim3((unsigned)blocks), dim3(256), 0, stream, work, n_frames, hop, n_out, total, out);
  MMK_HIP(hipGetLastError());
  return MMK_OK;
}

static int launch_stft_complex(const float* x, int64_t x_row_stride, int batch, int64_t n_samples, int hop, int center, int reflect,
                               int out_mode, float* out, float* tprev, float momentum, hipStream_t stream) {
  const int64_t n_frames = mmk_stft_n_frames(n_samples, 1024, hop, center);
  const int64_t total_pairs = (int64_t)batch * ((n_frames + 1) / 2);
  const dim3 grid(pair_grid(total_pairs)), block(64 * kIstftWaves);
#define MMK_STFT_LAUNCH(O) \
  hipLaunchKernelGGL((stft1024_complex_kernel<O>), grid, block, 0, stream, x, x_row_stride, n_samples, hop, center, reflect, n_frames, \
                     total_pairs, out, tprev, momentum)
  switch (out_mode) {
    case 0: MMK_STFT_LAUNCH(0); break;
    case 1: MMK_STFT_LAUNCH(1); break;
    case 2: MMK_STFT_LAUNCH(2); break;
    default: MMK_STFT_LAUNCH(3); break;
  }
#undef MMK_STFT_LAUNCH
  MMK_HIP(hipGetLastError());
  return MMK_OK;
}

}  // namespace mmk

extern "C" int mmk_stft_f32(const float* x, int64_t x_row_stride, int32_t batch, int64_t n_samples, int32_t n_fft, int32_t hop,
                            int32_t center, int32_t reflect, int32_t coordinate, float* out, mmk_stream_t stream) {
  using namespace mmk;
  if (!x || !out || batch <= 0) return fail(MMK_ERR_INVALID, "stft: bad arguments");
  if (int rc = check_1024("stft", n_fft, hop)) return rc;
  if (coordinate < 0 || coordinate > 2) return fail(MMK_ERR_INVALID, "stft: coordinate must be 0 (car), 1 (pol) or 2 (angle)");
  if (mmk_stft_n_frames(n_samples, n_fft, hop, center) <= 0)
    return fail(MMK_ERR_INVALID, "stft: input of %lld samples is shorter than one frame", (long long)n_samples);
  if (reflect && center && n_samples <= n_fft / 2)
    return fail(MMK_ERR_INVALID, "stft: reflect padding of %d needs more than %d samples, got %lld", n_fft / 2, n_fft / 2, (long long)n_samples);
  return launch_stft_complex(x, x_row_stride, batch, n_samples, hop, center, reflect, coordinate, out, nullptr, 0.f, (hipStream_t)stream);
}

extern "C" int64_t mmk_istft_n_samples(int64_t n_frames, int32_t n_fft, int32_t hop) {
  (void)n_fft;
  return n_frames > 0 ? (int64_t)hop * (n_frames - 1) : 0;
}

extern "C" size_t mmk_istft_workspace_floats(int32_t batch, int64_t n_frames, int32_t n_fft) {
  return (size_t)batch * (size_t)n_frames * (size_t)n_fft;
}

extern "C" int mmk_istft_f32(const float* spec, int32_t coordinate, int32_t batch, int64_t n_frames, int32_t n_fft, int32_t hop,
                             float* work, float* out, mmk_stream_t stream) {
  using namespace mmk;
  if (!spec || !work || !out || batch <= 0 || n_frames <= 0) return fail(MMK_ERR_INVALID, "istft: bad arguments");
  if (int rc = check_1024("istft", n_fft, hop)) return rc;
  if (coordinate != 0 && coordinate != 1) return fail(MMK_ERR_INVALID, "istft: coordinate must be 0 (re, im) or 1 (mag, angle)");
  if (n_frames < 2) return fail(MMK_ERR_INVALID, "istft: one frame leaves no samples after the centre trim");
  return launch_istft(spec, nullptr, coordinate, batch, n_frames, hop, work, out, (hipStream_t)stream);
}

extern "C" size_t mmk_gla_workspace_floats(int32_t batch, int64_t n_frames, int32_t n_fft, int32_t hop) {
  const size_t bins = (size_t)n_fft / 2 + 1;
  return (size_t)batch * ((size_t)n_frames * n_fft                      // windowed frames
                          + (size_t)hop * (size_t)(n_frames > 0 ? n_frames - 1 : 0)   // the current waveform
                          + 4 * (size_t)n_frames * bins);               // angles, previous rebuilt spectrum (complex)
}

extern "C" int mmk_gla_f32(const float* mag, const float* init, int32_t batch, int64_t n_frames, int32_t n_fft, int32_t hop, int32_t n_iter,
                           float momentum, float* work, float* out, mmk_stream_t stream) {
  using namespace mmk;
  if (!mag || !work || !out || batch <= 0 || n_frames <= 0 || n_iter < 0) return fail(MMK_ERR_INVALID, "gla: bad arguments");
  if (int rc = check_1024("gla", n_fft, hop)) return rc;
  if (!(momentum >= 0.f && momentum < 1.f)) return fail(MMK_ERR_INVALID, "gla: momentum must be in [0, 1), got %g", (double)momentum);
  const int64_t n_out = (int64_t)hop * (n_frames - 1);
  if (n_out <= n_fft / 2) return fail(MMK_ERR_INVALID, "gla: %lld frames give %lld samples, reflect padding needs more than %d",
                                      (long long)n_frames, (long long)n_out, n_fft / 2);
  hipStream_t s = (hipStream_t)stream;
  const size_t bins = 513;
  float* frames = work;
  float* wave = frames + (size_t)batch * n_frames * n_fft;
  float* angles = wave + (size_t)batch * n_out;
  float* tprev = angles + 2 * (size_t)batch * n_frames * bins;
  const size_t spec_bytes = 2 * (size_t)batch * n_frames * bins * sizeof(float);
  if (init) MMK_HIP(hipMemcpyAsync(angles, init, spec_bytes, hipMemcpyDeviceToDevice, s));
  else {                                                     // rand_init=False: every phase estimate starts at 1 + 0i
    const int64_t n = (int64_t)batch * n_frames * (int64_t)bins;
    hipLaunchKernelGGL(fill_complex_kernel, dim3((unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256)), dim3(256), 0, s, angles, n,
                       cf32{1.f, 0.f});
    MMK_HIP(hipGetLastError());
  }
  MMK_HIP(hipMemsetAsync(tprev, 0, spec_bytes, s));
  const float m = momentum / (1.f + momentum);
  for (int it = 0; it < n_iter; ++it) {
    if (int rc = launch_istft(angles, mag, 2, batch, n_frames, hop, frames, wave, s)) return rc;
    if (int rc = launch_stft_complex(wave, n_out, batch, n_out, hop, 1, 1, 3, angles, tprev, m, s)) return rc;
  }
  return launch_istft(angles, mag, 2, batch, n_frames, hop, frames, out, s);
}

// STFT, inverse STFT and Griffin-Lim for n_fft = 2048 (the reference's default N_FFT / HOP_LENGTH = 2048 / 512,
// features/functionals.py:23-24) on the register-resident 1024-point complex FFT of fft1024.h.
//
// A real frame of 2N = 2048 samples is ONE complex transform of N = 1024 points plus an untangling pass:
//       z[n] = x[2n] + i x[2n+1]                       Z = FFT_N(z)
//       E[k] = (Z[k] + conj Z[N-k]) / 2                (spectrum of the even samples)
//       O[k] = (Z[k] - conj Z[N-k]) / (2i)             (spectrum of the odd samples)
//       X[k] = E[k] + W^k O[k] ,  W = exp(-2 pi i / 2N) ,  k = 0 .. N        (Z[N] = Z[0])
// and backwards  E[k] = (X[k] + conj X[N-k]) / 2 ,  O[k] = conj(W^k) (X[k] - conj X[N-k]) / 2 ,  Z = E + i O ,
// z = IFFT_N(Z) = conj(FFT_N(conj Z)) / N.  So a wave that handles two 1024-sample frames per transform in istft.hip
// handles one 2048-sample frame here: same bytes, same arithmetic, one more twiddle per bin.  Kernel structure (one
// frame per wave; output segments with a wave-private overlap-add ring; one launch per Griffin-Lim iteration) is that of
// istft.hip; the workgroup-FFT kernels there served this size at a third of the speed.
#include "mmk_common.h"
#include "fft1024.h"
#include "spectral_util.h"

namespace mmk {

constexpr int kN = 1024, kN2 = 2048, kBins2 = 1025;
constexpr int kRing2 = 4096;                                // live window of a frame: n_fft + hop <= 4095 samples

// (the periodic Hann window of 2048 samples and W^k for the bins a lane touches: tables of spectral_util.h, T.hann2048 / T.w2048)

// the 2048 samples of a frame as z[n] = (x[2n], x[2n+1]), n = lane + 64 r; zero or reflect padding at the clip's ends
__device__ __forceinline__ void load_frame2048(cf32 (&v)[16], const float* __restrict__ xr, int64_t start, int64_t n_samples, int reflect,
                                               int lane) {
  if (start >= 0 && start + kN2 <= n_samples) {
    const float* p = xr + start + 2 * lane;
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = cf32{p[128 * r], p[128 * r + 1]};
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float s[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int64_t i = start + 2 * (lane + 64 * r) + h;
        bool in = i >= 0 && i < n_samples;
        if (reflect) {                                      // torch 'reflect': no repeat of the edge sample
          i = i < 0 ? -i : (i >= n_samples ? 2 * (n_samples - 1) - i : i);
          in = i >= 0 && i < n_samples;
        }
        const int64_t j = i < 0 ? 0 : (i >= n_samples ? n_samples - 1 : i);      // unconditional load, clamped address
        const float x = xr[j];
        s[h] = in ? x : 0.f;
      }
      v[r] = cf32{s[0], s[1]};
    }
  }
}

// X[k] of the frame whose Z sits in buf (natural order)
__device__ __forceinline__ cf32 untangle(const cf32* buf, int k, cf32 w) {
  const cf32 z = buf[k & (kN - 1)];
  const cf32 zc = buf[(kN - k) & (kN - 1)];
  const cf32 e = cf32{0.5f * (z.x + zc.x), 0.5f * (z.y - zc.y)};
  const cf32 o = cf32{0.5f * (z.y + zc.y), -0.5f * (z.x - zc.x)};
  return cf32{e.x + (w.x * o.x - w.y * o.y), e.y + (w.x * o.y + w.y * o.x)};
}

// conj(Z[n]) from X[n] (a) and X[N - n] (b):  the FFT input of the inverse transform
__device__ __forceinline__ cf32 tangle_conj(cf32 a, cf32 b, cf32 w) {
  const cf32 e = cf32{0.5f * (a.x + b.x), 0.5f * (a.y - b.y)};
  const cf32 d = cf32{0.5f * (a.x - b.x), 0.5f * (a.y + b.y)};
  const cf32 o = cf32{w.x * d.x + w.y * d.y, w.x * d.y - w.y * d.x};       // conj(W^n) d
  return cf32{e.x - o.y, -(e.y + o.x)};                                   // conj(E + i O)
}

// ---- STFT -----------------------------------------------------------------------------------------------------------------
// OUT 0: (re, im)   OUT 1: (|S|, angle S)   OUT 2: angle S   OUT 4: |S|  (MagSpec)
template <int OUT>
__global__ __launch_bounds__(64 * kIstftWaves) __attribute__((amdgpu_waves_per_eu(2, 2)))
void stft2048_kernel(const SpectralTables T, const float* __restrict__ x, int64_t x_row_stride, int64_t n_samples, int hop, int center, int reflect,
                     int64_t n_frames, int64_t total_frames, float* __restrict__ out) {
  __shared__ cf32 tw[kN];
  __shared__ cf32 bufs[kIstftWaves * kFftWaveLds];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  load_twiddles(tw, T.tw1024, tid, 64 * kIstftWaves);
  cf32 win[16];                                             // periodic Hann of 2048 at samples 2n, 2n + 1
#pragma unroll
  for (int r = 0; r < 16; ++r) win[r] = cf32{T.hann2048[2 * (lane + 64 * r)], T.hann2048[2 * (lane + 64 * r) + 1]};
  cf32 wk[17];                                              // W^k of this lane's bins k = lane + 64 j
#pragma unroll
  for (int j = 0; j < 17; ++j) wk[j] = T.w2048[lane + 64 * j];
  __syncthreads();
  cf32* buf = bufs + wave * kFftWaveLds;
  const int64_t pad = center ? kN2 / 2 : 0;
  for (int64_t fr = (int64_t)blockIdx.x * kIstftWaves + wave; fr < total_frames; fr += (int64_t)gridDim.x * kIstftWaves) {
    const int64_t b = fr / n_frames, f = fr - b * n_frames;
    cf32 v[16];
    load_frame2048(v, x + b * x_row_stride, f * hop - pad, n_samples, reflect, lane);
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = v[r] * win[r];
    fft1024_wave(v, buf, tw, lane);
    const int64_t e0 = fr * kBins2;
#pragma unroll
    for (int j = 0; j < 17; ++j) {
      const int k = lane + 64 * j;
      if (k < kBins2) {
        const cf32 s = untangle(buf, k, wk[j]);
        const int64_t e = e0 + k;
        if (OUT == 0) *reinterpret_cast<cf32*>(out + 2 * e) = s;
        if (OUT == 1) *reinterpret_cast<cf32*>(out + 2 * e) = cf32{sqrtf(s.x * s.x + s.y * s.y), atan2f(s.y, s.x)};
        if (OUT == 2) out[e] = atan2f(s.y, s.x);
        if (OUT == 4) out[e] = sqrtf(s.x * s.x + s.y * s.y);
      }
    }
    __builtin_amdgcn_wave_barrier();                        // buf is rewritten by the next frame
  }
}

// ---- overlap-add of one transformed frame + emission of what is final ---------------------------------------------------------
// buf holds Y = FFT(conj Z): x[2m] = Re Y[m] / N, x[2m + 1] = -Im Y[m] / N; win = the window of those samples / N
__device__ __forceinline__ void ola_frame2048(float* ring, const cf32* buf, const cf32 (&win)[16], const float* envt, const float* __restrict__ hann, int64_t f, int hop,
                                              int64_t frontier, int64_t upto, int64_t t0, int64_t t1, int64_t n_frames,
                                              float* __restrict__ orow, int lane) {
  const int pa = (int)((f * hop) & (kRing2 - 1));
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int q = (pa + 2 * (lane + 64 * j)) & (kRing2 - 1);
    const cf32 y = buf[lane + 64 * j];
    ring[q] += y.x * win[j].x;
    ring[(q + 1) & (kRing2 - 1)] -= y.y * win[j].y;
  }
  __builtin_amdgcn_wave_barrier();
  // frontier = f hop and upto - frontier <= hop: t mod hop is the offset itself
  const int64_t t_int_lo = kN2 - 1, t_int_hi = n_frames * hop;              // inside: every frame that covers t exists
  for (int xo = lane; frontier + xo < upto; xo += 64) {
    const int64_t t = frontier + xo;
    const int q = (int)(t & (kRing2 - 1));
    const float acc = ring[q];
    ring[q] = 0.f;
    if (t >= t0 && t < t1) {
      float env;
      if (t >= t_int_lo && t < t_int_hi) {
        env = envt[xo];
      } else {                                              // the first / last n_fft samples of a clip
        int64_t g_hi = t / hop;
        g_hi = g_hi < n_frames - 1 ? g_hi : n_frames - 1;
        const int64_t g_lo = (t - kN2 + 1 <= 0) ? 0 : (t - kN2 + hop) / hop;
        env = 0.f;
        for (int64_t g = g_lo; g <= g_hi; ++g) {
          const float w = hann[t - g * hop];
          env += w * w;
        }
      }
      orow[t] = acc / env;
    }
  }
}

__device__ __forceinline__ void envelope_table2048(float* envt, const float* __restrict__ hann, int hop, int tid, int nthreads) {
  for (int r = tid; r < hop; r += nthreads) {
    float e = 0.f;
    for (int o = r; o < kN2; o += hop) {
      const float w = hann[o];
      e += w * w;
    }
    envt[r] = e;
  }
}

struct Seg2048 { int64_t b, t0, t1, f_lo, f_hi; };
__device__ __forceinline__ bool segment2048(Seg2048& s, int64_t task, int segs_per_clip, int seg_hops, int hop, int64_t n_frames,
                                            int64_t n_out) {
  s.b = task / segs_per_clip;
  const int64_t sgm = task - s.b * segs_per_clip;
  const int64_t t_end = kN2 / 2 + n_out;                    // positions t are in the untrimmed overlap-add signal
  s.t0 = kN2 / 2 + sgm * seg_hops * hop;
  s.t1 = s.t0 + (int64_t)seg_hops * hop;
  s.t1 = s.t1 < t_end ? s.t1 : t_end;
  s.f_lo = (s.t0 - kN2 + 1 <= 0) ? 0 : (s.t0 - kN2 + hop) / hop;              // first frame that covers t0
  s.f_hi = (s.t1 - 1) / hop;                                                  // last frame that covers t1 - 1
  s.f_hi = s.f_hi < n_frames - 1 ? s.f_hi : n_frames - 1;
  return s.t0 < s.t1;
}

// ---- inverse STFT, overlap-add fused ---------------------------------------------------------------------------------------------
// MODE 0: spec = (batch, frames, 1025) complex (re, im);  MODE 1: (abs, angle) pairs;  MODE 2: mag plane x complex plane.
template <int MODE>
__global__ __launch_bounds__(64 * kIstftWaves) __attribute__((amdgpu_waves_per_eu(1, 1)))
void istft2048_kernel(const SpectralTables T, const float* __restrict__ spec, const float* __restrict__ mag, int64_t n_frames, int hop, int seg_hops,
                      int segs_per_clip, int64_t total_tasks, int64_t n_out, float* __restrict__ out) {
  __shared__ cf32 tw[kN];
  __shared__ float envt[kN2];
  __shared__ cf32 bufs[kIstftWaves * kFftWaveLds];
  __shared__ float rings[kIstftWaves * kRing2];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  load_twiddles(tw, T.tw1024, tid, 64 * kIstftWaves);
  envelope_table2048(envt, T.hann2048, hop, tid, 64 * kIstftWaves);
  cf32 win[16], wn[16];                                     // window / N at samples 2n, 2n + 1; W^n, n = lane + 64 r
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int n = lane + 64 * r;
    win[r] = cf32{T.hann2048[2 * n] * (1.0f / kN), T.hann2048[2 * n + 1] * (1.0f / kN)};
    wn[r] = T.w2048[n];
  }
  __syncthreads();
  cf32* buf = bufs + wave * kFftWaveLds;
  float* ring = rings + wave * kRing2;

  for (int64_t task = (int64_t)blockIdx.x * kIstftWaves + wave; task < total_tasks; task += (int64_t)gridDim.x * kIstftWaves) {
    Seg2048 sg;
    if (!segment2048(sg, task, segs_per_clip, seg_hops, hop, n_frames, n_out)) continue;
    float* orow = out + sg.b * n_out - kN2 / 2;
#pragma unroll
    for (int j = 0; j < kRing2 / 64; ++j) ring[lane + 64 * j] = 0.f;
    int64_t frontier = sg.f_lo * hop;
    cf32 ra[16], rb[16];
    float ma[16], mb[16];
    auto load_bins = [&](int64_t f) {                       // X[n] and X[N - n] of frame f, n = lane + 64 r
      const int64_t e0 = (sg.b * n_frames + f) * kBins2;
      const cf32* sa = reinterpret_cast<const cf32*>(spec) + e0 + lane;
      const cf32* sb = reinterpret_cast<const cf32*>(spec) + e0 + (kN - lane);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        ra[r] = sa[64 * r];
        rb[r] = sb[-64 * r];
      }
      if (MODE == 2) {
        const float* pa = mag + e0 + lane;
        const float* pb = mag + e0 + (kN - lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          ma[r] = pa[64 * r];
          mb[r] = pb[-64 * r];
        }
      }
    };
    load_bins(sg.f_lo);
    for (int64_t f = sg.f_lo; f <= sg.f_hi; ++f) {
      cf32 v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        cf32 A = istft_bin<MODE>(ra[r], ma[r]);
        cf32 B = istft_bin<MODE>(rb[r], mb[r]);
        if (r == 0 && lane == 0) A.y = 0.f, B.y = 0.f;       // bins 0 and 1024 of a real signal: the C2R transform ignores them
        v[r] = tangle_conj(A, B, wn[r]);
      }
      load_bins(f + 1 <= sg.f_hi ? f + 1 : sg.f_hi);         // next frame's bins: in flight under this transform
      fft1024_wave(v, buf, tw, lane);
      const int64_t upto = f + 1 <= sg.f_hi ? (f + 1) * hop : sg.t1;
      ola_frame2048(ring, buf, win, envt, T.hann2048, f, hop, frontier, upto, sg.t0, sg.t1, n_frames, orow, lane);
      frontier = upto;
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// ---- one whole Griffin-Lim iteration: stft -> phase update -> istft, per output segment ---------------------------------------
//   rebuilt = stft(wave_in) ; angles = normalise(rebuilt - m tprev_in) ; tprev_out = rebuilt ; wave_out = istft(mag angles)
__global__ __launch_bounds__(64 * kIstftWaves) __attribute__((amdgpu_waves_per_eu(1, 1)))
void gla2048_iter_kernel(const SpectralTables T, const float* __restrict__ wave_in, const float* __restrict__ mag, const float* __restrict__ tprev_in,
                         float* __restrict__ tprev_out, float momentum, int64_t n_frames, int hop, int seg_hops, int segs_per_clip,
                         int64_t total_tasks, int64_t n_out, float* __restrict__ wave_out) {
  __shared__ cf32 tw[kN];
  __shared__ float envt[kN2];
  __shared__ cf32 bufs[kIstftWaves * kFftWaveLds];
  __shared__ float rings[kIstftWaves * kRing2];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  load_twiddles(tw, T.tw1024, tid, 64 * kIstftWaves);
  envelope_table2048(envt, T.hann2048, hop, tid, 64 * kIstftWaves);
  cf32 win[16], win_n[16], wn[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int n = lane + 64 * r;
    win[r] = cf32{T.hann2048[2 * n], T.hann2048[2 * n + 1]};
    win_n[r] = win[r] * (1.0f / kN);
    wn[r] = T.w2048[n];                                       // W^k for k = lane + 64 j, j < 16, as well
  }
  const cf32 w_last = T.w2048[1024];                           // bin 1024 (lane 0 only)
  __syncthreads();
  cf32* buf = bufs + wave * kFftWaveLds;
  float* ring = rings + wave * kRing2;

  for (int64_t task = (int64_t)blockIdx.x * kIstftWaves + wave; task < total_tasks; task += (int64_t)gridDim.x * kIstftWaves) {
    Seg2048 sg;
    if (!segment2048(sg, task, segs_per_clip, seg_hops, hop, n_frames, n_out)) continue;
    const float* xr = wave_in + sg.b * n_out;
    float* orow = wave_out + sg.b * n_out - kN2 / 2;
#pragma unroll
    for (int j = 0; j < kRing2 / 64; ++j) ring[lane + 64 * j] = 0.f;
    int64_t frontier = sg.f_lo * hop;
    for (int64_t f = sg.f_lo; f <= sg.f_hi; ++f) {
      const int64_t e0 = (sg.b * n_frames + f) * kBins2;
      // ---- forward: frame f of the current waveform (center, reflect) ---------------------------------------------------------
      cf32 v[16];
      load_frame2048(v, xr, f * hop - kN2 / 2, n_out, 1, lane);
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = v[r] * win[r];
      // previous spectrum and magnitudes of the bins this lane updates: in flight under the transform
      cf32 tp[17];
      float mg[17];
#pragma unroll
      for (int j = 0; j < 17; ++j) {
        const int k = lane + 64 * j;
        const int kc = k < kBins2 ? k : 0;                  // clamped: unconditional loads
        tp[j] = *reinterpret_cast<const cf32*>(tprev_in + 2 * (e0 + kc));
        mg[j] = mag[e0 + kc];
      }
      fft1024_wave(v, buf, tw, lane);
      // ---- phase update per bin; the new spectrum mag * angles goes back to LDS (bins 0 .. 1024) -------------------------------
      cf32 zn[17];
#pragma unroll
      for (int j = 0; j < 17; ++j) {
        const int k = lane + 64 * j;
        if (k < kBins2) {
          const cf32 s = untangle(buf, k, j < 16 ? wn[j] : w_last);
          const cf32 g = cf32{s.x - momentum * tp[j].x, s.y - momentum * tp[j].y};
          const float d = sqrtf(g.x * g.x + g.y * g.y) + 1e-16f;
          zn[j] = cf32{mg[j] * (g.x / d), mg[j] * (g.y / d)};
          if (k == 0 || k == kN) zn[j].y = 0.f;             // the C2R transform ignores them
          *reinterpret_cast<cf32*>(tprev_out + 2 * (e0 + k)) = s;
        }
      }
      __builtin_amdgcn_wave_barrier();                      // every lane has read its bins of the forward transform
#pragma unroll
      for (int j = 0; j < 17; ++j) {
        const int k = lane + 64 * j;
        if (k < kBins2) buf[k] = zn[j];
      }
      __builtin_amdgcn_wave_barrier();
      // ---- inverse ------------------------------------------------------------------------------------------------------------------
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = lane + 64 * r;
        const cf32 A = buf[n], B = buf[kN - n];
        v[r] = tangle_conj(A, B, wn[r]);
      }
      __builtin_amdgcn_wave_barrier();
      fft1024_wave(v, buf, tw, lane);
      const int64_t upto = f + 1 <= sg.f_hi ? (f + 1) * hop : sg.t1;
      ola_frame2048(ring, buf, win_n, envt, T.hann2048, f, hop, frontier, upto, sg.t0, sg.t1, n_frames, orow, lane);
      frontier = upto;
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// ---- launchers --------------------------------------------------------------------------------------------------------------------
static void geometry2048(int batch, int64_t n_frames, int* seg_hops, int* segs_per_clip) {
  const int64_t hops = n_frames - 1;                        // output hops per clip
  const int64_t slots = 1024;                               // one workgroup of 4 waves per CU
  const int64_t rounds = ((int64_t)batch * hops + slots * 48 - 1) / (slots * 48);
  int64_t per_clip = (slots * rounds + batch - 1) / batch;
  per_clip = per_clip < 1 ? 1 : per_clip;
  int64_t sh = (hops + per_clip - 1) / per_clip;
  sh = sh < 4 ? 4 : sh;
  sh = sh > hops ? hops : sh;
  *seg_hops = (int)sh;
  *segs_per_clip = (int)((hops + sh - 1) / sh);
}

int launch_stft2048(const float* x, int64_t x_row_stride, int batch, int64_t n_samples, int hop, int center, int reflect, int out_mode,
                    float* out, hipStream_t stream) {
  const int64_t n_frames = mmk_stft_n_frames(n_samples, kN2, hop, center);
  const int64_t total = (int64_t)batch * n_frames;
  const int64_t wgs = (total + kIstftWaves - 1) / kIstftWaves;
  const dim3 grid((unsigned)(wgs < 512 ? wgs : 512)), block(64 * kIstftWaves);   // 2 workgroups per CU (200 registers)
  SpectralTables T;
  MMK_TRY(spectral_tables(stream, &T));
#define MMK_STFT_LAUNCH(O) \
  hipLaunchKernelGGL((stft2048_kernel<O>), grid, block, 0, stream, T, x, x_row_stride, n_samples, hop, center, reflect, n_frames, total, out)
  switch (out_mode) {
    case 0: MMK_STFT_LAUNCH(0); break;
    case 1: MMK_STFT_LAUNCH(1); break;
    case 2: MMK_STFT_LAUNCH(2); break;
    default: MMK_STFT_LAUNCH(4); break;
  }
#undef MMK_STFT_LAUNCH
  MMK_HIP(hipGetLastError());
  return MMK_OK;
}

int launch_istft2048(const float* spec, const float* mag, int mode, int batch, int64_t n_frames, int hop, float* out, hipStream_t stream) {
  const int64_t n_out = (int64_t)hop * (n_frames - 1);
  if (n_out <= 0) return MMK_OK;
  int seg_hops, segs_per_clip;
  geometry2048(batch, n_frames, &seg_hops, &segs_per_clip);
  const int64_t total_tasks = (int64_t)batch * segs_per_clip;
  const int64_t wgs = (total_tasks + kIstftWaves - 1) / kIstftWaves;
  const dim3 grid((unsigned)(wgs < 256 ? wgs : 256)), block(64 * kIstftWaves);
  SpectralTables T;
  MMK_TRY(spectral_tables(stream, &T));
#define MMK_ISTFT_LAUNCH(M) \
  hipLaunchKernelGGL((istft2048_kernel<M>), grid, block, 0, stream, T, spec, mag, n_frames, hop, seg_hops, segs_per_clip, total_tasks, n_out, out)
  if (mode == 0) MMK_ISTFT_LAUNCH(0);
  else if (mode == 1) MMK_ISTFT_LAUNCH(1);
  else MMK_ISTFT_LAUNCH(2);
#undef MMK_ISTFT_LAUNCH
  MMK_HIP(hipGetLastError());
  return MMK_OK;
}

int launch_gla2048_iter(const float* wave_in, const float* mag, const float* tprev_in, float* tprev_out, float momentum, int batch,
                        int64_t n_frames, int hop, float* wave_out, hipStream_t stream) {
  const int64_t n_out = (int64_t)hop * (n_frames - 1);
  int seg_hops, segs_per_clip;
  geometry2048(batch, n_frames, &seg_hops, &segs_per_clip);
  const int64_t total_tasks = (int64_t)batch * segs_per_clip;
  const int64_t wgs = (total_tasks + kIstftWaves - 1) / kIstftWaves;
  const dim3 grid((unsigned)(wgs < 256 ? wgs : 256)), block(64 * kIstftWaves);
  SpectralTables T;
  MMK_TRY(spectral_tables(stream, &T));
  hipLaunchKernelGGL(gla2048_iter_kernel, grid, block, 0, stream, T, wave_in, mag, tprev_in, tprev_out, momentum, n_frames, hop, seg_hops,
                     segs_per_clip, total_tasks, n_out, wave_out);
  MMK_HIP(hipGetLastError());
  return MMK_OK;
}

}  // namespace mmk

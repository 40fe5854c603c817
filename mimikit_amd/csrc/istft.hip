// STFT, inverse STFT and Griffin-Lim (the output side of the Seq2Seq / spectral generate
// path).  Periodic Hann.  n_fft = 1024: one frame PAIR per wave through the register-resident FFT of fft1024.h; any other
// power of two in [64, 4096]: one pair per workgroup, Stockham passes through LDS.
//
//   ISTFT.torch_func (features/functionals.py:553-564)  = torch.istft(spec^T, n_fft, hop, window=hann)   (center=True)
//   STFT.torch_func  (features/functionals.py:506-523)  = torch.stft(..., return_complex=True) in 'car' / 'pol' / 'angle'
//   GLA.torch_func   (features/functionals.py:634-642)  = torchaudio.transforms.GriffinLim(n_fft, hop, power=1.)
//                                                          (torchaudio 2.0.1 functional.griffinlim: momentum 0.99, 32 its)
//
// All three are HBM-bound: a frame pair is 8 KiB of spectrum in, 8 KiB of windowed frames out (ISTFT), then the overlap-add
// reads every windowed sample once.  Two real inverse transforms share one complex FFT:
//       Z = A + i B  (A, B Hermitian-extended half spectra)   =>   ifft(Z) = a + i b ,   ifft(Z) = conj(fft(conj Z)) / N.
#include <mutex>
#include <vector>

#include "mmk_common.h"
#include "fft1024.h"
#include "spectral_util.h"

namespace mmk {

// ---- spectrum -> waveform, overlap-add fused ----------------------------------------------------------------------------
// MODE 0: spec = (batch, frames, 513) complex (re, im);  MODE 1: (abs, angle) pairs;  MODE 2: mag plane x complex plane.
//
// A wave owns a SEGMENT of one clip's output (seg_hops hops) and walks the frames that cover it in order, a pair per
// FFT.  The windowed frames are summed into a wave-private ring of 2048 samples in LDS (the live window of a pair is
// n_fft + hop wide); whatever lies below the next pair's first sample is final: divided by the window envelope, written
// once, zeroed.  Frames are never written to HBM; the first ceil(n_fft / hop) - 1 frames of a segment are recomputed by
// its left neighbour (11 % more transforms at hop = n_fft / 4 with 27-hop segments).  Summation order = frame order and
// the pairing of frames is fixed, so the result does not depend on the launch geometry.

template <int MODE>
struct IstftRaw {
  cf32 a[16], b[16];
  float ma[16], mb[16];                                      // MODE 2 only (dead otherwise)
};

// bins of frames fa, fb for the FFT input slots n = lane + 64 r: bin n for r < 8, bin N - n (to be conjugated) for r >= 8
template <int MODE>
__device__ __forceinline__ void istft_load(IstftRaw<MODE>& raw, const float* __restrict__ spec, const float* __restrict__ mag, int64_t fa,
                                           int64_t fb, int lane) {
  const cf32* sa_lo = reinterpret_cast<const cf32*>(spec) + fa * 513 + lane;
  const cf32* sb_lo = reinterpret_cast<const cf32*>(spec) + fb * 513 + lane;
  const cf32* sa_hi = reinterpret_cast<const cf32*>(spec) + fa * 513 + (1024 - lane);
  const cf32* sb_hi = reinterpret_cast<const cf32*>(spec) + fb * 513 + (1024 - lane);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    raw.a[r] = (r < 8) ? sa_lo[64 * r] : sa_hi[-64 * r];
    raw.b[r] = (r < 8) ? sb_lo[64 * r] : sb_hi[-64 * r];
  }
  if (MODE == 2) {
    const float* ma_lo = mag + fa * 513 + lane;
    const float* mb_lo = mag + fb * 513 + lane;
    const float* ma_hi = mag + fa * 513 + (1024 - lane);
    const float* mb_hi = mag + fb * 513 + (1024 - lane);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      raw.ma[r] = (r < 8) ? ma_lo[64 * r] : ma_hi[-64 * r];
      raw.mb[r] = (r < 8) ? mb_lo[64 * r] : mb_hi[-64 * r];
    }
  }
}

constexpr int kRing = 2048;

// ---- the tables of spectral_util.h ------------------------------------------------------------------------------------------------
__device__ cf32 g_tw1024[1024];
__device__ float g_hann1024[1024];
__device__ float g_hann2048[2048];
__device__ cf32 g_w2048[1088];

__global__ void spectral_tables_kernel() {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m < 1024) {
    float sn, cs;
    sincospif(-2.0f * (float)m / 1024.0f, &sn, &cs);
    g_tw1024[m] = cf32{cs, sn};
    g_hann1024[m] = 0.5f - 0.5f * cospif(2.0f * (float)m / 1024.0f);          // periodic Hann (functionals.py:513)
  }
  if (m < 2048) g_hann2048[m] = 0.5f - 0.5f * cospif((float)m / 1024.0f);
  if (m < 1088) {
    float sn, cs;
    sincospif(-(float)m / 1024.0f, &sn, &cs);
    g_w2048[m] = cf32{cs, sn};
  }
}

int spectral_tables(hipStream_t stream, SpectralTables* out) {
  constexpr int kMaxDevices = 64;
  static SpectralTables tables[kMaxDevices];
  static bool ready[kMaxDevices];
  static std::mutex lock;
  int dev = 0;
  MMK_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxDevices) return fail(MMK_ERR_INVALID, "spectral tables: device %d", dev);
  std::lock_guard<std::mutex> guard(lock);
  if (!ready[dev]) {
    SpectralTables t;
    MMK_HIP(hipGetSymbolAddress((void**)&t.tw1024, HIP_SYMBOL(g_tw1024)));
    MMK_HIP(hipGetSymbolAddress((void**)&t.hann1024, HIP_SYMBOL(g_hann1024)));
    MMK_HIP(hipGetSymbolAddress((void**)&t.hann2048, HIP_SYMBOL(g_hann2048)));
    MMK_HIP(hipGetSymbolAddress((void**)&t.w2048, HIP_SYMBOL(g_w2048)));
    hipLaunchKernelGGL(spectral_tables_kernel, dim3(8), dim3(256), 0, stream);
    MMK_HIP(hipGetLastError());
    MMK_HIP(hipStreamSynchronize(stream));      // once per device and process: later launches may come on any stream
    tables[dev] = t;
    ready[dev] = true;
  }
  *out = tables[dev];
  return MMK_OK;
}

// One transformed pair (a[m] = Re(Y[m]) / N, b[m] = -Im(Y[m]) / N in buf) into the ring at its positions, then
// everything below `upto` (the next pair's first sample) is final: divided by the window envelope, written, zeroed.
__device__ __forceinline__ void ola_pair(float* ring, const cf32* buf, const float (&win)[16], const float* envt, const float* __restrict__ hann, int64_t f, int hop,
                                         bool has_b, int64_t frontier, int64_t upto, int64_t t0, int64_t t1, int64_t n_frames,
                                         float* __restrict__ orow, int lane) {
  constexpr int N = 1024;
  const int pa = (int)((f * hop) & (kRing - 1));
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int q = (pa + lane + 64 * j) & (kRing - 1);
    ring[q] += buf[fft_swz(lane) + 64 * j].x * win[j];
  }
  if (has_b) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int q = (pa + hop + lane + 64 * j) & (kRing - 1);
      ring[q] -= buf[fft_swz(lane) + 64 * j].y * win[j];
    }
  }
  __builtin_amdgcn_wave_barrier();
  // frontier is a multiple of hop and upto - frontier <= 2 hop: t mod hop without a division
  const int64_t t_int_lo = N - 1, t_int_hi = n_frames * hop;                  // inside: every frame that covers t exists
  for (int x = lane; frontier + x < upto; x += 64) {
    const int64_t t = frontier + x;
    const int q = (int)(t & (kRing - 1));
    const float acc = ring[q];
    ring[q] = 0.f;
    if (t >= t0 && t < t1) {
      float env;
      if (t >= t_int_lo && t < t_int_hi) {
        int r = x;
        r = r >= hop ? r - hop : r;
        r = r >= hop ? r - hop : r;
        env = envt[r];
      } else {                                              // the first / last n_fft samples of a clip
        int64_t g_hi = t / hop;
        g_hi = g_hi < n_frames - 1 ? g_hi : n_frames - 1;
        const int64_t g_lo = (t - N + 1 <= 0) ? 0 : (t - N + hop) / hop;
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

template <int MODE>
__global__ __launch_bounds__(64 * kIstftWaves) __attribute__((amdgpu_waves_per_eu(2, 2)))
void istft1024_kernel(const SpectralTables T, const float* __restrict__ spec, const float* __restrict__ mag, int64_t n_frames, int hop, int seg_hops,
                      int segs_per_clip, int64_t total_tasks, int64_t n_out, float* __restrict__ out) {
  constexpr int N = 1024;
  __shared__ cf32 tw[N];
  __shared__ float envt[N];                                 // window envelope where every covering frame exists, by t mod hop
  __shared__ cf32 bufs[kIstftWaves * kFftWaveLds];
  __shared__ float rings[kIstftWaves * kRing];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  load_twiddles(tw, T.tw1024, tid, 64 * kIstftWaves);
  for (int r = tid; r < hop; r += 64 * kIstftWaves) {       // (two workgroups per CU: no room for a separate w^2 table)
    float e = 0.f;
    for (int o = r; o < N; o += hop) {
      const float w = T.hann1024[o];
      e += w * w;
    }
    envt[r] = e;
  }
  float win[16];                                            // periodic Hann / N at n = lane + 64 r
#pragma unroll
  for (int r = 0; r < 16; ++r) win[r] = T.hann1024[lane + 64 * r] * (1.0f / (float)N);
  __syncthreads();
  cf32* buf = bufs + wave * kFftWaveLds;
  float* ring = rings + wave * kRing;
  const int64_t t_end = N / 2 + n_out;                      // positions t are in the untrimmed overlap-add signal

  for (int64_t task = (int64_t)blockIdx.x * kIstftWaves + wave; task < total_tasks; task += (int64_t)gridDim.x * kIstftWaves) {
    const int64_t b = task / segs_per_clip;
    const int64_t sgm = task - b * segs_per_clip;
    const int64_t t0 = N / 2 + sgm * seg_hops * hop;
    int64_t t1 = t0 + (int64_t)seg_hops * hop;
    t1 = t1 < t_end ? t1 : t_end;
    if (t0 >= t1) continue;
    const int64_t f_lo = (t0 - N + 1 <= 0) ? 0 : (t0 - N + hop) / hop;          // first frame that covers t0
    int64_t f_hi = (t1 - 1) / hop;                                             // last frame that covers t1 - 1
    f_hi = f_hi < n_frames - 1 ? f_hi : n_frames - 1;
    const int64_t fbase = b * n_frames;
    float* orow = out + b * n_out - N / 2;
#pragma unroll
    for (int j = 0; j < kRing / 64; ++j) ring[lane + 64 * j] = 0.f;
    // pairs are always (even, odd) frames, whatever the segment: two real transforms that share a complex one pick up
    // each other's rounding, so a fixed pairing makes every frame's samples independent of the launch geometry
    const int64_t f_first = f_lo & ~(int64_t)1, f_last = n_frames - 1;
    int64_t frontier = f_first * hop;
    IstftRaw<MODE> raw;
    istft_load<MODE>(raw, spec, mag, fbase + f_first, fbase + (f_first + 1 <= f_last ? f_first + 1 : f_first), lane);
    for (int64_t f = f_first; f <= f_hi; f += 2) {
      const bool has_b = f + 1 <= f_last;
      cf32 v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        cf32 A = istft_bin<MODE>(raw.a[r], raw.ma[r]);
        cf32 B = istft_bin<MODE>(raw.b[r], raw.mb[r]);
        if ((r == 0 || r == 8) && lane == 0) A.y = 0.f, B.y = 0.f;   // DC / Nyquist of a real signal: the C2R transform ignores them
        if (!has_b) B = cf32{0.f, 0.f};
        // Z = A + i B (n <= 512) or conj(A) + i conj(B); the FFT input is conj(Z)
        if (r < 8) v[r] = cf32{A.x - B.y, -(A.y + B.x)};
        else v[r] = cf32{A.x + B.y, -(B.x - A.y)};
      }
      {   // next pair's bins: in flight under this pair's transform (clamped, so unconditional)
        const int64_t na = f + 2 <= f_last ? f + 2 : f_last;
        const int64_t nb = f + 3 <= f_last ? f + 3 : f_last;
        istft_load<MODE>(raw, spec, mag, fbase + na, fbase + nb, lane);
      }
      fft1024_wave<true>(v, buf, tw, lane);
      const bool more = f + 2 <= f_hi;
      const int64_t upto = more ? (f + 2) * hop : t1;             // the ring is cleared again by the next segment
      ola_pair(ring, buf, win, envt, T.hann1024, f, hop, has_b, frontier, upto, t0, t1, n_frames, orow, lane);
      frontier = upto;
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// ---- the same for hop = n_fft / 4 (the reference's ratio), without the ring ------------------------------------------------------------------
// With hop = 256 a lane that reads the transformed frame as samples n = 4 lane + i + 256 j (i, j < 4) holds, for EVERY frame, the same four
// positions of each of the four output blocks the frame covers: the overlap-add is a shift register of 3 x 4 partial sums per lane - no LDS ring,
// no read-modify-write (64 LDS accesses per pair), same summation order as the ring (frame order).  A block is final when its fourth frame has
// been added: times the reciprocal of the window envelope (one IEEE division per lane and kernel instead of one per output sample), one
// 16-byte store per lane.  The spectrum side: every bin is loaded and turned from polar to cartesian ONCE, by the lane that owns it
// (k = lane + 64 j); the mirrored half of the transform's input (bins N - n) comes from lane 64 - lane through ds_bpermute - the kernel above
// loads and converts the bins 1 .. 511 twice (32 sincos per pair and lane instead of 18).
template <int MODE>
struct IstftBins {
  cf32 a[9], b[9];
  float ma[9], mb[9];                                        // MODE 2 only (dead otherwise)
};
template <int MODE>
__device__ __forceinline__ void istft_load_bins(IstftBins<MODE>& raw, const float* __restrict__ spec, const float* __restrict__ mag, int64_t fa,
                                                int64_t fb, int lane) {
  const cf32* sa = reinterpret_cast<const cf32*>(spec) + fa * 513 + lane;
  const cf32* sb = reinterpret_cast<const cf32*>(spec) + fb * 513 + lane;
#pragma unroll
  for (int j = 0; j < 9; ++j) {
    const int o = (j < 8 || lane == 0) ? 64 * j : 0;          // bin 512 exists in lane 0 only (clamped: an unconditional load)
    raw.a[j] = sa[o];
    raw.b[j] = sb[o];
    if (MODE == 2) {
      raw.ma[j] = mag[fa * 513 + lane + o];
      raw.mb[j] = mag[fb * 513 + lane + o];
    }
  }
}

template <int MODE>
__global__ __launch_bounds__(64 * kIstftWaves) __attribute__((amdgpu_waves_per_eu(2, 2)))
void istft1024q_kernel(const SpectralTables T, const float* __restrict__ spec, const float* __restrict__ mag, int64_t n_frames, int seg_hops,
                       int segs_per_clip, int64_t total_tasks, int64_t n_out, float* __restrict__ out) {
  constexpr int N = 1024, hop = N / 4;
  typedef float f32x4q __attribute__((ext_vector_type(4)));
  __shared__ cf32 tw[kFftTwLds];
  __shared__ cf32 bufs[kIstftWaves * kFftWaveLds];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  fill_twiddles_by_pass(tw, T.tw1024, tid, 64 * kIstftWaves);
  float win[4][4], renv[4];                                 // periodic Hann / N at n = 256 j + 4 lane + i; 1 / envelope at t mod hop = 4 lane + i
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float e = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float w = T.hann1024[256 * j + 4 * lane + i];
      win[j][i] = w * (1.0f / (float)N);
      e += w * w;                                            // (the order of the table the ring kernel builds: o = r, r + hop, ...)
    }
    renv[i] = 1.0f / e;
  }
  __syncthreads();
  cf32* buf = bufs + wave * kFftWaveLds;
  const int64_t t_end = N / 2 + n_out;                      // positions t are in the untrimmed overlap-add signal
  const int mirror = ((64 - lane) & 63) << 2;               // ds_bpermute address of the lane that owns bin N - n

  for (int64_t task = (int64_t)blockIdx.x * kIstftWaves + wave; task < total_tasks; task += (int64_t)gridDim.x * kIstftWaves) {
    const int64_t b = task / segs_per_clip;
    const int64_t sgm = task - b * segs_per_clip;
    const int64_t t0 = N / 2 + sgm * seg_hops * hop;
    int64_t t1 = t0 + (int64_t)seg_hops * hop;
    t1 = t1 < t_end ? t1 : t_end;
    if (t0 >= t1) continue;
    const int64_t f_lo = (t0 - N + 1 <= 0) ? 0 : (t0 - N + hop) / hop;          // first frame that covers t0
    int64_t f_hi = (t1 - 1) / hop;                                             // last frame that covers t1 - 1
    f_hi = f_hi < n_frames - 1 ? f_hi : n_frames - 1;
    const int64_t fbase = b * n_frames;
    float* orow = out + b * n_out - N / 2;
    float R[3][4];                                           // partial sums of the three blocks still waiting for frames
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i) R[q][i] = 0.f;
    // a block of hop samples is final once frame `blk` has been added: envelope, one 16-byte store per lane
    auto finish = [&](int64_t blk, const float (&e)[4]) {
      const int64_t t = blk * hop + 4 * lane;
      if (t < t0 || t >= t1) return;                          // (t0, t1 are multiples of hop: whole blocks)
      f32x4q o;
      if (blk >= 4 && blk <= n_frames - 1) {                  // every frame that covers the block exists
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = e[i] * renv[i];
      } else {                                                // the first / last n_fft samples of a clip
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int64_t ti = t + i;
          int64_t g_hi = ti / hop;
          g_hi = g_hi < n_frames - 1 ? g_hi : n_frames - 1;
          const int64_t g_lo = (ti - N + 1 <= 0) ? 0 : (ti - N + hop) / hop;
          float env = 0.f;
          for (int64_t g = g_lo; g <= g_hi; ++g) {
            const float w = T.hann1024[ti - g * hop];
            env += w * w;
          }
          o[i] = e[i] / env;
        }
      }
      *reinterpret_cast<f32x4q*>(orow + t) = o;
    };
    // pairs are always (even, odd) frames, whatever the segment: two real transforms that share a complex one pick up
    // each other's rounding, so a fixed pairing makes every frame's samples independent of the launch geometry
    const int64_t f_first = f_lo & ~(int64_t)1, f_last = n_frames - 1;
    int64_t next_blk = f_first;                               // the first block no frame has completed yet
    IstftBins<MODE> raw;
    istft_load_bins<MODE>(raw, spec, mag, fbase + f_first, fbase + (f_first + 1 <= f_last ? f_first + 1 : f_first), lane);
    for (int64_t f = f_first; f <= f_hi; f += 2) {
      const bool has_b = f + 1 <= f_last;
      // Z = A + i B (n <= 512) or conj(A) + i conj(B) (bins N - n); the FFT input is conj(Z)
      cf32 v[16], mz[9];
#pragma unroll
      for (int j = 0; j < 9; ++j) {
        cf32 A = istft_bin<MODE>(raw.a[j], raw.ma[j]);
        cf32 B = istft_bin<MODE>(raw.b[j], raw.mb[j]);
        if ((j == 0 || j == 8) && lane == 0) A.y = 0.f, B.y = 0.f;   // DC / Nyquist of a real signal: the C2R transform ignores them
        if (!has_b) B = cf32{0.f, 0.f};
        if (j < 8) v[j] = cf32{A.x - B.y, -(A.y + B.x)};
        mz[j] = cf32{A.x + B.y, -(B.x - A.y)};                 // what the slot of bin N - k takes
      }
      {   // next pair's bins: in flight under this pair's transform (clamped, so unconditional)
        const int64_t na = f + 2 <= f_last ? f + 2 : f_last;
        const int64_t nb = f + 3 <= f_last ? f + 3 : f_last;
        istft_load_bins<MODE>(raw, spec, mag, fbase + na, fbase + nb, lane);
      }
#pragma unroll
      for (int r = 8; r < 16; ++r) {
        // slot n = lane + 64 r takes bin k = N - n = (64 - lane) + 64 (15 - r): lane 64 - lane's bin 15 - r; lane 0: its own bin 64 (16 - r)
        const cf32 src = mz[15 - r];
        const float x = __int_as_float(__builtin_amdgcn_ds_bpermute(mirror, __float_as_int(src.x)));
        const float y = __int_as_float(__builtin_amdgcn_ds_bpermute(mirror, __float_as_int(src.y)));
        v[r] = lane == 0 ? mz[16 - r] : cf32{x, y};
      }
      fft1024_wave_bypass<true>(v, buf, tw, lane);
      // the two frames' samples n = 256 j + 4 lane + i: a[n] = Re(Y[n]) / N, b[n] = -Im(Y[n]) / N
      cf32 y[4][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const cf32* src = buf + fft_swz(4 * lane) + 256 * j;       // (the swizzle keeps four consecutive points together)
#pragma unroll
        for (int i = 0; i < 4; ++i) y[j][i] = src[i];
      }
      float e[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        e[i] = fmaf(y[0][i].x, win[0][i], R[0][i]);
        R[0][i] = fmaf(y[1][i].x, win[1][i], R[1][i]);
        R[1][i] = fmaf(y[2][i].x, win[2][i], R[2][i]);
        R[2][i] = y[3][i].x * win[3][i];
      }
      finish(f, e);
      if (has_b) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          e[i] = fmaf(-y[0][i].y, win[0][i], R[0][i]);
          R[0][i] = fmaf(-y[1][i].y, win[1][i], R[1][i]);
          R[1][i] = fmaf(-y[2][i].y, win[2][i], R[2][i]);
          R[2][i] = -y[3][i].y * win[3][i];
        }
        finish(f + 1, e);
      }
      __builtin_amdgcn_wave_barrier();                      // buf is rewritten by the next pair
      next_blk = f + (has_b ? 2 : 1);
    }
    // the clip's end: the blocks behind its last frame are what the shift register still holds
    for (; next_blk * hop < t1; ++next_blk) {
      float e[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        e[i] = R[0][i];
        R[0][i] = R[1][i];
        R[1][i] = R[2][i];
        R[2][i] = 0.f;
      }
      finish(next_blk, e);
    }
  }
}

// ---- STFT (center or not, constant or reflect padding) with five epilogues --------------------------------------------------
// OUT 0: (re, im)   OUT 1: (|S|, angle S)   OUT 2: angle S   OUT 4: |S|  (MagSpec)   OUT 3: the Griffin-Lim phase update
//   angles = S - m tprev ; angles /= |angles| + 1e-16 ; tprev = S          (torchaudio functional.griffinlim, 2.0.1)
// Two real frames per complex FFT, one pair per wave.
struct StftRaw { float a[16], b[16]; };

__device__ __forceinline__ void stft_load(StftRaw& raw, const float* __restrict__ xr, int64_t start, int hop, int64_t n_samples,
                                          int reflect, int lane) {
  constexpr int N = 1024;
  if (start >= 0 && start + hop + N <= n_samples) {
    // interior pair (all but the first / last few): one base address per frame, compile-time offsets
    const float* pa = xr + start + lane;
    if (hop == N / 4) {
      // frame B starts a quarter frame later: its first 12 register rows ARE frame A's rows 4 .. 15 - 20 loads instead of 32
#pragma unroll
      for (int r = 0; r < 16; ++r) raw.a[r] = pa[64 * r];
#pragma unroll
      for (int r = 0; r < 12; ++r) raw.b[r] = raw.a[r + 4];
#pragma unroll
      for (int r = 12; r < 16; ++r) raw.b[r] = pa[64 * (r + 4)];
    } else {
      const float* pb = pa + hop;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        raw.a[r] = pa[64 * r];
        raw.b[r] = pb[64 * r];
      }
    }
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t ia = start + lane + 64 * r, ib = ia + hop;
      const bool ina = ia >= 0 && ia < n_samples, inb = ib >= 0 && ib < n_samples;
      int64_t ja = ia, jb = ib;
      if (reflect) {                                        // torch 'reflect': no repeat of the edge sample
        ja = ia < 0 ? -ia : (ia >= n_samples ? 2 * (n_samples - 1) - ia : ia);
        jb = ib < 0 ? -ib : (ib >= n_samples ? 2 * (n_samples - 1) - ib : ib);
      }
      ja = ja < 0 ? 0 : (ja >= n_samples ? n_samples - 1 : ja);   // unconditional loads from clamped addresses
      jb = jb < 0 ? 0 : (jb >= n_samples ? n_samples - 1 : jb);
      const float a = xr[ja], bb = xr[jb];
      raw.a[r] = (reflect || ina) ? a : 0.f;                // pad_mode="constant": zeros
      raw.b[r] = (reflect || inb) ? bb : 0.f;
    }
  }
}

// ---- one whole Griffin-Lim iteration: stft -> phase update -> istft, per output segment ---------------------------------------
//   rebuilt = stft(wave_in) ; angles = normalise(rebuilt - m tprev_in) ; tprev_out = rebuilt ; wave_out = istft(mag angles)
// The segment walk of istft1024_kernel with a forward transform in front of each inverse one: the phase estimates never
// exist in HBM (12 KB per frame-iteration instead of 20.5).  wave and tprev are ping-pong buffers, so the frames a segment
// recomputes for its left edge read the same inputs as their owner and write the same values.
__global__ __launch_bounds__(64 * kIstftWaves) __attribute__((amdgpu_waves_per_eu(2, 2)))
void gla1024_iter_kernel(const SpectralTables T, const float* __restrict__ wave_in, const float* __restrict__ mag, const float* __restrict__ tprev_in,
                         float* __restrict__ tprev_out, float momentum, int64_t n_frames, int hop, int seg_hops, int segs_per_clip,
                         int64_t total_tasks, int64_t n_out, float* __restrict__ wave_out) {
  constexpr int N = 1024, bins = 513;
  __shared__ cf32 tw[N];
  __shared__ float envt[N];
  __shared__ cf32 bufs[kIstftWaves * kFftWaveLds];
  __shared__ float rings[kIstftWaves * kRing];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  load_twiddles(tw, T.tw1024, tid, 64 * kIstftWaves);
  for (int r = tid; r < hop; r += 64 * kIstftWaves) {
    float e = 0.f;
    for (int o = r; o < N; o += hop) {
      const float w = T.hann1024[o];
      e += w * w;
    }
    envt[r] = e;
  }
  float win[16], win_n[16];                                 // periodic Hann at n = lane + 64 r, and the same / N
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    win[r] = T.hann1024[lane + 64 * r];
    win_n[r] = win[r] * (1.0f / (float)N);
  }
  __syncthreads();
  cf32* buf = bufs + wave * kFftWaveLds;
  float* ring = rings + wave * kRing;
  const int64_t t_end = N / 2 + n_out;

  for (int64_t task = (int64_t)blockIdx.x * kIstftWaves + wave; task < total_tasks; task += (int64_t)gridDim.x * kIstftWaves) {
    const int64_t b = task / segs_per_clip;
    const int64_t sgm = task - b * segs_per_clip;
    const int64_t t0 = N / 2 + sgm * seg_hops * hop;
    int64_t t1 = t0 + (int64_t)seg_hops * hop;
    t1 = t1 < t_end ? t1 : t_end;
    if (t0 >= t1) continue;
    const int64_t f_lo = (t0 - N + 1 <= 0) ? 0 : (t0 - N + hop) / hop;
    int64_t f_hi = (t1 - 1) / hop;
    f_hi = f_hi < n_frames - 1 ? f_hi : n_frames - 1;
    const float* xr = wave_in + b * n_out;
    float* orow = wave_out + b * n_out - N / 2;
#pragma unroll
    for (int j = 0; j < kRing / 64; ++j) ring[lane + 64 * j] = 0.f;
    // (even, odd) pairs whatever the segment: a frame that two segments compute gets bit-identical values in both
    const int64_t f_first = f_lo & ~(int64_t)1;
    int64_t frontier = f_first * hop;
    for (int64_t f = f_first; f <= f_hi; f += 2) {
      const bool has_b = f + 1 < n_frames;
      const int64_t ea = (b * n_frames + f) * bins;
      // ---- forward: frames f, f + 1 of the current waveform (center, reflect) -------------------------------------------------
      StftRaw raw;
      stft_load(raw, xr, f * hop - N / 2, hop, n_out, 1, lane);
      cf32 v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = cf32{raw.a[r] * win[r], has_b ? raw.b[r] * win[r] : 0.f};
      // previous spectra and magnitudes of the bins this lane updates: in flight under the transform
      cf32 tpa[9], tpb[9];
      float mga[9], mgb[9];
#pragma unroll
      for (int jj = 0; jj < 9; ++jj) {
        const int k = lane + 64 * jj;
        const int kc = k < bins ? k : 0;                    // clamped: unconditional loads
        const int64_t eb = has_b ? ea + bins : ea;
        tpa[jj] = *reinterpret_cast<const cf32*>(tprev_in + 2 * (ea + kc));
        tpb[jj] = *reinterpret_cast<const cf32*>(tprev_in + 2 * (eb + kc));
        mga[jj] = mag[ea + kc];
        mgb[jj] = mag[eb + kc];
      }
      fft1024_wave<true>(v, buf, tw, lane);
      // ---- phase update on the two real spectra; the new spectra mag * angles go back to LDS as A (0..512), B (513..1025) ---
      cf32 za[9], zb[9];
#pragma unroll
      for (int jj = 0; jj < 9; ++jj) {
        const int k = lane + 64 * jj;
        if (k < bins) {
          const cf32 z = buf[fft_swz(lane) + 64 * jj];
          const cf32 zc = buf[(((N - k) & (N - 1)) & ~63) | fft_swz((64 - lane) & 63)];
          const cf32 sa = cf32{0.5f * (z.x + zc.x), 0.5f * (z.y - zc.y)};
          const cf32 sb = cf32{0.5f * (z.y + zc.y), -0.5f * (z.x - zc.x)};
          const cf32 ga = cf32{sa.x - momentum * tpa[jj].x, sa.y - momentum * tpa[jj].y};
          const cf32 gb = cf32{sb.x - momentum * tpb[jj].x, sb.y - momentum * tpb[jj].y};
          const float da = sqrtf(ga.x * ga.x + ga.y * ga.y) + 1e-16f, db = sqrtf(gb.x * gb.x + gb.y * gb.y) + 1e-16f;
          za[jj] = cf32{mga[jj] * (ga.x / da), mga[jj] * (ga.y / da)};
          zb[jj] = cf32{mgb[jj] * (gb.x / db), mgb[jj] * (gb.y / db)};
          if (k == 0 || k == 512) za[jj].y = 0.f, zb[jj].y = 0.f;    // DC / Nyquist: the C2R transform ignores them
          *reinterpret_cast<cf32*>(tprev_out + 2 * (ea + k)) = sa;
          if (has_b) *reinterpret_cast<cf32*>(tprev_out + 2 * (ea + bins + k)) = sb;
          else zb[jj] = cf32{0.f, 0.f};
        }
      }
      __builtin_amdgcn_wave_barrier();                      // every lane has read its bins of the forward transform
#pragma unroll
      for (int jj = 0; jj < 9; ++jj) {
        const int k = lane + 64 * jj;
        if (k < bins) buf[k] = za[jj], buf[bins + k] = zb[jj];
      }
      __builtin_amdgcn_wave_barrier();
      // ---- inverse: Z = A + i B (n <= 512) or conj(A) + i conj(B); the FFT input is conj(Z) ------------------------------------
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = lane + 64 * r;
        const int k = (r < 8) ? n : N - n;
        const cf32 A = buf[k], B = buf[bins + k];
        if (r < 8) v[r] = cf32{A.x - B.y, -(A.y + B.x)};
        else v[r] = cf32{A.x + B.y, -(B.x - A.y)};
      }
      __builtin_amdgcn_wave_barrier();
      fft1024_wave<true>(v, buf, tw, lane);
      const bool more = f + 2 <= f_hi;
      const int64_t upto = more ? (f + 2) * hop : t1;
      ola_pair(ring, buf, win_n, envt, T.hann1024, f, hop, has_b, frontier, upto, t0, t1, n_frames, orow, lane);
      frontier = upto;
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// ---- the same for hop = n_fft / 4: no ring, no second trip of the new spectra through LDS ------------------------------------------------------
// The inverse half as in istft1024q_kernel (shift-register overlap-add, reciprocal envelope, 16-byte stores, the mirrored bins from lane
// 64 - lane through ds_bpermute instead of an LDS image of both spectra); the phase normalisation multiplies by one reciprocal per bin instead of
// dividing both components.
__global__ __launch_bounds__(64 * kIstftWaves) __attribute__((amdgpu_waves_per_eu(2, 2)))
void gla1024q_iter_kernel(const SpectralTables T, const float* __restrict__ wave_in, const float* __restrict__ mag, const float* __restrict__ tprev_in,
                          float* __restrict__ tprev_out, float momentum, int64_t n_frames, int seg_hops, int segs_per_clip, int64_t total_tasks,
                          int64_t n_out, float* __restrict__ wave_out) {
  constexpr int N = 1024, bins = 513, hop = N / 4;
  typedef float f32x4q __attribute__((ext_vector_type(4)));
  __shared__ cf32 tw[kFftTwLds];
  __shared__ cf32 bufs[kIstftWaves * kFftWaveLds];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  fill_twiddles_by_pass(tw, T.tw1024, tid, 64 * kIstftWaves);
  float win[16];                                            // periodic Hann at n = lane + 64 r (the forward transform's load order)
#pragma unroll
  for (int r = 0; r < 16; ++r) win[r] = T.hann1024[lane + 64 * r];
  float win_n[4][4], renv[4];                               // Hann / N at n = 256 j + 4 lane + i; 1 / envelope at t mod hop = 4 lane + i
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float e = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float w = T.hann1024[256 * j + 4 * lane + i];
      win_n[j][i] = w * (1.0f / (float)N);
      e += w * w;
    }
    renv[i] = 1.0f / e;
  }
  __syncthreads();
  cf32* buf = bufs + wave * kFftWaveLds;
  const int64_t t_end = N / 2 + n_out;
  const int mirror = ((64 - lane) & 63) << 2;

  for (int64_t task = (int64_t)blockIdx.x * kIstftWaves + wave; task < total_tasks; task += (int64_t)gridDim.x * kIstftWaves) {
    const int64_t b = task / segs_per_clip;
    const int64_t sgm = task - b * segs_per_clip;
    const int64_t t0 = N / 2 + sgm * seg_hops * hop;
    int64_t t1 = t0 + (int64_t)seg_hops * hop;
    t1 = t1 < t_end ? t1 : t_end;
    if (t0 >= t1) continue;
    const int64_t f_lo = (t0 - N + 1 <= 0) ? 0 : (t0 - N + hop) / hop;
    int64_t f_hi = (t1 - 1) / hop;
    f_hi = f_hi < n_frames - 1 ? f_hi : n_frames - 1;
    const float* xr = wave_in + b * n_out;
    float* orow = wave_out + b * n_out - N / 2;
    float R[3][4];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i) R[q][i] = 0.f;
    auto finish = [&](int64_t blk, const float (&e)[4]) {
      const int64_t t = blk * hop + 4 * lane;
      if (t < t0 || t >= t1) return;
      f32x4q o;
      if (blk >= 4 && blk <= n_frames - 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = e[i] * renv[i];
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int64_t ti = t + i;
          int64_t g_hi = ti / hop;
          g_hi = g_hi < n_frames - 1 ? g_hi : n_frames - 1;
          const int64_t g_lo = (ti - N + 1 <= 0) ? 0 : (ti - N + hop) / hop;
          float env = 0.f;
          for (int64_t g = g_lo; g <= g_hi; ++g) {
            const float w = T.hann1024[ti - g * hop];
            env += w * w;
          }
          o[i] = e[i] / env;
        }
      }
      *reinterpret_cast<f32x4q*>(orow + t) = o;
    };
    // (even, odd) pairs whatever the segment: a frame that two segments compute gets bit-identical values in both
    const int64_t f_first = f_lo & ~(int64_t)1;
    int64_t next_blk = f_first;
    for (int64_t f = f_first; f <= f_hi; f += 2) {
      const bool has_b = f + 1 < n_frames;
      const int64_t ea = (b * n_frames + f) * bins;
      // ---- forward: frames f, f + 1 of the current waveform (center, reflect) -------------------------------------------------
      StftRaw raw;
      stft_load(raw, xr, f * hop - N / 2, hop, n_out, 1, lane);
      cf32 v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = cf32{raw.a[r] * win[r], has_b ? raw.b[r] * win[r] : 0.f};
      // previous spectra and magnitudes of the bins this lane updates: in flight under the transform
      cf32 tpa[9], tpb[9];
      const int64_t eb = has_b ? ea + bins : ea;
#pragma unroll
      for (int jj = 0; jj < 9; ++jj) {
        const int k = lane + 64 * jj;
        const int kc = k < bins ? k : 0;                    // clamped: unconditional loads
        tpa[jj] = *reinterpret_cast<const cf32*>(tprev_in + 2 * (ea + kc));
        tpb[jj] = *reinterpret_cast<const cf32*>(tprev_in + 2 * (eb + kc));
      }
      fft1024_wave_bypass<true>(v, buf, tw, lane);
      float mga[9], mgb[9];                                  // (asked for behind the transform: 18 registers less to carry through it)
#pragma unroll
      for (int jj = 0; jj < 9; ++jj) {
        const int k = lane + 64 * jj;
        const int kc = k < bins ? k : 0;
        mga[jj] = mag[ea + kc];
        mgb[jj] = mag[eb + kc];
      }
      // ---- phase update on the two real spectra; the inverse transform's input straight from the registers ------------------------
      cf32 mz[9];
#pragma unroll
      for (int jj = 0; jj < 9; ++jj) {
        const int k = lane + 64 * jj;
        cf32 za = cf32{0.f, 0.f}, zb = cf32{0.f, 0.f};
        if (k < bins) {
          const cf32 z = buf[fft_swz(lane) + 64 * jj];
          const cf32 zc = buf[(((N - k) & (N - 1)) & ~63) | fft_swz((64 - lane) & 63)];
          const cf32 sa = cf32{0.5f * (z.x + zc.x), 0.5f * (z.y - zc.y)};
          const cf32 sb = cf32{0.5f * (z.y + zc.y), -0.5f * (z.x - zc.x)};
          const cf32 ga = cf32{sa.x - momentum * tpa[jj].x, sa.y - momentum * tpa[jj].y};
          const cf32 gb = cf32{sb.x - momentum * tpb[jj].x, sb.y - momentum * tpb[jj].y};
          const float ra = mga[jj] / (sqrtf(ga.x * ga.x + ga.y * ga.y) + 1e-16f), rb = mgb[jj] / (sqrtf(gb.x * gb.x + gb.y * gb.y) + 1e-16f);
          za = cf32{ga.x * ra, ga.y * ra};
          zb = cf32{gb.x * rb, gb.y * rb};
          if (k == 0 || k == 512) za.y = 0.f, zb.y = 0.f;    // DC / Nyquist: the C2R transform ignores them
          *reinterpret_cast<cf32*>(tprev_out + 2 * (ea + k)) = sa;
          if (has_b) *reinterpret_cast<cf32*>(tprev_out + 2 * (ea + bins + k)) = sb;
          else zb = cf32{0.f, 0.f};
        }
        // Z = A + i B (n <= 512) or conj(A) + i conj(B) (bins N - n); the FFT input is conj(Z)
        if (jj < 8) v[jj] = cf32{za.x - zb.y, -(za.y + zb.x)};
        mz[jj] = cf32{za.x + zb.y, -(zb.x - za.y)};
      }
      __builtin_amdgcn_wave_barrier();                      // every lane has read its bins of the forward transform
#pragma unroll
      for (int r = 8; r < 16; ++r) {
        const cf32 src = mz[15 - r];
        const float x = __int_as_float(__builtin_amdgcn_ds_bpermute(mirror, __float_as_int(src.x)));
        const float y = __int_as_float(__builtin_amdgcn_ds_bpermute(mirror, __float_as_int(src.y)));
        v[r] = lane == 0 ? mz[16 - r] : cf32{x, y};
      }
      fft1024_wave_bypass<true>(v, buf, tw, lane);
      cf32 y[4][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const cf32* src = buf + fft_swz(4 * lane) + 256 * j;
#pragma unroll
        for (int i = 0; i < 4; ++i) y[j][i] = src[i];
      }
      float e[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        e[i] = fmaf(y[0][i].x, win_n[0][i], R[0][i]);
        R[0][i] = fmaf(y[1][i].x, win_n[1][i], R[1][i]);
        R[1][i] = fmaf(y[2][i].x, win_n[2][i], R[2][i]);
        R[2][i] = y[3][i].x * win_n[3][i];
      }
      finish(f, e);
      if (has_b) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          e[i] = fmaf(-y[0][i].y, win_n[0][i], R[0][i]);
          R[0][i] = fmaf(-y[1][i].y, win_n[1][i], R[1][i]);
          R[1][i] = fmaf(-y[2][i].y, win_n[2][i], R[2][i]);
          R[2][i] = -y[3][i].y * win_n[3][i];
        }
        finish(f + 1, e);
      }
      __builtin_amdgcn_wave_barrier();
      next_blk = f + (has_b ? 2 : 1);
    }
    for (; next_blk * hop < t1; ++next_blk) {                // the clip's end: the blocks behind its last frame
      float e[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        e[i] = R[0][i];
        R[0][i] = R[1][i];
        R[1][i] = R[2][i];
        R[2][i] = 0.f;
      }
      finish(next_blk, e);
    }
  }
}

#ifndef MMK_STFT_WAVES
#define MMK_STFT_WAVES 4
#endif
#ifndef MMK_STFT_WPE
#define MMK_STFT_WPE 3
#endif
constexpr int kStftWaves = MMK_STFT_WAVES;      // waves (= frame pairs in flight) per workgroup of the n_fft = 1024 STFT kernel

template <int OUT>
__global__ __launch_bounds__(64 * kStftWaves) __attribute__((amdgpu_waves_per_eu(MMK_STFT_WPE, MMK_STFT_WPE)))
void stft1024_kernel(const SpectralTables T, const float* __restrict__ x, int64_t x_row_stride, int64_t n_samples, int hop, int center, int reflect,
                     int64_t n_frames, int64_t total_pairs, int runs_per_row, float* __restrict__ out, float* __restrict__ tprev, float momentum) {
  constexpr int N = 1024, bins = 513;
#ifndef MMK_STFT_REGTW
#define MMK_STFT_REGTW 1      // the lane's 30 twiddles in registers for the whole kernel (no table in LDS at all)
#endif
  __shared__ cf32 tw[MMK_STFT_REGTW ? 1 : N];
  __shared__ cf32 bufs[kStftWaves * kFftWaveLds];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float win[16];                                            // periodic Hann at n = lane + 64 r (functionals.py:513)
#pragma unroll
  for (int r = 0; r < 16; ++r) win[r] = T.hann1024[lane + 64 * r];
  FftLaneTw ltw;
  if (MMK_STFT_REGTW) {
    ltw.load(T.tw1024, lane);
  } else {
    load_twiddles(tw, T.tw1024, tid, 64 * kStftWaves);
    __syncthreads();
  }
  cf32* buf = bufs + wave * kFftWaveLds;
  const int64_t pairs_per_row = (n_frames + 1) >> 1;
  const int64_t pad = center ? N / 2 : 0;
  const int64_t stride = (int64_t)gridDim.x * kStftWaves;

  // A wave takes RUNS of consecutive pairs of one clip.  The pairs of a clip overlap - a pair reads 1280 samples, 768 of them the next pair's too -
  // and with the pairs dealt out one by one, neighbours landed on different XCDs: every L2 fetched the overlap for itself, 2.5 x the bytes over the
  // fabric (141 MB for 56 MB of samples; that load phase alone took 22 of the kernel's 55 us - its bytes at the rate the fabric gives).  Inside a run
  // with hop = n_fft / 4 the window slides through the wave's registers: the 20 rows of 64 samples of a pair are rows 8 .. 19 of the pair before and
  // 8 new ones - 8 loads per pair instead of 20.
  const int run_len = (int)((pairs_per_row + runs_per_row - 1) / runs_per_row);
  const int64_t total_runs = (int64_t)(total_pairs / pairs_per_row) * runs_per_row;
  for (int64_t run = (int64_t)blockIdx.x * kStftWaves + wave; run < total_runs; run += stride) {
    const int64_t b = (int64_t)((unsigned)run / (unsigned)runs_per_row);
    const int64_t p0 = (run - b * runs_per_row) * run_len;
    const int64_t p1 = p0 + run_len < pairs_per_row ? p0 + run_len : pairs_per_row;
    const float* xr = x + b * x_row_stride;
    float rows[20];                                          // rows[r] = x[start + lane + 64 r] of the pair at hand (hop = N / 4)
    bool slid = false;                                       // rows holds the pair before this one
  for (int64_t pr = p0; pr < p1; ++pr) {
    const int64_t f0 = pr * 2;
    const bool has_b = (f0 + 1) < n_frames;
    const int64_t start = f0 * hop - pad;
    StftRaw raw;
    if (hop == N / 4 && start >= 0 && start + hop + N <= n_samples) {
      const float* pa = xr + start + lane;
      if (slid) {
#pragma unroll
        for (int r = 0; r < 12; ++r) rows[r] = rows[r + 8];
#pragma unroll
        for (int r = 12; r < 20; ++r) rows[r] = pa[64 * r];
      } else {
#pragma unroll
        for (int r = 0; r < 20; ++r) rows[r] = pa[64 * r];
      }
      slid = true;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        raw.a[r] = rows[r];
        raw.b[r] = rows[r + 4];                               // frame B starts a quarter frame later
      }
    } else {
      stft_load(raw, xr, start, hop, n_samples, reflect, lane);
      slid = false;
    }
    cf32 v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = cf32{raw.a[r] * win[r], has_b ? raw.b[r] * win[r] : 0.f};   // frame f0 -> re, f0 + 1 -> im
    const int64_t ea = (b * n_frames + f0) * bins;
#ifndef MMK_STFT_ABL
#define MMK_STFT_ABL 0        // timing experiments only: 1 no stores, 2 no transform (the window's values written straight to LDS), 3 both
#endif
    if (MMK_STFT_ABL & 2) {
#pragma unroll
      for (int r = 0; r < 16; ++r) buf[lane + 64 * r] = v[r];
      __builtin_amdgcn_wave_barrier();
    } else if (MMK_STFT_REGTW) fft1024_wave_regtw<true>(v, buf, ltw, lane);
    else fft1024_wave<true>(v, buf, tw, lane);
    // the two real spectra:  A[k] = (Z[k] + conj(Z[N-k])) / 2 ,  B[k] = (Z[k] - conj(Z[N-k])) / (2i)
#pragma unroll
    for (int jj = 0; jj < 9; ++jj) {
      const int k = lane + 64 * jj;
      if (k < bins) {
        const cf32 z = buf[fft_swz(lane) + 64 * jj];
        const cf32 zc = buf[(((N - k) & (N - 1)) & ~63) | fft_swz((64 - lane) & 63)];
        if (OUT == 4) {
          // MagSpec: with p = Z[k] + Z[N-k], m = Z[k] - Z[N-k] (two packed ops), |A| = sqrt(p.x^2 + m.y^2) / 2 and
          // |B| = sqrt(p.y^2 + m.x^2) / 2: packed squares, the hardware square root (1 ulp; the reference's abs() of a complex64 is
          // held to 2e-5 of the largest magnitude in the tests), 32-bit offsets from the pair's base - the epilogue was a third of the
          // kernel's vector instructions
          const cf32 pz = z + zc, mz = z - zc;
          const cf32 pp = pz * pz, mm = mz * mz;
          float* o = out + ea;
          if ((MMK_STFT_ABL & 1) && pp.x != 1.2345f) continue;
          o[k] = 0.5f * __builtin_amdgcn_sqrtf(pp.x + mm.y);
          if (has_b) o[bins + k] = 0.5f * __builtin_amdgcn_sqrtf(pp.y + mm.x);
          continue;
        }
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
    __builtin_amdgcn_wave_barrier();                        // buf is rewritten by the next pair
  }
  }
}

// ---- any power-of-two n_fft in [64, 4096] -------------------------------------------------------------------------------------
// A workgroup walks over frame PAIRS: two real frames go through ONE complex FFT of size n_fft (frame A in the real lane,
// frame B in the imaginary lane), radix-4 Stockham autosort passes (+ one radix-2 pass when log2 n_fft is odd) between two
// LDS buffers, natural-order output.  The twiddle table exp(-2 pi i m / N) and the periodic Hann window live in LDS and
// are built once per workgroup (sincospif), amortised over all the pairs it processes.

// src -> natural-order DFT; returns the buffer that holds it (src or dst).  All threads of the workgroup call it.
__device__ __forceinline__ float2* stockham_fft(float2* src, float2* dst, const float2* __restrict__ tw, int N, int tid, int nt) {
  const int half = N >> 1, quarter = N >> 2;
  int p = 1;
  for (; p * 4 <= N; p *= 4) {
    const int tstep = N / (4 * p);
    for (int i = tid; i < quarter; i += nt) {
      const int k = i & (p - 1);
      const int j = ((i - k) << 2) + k;
      const int m = k * tstep;
      const float2 w1 = tw[m], w2 = tw[2 * m], w3 = tw[3 * m];
      const float2 u0 = src[i];
      const float2 a1 = src[i + quarter], a2 = src[i + 2 * quarter], a3 = src[i + 3 * quarter];
      const float2 u1 = make_float2(a1.x * w1.x - a1.y * w1.y, a1.x * w1.y + a1.y * w1.x);
      const float2 u2 = make_float2(a2.x * w2.x - a2.y * w2.y, a2.x * w2.y + a2.y * w2.x);
      const float2 u3 = make_float2(a3.x * w3.x - a3.y * w3.y, a3.x * w3.y + a3.y * w3.x);
      const float2 v0 = make_float2(u0.x + u2.x, u0.y + u2.y), v1 = make_float2(u0.x - u2.x, u0.y - u2.y);
      const float2 v2 = make_float2(u1.x + u3.x, u1.y + u3.y);
      const float2 v3 = make_float2(u1.y - u3.y, -(u1.x - u3.x));       // (u1 - u3) * (-i)
      dst[j] = make_float2(v0.x + v2.x, v0.y + v2.y);
      dst[j + p] = make_float2(v1.x + v3.x, v1.y + v3.y);
      dst[j + 2 * p] = make_float2(v0.x - v2.x, v0.y - v2.y);
      dst[j + 3 * p] = make_float2(v1.x - v3.x, v1.y - v3.y);
    }
    __syncthreads();
    float2* t = src; src = dst; dst = t;
  }
  if (p < N) {   // log2 n_fft odd: one radix-2 pass with p = N/2
    for (int i = tid; i < half; i += nt) {
      const int k = i & (p - 1);
      const int j = ((i - k) << 1) + k;
      const float2 w = tw[k * (N / (2 * p))];
      const float2 u0 = src[i], a1 = src[i + half];
      const float2 u1 = make_float2(a1.x * w.x - a1.y * w.y, a1.x * w.y + a1.y * w.x);
      dst[j] = make_float2(u0.x + u1.x, u0.y + u1.y);
      dst[j + p] = make_float2(u0.x - u1.x, u0.y - u1.y);
    }
    __syncthreads();
    float2* t = src; src = dst; dst = t;
  }
  return src;
}

__device__ __forceinline__ void generic_tables(float2* tw, float* win, int N, float win_scale, int tid, int nt) {
  for (int m = tid; m < N; m += nt) {
    float sn, cs;
    sincospif(-2.0f * (float)m / (float)N, &sn, &cs);
    tw[m] = make_float2(cs, sn);
    win[m] = (0.5f - 0.5f * cospif(2.0f * (float)m / (float)N)) * win_scale;   // periodic Hann, as torch.hann_window(n_fft)
  }
}

template <int OUT>   // the epilogues of stft1024_kernel
__global__ __launch_bounds__(256) void stft_generic_kernel(const float* __restrict__ x, int64_t x_row_stride, int64_t n_samples, int n_fft,
                                                          int hop, int center, int reflect, int64_t n_frames, int64_t total_pairs,
                                                          float* __restrict__ out, float* __restrict__ tprev, float momentum) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int N = n_fft, half = n_fft >> 1;
  float2* buf0 = reinterpret_cast<float2*>(smem_raw);
  float2* buf1 = buf0 + N;
  float2* tw = buf1 + N;
  float* win = reinterpret_cast<float*>(tw + N);
  const int tid = threadIdx.x, nt = blockDim.x;
  generic_tables(tw, win, N, 1.0f, tid, nt);
  __syncthreads();
  const int64_t pairs_per_row = (n_frames + 1) >> 1;
  const int64_t pad = center ? half : 0;
  const int bins = half + 1;

  for (int64_t pair = blockIdx.x; pair < total_pairs; pair += gridDim.x) {
    const int64_t b = pair / pairs_per_row;
    const int64_t f0 = (pair - b * pairs_per_row) * 2;
    const bool has_b = (f0 + 1) < n_frames;
    const float* xr = x + b * x_row_stride;
    for (int n = tid; n < N; n += nt) {
      const float w = win[n];
      int64_t ia = f0 * hop + n - pad, ib = ia + hop;
      bool ina = ia >= 0 && ia < n_samples, inb = ib >= 0 && ib < n_samples;
      if (reflect) {                                        // torch 'reflect': no repeat of the edge sample
        ia = ia < 0 ? -ia : (ia >= n_samples ? 2 * (n_samples - 1) - ia : ia);
        ib = ib < 0 ? -ib : (ib >= n_samples ? 2 * (n_samples - 1) - ib : ib);
        ina = ia >= 0 && ia < n_samples, inb = ib >= 0 && ib < n_samples;
      }
      const float a = ina ? xr[ia] : 0.f;                   // pad_mode="constant": zeros
      const float bb = (has_b && inb) ? xr[ib] : 0.f;
      buf0[n] = make_float2(a * w, bb * w);
    }
    __syncthreads();
    const float2* src = stockham_fft(buf0, buf1, tw, N, tid, nt);
    const int64_t ea = (b * n_frames + f0) * bins;
    for (int k = tid; k < bins; k += nt) {
      const float2 z = src[k];
      const float2 zc = src[(N - k) & (N - 1)];
      float2 s2[2];
      s2[0] = make_float2(0.5f * (z.x + zc.x), 0.5f * (z.y - zc.y));
      s2[1] = make_float2(0.5f * (z.y + zc.y), -0.5f * (z.x - zc.x));
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        if (q == 1 && !has_b) break;
        const int64_t e = ea + (int64_t)q * bins + k;
        const float2 v = s2[q];
        if (OUT == 0) *reinterpret_cast<float2*>(out + 2 * e) = v;
        if (OUT == 1) *reinterpret_cast<float2*>(out + 2 * e) = make_float2(sqrtf(v.x * v.x + v.y * v.y), atan2f(v.y, v.x));
        if (OUT == 2) out[e] = atan2f(v.y, v.x);
        if (OUT == 4) out[e] = sqrtf(v.x * v.x + v.y * v.y);
        if (OUT == 3) {
          const float2 tp = *reinterpret_cast<const float2*>(tprev + 2 * e);
          const float2 g = make_float2(v.x - momentum * tp.x, v.y - momentum * tp.y);
          const float d = sqrtf(g.x * g.x + g.y * g.y) + 1e-16f;
          *reinterpret_cast<float2*>(out + 2 * e) = make_float2(g.x / d, g.y / d);
          *reinterpret_cast<float2*>(tprev + 2 * e) = v;
        }
      }
    }
    __syncthreads();   // the buffers are rewritten by the next pair
  }
}

// spectrum -> windowed frames (batch, frames, n_fft) in `frames`; the inputs of istft1024_kernel's three modes
template <int MODE>
__global__ __launch_bounds__(256) void istft_frames_generic_kernel(const float* __restrict__ spec, const float* __restrict__ mag, int n_fft,
                                                                  int64_t n_frames, int64_t total_pairs, float* __restrict__ frames) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int N = n_fft, half = n_fft >> 1, bins = half + 1;
  float2* buf0 = reinterpret_cast<float2*>(smem_raw);
  float2* buf1 = buf0 + N;
  float2* tw = buf1 + N;
  float* win = reinterpret_cast<float*>(tw + N);
  const int tid = threadIdx.x, nt = blockDim.x;
  generic_tables(tw, win, N, 1.0f / (float)N, tid, nt);
  __syncthreads();
  const int64_t pairs_per_row = (n_frames + 1) >> 1;
  for (int64_t pair = blockIdx.x; pair < total_pairs; pair += gridDim.x) {
    const int64_t b = pair / pairs_per_row;
    const int64_t f0 = (pair - b * pairs_per_row) * 2;
    const bool has_b = (f0 + 1) < n_frames;
    const int64_t fa = b * n_frames + f0, fb = has_b ? fa + 1 : fa;
    for (int n = tid; n < N; n += nt) {
      const int k = n <= half ? n : N - n;
      const int64_t e_a = fa * bins + k, e_b = fb * bins + k;
      const cf32 ca = *reinterpret_cast<const cf32*>(spec + 2 * e_a), cb = *reinterpret_cast<const cf32*>(spec + 2 * e_b);
      cf32 A = istft_bin<MODE>(ca, MODE == 2 ? mag[e_a] : 0.f);
      cf32 B = istft_bin<MODE>(cb, MODE == 2 ? mag[e_b] : 0.f);
      if (k == 0 || k == half) A.y = 0.f, B.y = 0.f;        // DC / Nyquist of a real signal
      if (!has_b) B = cf32{0.f, 0.f};
      // Z = A + i B (n <= N/2) or conj(A) + i conj(B); the FFT input is conj(Z)
      buf0[n] = n <= half ? make_float2(A.x - B.y, -(A.y + B.x)) : make_float2(A.x + B.y, -(B.x - A.y));
    }
    __syncthreads();
    const float2* y = stockham_fft(buf0, buf1, tw, N, tid, nt);
    float* oa = frames + fa * N;                            // a[m] = Re(Y[m]) / N , b[m] = -Im(Y[m]) / N ; windowed
    for (int m = tid; m < N; m += nt) {
      oa[m] = y[m].x * win[m];
      if (has_b) oa[N + m] = -y[m].y * win[m];
    }
    __syncthreads();
  }
}

// out[b][n] = sum_f wf[b][f][t - f hop] / sum_f w^2[t - f hop],  t = n + N/2, over the frames that cover t, in frame order
// (torch.istft: fold, divide by the folded window^2, trim n_fft/2 on both sides)
template <int V>
__global__ __launch_bounds__(256) void istft_ola_generic_kernel(const float* __restrict__ frames, int n_fft, int64_t n_frames, int hop,
                                                               int64_t n_out, int64_t total, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* w2 = reinterpret_cast<float*>(smem_raw);
  const int N = n_fft;
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
    int64_t f_hi = t / hop;
    if (f_hi > n_frames - 1) f_hi = n_frames - 1;
    const int64_t f_lo = (t - N + 1 <= 0) ? 0 : (t - N + hop) / hop;           // ceil((t - N + 1) / hop)
    float acc[V], env[V];
#pragma unroll
    for (int i = 0; i < V; ++i) acc[i] = 0.f, env[i] = 0.f;
    const float* fr = frames + b * n_frames * N;
    for (int64_t f = f_lo; f <= f_hi; ++f) {
      const int64_t o = t - f * hop;                        // offset of sample n in frame f; hop % 4 == 0: four samples share it
      if (o < 0 || o >= N) continue;
      if (V == 4) {
        const float4 xv = *reinterpret_cast<const float4*>(fr + f * N + o);
        acc[0] += xv.x, acc[1] += xv.y, acc[2] += xv.z, acc[3] += xv.w;
        env[0] += w2[o], env[1] += w2[o + 1], env[2] += w2[o + 2], env[3] += w2[o + 3];
      } else {
        acc[0] += fr[f * N + o];
        env[0] += w2[o];
      }
    }
    if (V == 4) *reinterpret_cast<float4*>(out + b * n_out + n) = make_float4(acc[0] / env[0], acc[1] / env[1], acc[2] / env[2], acc[3] / env[3]);
    else out[b * n_out + n] = acc[0] / env[0];
  }
}

__global__ void fill_complex_kernel(float* __restrict__ dst, int64_t n, cf32 v) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
    *reinterpret_cast<cf32*>(dst + 2 * e) = v;
}

static int check_fft(const char* what, int n_fft, int hop) {
  int log2n = 0;
  while ((1 << log2n) < n_fft) ++log2n;
  if ((1 << log2n) != n_fft || n_fft < 64 || n_fft > 4096)
    return fail(MMK_ERR_UNSUPPORTED, "%s: n_fft must be a power of two in [64, 4096], got %d", what, n_fft);
  if (hop <= 0 || hop >= n_fft) return fail(MMK_ERR_INVALID, "%s: hop must be in [1, n_fft), got %d", what, hop);
  return MMK_OK;
}

// torch.istft refuses a window whose overlap-added square vanishes somewhere in the kept range ("window overlap add
// min: 1", e.g. hop = n_fft - 1).  Same test on the host: the envelope is periodic in hop away from the clip's ends, so
// the first and last n_fft kept positions and one period in between are all the distinct values.
static int check_envelope(const char* what, int n_fft, int hop, int64_t n_frames) {
  std::vector<float> w2((size_t)n_fft);
  for (int m = 0; m < n_fft; ++m) {
    const float w = 0.5f - 0.5f * cosf(6.283185307179586f * (float)m / (float)n_fft);
    w2[(size_t)m] = w * w;
  }
  const int64_t t_lo = n_fft / 2, t_hi = n_fft / 2 + (int64_t)hop * (n_frames - 1);   // kept positions [t_lo, t_hi)
  auto env = [&](int64_t t) {
    int64_t g_hi = t / hop;
    g_hi = g_hi < n_frames - 1 ? g_hi : n_frames - 1;
    const int64_t g_lo = (t - n_fft + 1 <= 0) ? 0 : (t - n_fft + hop) / hop;
    float e = 0.f;
    for (int64_t g = g_lo; g <= g_hi; ++g) e += w2[(size_t)(t - g * hop)];
    return e;
  };
  float mn = 1e30f;
  const int64_t span = 2 * (int64_t)n_fft + hop;
  if (t_hi - t_lo <= 2 * span) {
    for (int64_t t = t_lo; t < t_hi; ++t) mn = fminf(mn, env(t));
  } else {
    for (int64_t t = t_lo; t < t_lo + span; ++t) mn = fminf(mn, env(t));
    for (int64_t t = t_hi - span; t < t_hi; ++t) mn = fminf(mn, env(t));
  }
  if (!(mn >= 1e-11f))
    return fail(MMK_ERR_INVALID, "%s: the overlap-added squared window vanishes (min %g) for n_fft=%d, hop=%d: torch.istft raises here too",
                what, (double)mn, n_fft, hop);
  return MMK_OK;
}

static unsigned pair_grid(int64_t total_pairs) {
  const int64_t wgs = (total_pairs + kIstftWaves - 1) / kIstftWaves;
  return (unsigned)(wgs < 768 ? wgs : 768);                 // 3 workgroups of 4 waves per CU, all resident
}

// segment length: one wave per segment, about two waves' worth of segments per SIMD pair (2 workgroups of 4 waves per CU)
static void istft_geometry(int batch, int64_t n_frames, int* seg_hops, int* segs_per_clip) {
  const int64_t hops = n_frames - 1;                        // output hops per clip
  const int64_t slots = 2048;
  const int64_t rounds = ((int64_t)batch * hops + slots * 48 - 1) / (slots * 48);
  int64_t per_clip = (slots * rounds + batch - 1) / batch;
  per_clip = per_clip < 1 ? 1 : per_clip;
  int64_t sh = (hops + per_clip - 1) / per_clip;
  sh = sh < 4 ? 4 : sh;
  sh = sh > hops ? hops : sh;
  *seg_hops = (int)sh;
  *segs_per_clip = (int)((hops + sh - 1) / sh);
}

static int launch_istft(const float* spec, const float* mag, int mode, int batch, int64_t n_frames, int hop, float* out, hipStream_t stream) {
  const int64_t n_out = (int64_t)hop * (n_frames - 1);
  if (n_out <= 0) return MMK_OK;
  int seg_hops, segs_per_clip;
  istft_geometry(batch, n_frames, &seg_hops, &segs_per_clip);
  const int64_t total_tasks = (int64_t)batch * segs_per_clip;
  const int64_t wgs = (total_tasks + kIstftWaves - 1) / kIstftWaves;
  const dim3 grid((unsigned)(wgs < 512 ? wgs : 512)), block(64 * kIstftWaves);   // 2 workgroups per CU, all resident
  SpectralTables T;
  MMK_TRY(spectral_tables(stream, &T));
#define MMK_ISTFT_LAUNCH(M)                                                                                                                     \
  do {                                                                                                                                        \
    if (hop == 256 && (reinterpret_cast<uintptr_t>(out) & 15) == 0)                                                                            \
      hipLaunchKernelGGL((istft1024q_kernel<M>), grid, block, 0, stream, T, spec, mag, n_frames, seg_hops, segs_per_clip, total_tasks, n_out, out); \
    else                                                                                                                                      \
      hipLaunchKernelGGL((istft1024_kernel<M>), grid, block, 0, stream, T, spec, mag, n_frames, hop, seg_hops, segs_per_clip, total_tasks, n_out, out); \
  } while (0)
  if (mode == 0) MMK_ISTFT_LAUNCH(0);
  else if (mode == 1) MMK_ISTFT_LAUNCH(1);
  else MMK_ISTFT_LAUNCH(2);
#undef MMK_ISTFT_LAUNCH
  MMK_HIP(hipGetLastError());
  return MMK_OK;
}

int launch_stft1024(const float* x, int64_t x_row_stride, int batch, int64_t n_samples, int hop, int center, int reflect, int out_mode,
                    float* out, float* tprev, float momentum, hipStream_t stream) {
  const int64_t n_frames = mmk_stft_n_frames(n_samples, 1024, hop, center);
  const int64_t pairs_per_row = (n_frames + 1) / 2;
  const int64_t total_pairs = (int64_t)batch * pairs_per_row;
  const int64_t resident = 256 * (4 * MMK_STFT_WPE / kStftWaves);          // workgroups that are resident together on the chip
  // runs of consecutive pairs per clip: about one run per resident wave (at least 2 pairs per run where a clip has them, at most the clip)
  int64_t runs_per_row = (resident * kStftWaves + batch - 1) / batch;
  runs_per_row = runs_per_row > (pairs_per_row + 1) / 2 ? (pairs_per_row + 1) / 2 : runs_per_row;
  runs_per_row = runs_per_row < 1 ? 1 : runs_per_row;
  {   // (whole run lengths: no run is empty)
    const int64_t run_len = (pairs_per_row + runs_per_row - 1) / runs_per_row;
    runs_per_row = (pairs_per_row + run_len - 1) / run_len;
  }
  const int64_t total_runs = (int64_t)batch * runs_per_row;
  const int64_t wgs = (total_runs + kStftWaves - 1) / kStftWaves;
  const dim3 grid((unsigned)(wgs < resident ? wgs : resident)), block(64 * kStftWaves);
  SpectralTables T;
  MMK_TRY(spectral_tables(stream, &T));
#define MMK_STFT_LAUNCH(O) \
  hipLaunchKernelGGL((stft1024_kernel<O>), grid, block, 0, stream, T, x, x_row_stride, n_samples, hop, center, reflect, n_frames, \
                     total_pairs, (int)runs_per_row, out, tprev, momentum)
  switch (out_mode) {
    case 0: MMK_STFT_LAUNCH(0); break;
    case 1: MMK_STFT_LAUNCH(1); break;
    case 2: MMK_STFT_LAUNCH(2); break;
    case 3: MMK_STFT_LAUNCH(3); break;
    default: MMK_STFT_LAUNCH(4); break;
  }
#undef MMK_STFT_LAUNCH
  MMK_HIP(hipGetLastError());
  return MMK_OK;
}

int launch_stft_generic(const float* x, int64_t x_row_stride, int batch, int64_t n_samples, int n_fft, int hop, int center, int reflect,
                        int out_mode, float* out, float* tprev, float momentum, hipStream_t stream) {
  const int64_t n_frames = mmk_stft_n_frames(n_samples, n_fft, hop, center);
  const int64_t total_pairs = (int64_t)batch * ((n_frames + 1) / 2);
  const size_t lds = (size_t)n_fft * (3 * sizeof(float2) + sizeof(float));
  const dim3 grid((unsigned)(total_pairs < 2048 ? total_pairs : 2048)), block(n_fft >= 1024 ? 256 : (n_fft / 4 < 64 ? 64 : n_fft / 4));
#define MMK_STFT_LAUNCH(O) \
  hipLaunchKernelGGL((stft_generic_kernel<O>), grid, block, lds, stream, x, x_row_stride, n_samples, n_fft, hop, center, reflect, n_frames, \
                     total_pairs, out, tprev, momentum)
  switch (out_mode) {
    case 0: MMK_STFT_LAUNCH(0); break;
    case 1: MMK_STFT_LAUNCH(1); break;
    case 2: MMK_STFT_LAUNCH(2); break;
    case 3: MMK_STFT_LAUNCH(3); break;
    default: MMK_STFT_LAUNCH(4); break;
  }
#undef MMK_STFT_LAUNCH
  MMK_HIP(hipGetLastError());
  return MMK_OK;
}

static int launch_istft_generic(const float* spec, const float* mag, int mode, int batch, int64_t n_frames, int n_fft, int hop, float* work,
                                float* out, hipStream_t stream) {
  if (!work) return fail(MMK_ERR_INVALID, "istft: n_fft = %d needs the frame workspace (mmk_istft_workspace_floats)", n_fft);
  const int64_t total_pairs = (int64_t)batch * ((n_frames + 1) / 2);
  const size_t lds = (size_t)n_fft * (3 * sizeof(float2) + sizeof(float));
  const dim3 grid((unsigned)(total_pairs < 2048 ? total_pairs : 2048)), block(n_fft >= 1024 ? 256 : (n_fft / 4 < 64 ? 64 : n_fft / 4));
  if (mode == 0) hipLaunchKernelGGL((istft_frames_generic_kernel<0>), grid, block, lds, stream, spec, mag, n_fft, n_frames, total_pairs, work);
  else if (mode == 1) hipLaunchKernelGGL((istft_frames_generic_kernel<1>), grid, block, lds, stream, spec, mag, n_fft, n_frames, total_pairs, work);
  else hipLaunchKernelGGL((istft_frames_generic_kernel<2>), grid, block, lds, stream, spec, mag, n_fft, n_frames, total_pairs, work);
  MMK_HIP(hipGetLastError());
  const int64_t n_out = (int64_t)hop * (n_frames - 1);
  if (n_out <= 0) return MMK_OK;
  const bool v4 = (hop % 4) == 0;
  const int64_t total = (int64_t)batch * (v4 ? n_out / 4 : n_out);
  int64_t blocks = (total + 255) / 256;
  blocks = blocks > 4096 ? 4096 : blocks;
  const size_t lds2 = (size_t)n_fft * sizeof(float);
  if (v4) hipLaunchKernelGGL((istft_ola_generic_kernel<4>), dim3((unsigned)blocks), dim3(256), lds2, stream, work, n_fft, n_frames, hop, n_out, total, out);
  else hipLaunchKernelGGL((istft_ola_generic_kernel<1>), dim3((unsigned)blocks), dim3(256), lds2, stream, work, n_fft, n_frames, hop, n_out, total, out);
  MMK_HIP(hipGetLastError());
  return MMK_OK;
}

// n_fft = 1024 takes the fused register-FFT kernel, every other size the workgroup-FFT kernels and the frame workspace
static int istft_any(const float* spec, const float* mag, int mode, int batch, int64_t n_frames, int n_fft, int hop, float* work, float* out,
                     hipStream_t stream) {
  if (n_fft == 1024) return launch_istft(spec, mag, mode, batch, n_frames, hop, out, stream);
  if (n_fft == 2048) return launch_istft2048(spec, mag, mode, batch, n_frames, hop, out, stream);
  return launch_istft_generic(spec, mag, mode, batch, n_frames, n_fft, hop, work, out, stream);
}
static int stft_any(const float* x, int64_t x_row_stride, int batch, int64_t n_samples, int n_fft, int hop, int center, int reflect, int out_mode,
                    float* out, float* tprev, float momentum, hipStream_t stream) {
  if (n_fft == 1024) return launch_stft1024(x, x_row_stride, batch, n_samples, hop, center, reflect, out_mode, out, tprev, momentum, stream);
  if (n_fft == 2048 && out_mode != 3) return launch_stft2048(x, x_row_stride, batch, n_samples, hop, center, reflect, out_mode, out, stream);
  return launch_stft_generic(x, x_row_stride, batch, n_samples, n_fft, hop, center, reflect, out_mode, out, tprev, momentum, stream);
}

}  // namespace mmk

extern "C" int mmk_stft_f32(const float* x, int64_t x_row_stride, int32_t batch, int64_t n_samples, int32_t n_fft, int32_t hop,
                            int32_t center, int32_t reflect, int32_t coordinate, float* out, mmk_stream_t stream) {
  using namespace mmk;
  if (!x || !out || batch <= 0) return fail(MMK_ERR_INVALID, "stft: bad arguments");
  if (int rc = check_fft("stft", n_fft, hop)) return rc;
  if (coordinate < 0 || coordinate > 2) return fail(MMK_ERR_INVALID, "stft: coordinate must be 0 (car), 1 (pol) or 2 (angle)");
  if (mmk_stft_n_frames(n_samples, n_fft, hop, center) <= 0)
    return fail(MMK_ERR_INVALID, "stft: input of %lld samples is shorter than one frame", (long long)n_samples);
  if (reflect && center && n_samples <= n_fft / 2)
    return fail(MMK_ERR_INVALID, "stft: reflect padding of %d needs more than %d samples, got %lld", n_fft / 2, n_fft / 2, (long long)n_samples);
  return stft_any(x, x_row_stride, batch, n_samples, n_fft, hop, center, reflect, coordinate, out, nullptr, 0.f, (hipStream_t)stream);
}

extern "C" int64_t mmk_istft_n_samples(int64_t n_frames, int32_t n_fft, int32_t hop) {
  (void)n_fft;
  return n_frames > 0 ? (int64_t)hop * (n_frames - 1) : 0;
}

// windowed frames of the two-kernel path; the n_fft = 1024 kernel overlap-adds in LDS and needs none
extern "C" size_t mmk_istft_workspace_floats(int32_t batch, int64_t n_frames, int32_t n_fft) {
  return (n_fft == 1024 || n_fft == 2048) ? 0 : (size_t)batch * (size_t)n_frames * (size_t)n_fft;
}

extern "C" int mmk_istft_f32(const float* spec, int32_t coordinate, int32_t batch, int64_t n_frames, int32_t n_fft, int32_t hop,
                             float* work, float* out, mmk_stream_t stream) {
  using namespace mmk;
  if (!spec || !out || batch <= 0 || n_frames <= 0) return fail(MMK_ERR_INVALID, "istft: bad arguments");
  if (int rc = check_fft("istft", n_fft, hop)) return rc;
  if (coordinate != 0 && coordinate != 1) return fail(MMK_ERR_INVALID, "istft: coordinate must be 0 (re, im) or 1 (mag, angle)");
  if (n_frames < 2) return fail(MMK_ERR_INVALID, "istft: one frame leaves no samples after the centre trim");
  if (int rc = check_envelope("istft", n_fft, hop, n_frames)) return rc;
  return istft_any(spec, nullptr, coordinate, batch, n_frames, n_fft, hop, work, out, (hipStream_t)stream);
}

extern "C" size_t mmk_gla_workspace_floats(int32_t batch, int64_t n_frames, int32_t n_fft, int32_t hop) {
  const size_t bins = (size_t)n_fft / 2 + 1;
  const size_t wave = (size_t)hop * (size_t)(n_frames > 0 ? n_frames - 1 : 0);
  // n_fft = 1024 / 2048: two waveforms and two previous spectra (ping-pong of the fused iteration kernels);
  // other sizes: the waveform, phase estimates, previous spectrum and the windowed frames
  return (size_t)batch * (((n_fft == 1024 || n_fft == 2048) ? 2 : 1) * wave + 4 * (size_t)n_frames * bins) +
         mmk_istft_workspace_floats(batch, n_frames, n_fft);
}

extern "C" int mmk_gla_f32(const float* mag, const float* init, int32_t batch, int64_t n_frames, int32_t n_fft, int32_t hop, int32_t n_iter,
                           float momentum, float* work, float* out, mmk_stream_t stream) {
  using namespace mmk;
  if (!mag || !work || !out || batch <= 0 || n_frames <= 0 || n_iter < 0) return fail(MMK_ERR_INVALID, "gla: bad arguments");
  if (int rc = check_fft("gla", n_fft, hop)) return rc;
  if (!(momentum >= 0.f && momentum < 1.f)) return fail(MMK_ERR_INVALID, "gla: momentum must be in [0, 1), got %g", (double)momentum);
  const int64_t n_out = (int64_t)hop * (n_frames - 1);
  if (n_out <= n_fft / 2) return fail(MMK_ERR_INVALID, "gla: %lld frames give %lld samples, reflect padding needs more than %d",
                                      (long long)n_frames, (long long)n_out, n_fft / 2);
  if (int rc = check_envelope("gla", n_fft, hop, n_frames)) return rc;
  hipStream_t s = (hipStream_t)stream;
  const size_t bins = (size_t)n_fft / 2 + 1;
  const size_t spec_floats = 2 * (size_t)batch * n_frames * bins;
  const float m = momentum / (1.f + momentum);
  auto fill_ones = [&](float* dst) -> int {                  // rand_init=False: every phase estimate starts at 1 + 0i
    const int64_t n = (int64_t)batch * n_frames * (int64_t)bins;
    hipLaunchKernelGGL(fill_complex_kernel, dim3((unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256)), dim3(256), 0, s, dst, n,
                       cf32{1.f, 0.f});
    MMK_HIP(hipGetLastError());
    return MMK_OK;
  };
  if (n_fft == 1024 || n_fft == 2048) {
    // one launch per iteration: stft -> phase update -> istft fused per output segment (gla1024_iter_kernel / gla2048_iter_kernel)
    float* wave_a = work;
    float* wave_b = wave_a + (size_t)batch * n_out;
    float* tprev_a = wave_b + (size_t)batch * n_out;
    float* tprev_b = tprev_a + spec_floats;
    const float* angles0 = init;
    if (!init) {                                             // tprev_b holds the initial estimates until iteration 0 overwrites it
      if (int rc = fill_ones(tprev_b)) return rc;
      angles0 = tprev_b;
    }
    if (int rc = istft_any(angles0, mag, 2, batch, n_frames, n_fft, hop, nullptr, n_iter ? wave_a : out, s)) return rc;
    if (!n_iter) return MMK_OK;
    MMK_HIP(hipMemsetAsync(tprev_a, 0, spec_floats * sizeof(float), s));
    int seg_hops = 0, segs_per_clip = 0;
    int64_t total_tasks = 0;
    dim3 grid(1), block(64 * kIstftWaves);
    if (n_fft == 1024) {
      istft_geometry(batch, n_frames, &seg_hops, &segs_per_clip);
      total_tasks = (int64_t)batch * segs_per_clip;
      const int64_t wgs = (total_tasks + kIstftWaves - 1) / kIstftWaves;
      grid = dim3((unsigned)(wgs < 512 ? wgs : 512));
    }
    const float* wave_in = wave_a;
    float* wave_other = wave_b;
    const float* tin = tprev_a;
    float* tout = tprev_b;
    SpectralTables T;
    MMK_TRY(spectral_tables(s, &T));
    for (int it = 0; it < n_iter; ++it) {
      float* dst = it == n_iter - 1 ? out : wave_other;
      if (n_fft == 1024 && hop == 256 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
        hipLaunchKernelGGL(gla1024q_iter_kernel, grid, block, 0, s, T, wave_in, mag, tin, tout, m, n_frames, seg_hops, segs_per_clip, total_tasks, n_out, dst);
        MMK_HIP(hipGetLastError());
      } else if (n_fft == 1024) {
        hipLaunchKernelGGL(gla1024_iter_kernel, grid, block, 0, s, T, wave_in, mag, tin, tout, m, n_frames, hop, seg_hops, segs_per_clip,
                           total_tasks, n_out, dst);
        MMK_HIP(hipGetLastError());
      } else if (int rc = launch_gla2048_iter(wave_in, mag, tin, tout, m, batch, n_frames, hop, dst, s)) {
        return rc;
      }
      wave_other = const_cast<float*>(wave_in);
      wave_in = dst;
      float* t = const_cast<float*>(tin);
      tin = tout;
      tout = t;
    }
    return MMK_OK;
  }
  float* wave = work;
  float* angles = wave + (size_t)batch * n_out;
  float* tprev = angles + spec_floats;
  float* frames = tprev + spec_floats;
  if (init) MMK_HIP(hipMemcpyAsync(angles, init, spec_floats * sizeof(float), hipMemcpyDeviceToDevice, s));
  else if (int rc = fill_ones(angles)) return rc;
  MMK_HIP(hipMemsetAsync(tprev, 0, spec_floats * sizeof(float), s));
  for (int it = 0; it < n_iter; ++it) {
    if (int rc = istft_any(angles, mag, 2, batch, n_frames, n_fft, hop, frames, wave, s)) return rc;
    if (int rc = stft_any(wave, n_out, batch, n_out, n_fft, hop, 1, 1, 3, angles, tprev, m, s)) return rc;
  }
  return istft_any(angles, mag, 2, batch, n_frames, n_fft, hop, frames, out, s);
}

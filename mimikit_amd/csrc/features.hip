// Feature functionals of the generate path: mu-law quantise / expand and the
// framed periodic-Hann STFT magnitude (MagSpec).  All HBM-bound streaming work:
// coalesced 16 B/lane accesses, tables and frame tiles staged in LDS.
#include <stdlib.h>

#include "mmk_common.h"

namespace mmk {

// ---- mu-law -----------------------------------------------------------------------
// The reference formula (features/functionals.py:330-338) is evaluated in fp32
// with torch's CPU log1p; a 1-ulp different log1p flips codes at bin edges.  The
// kernel therefore only uses the device log1pf to get a candidate code and then
// settles it against the q-1 decision thresholds of the reference formula
// (built once on the host, mimikit_amd/features/functionals.py), which makes
// in-range codes exact by construction.
constexpr int kMuLawLdsLevels = 8192;   // tables up to 32 KiB are staged in LDS (the launch asks for exactly that much)

__device__ __forceinline__ int64_t mulaw_code(float x, float mu, float C, float inv_log, const float* edges, int q) {
  const float ax = fabsf(x);
  const float sgn = (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f);
  if (edges != nullptr && ax <= 1.f) {
    // in range: the hardware log only proposes a candidate (within one code); the decision thresholds
    // of the reference formula settle it, so the result does not depend on this log's rounding
    const float y = sgn * __logf(1.f + mu * ax * C) * inv_log;
    int ci = (int)((y + 1.f) * 0.5f * mu + 0.5f);
    ci = ci < 0 ? 0 : (ci > q - 1 ? q - 1 : ci);
    while (ci > 0 && x < edges[ci - 1]) --ci;
    while (ci < q - 1 && x >= edges[ci]) ++ci;
    return ci;
  }
  // out of range (or no table): direct fp32 evaluation, no clamp, as the reference
  const float y = sgn * log1pf(mu * ax * C) / log1pf(mu * C);
  const float v = (y + 1.f) / 2.f * mu + 0.5f;
  return (int64_t)v;  // trunc toward zero, as tensor.to(int64)
}

__global__ __launch_bounds__(256) void mulaw_compress_kernel(const float* __restrict__ x, int64_t* __restrict__ codes,
                                                            int64_t n, int q, float C, const float* __restrict__ edges) {
  extern __shared__ float s_edges[];
  const bool have = edges != nullptr;
  const bool staged = have && q <= kMuLawLdsLevels;   // larger tables stay in global memory (L1 / L2 serve them)
  if (staged)
    for (int i = threadIdx.x; i < q - 1; i += blockDim.x) s_edges[i] = edges[i];
  __syncthreads();
  const float mu = (float)(q - 1);
  const float inv_log = 1.f / log1pf(mu * C);
  const float* e = staged ? s_edges : edges;
  const int64_t n4 = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const bool aligned = ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && ((reinterpret_cast<uintptr_t>(codes) & 15) == 0);
  if (aligned) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += stride) {
      const float4 v = reinterpret_cast<const float4*>(x)[i];
      longlong2 o0, o1;
      o0.x = mulaw_code(v.x, mu, C, inv_log, e, q);
      o0.y = mulaw_code(v.y, mu, C, inv_log, e, q);
      o1.x = mulaw_code(v.z, mu, C, inv_log, e, q);
      o1.y = mulaw_code(v.w, mu, C, inv_log, e, q);
      reinterpret_cast<longlong2*>(codes)[2 * i] = o0;
      reinterpret_cast<longlong2*>(codes)[2 * i + 1] = o1;
    }
    for (int64_t i = (n4 << 2) + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += stride)
      codes[i] = mulaw_code(x[i], mu, C, inv_log, e, q);
  } else {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += stride)
      codes[i] = mulaw_code(x[i], mu, C, inv_log, e, q);
  }
}

// The streaming form for the usual case (16-byte aligned buffers, table in LDS): four 16-byte loads per lane in flight before any
// arithmetic (64 KB per CU at full occupancy instead of 16), the two thresholds around a candidate as ONE 8-byte LDS read
// (pairs (edges[c - 1], edges[c]) with -inf / +inf at the ends, so the settle loop needs no index checks and usually runs once),
// plain loads and stores: with the non-temporal hint on both the kernel ran at 3.96 TB/s instead of 4.80 (the general kernel: 4.72).
__device__ __forceinline__ int64_t mulaw_code_pairs(float x, float mu, float C, float inv_log, const float2* ep, const float* edges, int q) {
  const float ax = fabsf(x);
  if (!(ax <= 1.f)) return mulaw_code(x, mu, C, inv_log, nullptr, q);      // out of range / NaN: the direct formula, as the reference
  const float sgn = (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f);
  const float y = sgn * __logf(1.f + mu * ax * C) * inv_log;
  int ci = (int)((y + 1.f) * 0.5f * mu + 0.5f);
  ci = ci < 0 ? 0 : (ci > q - 1 ? q - 1 : ci);
  for (;;) {
    const float2 b = ep[ci];
    if (x < b.x) --ci;
    else if (x >= b.y) ++ci;
    else break;
  }
  return ci;
}

typedef float mu_f32x4 __attribute__((ext_vector_type(4)));
typedef long long mu_i64x2 __attribute__((ext_vector_type(2)));

// Stores: a lane's four codes are 32 bytes of the output; written as the lane's own two 16-byte stores, a store INSTRUCTION covers 64 pieces of
// 16 bytes that lie 32 bytes apart - half of every 128-byte line, the other half coming with the next instruction.  A read-4-write-8 stream with
// that pattern moves 5.2 - 5.4 TB/s on this chip, with store instructions that cover 1 KB of consecutive lanes each 6.0 - 6.2
// (scripts/probes/rw_mix.hip; copy 5.9, fill 6.8, read 7.2).  So the wave first passes the codes between lanes (in-range codes are < q <= 4096: two
// per 32-bit word, two ds_bpermute per store): store A takes samples 2 l, 2 l + 1 of the wave's 256 from lane l / 2, store B samples
// 128 + 2 l .. from lane 32 + l / 2.  A wave that holds an out-of-range sample (codes beyond 16 bits: the reference does not clamp), or
// that reaches past the end, keeps the lane's own stores for that round.
__global__ __launch_bounds__(256) void mulaw_compress_stream_kernel(const mu_f32x4* __restrict__ x4, mu_i64x2* __restrict__ codes2, int64_t n4, int q,
                                                                   float C, const float* __restrict__ edges) {
  extern __shared__ float2 s_pairs[];
  for (int i = threadIdx.x; i < q; i += blockDim.x)
    s_pairs[i] = make_float2(i > 0 ? edges[i - 1] : -__builtin_inff(), i < q - 1 ? edges[i] : __builtin_inff());
  __syncthreads();
  const float mu = (float)(q - 1);
  const float inv_log = 1.f / log1pf(mu * C);
  constexpr int kU = 4;
  const int lane = threadIdx.x & 63;
  const int64_t tile = (int64_t)blockDim.x * kU;                    // float4s per workgroup and round
  for (int64_t base = (int64_t)blockIdx.x * tile; base < n4; base += (int64_t)gridDim.x * tile) {
    mu_f32x4 v[kU];
#pragma unroll
    for (int k = 0; k < kU; ++k) {
      const int64_t i = base + k * blockDim.x + threadIdx.x;
      v[k] = i < n4 ? x4[i] : mu_f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int k = 0; k < kU; ++k) {
      const int64_t i = base + k * blockDim.x + threadIdx.x;
      const int64_t w0 = base + k * blockDim.x + (threadIdx.x & ~63);       // the wave's first float4 of this round
      const bool in_range = fabsf(v[k][0]) <= 1.f && fabsf(v[k][1]) <= 1.f && fabsf(v[k][2]) <= 1.f && fabsf(v[k][3]) <= 1.f;
      const bool exchange = w0 + 64 <= n4 && __all(in_range);               // (wave-uniform)
      long long c[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) c[e] = mulaw_code_pairs(v[k][e], mu, C, inv_log, s_pairs, edges, q);
      if (exchange) {
        const int lo = (int)c[0] | ((int)c[1] << 16), hi = (int)c[2] | ((int)c[3] << 16);
        const int src_a = (lane >> 1) << 2, src_b = (32 + (lane >> 1)) << 2;
        const int a_lo = __builtin_amdgcn_ds_bpermute(src_a, lo), a_hi = __builtin_amdgcn_ds_bpermute(src_a, hi);
        const int b_lo = __builtin_amdgcn_ds_bpermute(src_b, lo), b_hi = __builtin_amdgcn_ds_bpermute(src_b, hi);
        const int pa = (lane & 1) ? a_hi : a_lo, pb = (lane & 1) ? b_hi : b_lo;
        codes2[2 * w0 + lane] = mu_i64x2{(long long)(pa & 0xffff), (long long)((unsigned)pa >> 16)};
        codes2[2 * w0 + 64 + lane] = mu_i64x2{(long long)(pb & 0xffff), (long long)((unsigned)pb >> 16)};
      } else if (i < n4) {
        codes2[2 * i] = mu_i64x2{c[0], c[1]};
        codes2[2 * i + 1] = mu_i64x2{c[2], c[3]};
      }
    }
  }
}

__device__ __forceinline__ float mulaw_value(int64_t code, float mu, float C, float logv, const float* table, int q) {
  if (table != nullptr && code >= 0 && code < q) return table[code];
  // x = code/mu*2-1 ; sign(x)*(exp(|x|*log1p(mu*C))-1)/(mu*C)     (functionals.py:361-369)
  const float xx = ((float)code / mu) * 2.f - 1.0f;
  const float sgn = (xx > 0.f) ? 1.f : ((xx < 0.f) ? -1.f : 0.f);
  return sgn * (expf(fabsf(xx) * logv) - 1.0f) / (mu * C);
}

__global__ __launch_bounds__(256) void mulaw_expand_kernel(const int64_t* __restrict__ codes, float* __restrict__ x,
                                                          int64_t n, int q, float C, const float* __restrict__ table) {
  extern __shared__ float s_table[];
  const bool have = table != nullptr;
  const bool staged = have && q <= kMuLawLdsLevels;
  if (staged)
    for (int i = threadIdx.x; i < q; i += blockDim.x) s_table[i] = table[i];
  __syncthreads();
  const float mu = (float)(q - 1);
  const float logv = log1pf(mu * C);
  const float* t = staged ? s_table : table;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t n2 = n >> 1;
  const bool aligned = ((reinterpret_cast<uintptr_t>(x) & 7) == 0) && ((reinterpret_cast<uintptr_t>(codes) & 15) == 0);
  if (aligned) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n2; i += stride) {
      const longlong2 c = reinterpret_cast<const longlong2*>(codes)[i];
      float2 o;
      o.x = mulaw_value(c.x, mu, C, logv, t, q);
      o.y = mulaw_value(c.y, mu, C, logv, t, q);
      reinterpret_cast<float2*>(x)[i] = o;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) x[n - 1] = mulaw_value(codes[n - 1], mu, C, logv, t, q);
  } else {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += stride)
      x[i] = mulaw_value(codes[i], mu, C, logv, t, q);
  }
}

// ---- band-limited resampling ---------------------------------------------------------
// Resample.torch_func (features/functionals.py:305-306) = torchaudio.functional.resample: a polyphase windowed-sinc FIR.
// With orig = orig_sr / gcd and new = new_sr / gcd, output sample n * new + j of a row is
//     sum_k kernel[j][k] * x[n * orig + k - width]            (zero outside the row), k < 2 width + orig,
// for n < ceil(T / orig), cut to ceil(new * T / orig) samples.  The (new, 2 width + orig) kernel table is built on the host
// exactly as torchaudio builds it (float64, Hann-windowed sinc, rolloff 0.99, lowpass_filter_width 6;
// mimikit_amd/features/functionals.py) - a filter table like the FFT's twiddles.  One workgroup per (row, block of input
// steps): the input span is staged in LDS once and every output phase reads it from there; the table streams from L2.
constexpr int kRsSteps = 8;          // input steps (of `orig` samples) per workgroup

__global__ __launch_bounds__(256) void resample_kernel(const float* __restrict__ x, int64_t x_row_stride, int64_t n_in,
                                                      const float* __restrict__ table, int orig, int nnew, int width,
                                                      float* __restrict__ out, int64_t out_row_stride, int64_t n_out) {
  extern __shared__ float s_x[];
  const int row = blockIdx.y;
  const int64_t step0 = (int64_t)blockIdx.x * kRsSteps;
  const int taps = 2 * width + orig;
  const int span = (kRsSteps - 1) * orig + taps;                 // input samples the block's outputs read
  const float* xr = x + (int64_t)row * x_row_stride;
  const int64_t first = step0 * orig - width;
  for (int i = threadIdx.x; i < span; i += blockDim.x) {
    const int64_t t = first + i;
    s_x[i] = (t >= 0 && t < n_in) ? xr[t] : 0.f;
  }
  __syncthreads();
  float* orow = out + (int64_t)row * out_row_stride;
  for (int o = threadIdx.x; o < kRsSteps * nnew; o += blockDim.x) {
    const int st = o / nnew, j = o - st * nnew;
    const int64_t idx = (step0 + st) * nnew + j;
    if (idx >= n_out) continue;
    const float* w = table + (int64_t)j * taps;
    const float* xs = s_x + st * orig;
    float acc = 0.f;
    for (int k = 0; k < taps; ++k) acc = fmaf(w[k], xs[k], acc);
    orow[idx] = acc;
  }
}

// (the framed STFT kernels live in istft.hip)

}  // namespace mmk

extern "C" int mmk_mulaw_compress_f32_i64(const float* x, int64_t* codes, int64_t n, int32_t q_levels,
                                          float compression, const float* edges, mmk_stream_t stream) {
  using namespace mmk;
  if (n == 0) return MMK_OK;
  if (!x || !codes || n < 0 || q_levels < 2 || q_levels > 65536)
    return fail(MMK_ERR_INVALID, "mulaw_compress: bad arguments (n=%lld, q_levels=%d)", (long long)n, q_levels);
  const bool aligned = ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && ((reinterpret_cast<uintptr_t>(codes) & 15) == 0);
  if (edges && aligned && q_levels <= kMuLawLdsLevels / 2 && n >= (1 << 16)) {
    const int64_t n4 = n >> 2;
    int64_t sblocks = (n4 + 1023) / 1024;
    sblocks = sblocks > 8192 ? 8192 : sblocks;       // (1024 .. 8192 workgroups: +3 % from the first to the last, scripts/probes/rw_mix.hip)
    hipLaunchKernelGGL(mulaw_compress_stream_kernel, dim3((unsigned)sblocks), dim3(256), (size_t)q_levels * sizeof(float2), (hipStream_t)stream,
                       reinterpret_cast<const mu_f32x4*>(x), reinterpret_cast<mu_i64x2*>(codes), n4, q_levels, compression, edges);
    MMK_HIP(hipGetLastError());
    if ((n & 3) == 0) return MMK_OK;
    x += n4 * 4; codes += n4 * 4; n &= 3;            // the last 1 - 3 samples: the general kernel
  }
  int64_t blocks = (n / 4 + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
  hipLaunchKernelGGL(mulaw_compress_kernel, dim3((unsigned)blocks), dim3(256),
                     (size_t)(q_levels <= kMuLawLdsLevels ? q_levels : 1) * sizeof(float), (hipStream_t)stream, x, codes, n, q_levels,
                     compression, edges);
  MMK_HIP(hipGetLastError());
  return MMK_OK;
}

extern "C" int mmk_mulaw_expand_i64_f32(const int64_t* codes, float* x, int64_t n, int32_t q_levels, float compression,
                                        const float* table, mmk_stream_t stream) {
  using namespace mmk;
  if (n == 0) return MMK_OK;
  if (!x || !codes || n < 0 || q_levels < 2 || q_levels > 65536)
    return fail(MMK_ERR_INVALID, "mulaw_expand: bad arguments (n=%lld, q_levels=%d)", (long long)n, q_levels);
  int64_t blocks = (n / 2 + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
  hipLaunchKernelGGL(mulaw_expand_kernel, dim3((unsigned)blocks), dim3(256),
                     (size_t)(q_levels <= kMuLawLdsLevels ? q_levels : 1) * sizeof(float), (hipStream_t)stream, codes, x, n, q_levels,
                     compression, table);
  MMK_HIP(hipGetLastError());
  return MMK_OK;
}

extern "C" int64_t mmk_resample_n_out(int64_t n_in, int32_t orig, int32_t nnew) {
  if (orig <= 0 || nnew <= 0 || n_in < 0) return 0;
  return (nnew * n_in + orig - 1) / orig;          // ceil(new * T / orig), as torchaudio cuts its output
}

extern "C" int mmk_resample_f32(const float* x, int64_t x_row_stride, int32_t batch, int64_t n_in, const float* table, int32_t orig,
                                int32_t nnew, int32_t width, float* out, int64_t out_row_stride, mmk_stream_t stream) {
  using namespace mmk;
  if (!x || !out || !table || batch <= 0 || n_in <= 0 || orig <= 0 || nnew <= 0 || width < 0)
    return fail(MMK_ERR_INVALID, "resample: bad arguments");
  const int taps = 2 * width + orig;
  const size_t lds = (size_t)((kRsSteps - 1) * orig + taps) * sizeof(float);
  if (lds > 160 * 1024) return fail(MMK_ERR_UNSUPPORTED, "resample: %d / %d needs %zu bytes of LDS per workgroup", orig, nnew, lds);
  const int64_t n_out = mmk_resample_n_out(n_in, orig, nnew);
  const int64_t steps = (n_in + orig - 1) / orig;
  dim3 grid((unsigned)((steps + kRsSteps - 1) / kRsSteps), (unsigned)batch);
  hipLaunchKernelGGL(resample_kernel, grid, dim3(256), lds, (hipStream_t)stream, x, x_row_stride, n_in, table, orig, nnew, width, out,
                     out_row_stride, n_out);
  MMK_HIP(hipGetLastError());
  return MMK_OK;
}

extern "C" int64_t mmk_stft_n_frames(int64_t n_samples, int32_t n_fft, int32_t hop, int32_t center) {
  if (hop <= 0 || n_fft <= 0) return 0;
  const int64_t padded = n_samples + (center ? (int64_t)(n_fft / 2) * 2 : 0);
  if (padded < n_fft) return 0;
  return 1 + (padded - n_fft) / hop;
}

extern "C" int mmk_stft_mag_f32(const float* x, int64_t x_row_stride, int32_t batch, int64_t n_samples, int32_t n_fft,
                                int32_t hop, int32_t center, float* out, mmk_stream_t stream) {
  using namespace mmk;
  if (!x || !out || batch <= 0 || hop <= 0) return fail(MMK_ERR_INVALID, "stft: bad arguments");
  int log2n = 0;
  while ((1 << log2n) < n_fft) ++log2n;
  if ((1 << log2n) != n_fft || n_fft < 64 || n_fft > 4096)
    return fail(MMK_ERR_UNSUPPORTED, "stft: n_fft must be a power of two in [64, 4096], got %d", n_fft);
  const int64_t n_frames = mmk_stft_n_frames(n_samples, n_fft, hop, center);
  if (n_frames <= 0) return fail(MMK_ERR_INVALID, "stft: input of %lld samples is shorter than one frame", (long long)n_samples);
  if (n_fft == 1024)     // register-resident variant, one pair per wave (istft.hip)
    return launch_stft1024(x, x_row_stride, batch, n_samples, hop, center, 0, 4, out, nullptr, 0.f, (hipStream_t)stream);
  if (n_fft == 2048)     // one frame per wave through the same 1024-point transform (spectral2048.hip)
    return launch_stft2048(x, x_row_stride, batch, n_samples, hop, center, 0, 4, out, (hipStream_t)stream);
  return launch_stft_generic(x, x_row_stride, batch, n_samples, n_fft, hop, center, 0, 4, out, nullptr, 0.f, (hipStream_t)stream);
}

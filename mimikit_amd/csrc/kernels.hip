// Small per-step kernels of the generate path: embedding gather, categorical
// head (temperature column + argmax / inverse-CDF sampling), recurrent cells,
// and the position counter every step kernel reads.
#include <stdarg.h>

#include "mmk_common.h"

namespace mmk {

// ---- error plumbing ---------------------------------------------------------
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

// ---- position counter ---------------------------------------------------------
__global__ void bump_kernel(int64_t* p, int64_t inc) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *p += inc;
}
__global__ void set_i64_kernel(int64_t* p, int64_t v) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *p = v;
}
int launch_bump(int64_t* tau_ptr, int64_t inc, hipStream_t stream) {
  hipLaunchKernelGGL(bump_kernel, dim3(1), dim3(64), 0, stream, tau_ptr, inc);
  MMK_HIP(hipGetLastError());
  return MMK_OK;
}
int launch_set_i64(int64_t* p, int64_t v, hipStream_t stream) {
  hipLaunchKernelGGL(set_i64_kernel, dim3(1), dim3(64), 0, stream, p, v);
  MMK_HIP(hipGetLastError());
  return MMK_OK;
}

__global__ void fill_kernel(float* p, float v, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    p[i] = v;
}
int launch_fill(float* p, float v, int64_t n, hipStream_t stream) {
  if (n <= 0) return MMK_OK;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(fill_kernel, dim3(blocks), dim3(256), 0, stream, p, v, n);
  MMK_HIP(hipGetLastError());
  return MMK_OK;
}

// ---- nn.Embedding row gather (modules/io.py:148-154) ---------------------------
__global__ void embed_kernel(const int64_t* __restrict__ idx, int64_t idx_row_stride, int64_t idx_tau_off,
                             const float* __restrict__ table, int C, int q_levels, Addr out, int64_t out_ld,
                             int M, const int64_t* tau_ptr, int64_t tau_off) {
  const int64_t tau = (tau_ptr ? *tau_ptr : 0) + tau_off;
  const int m = blockIdx.x;
  if (m >= M) return;
  int64_t cls = idx[m * idx_row_stride + tau + idx_tau_off];
  // torch raises on out-of-range indices; keep memory safe and make it visible (NaN row)
  const bool ok = cls >= 0 && cls < q_levels;
  float* o = (float*)out.base + addr_elems(out, tau) + (int64_t)m * out_ld;
  for (int c = threadIdx.x; c < C; c += blockDim.x) o[c] = ok ? table[cls * C + c] : __builtin_nanf("");
}

int launch_embed(const int64_t* idx, int64_t idx_row_stride, int64_t idx_tau_off, const float* table, int C,
                 int q_levels, Addr out, int64_t out_ld, int M, const int64_t* tau_ptr, int64_t tau_off,
                 hipStream_t stream) {
  int threads = C >= 256 ? 256 : (C > 64 ? 128 : 64);
  hipLaunchKernelGGL(embed_kernel, dim3(M), dim3(threads), 0, stream, idx, idx_row_stride, idx_tau_off, table, C,
                     q_levels, out, out_ld, M, tau_ptr, tau_off);
  MMK_HIP(hipGetLastError());
  return MMK_OK;
}

__global__ void copy_rows_kernel(Addr src, int64_t src_ld, Addr dst, int64_t dst_ld, int M, int C,
                                 const int64_t* tau_ptr, int64_t tau_off) {
  const int64_t tau = (tau_ptr ? *tau_ptr : 0) + tau_off;
  const int m = blockIdx.x;
  const float* s = (const float*)src.base + addr_elems(src, tau) + (int64_t)m * src_ld;
  float* d = (float*)dst.base + addr_elems(dst, tau) + (int64_t)m * dst_ld;
  for (int c = threadIdx.x; c < C; c += blockDim.x) d[c] = s[c];
}

__global__ void add_rows_kernel(Addr src, int64_t src_ld, Addr dst, int64_t dst_ld, int M, int C, const int64_t* tau_ptr,
                                int64_t tau_off) {
  const int64_t tau = (tau_ptr ? *tau_ptr : 0) + tau_off;
  const int m = blockIdx.x;
  const float* s = (const float*)src.base + addr_elems(src, tau) + (int64_t)m * src_ld;
  float* d = (float*)dst.base + addr_elems(dst, tau) + (int64_t)m * dst_ld;
  for (int c = threadIdx.x; c < C; c += blockDim.x) d[c] = d[c] + s[c];
}

// dst[m][:] = dst[m][:] + src[m][:]   (WaveNet layerwise_inputs: every layer output + the embedded input, wavenet_v2.py:285-286)
int launch_add_rows(Addr src, int64_t src_ld, Addr dst, int64_t dst_ld, int M, int C, const int64_t* tau_ptr, int64_t tau_off,
                    hipStream_t stream) {
  hipLaunchKernelGGL(add_rows_kernel, dim3(M), dim3(C >= 256 ? 256 : 64), 0, stream, src, src_ld, dst, dst_ld, M, C, tau_ptr, tau_off);
  MMK_HIP(hipGetLastError());
  return MMK_OK;
}

// dst[m][c] = P[m][c] * P[m][C + c] + P[m][2 C + c]      (ParametrizedLinear, networks/parametrized.py:45-47)
__global__ void affine_rows_kernel(const float* __restrict__ P, int64_t p_ld, Addr dst, int64_t dst_ld, int M, int C, const int64_t* tau_ptr,
                                   int64_t tau_off) {
  const int64_t tau = (tau_ptr ? *tau_ptr : 0) + tau_off;
  const int m = blockIdx.x;
  const float* s = P + (int64_t)m * p_ld;
  float* d = (float*)dst.base + addr_elems(dst, tau) + (int64_t)m * dst_ld;
  for (int c = threadIdx.x; c < C; c += blockDim.x) d[c] = s[c] * s[C + c] + s[2 * C + c];
}

int launch_affine_rows(const float* P, int64_t p_ld, Addr dst, int64_t dst_ld, int M, int C, const int64_t* tau_ptr, int64_t tau_off,
                       hipStream_t stream) {
  hipLaunchKernelGGL(affine_rows_kernel, dim3(M), dim3(C >= 256 ? 256 : 64), 0, stream, P, p_ld, dst, dst_ld, M, C, tau_ptr, tau_off);
  MMK_HIP(hipGetLastError());
  return MMK_OK;
}

int launch_copy_rows(Addr src, int64_t src_ld, Addr dst, int64_t dst_ld, int M, int C, const int64_t* tau_ptr,
                     int64_t tau_off, hipStream_t stream) {
  hipLaunchKernelGGL(copy_rows_kernel, dim3(M), dim3(C >= 256 ? 256 : 64), 0, stream, src, src_ld, dst, dst_ld, M,
                     C, tau_ptr, tau_off);
  MMK_HIP(hipGetLastError());
  return MMK_OK;
}

// ---- categorical head -----------------------------------------------------------
// One wavefront per row.  Lane i owns the contiguous classes [i*per, (i+1)*per),
// so the inclusive CDF is a lane-local running sum plus a wave exclusive scan.
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

constexpr int kMaxPerLane = 16;  // classes <= 1024

__global__ __launch_bounds__(64) void sample_kernel(const SampleArgs a) {
  const int row = blockIdx.x;
  const int lane = threadIdx.x;
  const int64_t tau = (a.tau_ptr ? *a.tau_ptr : 0) + a.tau_off;
  const float* lg = a.logits + (int64_t)row * a.ld;
  const int nc = a.n_classes;
  const int per = (nc + 63) / 64;

  float denom = 1.f;
  if (a.has_temp_col) {
    // logits[..., :-1] / maximum(sigmoid(logits[..., -1:]), min_temp)      (mlp.py:60-62)
    const float t = sigmoidf_(lg[nc]);
    denom = fmaxf(t, a.min_temp);
  }
  float v[kMaxPerLane];
#pragma unroll
  for (int j = 0; j < kMaxPerLane; ++j) {
    const int c = lane * per + j;
    v[j] = (j < per && c < nc) ? (a.has_temp_col ? lg[c] / denom : lg[c]) : -INFINITY;
  }

  int64_t result;
  if (a.temperature == nullptr) {
    // argmax, first maximum wins (torch.argmax)
    float best = -INFINITY;
    int bi = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < kMaxPerLane; ++j) {
      const int c = lane * per + j;
      if (j < per && c < nc && (v[j] > best || bi == 0x7fffffff)) {
        if (v[j] > best || bi == 0x7fffffff) { best = v[j]; bi = c; }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o);
      const int oi = __shfl_xor(bi, o);
      if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    result = bi;
  } else {
    // logits / T ; softmax ; inverse CDF at u            (targets.py:43-52)
    const float T = a.temperature[row];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < kMaxPerLane; ++j) {
      if (j < per) { v[j] = v[j] / T; mx = fmaxf(mx, v[j]); }
    }
    mx = wave_max(mx);
    float local = 0.f;
#pragma unroll
    for (int j = 0; j < kMaxPerLane; ++j) {
      if (j < per) { v[j] = expf(v[j] - mx); local += v[j]; }
    }
    // inclusive scan across lanes
    float incl = local;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const float up = __shfl_up(incl, o);
      if (lane >= o) incl += up;
    }
    const float total = __shfl(incl, 63);
    const float u = a.uniforms[(int64_t)row * a.uniform_ld + tau + a.uni_off];
    const float target = u * total;
    float run = incl - local;
    int pick = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < kMaxPerLane; ++j) {
      const int c = lane * per + j;
      if (j < per && c < nc) {
        run += v[j];
        if (pick == 0x7fffffff && run > target && v[j] > 0.f) pick = c;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const int other = __shfl_xor(pick, o);
      pick = other < pick ? other : pick;
    }
    if (pick == 0x7fffffff) {
      // u*total rounded above the last partial sum: take the last class with mass
      int last = -1;
#pragma unroll
      for (int j = 0; j < kMaxPerLane; ++j) {
        const int c = lane * per + j;
        if (j < per && c < nc && v[j] > 0.f) last = c;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const int other = __shfl_xor(last, o);
        last = other > last ? other : last;
      }
      pick = last < 0 ? 0 : last;
    }
    result = pick;
  }
  if (lane == 0) {
    if (a.group > 0) {
      if (row % a.group < a.kept) a.out[(int64_t)(row / a.group) * a.group_stride + (int64_t)(row % a.group) * a.out_row_stride + tau + a.out_tau_off] = result;
    } else {
      a.out[(int64_t)row * a.out_row_stride + tau + a.out_tau_off] = result;
    }
  }
}

int launch_sample(const SampleArgs& a, hipStream_t stream) {
  if (a.rows <= 0) return MMK_OK;
  if (a.n_classes > 64 * kMaxPerLane) return fail(MMK_ERR_UNSUPPORTED, "sampler: at most %d classes", 64 * kMaxPerLane);
  hipLaunchKernelGGL(sample_kernel, dim3(a.rows), dim3(64), 0, stream, a);
  MMK_HIP(hipGetLastError());
  return MMK_OK;
}

// ---- recurrent cells --------------------------------------------------------------
// GRU (torch gate order r, z, n):  r = s(gi_r+gh_r), z = s(gi_z+gh_z),
// n = tanh(gi_n + r*gh_n), h' = (h - n)*z + n        (ATen RNN.cpp GRUCell)
__global__ void gru_cell_kernel(const float* __restrict__ gi, const float* __restrict__ gh, float* __restrict__ h,
                                int M, int H) {
  const int64_t total = (int64_t)M * H;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int m = (int)(i / H), j = (int)(i % H);
    const float* a = gi + (int64_t)m * 3 * H;
    const float* b = gh + (int64_t)m * 3 * H;
    const float r = sigmoidf_(b[j] + a[j]);
    const float z = sigmoidf_(b[H + j] + a[H + j]);
    const float n = tanhf(a[2 * H + j] + b[2 * H + j] * r);
    const float hp = h[i];
    h[i] = (hp - n) * z + n;
  }
}
int launch_gru_cell(const float* gi, const float* gh, float* h, int M, int H, hipStream_t stream) {
  const int64_t total = (int64_t)M * H;
  hipLaunchKernelGGL(gru_cell_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, gi, gh, h, M, H);
  MMK_HIP(hipGetLastError());
  return MMK_OK;
}

// LSTM (gate order i, f, g, o): gates (+ gadd) hold W_ih x + b_ih + W_hh h + b_hh
__global__ void lstm_cell_kernel(const float* __restrict__ gates, int64_t gates_ld, const float* __restrict__ gadd,
                                 int64_t gadd_ld, float* __restrict__ h, int64_t h_ld, float* __restrict__ c,
                                 int64_t c_ld, float* __restrict__ y, int64_t y_ld, int M, int H) {
  const int64_t total = (int64_t)M * H;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int m = (int)(i / H), j = (int)(i % H);
    const float* g = gates + (int64_t)m * gates_ld;
    float gi_ = g[j], gf = g[H + j], gg = g[2 * H + j], go = g[3 * H + j];
    if (gadd) {
      const float* e = gadd + (int64_t)m * gadd_ld;
      gi_ += e[j]; gf += e[H + j]; gg += e[2 * H + j]; go += e[3 * H + j];
    }
    const float ig = sigmoidf_(gi_), fg = sigmoidf_(gf), cg = tanhf(gg), og = sigmoidf_(go);
    const float cn = fg * c[(int64_t)m * c_ld + j] + ig * cg;
    const float hn = og * tanhf(cn);
    c[(int64_t)m * c_ld + j] = cn;
    h[(int64_t)m * h_ld + j] = hn;
    if (y) y[(int64_t)m * y_ld + j] = hn;
  }
}
int launch_lstm_cell(const float* gates, int64_t gates_ld, const float* gadd, int64_t gadd_ld, float* h, int64_t h_ld,
                     float* c, int64_t c_ld, float* y, int64_t y_ld, int M, int H, hipStream_t stream) {
  const int64_t total = (int64_t)M * H;
  hipLaunchKernelGGL(lstm_cell_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, gates, gates_ld,
                     gadd, gadd_ld, h, h_ld, c, c_ld, y, y_ld, M, H);
  MMK_HIP(hipGetLastError());
  return MMK_OK;
}

__global__ void rnn_tanh_cell_kernel(const float* __restrict__ g, float* __restrict__ h, int M, int H) {
  const int64_t total = (int64_t)M * H;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    h[i] = tanhf(g[i]);
}
int launch_rnn_tanh_cell(const float* g, float* h, int M, int H, hipStream_t stream) {
  const int64_t total = (int64_t)M * H;
  hipLaunchKernelGGL(rnn_tanh_cell_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, g, h, M, H);
  MMK_HIP(hipGetLastError());
  return MMK_OK;
}

}  // namespace mmk

extern "C" int mmk_abi_version(void) { return MMK_ABI_VERSION; }
#ifndef MMK_SOURCE_DIGEST
#define MMK_SOURCE_DIGEST "unknown"
#endif
// (behind a marker: the build script reads the digest out of the FILE - a dlopen of a path that is loaded already returns the old image)
static const char kBuildDigest[] = "mmk-source-digest:" MMK_SOURCE_DIGEST;
extern "C" const char* mmk_build_digest(void) { return kBuildDigest + 18; }
extern "C" int64_t mmk_config_bytes(int which) {
  return which == 0 ? (int64_t)sizeof(mmk_wavenet_config) : which == 1 ? (int64_t)sizeof(mmk_srnn_config) : which == 2 ? (int64_t)sizeof(mmk_s2s_config) : -1;
}
extern "C" const char* mmk_last_error(void) { return mmk::g_err; }

extern "C" int mmk_categorical_sample_f32_i64(const float* logits, int64_t ld, int32_t rows, int32_t n_classes,
                                              int32_t has_temp_col, float min_temp, const float* temperature,
                                              const float* uniforms, int64_t* out, int64_t out_stride,
                                              mmk_stream_t stream) {
  using namespace mmk;
  if (!logits || !out || rows < 0 || n_classes <= 0) return fail(MMK_ERR_INVALID, "sample: bad arguments");
  if (temperature && !uniforms) return fail(MMK_ERR_INVALID, "sample: temperature given without uniforms");
  SampleArgs a = {};
  a.logits = logits; a.ld = ld; a.rows = rows; a.n_classes = n_classes; a.has_temp_col = has_temp_col;
  a.min_temp = min_temp; a.temperature = temperature; a.uniforms = uniforms; a.uniform_ld = 1; a.uni_off = 0;
  a.out = out; a.out_row_stride = out_stride; a.out_tau_off = 0; a.tau_ptr = nullptr; a.tau_off = 0;
  return launch_sample(a, (hipStream_t)stream);
}

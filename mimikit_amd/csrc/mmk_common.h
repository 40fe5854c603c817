// Shared host/device declarations of libmmk_hip (gfx950 only).
#pragma once
// A wave that waits for a hand-off inside a persistent kernel gives up after its OWN time-out and raises the plan's error word.  1: it also looks at that word
// every few hundred polls and follows a wave that has given up.  Measured (round 6): that second way out of every wait loop costs the loops their shape - a dozen scalar
// instructions and taken branches between the poll that succeeds and the next instruction (SampleRNN cfg 3 3.63 -> 3.44 us per step without it) - and buys little:
// the waves of a launch that has lost a hand-off run into their time-outs at about the same moment anyway.
#ifndef MMK_WAIT_ERR_LOOK
#define MMK_WAIT_ERR_LOOK 0
#endif
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>

#include "../../include/mmk.h"

namespace mmk {

// ---- error plumbing ---------------------------------------------------------
void set_error(const char* fmt, ...);
int fail(int code, const char* fmt, ...);

#define MMK_HIP(call)                                                                         \
  do {                                                                                        \
    hipError_t e__ = (call);                                                                  \
    if (e__ != hipSuccess)                                                                    \
      return ::mmk::fail(MMK_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), \
                         __FILE__, __LINE__);                                                 \
  } while (0)

#define MMK_TRY(expr)            \
  do {                           \
    int rc__ = (expr);           \
    if (rc__ != MMK_OK) return rc__; \
  } while (0)

static inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// ---- time-indexed addressing --------------------------------------------------
// Every step kernel reads the current position `tau` from device memory, so a
// captured hipGraph can be replayed for any step.  An Addr resolves to
//   base + (((tau + offset) / div) % mod) * slot_stride        (mod == 0: no wrap)
// in ELEMENTS of the pointed-to type; static buffers use slot_stride == 0.
struct Addr {
  const void* base;
  int64_t slot_stride;
  int32_t offset;
  int32_t div;    // >= 1
  int32_t mod;    // 0: no wrap
  int32_t shift;  // log2(div) when div is a power of two, else -1
  int32_t mask;   // mod - 1 when mod is a power of two, else -1
  int32_t pad_;
};

static inline int32_t log2_exact(int32_t v) {
  if (v <= 0 || (v & (v - 1))) return -1;
  int32_t l = 0;
  while ((1 << l) < v) ++l;
  return l;
}
static inline Addr addr_time(const void* p, int64_t slot_stride, int32_t offset, int32_t div, int32_t mod) {
  Addr a;
  a.base = p; a.slot_stride = slot_stride; a.offset = offset; a.div = div < 1 ? 1 : div; a.mod = mod;
  a.shift = log2_exact(a.div);
  a.mask = (mod > 0 && log2_exact(mod) >= 0) ? mod - 1 : -1;
  a.pad_ = 0;
  return a;
}
static inline Addr addr_static(const void* p) { return addr_time(p, 0, 0, 1, 0); }
static inline Addr addr_ring(const void* p, int64_t slot_stride, int32_t offset, int32_t mod) {
  return addr_time(p, slot_stride, offset, 1, mod);
}

// positions are < 2^31; everything here is 32-bit and, for power-of-two rings, shift/mask only
__device__ __forceinline__ int64_t addr_elems(const Addr& a, int tau) {
  if (a.slot_stride == 0) return 0;
  int s = tau + a.offset;
  if (a.div > 1) s = a.shift >= 0 ? (s >> a.shift) : (int)((unsigned)(s < 0 ? 0 : s) / (unsigned)a.div);
  if (a.mod > 0) {
    if (a.mask >= 0) {
      s &= a.mask;  // two's complement: also right for positions before the start of a warm-up
    } else {
      s %= a.mod;
      if (s < 0) s += a.mod;
    }
  }
  return (int64_t)s * a.slot_stride;
}

// ---- activations (match the torch CPU formulas the reference runs) ------------
// Reciprocal of the activations below: __frcp_rn is a correctly rounded division (v_div_scale / v_fma / v_div_fixup: ten instructions),
// MMK_FAST_RCP = 1 takes the hardware's v_rcp_f32 (1 ulp) - per translation unit, where the activation sits on a per-step chain.
#ifndef MMK_FAST_RCP
#define MMK_FAST_RCP 0
#endif
__device__ __forceinline__ float mmk_rcp(float x) {
#if MMK_FAST_RCP
  return __builtin_amdgcn_rcpf(x);
#else
  return __frcp_rn(x);
#endif
}

enum Act : int32_t { ACT_NONE = 0, ACT_TANH = 1, ACT_SIGMOID = 2, ACT_MISH = 3, ACT_ABS = 4, ACT_RELU = 5, ACT_SOFTPLUS = 6, ACT_SIN = 7, ACT_COS = 8 };      // (include/mmk.h: MMK_ACT_*)

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float mishf_(float x) { return x * tanhf(log1pf(expf(x))); }
// Mish with the hardware exp/rcp: tanh(log(1 + e)) = n / (n + 2) with n = e (e + 2), e = exp(x); for x > 20 the
// factor is 1 to fp32 precision (and e*e would overflow).  ~1e-6 relative, a dozen instructions instead of three
// libm calls - for the kernels where the activation sits on the per-sample chain.
__device__ __forceinline__ float mish_fast(float x) {
  const float e = __expf(fminf(x, 20.f));
  const float n = e * (e + 2.f);
  return x > 20.f ? x : x * (n * mmk_rcp(n + 2.f));
}

// sigmoid / tanh with the hardware exp2 / rcp (the gate arithmetic of the WaveNet step kernels): sigmoid(x) = 1 / (1 + 2^(-x log2 e)),
// tanh(x) = 2 sigmoid(2 x) - 1; ~1e-7 absolute, a handful of instructions instead of a libm call - for cells on a per-step chain
__device__ __forceinline__ float sigmoid_fast(float x) { return mmk_rcp(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f)); }
__device__ __forceinline__ float tanh_fast(float x) { return fmaf(mmk_rcp(1.0f + __builtin_amdgcn_exp2f(x * -2.8853900817779268f)), 2.f, -1.f); }

__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case ACT_TANH: return tanhf(v);
    case ACT_SIGMOID: return sigmoidf_(v);
    case ACT_MISH: return mishf_(v);
    case ACT_ABS: return fabsf(v);
    case ACT_RELU: return v > 0.f ? v : 0.f;
    case ACT_SOFTPLUS: return v > 20.f ? v : log1pf(expf(v));      // nn.Softplus(beta=1, threshold=20)
    case ACT_SIN: return sinf(v);
    case ACT_COS: return cosf(v);
    default: return v;
  }
}

// ---- linear (skinny GEMM) kernel interface ------------------------------------
constexpr int kMaxSeg = 4;

enum SegKind : int32_t { SEG_F32 = 0, SEG_I64_LINEARIZED = 1 };

// One K-segment of the A operand: rows m = 0..M-1 at x + m*ld (+ time slot).
struct Seg {
  Addr x;
  int64_t ld;       // elements between consecutive rows (clips)
  int32_t K;        // real columns
  int32_t kind;     // SegKind
  float class_size; // for SEG_I64_LINEARIZED: ((q / class_size) - .5) * 2   (modules/io.py:106-112)
  int32_t pad_;
};

enum Epilogue : int32_t {
  EPI_STORE = 0,     // y = act(v + bias (+ add))
  EPI_GATE = 1,      // packed rows interleave (f, g): y[n/2] = tanh(f) * sigmoid(g)   (wavenet_v2.py:151)
  EPI_RES_SKIP = 2,  // n < n_res: out = res_in + v ; n >= n_res_pad: skip (+)= v      (wavenet_v2.py:165-176)
};

struct LinearArgs {
  Seg seg[kMaxSeg];
  int32_t nseg;
  int32_t M;            // real rows
  int32_t N;            // real output columns (packed order)
  int32_t n_tiles;      // ceil(N_pad / 16)
  int32_t k_chunks;     // total 16-wide K chunks over all segments
  int32_t seg_chunk0[kMaxSeg + 1];
  const float* Wp;      // [n_tiles][k_chunks][64][4]
  const float* bias;    // packed order, n_tiles*16 floats, or nullptr
  const int64_t* tau_ptr;
  int64_t tau_off;
  int32_t epilogue;
  int32_t act;          // EPI_STORE: of the output; EPI_GATE: act_f, of the f columns
  int32_t act2;         // EPI_GATE: act_g, of the g columns
  // EPI_STORE
  Addr out; int64_t out_ld;
  Addr add; int64_t add_ld; int32_t has_add; int32_t accumulate;
  // EPI_GATE: out/out_ld receive N/2 columns
  // EPI_RES_SKIP
  int32_t n_res;        // real residual rows (0: none)
  int32_t n_res_pad;    // packed row where the skip rows start
  int32_t n_skip;       // real skip rows (0: none)
  int32_t skip_first;   // overwrite instead of accumulate
  Addr res_in; int64_t res_in_ld;
  Addr res_out; int64_t res_out_ld;
  Addr skip; int64_t skip_ld;
  // rows in groups (EPI_STORE, one segment): row m = group m / row_group, row m % row_group of it; a group's rows lie x_group_stride /
  // out_group_stride elements after the previous group's - the clips of a (clip, position, feature) tensor as ONE launch.  0: plain rows
  int32_t row_group;
  int64_t x_group_stride, out_group_stride;
};

int launch_linear(const LinearArgs& a, hipStream_t stream);

// measurement mode: when g_prof is set, every linear launch carries start/stop events tagged g_prof_tag
struct ProfRecord {
  hipEvent_t start, stop;
  int tag;
};
extern thread_local std::vector<ProfRecord>* g_prof;
extern thread_local int g_prof_tag;

// packing helpers (device side, enqueue on stream)
int64_t packed_floats(int n_rows, int k_cols);
// writes rows [row0 + r*row_step) r<n_rows, k-chunks starting at chunk0 of a packed matrix with k_chunks total
int pack_rect(float* Wp, int k_chunks_total, int row0, int row_step, int n_rows, int chunk0, int K_real,
              const float* src, int64_t src_row_stride, int64_t src_col_stride, hipStream_t stream);
int pack_bias(float* dst, int row0, int row_step, int n_rows, const float* src, int accumulate, hipStream_t stream);

// small kernels
int launch_embed(const int64_t* idx, int64_t idx_row_stride, int64_t idx_tau_off, const float* table, int C,
                 int q_levels, Addr out, int64_t out_ld, int M, const int64_t* tau_ptr, int64_t tau_off,
                 hipStream_t stream);
int launch_bump(int64_t* tau_ptr, int64_t inc, hipStream_t stream);
int launch_set_i64(int64_t* p, int64_t v, hipStream_t stream);
int launch_copy_rows(Addr src, int64_t src_ld, Addr dst, int64_t dst_ld, int M, int C, const int64_t* tau_ptr,
                     int64_t tau_off, hipStream_t stream);
int launch_add_rows(Addr src, int64_t src_ld, Addr dst, int64_t dst_ld, int M, int C, const int64_t* tau_ptr, int64_t tau_off,
                    hipStream_t stream);
int launch_affine_rows(const float* P, int64_t p_ld, Addr dst, int64_t dst_ld, int M, int C, const int64_t* tau_ptr, int64_t tau_off,
                       hipStream_t stream);
int launch_fill(float* p, float v, int64_t n, hipStream_t stream);

struct SampleArgs {
  const float* logits; int64_t ld; int32_t rows; int32_t n_classes; int32_t has_temp_col; float min_temp;
  const float* temperature;     // nullptr: argmax
  const float* uniforms;        // [rows][uniform_ld], column = (tau + uni_off)
  int64_t uniform_ld; int64_t uni_off;
  int64_t* out; int64_t out_row_stride; int64_t out_tau_off;  // out[r*stride + tau + out_tau_off]
  const int64_t* tau_ptr; int64_t tau_off;
  // group > 0: row r = (g, i) = (r / group, r % group) is written to out[g * group_stride + i * out_row_stride + ...], rows with
  // i >= kept are not written (the hop classes of a Seq2Seq step; clipped at the end of the tensor)
  int32_t group; int32_t kept; int64_t group_stride;
};
int launch_sample(const SampleArgs& a, hipStream_t stream);

// n_fft = 1024 STFT (istft.hip); out_mode 0 (re, im), 1 (abs, angle), 2 angle, 3 Griffin-Lim update, 4 abs
int launch_stft1024(const float* x, int64_t x_row_stride, int batch, int64_t n_samples, int hop, int center, int reflect, int out_mode,
                    float* out, float* tprev, float momentum, hipStream_t stream);
// n_fft = 2048 (spectral2048.hip): one real frame = one 1024-point complex transform + an untangling pass
int launch_stft2048(const float* x, int64_t x_row_stride, int batch, int64_t n_samples, int hop, int center, int reflect, int out_mode,
                    float* out, hipStream_t stream);
int launch_istft2048(const float* spec, const float* mag, int mode, int batch, int64_t n_frames, int hop, float* out, hipStream_t stream);
int launch_gla2048_iter(const float* wave_in, const float* mag, const float* tprev_in, float* tprev_out, float momentum, int batch,
                        int64_t n_frames, int hop, float* wave_out, hipStream_t stream);
// any power-of-two n_fft in [64, 4096]: one pair per workgroup, Stockham passes through LDS (istft.hip); same out modes
int launch_stft_generic(const float* x, int64_t x_row_stride, int batch, int64_t n_samples, int n_fft, int hop, int center, int reflect,
                        int out_mode, float* out, float* tprev, float momentum, hipStream_t stream);
// plain tiled GEMM on a packed weight matrix (gemm.hip): C[b][M, N] = A[b][M, K] . W^T, b < batch
int launch_gemm_f32(const float* A, int64_t lda, int64_t a_batch, const float* Wp, int n_tiles, int k_chunks, int N, int K,
                    float* C, int64_t ldc, int64_t c_batch, int M, int batch, hipStream_t stream);
// C[M, N] = act(A[M, K] . W^T + bias) on a packed weight matrix, 128 x 64 tiles with a pipelined K loop (gemm.hip); M >= 128
bool gemm_bias_act_supported(const float* A, int64_t lda, int M, int K);
// optional scatter of the output rows: row m = (g, i) with g = m / group, i = m % group goes to C + g group_stride + i row_stride,
// rows with i >= kept are not written (group == 0: plain rows at ldc)
struct GemmRowMap {
  int32_t group = 0, kept = 0;
  int64_t group_stride = 0, row_stride = 0;
};
// with a `partial` buffer of gemm_bias_act_partial_floats(M, n_tiles, k_chunks) floats a launch of few tiles splits K over more
// workgroups and adds the partial sums in a fixed order in a second launch (0 floats: the launch fills the chip as it is)
int64_t gemm_bias_act_partial_floats(int M, int n_tiles, int k_chunks);
// (forced_k_split > 0: that many K splits instead of the launch's own choice - the parity tests' way to every split width)
int launch_gemm_bias_act(const float* A, int64_t lda, const float* Wp, const float* bias, int n_tiles, int k_chunks, int N, int K, float* C,
                         int64_t ldc, int M, int act, hipStream_t stream, GemmRowMap rm = GemmRowMap(), float* partial = nullptr,
                         int64_t partial_floats = 0, int forced_k_split = 0);

// C[M <= 64, N] = act(A . W^T + bias) for few rows against a large packed matrix (skinny.hip)
bool skinny_linear_supported(const float* A, int64_t lda, int M, int K, int k_chunks);
int launch_skinny_linear(const float* A, int64_t lda, const float* Wp, const float* bias, int n_tiles, int k_chunks, int N, int K, float* C,
                         int64_t ldc, int M, int act, hipStream_t stream);

// recurrent cells (elementwise)
int launch_gru_cell(const float* gi, const float* gh, float* h, int M, int H, hipStream_t stream);
int launch_lstm_cell(const float* gates, int64_t gates_ld, const float* gadd, int64_t gadd_ld, float* h, int64_t h_ld,
                     float* c, int64_t c_ld, float* y, int64_t y_ld, int M, int H, hipStream_t stream);
int launch_rnn_tanh_cell(const float* g, float* h, int M, int H, hipStream_t stream);

}  // namespace mmk

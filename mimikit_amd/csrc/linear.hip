// Skinny fp32 GEMM on the f32-input matrix cores + fused epilogues (gfx950).
//
// Y[M,N] = X[M,K] · W[N,K]^T for M = clips (1..64 per workgroup row-block),
// the shape of every per-step op on the generate path: WaveNet dilated/1x1
// convs at one time position (wavenet_v2.py:131-176), SampleRNN tier
// projections / RNN gates / up-samplers (sample_rnn_v2.py:83-99), the MLP head
// (mlp.py:58-63) and the Seq2Seq LSTM gates.
//
// Mapping: one workgroup = one 16-column output tile x (MT x 16) rows; its
// waves split K between them (v_mfma_f32_16x16x4_f32, exact fp32 fmaf chains)
// and reduce through LDS in a fixed wave order, so results are run-to-run
// deterministic.  Weights are pre-packed once per commit into fragment order
// (one coalesced 1 KiB dwordx4 load per 16x16 block).  The A operand can be the
// concatenation of up to kMaxSeg row-major segments, each resolved through a
// time-indexed Addr (dilation queues, cond rows, int64 sample windows).
#include <hip/hip_ext.h>

#include <atomic>

#include "mmk_common.h"

namespace mmk {

thread_local std::vector<ProfRecord>* g_prof = nullptr;
thread_local int g_prof_tag = 0;

typedef float f32x4 __attribute__((ext_vector_type(4)));

// per-chunk view of the A operand: which segment a K-chunk belongs to is resolved with selects on
// wave-uniform scalars (no dynamic indexing of the kernel-argument struct)
struct SegView {
  const char* base;  // segment base at the current time slot (bytes)
  int64_t ld;        // row stride in elements
  int K;
  int kind;
  float class_size;
  int chunk0;
};

template <int MT, bool VEC>
__global__ __launch_bounds__(1024) void linear_kernel(const LinearArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  f32x4* red = reinterpret_cast<f32x4*>(smem_raw);
  constexpr int UNR = MT <= 2 ? 4 : 2;  // K-chunks whose loads are in flight together

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = blockDim.x >> 6;
  const int tile = blockIdx.x;
  const int m0 = blockIdx.y * (MT * 16);
  const int tau = (int)((a.tau_ptr ? *a.tau_ptr : 0) + a.tau_off);

  // ---- resolve the (<= 4) segments once, with compile-time indices ----------------
  SegView sv[kMaxSeg];
#pragma unroll
  for (int s = 0; s < kMaxSeg; ++s) {
    const int esz = a.seg[s].kind == SEG_I64_LINEARIZED ? 8 : 4;
    sv[s].base = (const char*)a.seg[s].x.base + addr_elems(a.seg[s].x, tau) * esz;
    sv[s].ld = a.seg[s].ld;
    sv[s].K = a.seg[s].K;
    sv[s].kind = a.seg[s].kind;
    sv[s].class_size = a.seg[s].class_size;
    sv[s].chunk0 = a.seg_chunk0[s];
  }
  const int c1 = a.seg_chunk0[1], c2 = a.seg_chunk0[2], c3 = a.seg_chunk0[3];

  const int c_begin = (int)(((long)a.k_chunks * wave) / nw);
  const int c_end = (int)(((long)a.k_chunks * (wave + 1)) / nw);
  const int r = lane & 15, q = lane >> 4;

  int64_t row[MT], gx[MT];       // gx: what a row's group adds to its offset (elements), beyond row * ld
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m0 + mt * 16 + r;
    row[mt] = m < a.M ? m : a.M - 1;
    gx[mt] = a.row_group > 0 ? (row[mt] / a.row_group) * (a.x_group_stride - (int64_t)a.row_group * a.seg[0].ld) : 0;
  }

  f32x4 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const f32x4* wp = reinterpret_cast<const f32x4*>(a.Wp) + ((int64_t)tile * a.k_chunks) * 64 + lane;

  for (int cb = c_begin; cb < c_end; cb += UNR) {
    f32x4 w[UNR];
    f32x4 x[UNR][MT];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int c = cb + u;
      const bool live = c < c_end;
      const int cc = live ? c : c_begin;
      // segment of this chunk (wave-uniform selects)
      const int si = (cc >= c1) + (cc >= c2) + (cc >= c3);
      const char* base = si == 0 ? sv[0].base : (si == 1 ? sv[1].base : (si == 2 ? sv[2].base : sv[3].base));
      const int64_t ld = si == 0 ? sv[0].ld : (si == 1 ? sv[1].ld : (si == 2 ? sv[2].ld : sv[3].ld));
      const int K = si == 0 ? sv[0].K : (si == 1 ? sv[1].K : (si == 2 ? sv[2].K : sv[3].K));
      const int ch0 = si == 0 ? 0 : (si == 1 ? c1 : (si == 2 ? c2 : c3));
      const int kk = (cc - ch0) * 16 + 4 * q;
      w[u] = live ? wp[(int64_t)cc * 64] : f32x4{0.f, 0.f, 0.f, 0.f};
      if (VEC) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          if (live && kk < K)
            x[u][mt] = *reinterpret_cast<const f32x4*>(base + (row[mt] * ld + gx[mt] + kk) * 4);
          else
            x[u][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      } else {
        const int kind = si == 0 ? sv[0].kind : (si == 1 ? sv[1].kind : (si == 2 ? sv[2].kind : sv[3].kind));
        const float cs = si == 0 ? sv[0].class_size
                                 : (si == 1 ? sv[1].class_size : (si == 2 ? sv[2].class_size : sv[3].class_size));
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float v = 0.f;
            if (live && kk + i < K) {
              if (kind == SEG_I64_LINEARIZED) {
                const int64_t cls = *reinterpret_cast<const int64_t*>(base + (row[mt] * ld + gx[mt] + kk + i) * 8);
                v = (((float)cls / cs) - .5f) * 2.f;  // Linearizer, modules/io.py:106-112
              } else {
                v = *reinterpret_cast<const float*>(base + (row[mt] * ld + gx[mt] + kk + i) * 4);
              }
            }
            x[u][mt][i] = v;
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[u][mt][i], w[u][i], acc[mt], 0, 0, 0);
      }
    }
  }

  // ---- split-K reduction across the workgroup's waves (fixed order) ----------
  if (nw > 1) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) red[(wave * MT + mt) * 64 + lane] = acc[mt];
    __syncthreads();
  }

  // ---- epilogue: lane holds column n, rows 4*(lane>>4)+j ----------------------
  const int n = tile * 16 + (lane & 15);
  const float b = (a.bias && n < a.N) ? a.bias[n] : 0.f;

  auto run_epilogue = [&](const f32x4& val, int mt) {
    const int mbase = m0 + mt * 16 + 4 * (lane >> 4);
    if (a.epilogue == EPI_GATE) {
      const int64_t ooff = addr_elems(a.out, tau);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float f = val[j] + b;
        const float g = __shfl_down(f, 1);
        const int m = mbase + j;
        if (!(n & 1) && n < a.N && m < a.M) {
          float* o = (float*)a.out.base + ooff + (int64_t)m * a.out_ld + (n >> 1);
          *o = apply_act(f, a.act) * apply_act(g, a.act2);      // act_f(z_f) act_g(z_g), wavenet_v2.py:151 (Tanh / Sigmoid: tanhf, the exact sigmoid - as before)
        }
      }
    } else if (a.epilogue == EPI_RES_SKIP) {
      if (n < a.n_res) {
        const int64_t ioff = addr_elems(a.res_in, tau), ooff = addr_elems(a.res_out, tau);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int m = mbase + j;
          if (m < a.M) {
            const float xin = ((const float*)a.res_in.base)[ioff + (int64_t)m * a.res_in_ld + n];
            ((float*)a.res_out.base)[ooff + (int64_t)m * a.res_out_ld + n] = xin + (val[j] + b);
          }
        }
      } else if (n >= a.n_res_pad && n - a.n_res_pad < a.n_skip) {
        const int ns = n - a.n_res_pad;
        const int64_t soff = addr_elems(a.skip, tau);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int m = mbase + j;
          if (m < a.M) {
            float* sp = (float*)a.skip.base + soff + (int64_t)m * a.skip_ld + ns;
            const float cv = val[j] + b;
            *sp = a.skip_first ? cv : cv + *sp;
          }
        }
      }
    } else {
      if (n < a.N) {
        const int64_t ooff = addr_elems(a.out, tau);
        const int64_t aoff = a.has_add ? addr_elems(a.add, tau) : 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int m = mbase + j;
          if (m < a.M) {
            float t = val[j] + b;
            if (a.has_add) t += ((const float*)a.add.base)[aoff + (int64_t)m * a.add_ld + n];
            t = apply_act(t, a.act);
            const int64_t go = a.row_group > 0 ? (int64_t)(m / a.row_group) * (a.out_group_stride - (int64_t)a.row_group * a.out_ld) : 0;
            float* o = (float*)a.out.base + ooff + (int64_t)m * a.out_ld + go + n;
            *o = a.accumulate ? (*o + t) : t;
          }
        }
      }
    }
  };

  if (nw > 1) {
    // wave w' finishes row tiles w', w'+nw, ...: sums the partials in wave order, then the epilogue
    for (int mt = wave; mt < MT; mt += nw) {
      f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int w = 0; w < nw; ++w) {
        const f32x4 p = red[(w * MT + mt) * 64 + lane];
        v[0] += p[0]; v[1] += p[1]; v[2] += p[2]; v[3] += p[3];
      }
      run_epilogue(v, mt);
    }
  } else {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) run_epilogue(acc[mt], mt);
  }
}

static bool seg_vec_ok(const Seg& s) {
  return s.kind == SEG_F32 && (s.K % 4 == 0) && (s.ld % 4 == 0) && (s.x.slot_stride % 4 == 0) &&
         ((reinterpret_cast<uintptr_t>(s.x.base) & 15) == 0);
}

int launch_linear(const LinearArgs& a_in, hipStream_t stream) {
  if (a_in.M <= 0 || a_in.n_tiles <= 0) return MMK_OK;
  if (a_in.nseg < 1 || a_in.nseg > kMaxSeg) return fail(MMK_ERR_INVALID, "linear: bad segment count %d", a_in.nseg);
  LinearArgs a = a_in;
  for (int s = a.nseg; s < kMaxSeg; ++s) {  // unused segments alias segment 0 (never selected: chunk0 == k_chunks)
    a.seg[s] = a.seg[0];
    a.seg[s].K = 0;
  }
  for (int s = a.nseg; s <= kMaxSeg; ++s) a.seg_chunk0[s] = a.k_chunks;
  if (a.row_group > 0 && (a.nseg != 1 || a.epilogue != EPI_STORE || a.has_add))
    return fail(MMK_ERR_INVALID, "linear: grouped rows take one segment and a plain store");
  bool vec = true;
  for (int s = 0; s < a.nseg; ++s) vec = vec && seg_vec_ok(a.seg[s]);
  if (a.row_group > 0) vec = vec && (a.x_group_stride % 4 == 0);
  // latency-bound shapes (few tiles): one 16-row tile per workgroup to spread over more CUs;
  // GEMM-like shapes: up to 64 rows per workgroup so a weight fragment is reused from registers
  const int m_tiles = (a.M + 15) / 16;
  int mt = 1;
  if ((long)a.n_tiles * m_tiles >= 512) mt = m_tiles >= 4 ? 4 : (m_tiles >= 2 ? 2 : 1);
  int nw = 1;
  while (nw < 16 && a.k_chunks > nw * 3) nw *= 2;  // <= 3 K-chunks (48 columns) per wave where possible
  dim3 grid(a.n_tiles, (a.M + mt * 16 - 1) / (mt * 16));
  dim3 block(64 * nw);
  size_t lds = nw > 1 ? (size_t)nw * mt * 64 * sizeof(f32x4) : 0;
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;
  if (g_prof) {  // measurement mode: exact kernel start/stop timestamps on this stream
    MMK_HIP(hipEventCreate(&ev_start));
    MMK_HIP(hipEventCreate(&ev_stop));
    g_prof->push_back(ProfRecord{ev_start, ev_stop, g_prof_tag});
  }
#define MMK_LAUNCH(MT_, VEC_)                                                                              \
  do {                                                                                                     \
    if (g_prof)                                                                                            \
      hipExtLaunchKernelGGL((linear_kernel<MT_, VEC_>), grid, block, lds, stream, ev_start, ev_stop, 0, a); \
    else                                                                                                   \
      hipLaunchKernelGGL((linear_kernel<MT_, VEC_>), grid, block, lds, stream, a);                          \
  } while (0)
  if (vec) {
    if (mt == 1) MMK_LAUNCH(1, true);
    else if (mt == 2) MMK_LAUNCH(2, true);
    else MMK_LAUNCH(4, true);
  } else {
    if (mt == 1) MMK_LAUNCH(1, false);
    else if (mt == 2) MMK_LAUNCH(2, false);
    else MMK_LAUNCH(4, false);
  }
#undef MMK_LAUNCH
  MMK_HIP(hipGetLastError());
  return MMK_OK;
}

// ---- packing ----------------------------------------------------------------
int64_t packed_floats(int n_rows, int k_cols) {
  return round_up(n_rows, 16) * round_up(k_cols, 16);
}

__global__ void pack_rect_kernel(float* __restrict__ Wp, int k_chunks_total, int row0, int row_step, int n_rows,
                                 int chunk0, int n_chunks, int K_real, const float* __restrict__ src,
                                 int64_t rs, int64_t cs) {
  const int64_t total = (int64_t)n_rows * n_chunks * 16;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int rr = (int)(idx / (n_chunks * 16));
    const int kk = (int)(idx % (n_chunks * 16));
    const int prow = row0 + rr * row_step;
    const int tile = prow >> 4, rin = prow & 15;
    const int c = chunk0 + (kk >> 4);
    const int k16 = kk & 15;
    const int lane = (k16 >> 2) * 16 + rin;
    const float val = kk < K_real ? src[rr * rs + kk * cs] : 0.f;
    Wp[(((int64_t)tile * k_chunks_total + c) * 64 + lane) * 4 + (k16 & 3)] = val;
  }
}

std::atomic<int64_t> g_pack_launches{0};   // weight re-packing launches since the library was loaded (mmk_pack_launch_count)

int pack_rect(float* Wp, int k_chunks_total, int row0, int row_step, int n_rows, int chunk0, int K_real,
              const float* src, int64_t src_row_stride, int64_t src_col_stride, hipStream_t stream) {
  const int n_chunks = (K_real + 15) / 16;
  const int64_t total = (int64_t)n_rows * n_chunks * 16;
  if (total <= 0) return MMK_OK;
  g_pack_launches.fetch_add(1, std::memory_order_relaxed);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_rect_kernel, dim3(blocks), dim3(256), 0, stream, Wp, k_chunks_total, row0, row_step,
                     n_rows, chunk0, n_chunks, K_real, src, src_row_stride, src_col_stride);
  MMK_HIP(hipGetLastError());
  return MMK_OK;
}

__global__ void pack_bias_kernel(float* dst, int row0, int row_step, int n_rows, const float* src, int accumulate) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_rows) {
    float* d = dst + row0 + i * row_step;
    *d = accumulate ? (*d + src[i]) : src[i];
  }
}

int pack_bias(float* dst, int row0, int row_step, int n_rows, const float* src, int accumulate, hipStream_t stream) {
  if (n_rows <= 0) return MMK_OK;
  g_pack_launches.fetch_add(1, std::memory_order_relaxed);
  hipLaunchKernelGGL(pack_bias_kernel, dim3((n_rows + 255) / 256), dim3(256), 0, stream, dst, row0, row_step,
                     n_rows, src, accumulate);
  MMK_HIP(hipGetLastError());
  return MMK_OK;
}

}  // namespace mmk

// ---- exported building blocks -------------------------------------------------
extern "C" int64_t mmk_pack_launch_count(void) { return mmk::g_pack_launches.load(std::memory_order_relaxed); }

// Content fingerprint of a buffer of 32-bit words: every word is mixed with its position (a multiplicative hash, so that sign flips,
// swaps and shifted copies all move the sum) and the results are added up as 64-bit integers - addition commutes, so the order the
// waves finish in does not matter.  One launch over the concatenated weights of a network; the host compares the number with the one
// it took when it last committed a plan (mimikit_amd/native.py: WeightsTracker).
namespace mmk {
__device__ __forceinline__ unsigned long long fingerprint_word(uint32_t w, uint32_t pos) {
  uint32_t h = (w ^ (pos * 0x9E3779B1u)) * 0x85EBCA77u;
  h ^= h >> 15;
  h *= 0xC2B2AE3Du;
  return (unsigned long long)(h ^ (h >> 13)) + ((unsigned long long)w << 20);
}

constexpr int kFpMaxBuffers = 96;
struct FingerprintArgs {
  const uint32_t* w[kFpMaxBuffers];
  int64_t n[kFpMaxBuffers];
  uint32_t pos0[kFpMaxBuffers];      // position of a buffer's first word in the concatenation (mod 2^32)
  uint32_t blk0[kFpMaxBuffers + 1];  // first workgroup of a buffer: workgroups are dealt out in proportion to the sizes
  int32_t count;
};

// 16 bytes per lane and load where the buffer allows it (every torch allocation does); one atomic per workgroup - thousands of
// atomics on the one result word were the whole cost of the first version
__global__ __launch_bounds__(256) void fingerprint_kernel(const FingerprintArgs a, unsigned long long* __restrict__ out) {
  __shared__ unsigned long long part[4];
  int bi = 0;
  while (bi + 1 < a.count && blockIdx.x >= a.blk0[bi + 1]) ++bi;
  const uint32_t* __restrict__ w = a.w[bi];
  const int64_t n = a.n[bi];
  const uint32_t pos0 = a.pos0[bi];
  unsigned long long acc = 0;
  const int64_t tid = (int64_t)(blockIdx.x - a.blk0[bi]) * blockDim.x + threadIdx.x;
  const int64_t nthreads = (int64_t)(a.blk0[bi + 1] - a.blk0[bi]) * blockDim.x;
  if ((reinterpret_cast<uintptr_t>(w) & 15) == 0) {
    const uint4* w4 = reinterpret_cast<const uint4*>(w);
    const int64_t n4 = n >> 2;
    for (int64_t i = tid; i < n4; i += nthreads) {
      const uint4 v = w4[i];
      const uint32_t p = pos0 + (uint32_t)(4 * i);
      acc += fingerprint_word(v.x, p) + fingerprint_word(v.y, p + 1) + fingerprint_word(v.z, p + 2) + fingerprint_word(v.w, p + 3);
    }
    for (int64_t i = 4 * n4 + tid; i < n; i += nthreads) acc += fingerprint_word(w[i], pos0 + (uint32_t)i);
  } else {
    for (int64_t i = tid; i < n; i += nthreads) acc += fingerprint_word(w[i], pos0 + (uint32_t)i);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long sum = part[0] + part[1] + part[2] + part[3];
    if (sum != 0) atomicAdd(out, sum);
  }
}
}  // namespace mmk

extern "C" int mmk_fingerprint_buffers_u32(const void* const* buffers, const int64_t* n_words, int32_t n_buffers, uint64_t* out,
                                           mmk_stream_t stream) {
  using namespace mmk;
  if (!out || n_buffers < 0 || (n_buffers > 0 && (!buffers || !n_words))) return fail(MMK_ERR_INVALID, "fingerprint: bad arguments");
  MMK_HIP(hipMemsetAsync(out, 0, sizeof(uint64_t), (hipStream_t)stream));
  uint64_t pos = 0;
  for (int first = 0; first < n_buffers; first += kFpMaxBuffers) {
    FingerprintArgs a = {};
    const int count = n_buffers - first < kFpMaxBuffers ? n_buffers - first : kFpMaxBuffers;
    int64_t total = 0;
    for (int i = 0; i < count; ++i) {
      if (n_words[first + i] < 0 || (n_words[first + i] > 0 && !buffers[first + i])) return fail(MMK_ERR_INVALID, "fingerprint: bad buffer %d", first + i);
      total += n_words[first + i];
    }
    if (total == 0) continue;
    // ~2048 workgroups over the launch, 8192 words (32 per thread) or more each, one at least per buffer
    const int64_t per_block = total / 2048 > 8192 ? total / 2048 : 8192;
    uint32_t blocks = 0;
    for (int i = 0; i < count; ++i) {
      a.w[i] = (const uint32_t*)buffers[first + i];
      a.n[i] = n_words[first + i];
      a.pos0[i] = (uint32_t)pos;
      a.blk0[i] = blocks;
      pos += (uint64_t)a.n[i];
      const int64_t nb = (a.n[i] + per_block - 1) / per_block;
      blocks += (uint32_t)(nb > 0 ? nb : 1);
    }
    a.blk0[count] = blocks;
    a.count = count;
    hipLaunchKernelGGL(fingerprint_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, (unsigned long long*)out);
    MMK_HIP(hipGetLastError());
  }
  return MMK_OK;
}

extern "C" int mmk_fingerprint_u32(const void* words, int64_t n_words, uint64_t* out, mmk_stream_t stream) {
  return mmk_fingerprint_buffers_u32(&words, &n_words, 1, out, stream);
}

extern "C" int64_t mmk_packed_weight_floats(int32_t n_rows, int32_t k_cols) {
  return mmk::packed_floats(n_rows, k_cols);
}

extern "C" int mmk_pack_weight_f32(const float* w, int64_t ldw, int32_t n_rows, int32_t k_cols, float* packed,
                                   mmk_stream_t stream) {
  using namespace mmk;
  if (!w || !packed || n_rows <= 0 || k_cols <= 0) return fail(MMK_ERR_INVALID, "pack_weight: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  MMK_HIP(hipMemsetAsync(packed, 0, packed_floats(n_rows, k_cols) * sizeof(float), st));
  return pack_rect(packed, (k_cols + 15) / 16, 0, 1, n_rows, 0, k_cols, w, ldw, 1, st);
}

extern "C" int mmk_linear_f32(const float* x, int64_t ldx, int32_t m_rows, const float* packed_w, const float* bias,
                              int32_t n_rows, int32_t k_cols, float* y, int64_t ldy, int32_t act,
                              mmk_stream_t stream) {
  using namespace mmk;
  if (!x || !packed_w || !y || m_rows <= 0 || n_rows <= 0 || k_cols <= 0)
    return fail(MMK_ERR_INVALID, "linear: bad arguments");
  LinearArgs a = {};
  a.nseg = 1;
  a.seg[0].x = addr_static(x);
  a.seg[0].ld = ldx;
  a.seg[0].K = k_cols;
  a.seg[0].kind = SEG_F32;
  a.seg_chunk0[0] = 0;
  a.seg_chunk0[1] = (k_cols + 15) / 16;
  a.k_chunks = a.seg_chunk0[1];
  a.M = m_rows;
  a.N = n_rows;
  a.n_tiles = (n_rows + 15) / 16;
  a.Wp = packed_w;
  a.bias = bias;
  a.tau_ptr = nullptr;
  a.epilogue = EPI_STORE;
  a.act = act;
  a.out = addr_static(y);
  a.out_ld = ldy;
  return launch_linear(a, (hipStream_t)stream);
}

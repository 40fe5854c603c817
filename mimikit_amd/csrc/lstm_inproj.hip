// The input half of a bidirectional LSTM layer for ALL frames of ALL clips in one launch (gfx950):
//     gi[d][row, :] = W_ih[d] x[row, :] + (b_ih[d] + b_hh[d])        row = (clip, frame), d = forward / reverse
//
// Reference: torch.nn.LSTM inside the Seq2Seq encoder / decoder (s2s_lstm_v2.py:90-171); the recurrent half is lstm_seq.hip, which
// adds these rows to W_hh h frame by frame.
//
// Same shape of work as the resident recurrent kernel, without its dependencies: a workgroup owns 16 hidden units x 4 gates of one
// direction and keeps that slice of W_ih in registers (K <= 1024: 128 registers per lane), its 8 waves split K, and the rows stream
// past in blocks of 16 - a block's fragments come straight from L2 in MFMA operand order, one block ahead of the products, the
// partial sums meet in LDS (two slots: one barrier per block) and every thread adds up two outputs.  No operand goes through LDS,
// no weight is fetched twice by a workgroup: the 64 x 64 tiles of gemm.hip spend 28 % of their K loop outside the MFMA pipe on this
// shape (512 x 4096 x 1024, 46 us per direction); here both directions are one launch of 64 x 2 x 2 workgroups, one per CU.
#include <type_traits>

#include "lstm_inproj.h"

namespace mmk {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4s __attribute__((ext_vector_type(4)));

#ifndef MMK_IP_NOLOAD
#define MMK_IP_NOLOAD 0      // timing experiments only (wrong results)
#endif
#ifndef MMK_IP_NOEPI
#define MMK_IP_NOEPI 0
#endif
#ifndef MMK_IP_NOBAR
#define MMK_IP_NOBAR 0
#endif
#ifndef MMK_IP_SPREAD
#define MMK_IP_SPREAD 1      // the next block's fragments are requested one per chunk of products (0: all in front of the first)
#endif
#ifndef MMK_IP_EPI_EARLY
#define MMK_IP_EPI_EARLY 1   // chunk of the next block in front of which waves 0 - 3 add up a block's partial sums ...
#endif
#ifndef MMK_IP_EPI_LATE
#define MMK_IP_EPI_LATE 5    // ... and waves 4 - 7 (the other wave of each SIMD)
#endif
constexpr int kIpThreads = 512;
constexpr int kIpWaves = kIpThreads / 64;

template <int CPW>
__device__ __forceinline__ void ip_request(u32x4s (&set)[CPW], const __amdgpu_buffer_rsrc_t& x, int byte_off, int n_valid) {
  if (MMK_IP_NOLOAD) return;
#pragma unroll
  for (int u = 0; u < CPW; ++u) {
    // (chunks beyond K carry zero weights: any finite fragment will do - the last valid one, so that nothing is read out of bounds)
    const int uu = u < n_valid ? u : (n_valid > 0 ? n_valid - 1 : 0);
    set[u] = __builtin_amdgcn_raw_buffer_load_b128(x, byte_off, uu * 64, 0);
  }
}

template <int CPW>   // K-chunks (of 16) per wave: K <= 128 CPW
__global__ __launch_bounds__(kIpThreads) void lstm_inproj_kernel(const LstmInProjArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int H = a.H, KCo = H / 16;                 // column tiles of one gate
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ub = blockIdx.x;                       // block of 16 hidden units
  const int di = blockIdx.z;
  const LstmInProjDir d = a.dir[di];
  f32x4* red = reinterpret_cast<f32x4*>(smem_raw);                      // split-K partials [slot][gate][wave][lane]

  // rows of this workgroup: whole blocks of 16, dealt out evenly over gridDim.y
  const int n_blocks = (a.rows + 15) / 16;
  const int b_first = (int)(((int64_t)blockIdx.y * n_blocks) / gridDim.y), b_end = (int)(((int64_t)(blockIdx.y + 1) * n_blocks) / gridDim.y);
  if (b_first >= b_end) return;

  // ---- this wave's slice of W_ih: 4 gate tiles x CPW chunks, resident for the launch (zeros beyond K) --------------------------------
  // K's chunks dealt out evenly: k_chunks / 8 per wave, the first k_chunks % 8 waves one more (CPW is the larger of the two counts)
  const int c_base = a.k_chunks / kIpWaves, c_extra = a.k_chunks % kIpWaves;
  const int c0 = wave * c_base + min(wave, c_extra);
  const int n_valid = c_base + (wave < c_extra ? 1 : 0);
  f32x4 w[CPW][4];
  {
    const f32x4* wsrc = reinterpret_cast<const f32x4*>(d.wih_wp) + ((int64_t)ub * a.k_chunks + c0) * 64 + lane;
    const int64_t gate_stride = (int64_t)KCo * a.k_chunks * 64;          // f32x4 elements between the gates' tile rows
#pragma unroll
    for (int u = 0; u < CPW; ++u)
#pragma unroll
      for (int g = 0; g < 4; ++g) w[u][g] = u < n_valid ? wsrc[g * gate_stride + u * 64] : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // this thread's two outputs of a block: gate g, unit n, rows 2 rp and 2 rp + 1
  const int e_n = lane & 15, e_rp = (lane >> 4) + 4 * (wave >> 2), e_g = wave & 3;
  const int col = e_g * H + ub * 16 + e_n;
  const float bias = d.bias ? d.bias[col] : 0.f;
  // (in their registers before the row loop: see lstm_seq.hip - a load pending at the loop's entry costs a drain in every pass)
#pragma unroll
  for (int u = 0; u < CPW; ++u)
#pragma unroll
    for (int g = 0; g < 4; ++g) asm volatile("" : "+v"(w[u][g]));

  const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)(a.x_floats * sizeof(float)), 0x00020000);   // reads past the end: zeros
  auto frag_off = [&](int blk) {    // MFMA A operand: row lane & 15 of the block (clamped), K offset 4 (lane >> 4)
    int r = blk * 16 + (lane & 15);
    r = r < a.rows ? r : a.rows - 1;
    int64_t at = (int64_t)r * a.x_ld;
    if (a.x_group > 0) {
      const int gq = r / a.x_group;
      at = (int64_t)gq * a.x_group_stride + (int64_t)(r - gq * a.x_group) * a.x_ld;
    }
    return (int)((at + (n_valid > 0 ? c0 : 0) * 16 + 4 * (lane >> 4)) * sizeof(float));   // (a wave without chunks: zero weights, the row's first chunks)
  };
  // Rows read where the caller's frames lie are not padded: the chunk that holds a row's last inputs (K % 16 != 0, e.g. 513 bins) also
  // covers the first floats of the NEXT frame or clip.  Their weights are zero, but 0 x NaN / Inf is NaN: those elements are replaced by
  // 0 before the product, so that a non-finite neighbour never reaches another row's gates.
  const int kv = a.k_valid > 0 ? a.k_valid : a.K;
  const int tail_u = (kv % 16 != 0) ? kv / 16 - c0 : -1;                // which of this wave's chunks that is (none: out of 0 .. n_valid - 1)
  const int tail_nv = kv - (kv / 16) * 16 - 4 * (lane >> 4);              // this lane's valid elements in it (<= 0: none, >= 4: all)
  u32x4s xa[2][CPW];
#pragma unroll
  for (int u = 0; u < CPW; ++u) xa[1][u] = u32x4s{0u, 0u, 0u, 0u};
  ip_request<CPW>(xa[0], xsrc, frag_off(b_first), n_valid);

  // A block's partial sums are added up during the NEXT block's products, and at different times by the two waves of a SIMD (waves w
  // and w + 4): while one of them reads LDS and stores, the other keeps the matrix pipe busy.  The same for the requests of the next
  // block's fragments: one per chunk of products, not eight at once (8 waves x 8 KB at the same moment queue up in the CU's address
  // unit for ~1000 clocks, and a wave whose request waits there issues no MFMA).
  auto epilogue = [&](int blk, int slot) {
    // rows 2 rp, 2 rp + 1 of column n sit side by side in accumulator element (rp >> 1) * 16 + n: one 8-byte read per wave's partial
    const float* f = reinterpret_cast<const float*>(red + (slot * 4 + e_g) * kIpWaves * 64) + ((e_rp >> 1) * 16 + e_n) * 4 + (e_rp & 1) * 2;
    f32x2 sum = *reinterpret_cast<const f32x2*>(f);
#pragma unroll
    for (int wv = 1; wv < kIpWaves; ++wv) sum += *reinterpret_cast<const f32x2*>(f + wv * 256);
    const int r0 = blk * 16 + 2 * e_rp;
    float* dst = d.out + (int64_t)r0 * a.out_ld + col;
    if (r0 < a.rows) dst[0] = sum.x + bias;
    if (r0 + 1 < a.rows) dst[a.out_ld] = sum.y + bias;
  };
  const bool late_half = wave >= 4;
  constexpr int kEpiEarly = MMK_IP_EPI_EARLY < CPW ? MMK_IP_EPI_EARLY : CPW - 1, kEpiLate = MMK_IP_EPI_LATE < CPW ? MMK_IP_EPI_LATE : CPW - 1;
  auto pass = [&](auto setc, int blk, bool has_prev) {
    constexpr int set = decltype(setc)::value;
    const int next_off = frag_off(blk + 1 < b_end ? blk + 1 : blk);   // (the last pass asks for its own once more: every pass defines the other set anew)
    if (!MMK_IP_SPREAD) {
      ip_request<CPW>(xa[set ^ 1], xsrc, next_off, n_valid);
      __builtin_amdgcn_sched_barrier(0);
    }
    f32x4 acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < CPW; ++u) {
      if (MMK_IP_SPREAD && !MMK_IP_NOLOAD) {
        const int uu = u < n_valid ? u : (n_valid > 0 ? n_valid - 1 : 0);
        xa[set ^ 1][u] = __builtin_amdgcn_raw_buffer_load_b128(xsrc, next_off, uu * 64, 0);
      }
      if (!MMK_IP_NOEPI && has_prev && ((u == kEpiEarly && !late_half) || (u == kEpiLate && late_half))) epilogue(blk - 1, set ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      if (u < n_valid) {      // (same for the whole wave)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float xv = __uint_as_float(xa[set][u][i]);
          if (u == tail_u) xv = i < tail_nv ? xv : 0.f;
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv, w[u][g][i], acc[g], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    const int slot = set;
#pragma unroll
    for (int g = 0; g < 4; ++g) red[((slot * 4 + g) * kIpWaves + wave) * 64 + lane] = acc[g];
    if (!MMK_IP_NOBAR) __syncthreads();
  };
  int blk = b_first;
  for (; blk < b_end; blk += 2) {
    pass(std::integral_constant<int, 0>{}, blk, blk > b_first);
    if (blk + 1 < b_end) pass(std::integral_constant<int, 1>{}, blk + 1, true);
  }
  if (!MMK_IP_NOEPI) epilogue(b_end - 1, (b_end - 1 - b_first) & 1);
}

bool lstm_inproj_supported(int rows, int K, int k_chunks, int H) {
  return rows >= 16 && H >= 16 && H % 16 == 0 && K >= 1 && k_chunks >= 1 && k_chunks <= 64;
}

int launch_lstm_inproj(const LstmInProjArgs& a, int n_cu, hipStream_t stream) {
  if (a.x_floats <= 0 || a.x_floats * (int64_t)sizeof(float) >= ((int64_t)1 << 31) || (reinterpret_cast<uintptr_t>(a.x) & 3) != 0)
    return fail(MMK_ERR_UNSUPPORTED, "lstm input projection kernel: %lld input floats", (long long)a.x_floats);
  const int n_blocks = (a.rows + 15) / 16;
  // one workgroup per CU where the rows allow it: units x directions workgroups per row share
  int split = n_cu / ((a.H / 16) * 2);
  split = split < 1 ? 1 : (split > n_blocks ? n_blocks : split);
  const size_t lds = (size_t)2 * 4 * kIpWaves * 64 * 16;
  dim3 grid(a.H / 16, split, 2), block(kIpThreads);
  const int cpw = (a.k_chunks + kIpWaves - 1) / kIpWaves;
#define MMK_IP(CPW_) hipLaunchKernelGGL((lstm_inproj_kernel<CPW_>), grid, block, lds, stream, a)
  switch (cpw) {
    case 1: MMK_IP(1); break;
    case 2: MMK_IP(2); break;
    case 3: MMK_IP(3); break;
    case 4: MMK_IP(4); break;
    case 5: MMK_IP(5); break;
    case 6: MMK_IP(6); break;
    case 7: MMK_IP(7); break;
    case 8: MMK_IP(8); break;
    default: return fail(MMK_ERR_UNSUPPORTED, "lstm input projection kernel: K of %d chunks", a.k_chunks);
  }
#undef MMK_IP
  MMK_HIP(hipGetLastError());
  return MMK_OK;
}

}  // namespace mmk

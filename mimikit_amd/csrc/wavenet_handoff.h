// Device-side helpers shared by the persistent WaveNet kernels (wavenet_persist.hip, wavenet_chain.hip): data-tagged
// 8-byte hand-off granules, the polling sweep, MFMA tile steps and fixed-order reductions.
#pragma once
#include "mmk_common.h"

namespace mmk {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;
// Pointers that come out of LDS tables lose their address space and compile to FLAT loads, which count against
// lgkmcnt as well: every later LDS wait would then also wait for the weight stream.  Force global loads.
typedef const __attribute__((address_space(1))) f32x4* gf32x4_ptr;

constexpr unsigned kSpinLimit = 1u << 22;

// XCD == false: agent-scope store (sc1, write-through to memory; readable from every XCD).
// XCD == true : plain 8-byte store that stays in the producer's XCD L2; only used when every consumer
//               of the granule was VERIFIED (at kernel start, from HW_REG_XCC_ID) to run on the same XCD,
//               whose L2 is the coherence point for its CUs.  Consumers always load with sc1 (L1 bypass).
template <bool XCD>
__device__ __forceinline__ void gran_store(u64* p, unsigned epoch, float v) {
  const u64 x = ((u64)epoch << 32) | (u64)__float_as_uint(v);
  if (XCD)
    __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else
    __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool XCD>
__device__ __forceinline__ void gran_store_u32(u64* p, unsigned epoch, unsigned v) {
  const u64 x = ((u64)epoch << 32) | (u64)v;
  if (XCD)
    __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else
    __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 gran_load(const u64* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Poll `count` granules until every tag == epoch and write the values as floats into an LDS matrix
// (`cols` values per row, leading dimension ld; cols and ld are multiples of 4).  Each thread keeps four 8-byte
// loads in flight per round, so a sweep whose data is already there costs ONE round trip.  dst0 (optional) is
// the LDS address of the thread's first four values, computed once by the caller: no division on the path.
// Returns a workgroup-uniform success flag.
template <int NT>
__device__ __forceinline__ bool sweep(const u64* gran, int count, unsigned epoch, float* dst0, float* dst, int cols,
                                      int ld, int* err_flag, int* s_fail) {
  const int tid = threadIdx.x;
  for (int base = 0; base < count; base += NT * 4) {
    const int i0 = base + tid * 4;
    if (i0 < count) {      // count is a multiple of 16, i0 of 4: all four granules exist and share a row
      u64 v[4];
      unsigned spins = 0;
      bool ok = true;
      for (;;) {
        bool all = true;
        {
          // two 16-byte agent-scope loads (sc1: L1 bypass) for the four 8-byte granules.  A granule is written by ONE
          // aligned 8-byte store and lies inside one aligned 16-byte read: both are single transactions on a cache
          // line, so a granule is seen entirely old or entirely new; its tag is checked either way.
          typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
          u32x4v lo, hi;
          const u64* gp = gran + i0;
          asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
                       : "=&v"(lo), "=&v"(hi)
                       : "v"(gp)
                       : "memory");
          v[0] = ((u64)lo[1] << 32) | lo[0];
          v[1] = ((u64)lo[3] << 32) | lo[2];
          v[2] = ((u64)hi[1] << 32) | hi[0];
          v[3] = ((u64)hi[3] << 32) | hi[2];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) all = all && ((unsigned)(v[k] >> 32) == epoch);
        if (all) break;
        ++spins;
        if (spins > kSpinLimit || (MMK_WAIT_ERR_LOOK && (spins & 255u) == 0 && __hip_atomic_load(err_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
          ok = false;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      if (!ok) {
        *s_fail = 1;
        atomicExch(err_flag, 1);
      }
      float* d;
      if (base == 0 && dst0) {
        d = dst0;
      } else {
        const int m = i0 / cols;
        d = dst + m * ld + (i0 - m * cols);
      }
      *reinterpret_cast<f32x4*>(d) = f32x4{__uint_as_float((unsigned)v[0]), __uint_as_float((unsigned)v[1]),
                                           __uint_as_float((unsigned)v[2]), __uint_as_float((unsigned)v[3])};
    }
  }
  // vmcnt(0) on EVERY path: waves without granules skip the polls, and without this the compiler has to assume
  // that older requests are still pending and guards recycled registers with waits between the next requests
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  return *s_fail == 0;
}

// one wave's share of  X[16 x K] (LDS, ld) . W^T  for a 16-column tile: chunks [c0, c1) after chunk_base
__device__ __forceinline__ f32x4 tile_mma(const float* x, int ld, const f32x4* wp, int chunk_base, int c0, int c1,
                                          int lane) {
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  const int r = lane & 15, q = lane >> 4;
  for (int c = c0; c < c1; ++c) {
    const f32x4 w = wp[(int64_t)(chunk_base + c) * 64 + lane];
    const f32x4 xv = *reinterpret_cast<const f32x4*>(x + r * ld + c * 16 + 4 * q);
#pragma unroll
    for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[i], w[i], acc, 0, 0, 0);
  }
  return acc;
}

// One step of the K loop of the per-layer products.
//   SMALL == false: v_mfma_f32_16x16x4_f32 - rows = 16 clips (lane & 15), K = 4 per instruction.
//   SMALL == true : v_mfma_f32_4x4x1_16b_f32 - 16 independent 4x4 blocks, K = 1 per instruction.  Block b = lane / 4
//     handles output columns 4 (b / 4) .. +3 for k sub-slice b % 4; A operand = x[clip lane % 4][k], B operand =
//     W[column 4 (lane / 16) + lane % 4][k], D register i of lane = (clip i, that column).  With at most 4 clips per group
//     a 16-row tile would be 3/4 padding; this form does the same arithmetic in a quarter of the matrix-pipe time
//     (layout measured with scripts/probes/mfma4x4.hip).
template <bool SMALL>
__device__ __forceinline__ f32x4 mma_step(float x, float w, f32x4 acc) {
  if (SMALL) return __builtin_amdgcn_mfma_f32_4x4x1f32(x, w, acc, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x4f32(x, w, acc, 0, 0, 0);
}

// 4x4-block mode: add the partial sums of the four K sub-slices of a column group (lanes 4 apart inside a row of 16):
// after two row shifts the lane of sub-slice 3 holds ((s3 + s2) + (s1 + s0)); shifted-in lanes read zero
__device__ __forceinline__ f32x4 reduce_subslices(f32x4 v) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[i] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[i]), 0x114, 0xf, 0xf, true));   // row_shr:4
    v[i] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[i]), 0x118, 0xf, 0xf, true));   // row_shr:8
  }
  return v;
}

// cross-wave reduction in fixed order; result valid in wave 0 only (head phases)
__device__ __forceinline__ f32x4 reduce_waves(f32x4 acc, f32x4* red, int wave, int lane, int nw) {
  red[wave * 64 + lane] = acc;
  __syncthreads();
  f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
  if (wave == 0) {
    for (int w = 0; w < nw; ++w) {
      const f32x4 p = red[w * 64 + lane];
      v[0] += p[0]; v[1] += p[1]; v[2] += p[2]; v[3] += p[3];
    }
  }
  __syncthreads();
  return v;
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const float* gcfloat_ptr;
typedef __attribute__((address_space(1))) f32x4* gf32x4_wptr;

__device__ __forceinline__ unsigned sgpr(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }

}  // namespace mmk

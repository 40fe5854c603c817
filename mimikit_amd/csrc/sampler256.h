// One wave draws one class from 256 logits by inverting the CDF of softmax(logits / T) at a given uniform - the head of every
// sampled decode step (modules/targets.py:37-52 restated for a prescribed uniform).
//
// Same arithmetic as the general loop of the step kernels (value = (logit [/ learned temperature]) / T, exponentials against
// the row maximum, running sum in class order, first class whose running sum exceeds u * total and whose own term is not
// zero; the last such class when rounding leaves none), but laid out for the wave: lane i owns classes 4 i .. 4 i + 3 (one
// 16-byte read), the four exponentials are kept instead of being recomputed, and every wave-wide step - maximum, inclusive
// scan, the two index reductions - is DPP inside a row of 16 lanes plus four scalar reads across the rows, where the general
// loop spends 24 LDS round trips (ds_bpermute).  On the chain of every sampled step: 2.7 us -> ~0.7 us per clip.
#pragma once
#include "mmk_common.h"

namespace mmk {


#define MMK_DPP_F(old_, v_, CTRL, ROWMASK) \
  __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old_), __float_as_int(v_), CTRL, ROWMASK, 0xf, false))
#define MMK_DPP_I(old_, v_, CTRL, ROWMASK) __builtin_amdgcn_update_dpp(old_, v_, CTRL, ROWMASK, 0xf, false)

// (the readlane builtin is an integer one: a float argument would be CONVERTED, not reinterpreted)
__device__ __forceinline__ float readlane_f(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }

// maximum over the wave, in every lane
__device__ __forceinline__ float wave_max_dpp(float v) {
  v = fmaxf(v, MMK_DPP_F(v, v, 0xB1, 0xf));     // quad_perm [1,0,3,2]
  v = fmaxf(v, MMK_DPP_F(v, v, 0x4E, 0xf));     // quad_perm [2,3,0,1]
  v = fmaxf(v, MMK_DPP_F(v, v, 0x141, 0xf));    // row_half_mirror
  v = fmaxf(v, MMK_DPP_F(v, v, 0x140, 0xf));    // row_mirror: every lane holds its row's maximum
  const float r0 = readlane_f(v, 0), r1 = readlane_f(v, 16), r2 = readlane_f(v, 32), r3 = readlane_f(v, 48);
  return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

// inclusive prefix sum over the lanes of the wave (fixed order: Kogge-Stone inside a row, then the rows' totals)
__device__ __forceinline__ float wave_scan_dpp(float x) {
  x += MMK_DPP_F(0.f, x, 0x111, 0xf);           // row_shr:1 (lanes without a source add 0)
  x += MMK_DPP_F(0.f, x, 0x112, 0xf);           // row_shr:2
  x += MMK_DPP_F(0.f, x, 0x114, 0xf);           // row_shr:4
  x += MMK_DPP_F(0.f, x, 0x118, 0xf);           // row_shr:8
  x += MMK_DPP_F(0.f, x, 0x142, 0xa);           // row_bcast:15 into rows 1 and 3
  x += MMK_DPP_F(0.f, x, 0x143, 0xc);           // row_bcast:31 into rows 2 and 3
  return x;
}

__device__ __forceinline__ int wave_min_dpp(int v) {
  v = min(v, MMK_DPP_I(v, v, 0xB1, 0xf));
  v = min(v, MMK_DPP_I(v, v, 0x4E, 0xf));
  v = min(v, MMK_DPP_I(v, v, 0x141, 0xf));
  v = min(v, MMK_DPP_I(v, v, 0x140, 0xf));
  return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
             min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

__device__ __forceinline__ int wave_max_dpp_i(int v) {
  v = max(v, MMK_DPP_I(v, v, 0xB1, 0xf));
  v = max(v, MMK_DPP_I(v, v, 0x4E, 0xf));
  v = max(v, MMK_DPP_I(v, v, 0x141, 0xf));
  v = max(v, MMK_DPP_I(v, v, 0x140, 0xf));
  return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
             max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

// argmax with torch.argmax's tie rule (the first maximum wins), every lane of the wave holding a candidate (value, index):
// DPP inside the rows, four scalar reads across them
__device__ __forceinline__ int wave_argmax_first(float best, int bi) {
  auto take = [&](float ob, int oi) {
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  };
#define MMK_ARGMAX_STEP(CTRL) take(MMK_DPP_F(best, best, CTRL, 0xf), MMK_DPP_I(bi, bi, CTRL, 0xf))
  MMK_ARGMAX_STEP(0xB1);
  MMK_ARGMAX_STEP(0x4E);
  MMK_ARGMAX_STEP(0x141);
  MMK_ARGMAX_STEP(0x140);
#undef MMK_ARGMAX_STEP
  float rb = readlane_f(best, 0);
  int ri = __builtin_amdgcn_readlane(bi, 0);
#pragma unroll
  for (int row = 1; row < 4; ++row) {
    const float ob = readlane_f(best, 16 * row);
    const int oi = __builtin_amdgcn_readlane(bi, 16 * row);
    if (ob > rb || (ob == rb && oi < ri)) { rb = ob; ri = oi; }
  }
  return ri;
}

// The greedy pick of 256 logits that a head divides by ONE positive number (the learned temperature, networks/mlp.py:60-62) before
// the argmax (modules/targets.py:37-52), lane i holding classes 4 i .. 4 i + 3: the division keeps the order, so the first maximum of the
// RAW logits is the answer - unless it rounds an earlier, slightly smaller logit onto the maximum's quotient (the first maximum wins).
// Only when another logit lies within 4 ulp of the maximum are the four quotients formed and compared.  A NaN logit: the first one
// (torch.argmax).  One DPP maximum, four ballots and scalar bit scans instead of four IEEE divisions and an (index, value) reduction.
__device__ __forceinline__ int greedy_256(const float* lg, bool divided, float temp_logit, float min_temp, int lane) {
  typedef float f32x4_g __attribute__((ext_vector_type(4)));
  const f32x4_g v4 = *reinterpret_cast<const f32x4_g*>(lg + lane * 4);
  const float m = wave_max_dpp(fmaxf(fmaxf(v4[0], v4[1]), fmaxf(v4[2], v4[3])));
  int result = 0x7fffffff;
  bool odd = false, near = false;
  const float lim = m - fmaxf(fabsf(m) * 4.8e-7f, 1e-37f);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const unsigned long long mk = __ballot(v4[k] == m);
    if (mk) result = min(result, 4 * (int)__builtin_ctzll(mk) + k);
    odd = odd || v4[k] != v4[k];
    near = near || (v4[k] != m && v4[k] >= lim);
  }
  if (__any(odd)) {
    int cand = 0x7fffffff;
#pragma unroll
    for (int k = 3; k >= 0; --k)
      if (v4[k] != v4[k]) cand = lane * 4 + k;
    result = wave_min_dpp(cand);
  } else if (divided && __any(near)) {
    const float denom = fmaxf(1.0f / (1.0f + expf(-temp_logit)), min_temp);
    float best = v4[0] / denom;
    int bi = lane * 4;
#pragma unroll
    for (int k = 1; k < 4; ++k) {
      const float q = v4[k] / denom;
      if (q > best) { best = q; bi = lane * 4 + k; }
    }
    result = wave_argmax_first(best, bi);
  }
  return result > 255 ? 255 : result;
}


// lg: the row's 256 class logits (16-byte aligned, LDS or global); scale_by_denom: the learned-temperature divisor applies
__device__ __forceinline__ int sample_256(const float* lg, bool scale_by_denom, float denom, float T, float uniform, int lane) {
  typedef float f32x4_ __attribute__((ext_vector_type(4)));
  const f32x4_ v4 = *reinterpret_cast<const f32x4_*>(lg + lane * 4);
  float v[4], e[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) v[q] = (scale_by_denom ? v4[q] / denom : v4[q]) / T;
  const float mx = wave_max_dpp(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
  float local = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    e[q] = expf(v[q] - mx);
    local += e[q];
  }
  const float incl = wave_scan_dpp(local);
  const float total = readlane_f(incl, 63);
  const float target = uniform * total;
  float run = incl - local;
  int pick = 0x7fffffff, last_c = -1;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    run += e[q];
    if (e[q] > 0.f) last_c = lane * 4 + q;
    if (pick == 0x7fffffff && run > target && e[q] > 0.f) pick = lane * 4 + q;
  }
  pick = wave_min_dpp(pick);
  if (pick != 0x7fffffff) return pick;
  last_c = wave_max_dpp_i(last_c);
  return last_c < 0 ? 0 : last_c;
}

}  // namespace mmk

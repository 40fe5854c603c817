// Register-resident 1024-point complex FFT, one transform per wave (gfx950).  Shared by the STFT magnitude kernel
// (features.hip) and the inverse-STFT / Griffin-Lim kernels (istft.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace mmk {

// ---- n_fft = 1024: one transform per WAVE, 16 points per lane in registers ---------------------------------------------
// 1024 = 16 x 16 x 4.  With n = 64 n1 + n' and k = k1 + 16 k':   W^{nk} = W16^{n1 k1} W1024^{n' k1} W64^{n' k'},
// and with n' = 4a + b, k' = c + 16 d:                            W64^{n'k'} = W16^{ac} W64^{bc} W4^{bd}.
//   pass 1  lane n' holds x[64 n1 + n'] (the load pattern)  -> 16-point DFT over n1 in registers, twiddle W1024^{n' k1}
//   (LDS)   lane (k1, b) collects a = 0..15                 -> 16-point DFT over a, twiddle W64^{bc}
//   (LDS)   lane (k1, c mod 4) collects b for 4 values of c -> four 4-point DFTs over b:  X[k1 + 16 c + 256 d]
//   (LDS)   natural order -> |A[k]|, |B[k]| of the two real frames packed as re/im, consecutive lanes on consecutive bins
// The LDS buffer is private to the wave (wave-synchronous, no workgroup barrier in the loop); row strides of 68 and
// 17 complex keep the transposes bank-conflict free.  The generic kernel above runs five radix-4 passes through LDS
// with a workgroup barrier each; this one is bound by vector ALU work instead.
// a complex number is a native 2-vector: the compiler keeps (re, im) in an aligned register pair and emits packed fp32
// instructions (v_pk_add / v_pk_mul / v_pk_fma) for it instead of pairing unrelated scalars and shuffling them back
typedef float cf32 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ cf32 cadd(cf32 a, cf32 b) { return a + b; }
__device__ __forceinline__ cf32 csub(cf32 a, cf32 b) { return a - b; }
// The packed instructions pick either half of each 64-bit source for either half of the result (op_sel / op_sel_hi)
// and negate per half (neg_lo / neg_hi), so a complex product is two of them and a rotation by -i costs nothing; the
// compiler does not find these forms by itself (it scalarises the product and builds rotated copies with v_mov).
__device__ __forceinline__ cf32 cmul(cf32 a, cf32 b) {   // (a.x b.x - a.y b.y, a.x b.y + a.y b.x)
  cf32 t, d;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(d) : "v"(a), "v"(b), "v"(t));
  return d;
}
__device__ __forceinline__ cf32 cmul_k(cf32 a, cf32 k) {   // the same with a wave-uniform constant (scalar register pair)
  cf32 t, d;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "s"(k));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(d) : "v"(a), "s"(k), "v"(t));
  return d;
}
__device__ __forceinline__ cf32 add_mi(cf32 a, cf32 u) {   // a + (-i) u = (a.x + u.y, a.y - u.x)
  cf32 d;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(u));
  return d;
}
__device__ __forceinline__ cf32 sub_mi(cf32 a, cf32 u) {   // a - (-i) u = (a.x - u.y, a.y + u.x)
  cf32 d;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(u));
  return d;
}
__device__ __forceinline__ void radix4(cf32& a0, cf32& a1, cf32& a2, cf32& a3) {   // out_k = sum_j a_j (-i)^{jk}
  const cf32 t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), u = csub(a1, a3);
  a0 = cadd(t0, t2); a1 = add_mi(t1, u); a2 = csub(t0, t2); a3 = sub_mi(t1, u);
}
// 16-point DFT in place.  Input v[n1]; output X[k1] lands in v[i] with k1 = (i >> 2) + 4 (i & 3).
__device__ __forceinline__ void dft16(cf32 (&v)[16]) {
  constexpr float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f, h = 0.70710678118654752f;
#pragma unroll
  for (int q = 0; q < 4; ++q) radix4(v[q], v[4 + q], v[8 + q], v[12 + q]);      // v[4r + q] = T[q][r]
  // T[q][r] *= W16^{qr}:  W16^1 = (c1,-s1), ^2 = (h,-h), ^3 = (s1,-c1), ^4 = (0,-1), ^6 = (-h,-h), ^9 = (-c1, s1)
  v[4 * 1 + 1] = cmul_k(v[4 * 1 + 1], cf32{c1, -s1});
  v[4 * 1 + 2] = cmul_k(v[4 * 1 + 2], cf32{h, -h});
  v[4 * 1 + 3] = cmul_k(v[4 * 1 + 3], cf32{s1, -c1});
  v[4 * 2 + 1] = cmul_k(v[4 * 2 + 1], cf32{h, -h});
  v[4 * 2 + 2] = cf32{v[4 * 2 + 2].y, -v[4 * 2 + 2].x};
  v[4 * 2 + 3] = cmul_k(v[4 * 2 + 3], cf32{-h, -h});
  v[4 * 3 + 1] = cmul_k(v[4 * 3 + 1], cf32{s1, -c1});
  v[4 * 3 + 2] = cmul_k(v[4 * 3 + 2], cf32{-h, -h});
  v[4 * 3 + 3] = cmul_k(v[4 * 3 + 3], cf32{-c1, s1});
#pragma unroll
  for (int r = 0; r < 4; ++r) radix4(v[4 * r], v[4 * r + 1], v[4 * r + 2], v[4 * r + 3]);   // v[4r + s] = X[r + 4s]
}


constexpr int kFftWaveLds = 16 * 68;      // complex elements of a wave's private buffer (>= 64 x 17 and >= 1024)

// Where X[k] lives in the buffer when the transform is asked for the swizzled natural order: bits 2-3 of k flipped by bits 4-5.
// The last pass stores X[k1 + 16 c + 256 d] from lane (k1, c mod 4): in plain natural order the four lanes of a k1 that differ in
// c mod 4 write the same bank pair (16 complex apart), a 4-way conflict on all 16 stores of a transform (SQ_LDS_BANK_CONFLICT: 41 %
// of the STFT kernel's LDS-active cycles); with the flip the 16 lanes of a store group write 16 different bank pairs, and a reader
// whose lanes walk k = lane + 64 j reads conflict-free as before (lane -> fft_swz(lane) is a permutation inside each group of 16).
__device__ __forceinline__ int fft_swz(int k) { return k ^ (((k >> 4) & 3) << 2); }

// The twiddles of a lane for the whole kernel: t1[k1 - 1] = W1024^{lane k1} (pass 1), t2[c - 1] = W64^{(lane & 3) c} (pass 2), k1, c = 1 .. 15.
// Read from the table once per kernel instead of 30 LDS reads (the pass-1 ones with strides lane k1: 2- to 8-way bank conflicts for even k1)
// and their address arithmetic per transform; 60 registers.
struct FftLaneTw {
  cf32 t1[15], t2[15];
  __device__ __forceinline__ void load(const cf32* tw, int lane) {
#pragma unroll
    for (int k = 1; k < 16; ++k) {
      t1[k - 1] = tw[(lane * k) & 1023];
      t2[k - 1] = tw[(16 * (lane & 3) * k) & 1023];
    }
  }
};

// The twiddles in LDS, laid out by what a lane reads: t1[(k1 - 1) 64 + lane] = W1024^{lane k1} (pass 1: consecutive lanes on consecutive entries - read
// out of the plain table exp(-2 pi i m / 1024) at m = lane k1 the even k1 are 2- to 8-way bank conflicts, 20 % of the inverse kernels' LDS cycles),
// then t2[(c - 1) 4 + b] = W64^{b c} (pass 2: four distinct entries per read).  kFftTwLds complex entries.
constexpr int kFftTwLds = 15 * 64 + 15 * 4;
__device__ __forceinline__ void fill_twiddles_by_pass(cf32* twl, const cf32* __restrict__ table, int tid, int nthreads) {
  for (int e = tid; e < kFftTwLds; e += nthreads) {
    int m;
    if (e < 15 * 64) m = ((e & 63) * ((e >> 6) + 1)) & 1023;
    else m = (16 * ((e - 15 * 64) & 3) * (((e - 15 * 64) >> 2) + 1)) & 1023;
    twl[e] = table[m];
  }
}

// Forward DFT of 1024 points.  In: v[r] = x[lane + 64 r].  Out: X[k] in natural order in buf[0..1023] (SWZ: at buf[fft_swz(k)];
// wave-private LDS, kFftWaveLds complex); tw[m] = exp(-2 pi i m / 1024), or ltw: the lane's twiddles in registers.  Wave-synchronous: no
// workgroup barrier.
template <bool SWZ, bool REGTW, bool BYPASS = false>
__device__ __forceinline__ void fft1024_wave_impl(cf32 (&v)[16], cf32* buf, const cf32* tw, int lane, const FftLaneTw& ltw) {
  constexpr int N = 1024;
  const int k1b = lane >> 2, lo2 = lane & 3;                  // (k1, b) of pass 2 = (k1, c mod 4) of pass 3
  // ---- pass 1: DFT over n1, twiddle W1024^{n' k1}, transpose ---------------------------------------------------------
  dft16(v);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int k1 = (i >> 2) + 4 * (i & 3);
    if (k1 != 0) v[i] = cmul(v[i], REGTW ? ltw.t1[k1 - 1 < 0 ? 0 : k1 - 1] : (BYPASS ? tw[(k1 - 1) * 64 + lane] : tw[(lane * k1) & (N - 1)]));
    buf[k1 * 68 + lane] = v[i];
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int a2 = 0; a2 < 16; ++a2) v[a2] = buf[k1b * 68 + 4 * a2 + lo2];
  __builtin_amdgcn_wave_barrier();
  // ---- pass 2: DFT over a, twiddle W64^{b c}, transpose -----------------------------------------------------------------
  dft16(v);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = (i >> 2) + 4 * (i & 3);
    if (c != 0) v[i] = cmul(v[i], REGTW ? ltw.t2[c - 1 < 0 ? 0 : c - 1] : (BYPASS ? tw[15 * 64 + (c - 1) * 4 + lo2] : tw[(16 * lo2 * c) & (N - 1)]));
    buf[lane * 17 + c] = v[i];                              // row (k1, b)
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int bq = 0; bq < 4; ++bq) v[4 * g + bq] = buf[(k1b * 4 + bq) * 17 + lo2 + 4 * g];   // c = (c mod 4) + 4 g
  __builtin_amdgcn_wave_barrier();
  // ---- pass 3: DFT over b -> X[k1 + 16 c + 256 d], natural order in LDS --------------------------------------------------
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    radix4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
#pragma unroll
    for (int d = 0; d < 4; ++d) buf[(SWZ ? (k1b ^ (lo2 << 2)) : k1b) + 16 * (lo2 + 4 * g) + 256 * d] = v[4 * g + d];
  }
  __builtin_amdgcn_wave_barrier();
}

template <bool SWZ = false>
__device__ __forceinline__ void fft1024_wave(cf32 (&v)[16], cf32* buf, const cf32* tw, int lane) {
  FftLaneTw none;
  fft1024_wave_impl<SWZ, false>(v, buf, tw, lane, none);
}
// (twl: the table of fill_twiddles_by_pass)
template <bool SWZ = false>
__device__ __forceinline__ void fft1024_wave_bypass(cf32 (&v)[16], cf32* buf, const cf32* twl, int lane) {
  FftLaneTw none;
  fft1024_wave_impl<SWZ, false, true>(v, buf, twl, lane, none);
}
template <bool SWZ = false>
__device__ __forceinline__ void fft1024_wave_regtw(cf32 (&v)[16], cf32* buf, const FftLaneTw& ltw, int lane) {
  fft1024_wave_impl<SWZ, true>(v, buf, nullptr, lane, ltw);
}

}  // namespace mmk

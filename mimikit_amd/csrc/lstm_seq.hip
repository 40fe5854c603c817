// All time steps of one bidirectional LSTM layer in ONE launch, the recurrent weights resident in registers (gfx950).
//
// Reference: torch.nn.LSTM(bidirectional=True) inside the Seq2Seq encoder / decoder (s2s_lstm_v2.py:90-171); per frame
//     gates = (W_ih x_t + b)  [one GEMM over all frames, done before this kernel]  +  W_hh h_{t-1}
//     i, f, o = sigmoid(.), g = tanh(.);  c' = f c + i g;  h' = o tanh(c')                    (ATen LSTMCell)
//
// The per-step kernel (lstm_step.hip) streams both W_hh (33.5 MB at H = 1024) through every CU once per frame: 17.6 us per frame
// against 6.8 us of fp32 MFMA work, and the stream and the MFMAs get in each other's way.  Here the launch lasts for the whole
// sequence and a workgroup keeps its slice of W_hh - 16 hidden units x 4 gates x all of K, 128 registers per lane at H = 1024 - for
// all frames.  Same split as the per-step kernel: a workgroup owns 16 units x 32 rows of one direction (64 rows x 1024 units x 2
// directions = 256 workgroups = one per CU), its 8 waves take 1/8 of K each, the partial sums meet in LDS, one thread runs the cell
// of one (row, unit) pair and keeps that pair's cell state in a register.
//
// What crosses workgroups is the new hidden state: the 64 workgroups of a (direction, 32-row half) group need each other's 16 units.
// No barrier and no counters: the state of sequence step s lives at its own address (`xch`, one image per step), every word of it
// starts out as a poison pattern (0xFFFFFFFF: a NaN no LSTM arithmetic produces) and the consumer's load of a K range IS the poll:
// a 16-byte fragment with no poison word in it is final, one with poison is requested again.  The set of images alternates between
// launches and a launch poisons the set of the next one, so no memset sits between the layers.  The two 16-row blocks of a workgroup
// are independent sequences: block 1's products run while block 0's new state travels, and the other way round.
//
// A consumer that sees poison for ~20 ms gives up, raises `err` and lets the launch drain (everybody else follows through `err`):
// the plan then repeats the call with the per-step kernel.  That covers a chip on which the 256 workgroups are not co-resident.
#include <type_traits>

#include "lstm_seq.h"

namespace mmk {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kSqThreads = 512;
constexpr int kSqWaves = kSqThreads / 64;
constexpr uint32_t kSqPoison = 0xFFFFFFFFu;
constexpr unsigned kSqSpinLimit = 1u << 14;     // re-requests of one fragment (1 - 2 us each) before giving up

#ifndef MMK_SQ_AHEAD
#define MMK_SQ_AHEAD 1         // request a phase's state fragments during the phase before (two row blocks)
#endif
#ifndef MMK_SQ_AHEAD_AT
#define MMK_SQ_AHEAD_AT 5      // ... in front of the products of chunk CPW * this / 8
#endif
#ifndef MMK_SQ_FAST_RCP
#define MMK_SQ_FAST_RCP 1      // v_rcp_f32 in the cell's activations
#endif
#ifndef MMK_SQ_CHECK_GROUP
#define MMK_SQ_CHECK_GROUP 1   // fragments whose poison check is one compare and one branch
#endif
#ifndef MMK_SQ_MAX3
#define MMK_SQ_MAX3 1
#endif
#ifndef MMK_SQ_SPREAD
#define MMK_SQ_SPREAD 1        // ... and dealt out over the chunks from there to the phase's last (0: all at once)
#endif
#ifndef MMK_SQ_FIRST_SC1
#define MMK_SQ_FIRST_SC1 1     // 1: a fragment's FIRST request goes past the L2 as well (re-requests always do)
#endif
#ifndef MMK_SQ_ACC2
#define MMK_SQ_ACC2 0          // experiment: two accumulator sets per gate (even / odd K quarter of a chunk)
#endif
#ifndef MMK_SQ_NOLOAD
#define MMK_SQ_NOLOAD 0        // timing experiment only (wrong results): no state fragments are requested
#endif
#ifndef MMK_SQ_LOCALSRC
#define MMK_SQ_LOCALSRC 0      // timing experiment only (wrong results): every fragment is read from the caller's state (never poisoned)
#endif
#ifndef MMK_SQ_NOCELL
#define MMK_SQ_NOCELL 0        // timing experiment only (wrong results): the cell does not read the partial sums
#endif
#ifndef MMK_SQ_NOCHECK
#define MMK_SQ_NOCHECK 0       // experiment only (wrong results possible): no poison check
#endif
#ifndef MMK_SQ_BUBBLE
#define MMK_SQ_BUBBLE 0        // s_nop behind every second MFMA (what the per-step kernel needs to let its weight stream land)
#endif

typedef unsigned u32x4s __attribute__((ext_vector_type(4)));

// any of the four words still the poison pattern (the largest unsigned value: one v_max3, one v_max, one compare)
__device__ __forceinline__ bool sq_poisoned(const u32x4s& v) {
#if !MMK_SQ_MAX3
  return (v.x == kSqPoison) | (v.y == kSqPoison) | (v.z == kSqPoison) | (v.w == kSqPoison);
#endif
  unsigned m;
  asm("v_max3_u32 %0, %1, %2, %3" : "=v"(m) : "v"(v.x), "v"(v.y), "v"(v.z));
  return max(m, v.w) == kSqPoison;
}

// the cell's activations on the hardware reciprocal (1 ulp; the correctly rounded one of sigmoid_fast / tanh_fast is a ten-instruction
// sequence, five of them per cell between a block's partial sums and its new state on the wire)
__device__ __forceinline__ float sq_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f)); }
__device__ __forceinline__ float sq_tanh(float x) { return fmaf(__builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -2.8853900817779268f)), 2.f, -1.f); }

// CPW - FROM state fragments of a wave's K range for one row block, 16 bytes per lane each, past the L2 (sc1): buffer loads the
// compiler counts itself - a fragment is in flight across the frame loop's back edge, where only the compiler knows which of its
// copies of a register is the live one
template <int CPW, int FROM = 0>
__device__ __forceinline__ void sq_request(u32x4s (&set)[CPW], const __amdgpu_buffer_rsrc_t& image, int byte_off) {
  if (MMK_SQ_NOLOAD) return;
#pragma unroll
  for (int u = FROM; u < CPW; ++u) set[u] = __builtin_amdgcn_raw_buffer_load_b128(image, byte_off, u * 64, MMK_SQ_FIRST_SC1 ? 16 : 0);
}
template <int CPW, int FROM>
__device__ __forceinline__ void sq_rerequest(u32x4s (&set)[CPW], const __amdgpu_buffer_rsrc_t& image, int byte_off) {
#pragma unroll
  for (int u = FROM; u < CPW; ++u) set[u] = __builtin_amdgcn_raw_buffer_load_b128(image, byte_off, u * 64, 16);
}

template <int CPW, int RB, bool STAMPS>   // K-chunks per wave: H = 128 CPW; 16-row blocks per workgroup; STAMPS: diagnostic build
__global__ __launch_bounds__(kSqThreads) void lstm_seq_kernel(const LstmSeqArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr int KC = CPW * kSqWaves;
  constexpr int H = KC * 16;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ub = blockIdx.x;                       // block of 16 hidden units
  const int m_first = blockIdx.y * (16 * RB);
  const int mg = min(16 * RB, a.M - m_first);
  const int di = blockIdx.z;
  const LstmSeqDir d = a.dir[di];
  f32x4* red = reinterpret_cast<f32x4*>(smem_raw);                      // split-K partials [slot][gate][wave][lane]

  const int e_m = tid >> 4, e_n = tid & 15;                             // this thread's (row, unit) pair
  const int unit = ub * 16 + e_n;
  const bool has_pair = e_m < 16 * RB;
  const bool cell = e_m < mg;
  const int my_rb = e_m >> 4;
  const int64_t row = m_first + (cell ? e_m : 0);
  const int64_t image = (int64_t)a.rows_pad * H;                        // floats of one (step, direction) state image

  // ---- poison the images of the next launch (all rows of the plan's largest batch, whatever this launch's M is) ----------------
  if (has_pair)
    for (int r = m_first + e_m; r < a.rows_pad; r += gridDim.y * 16 * RB)
      for (int s = 0; s + 1 < a.n_steps; ++s)
        reinterpret_cast<uint32_t*>(a.xch_next)[(int64_t)(s * 2 + di) * image + (int64_t)r * H + unit] = kSqPoison;

  // ---- this wave's slice of W_hh: 4 gate tiles x CPW chunks, resident for the launch -----------------------------------------------
  f32x4 w[CPW][4];
  const int c0 = wave * CPW;
  {
    const f32x4* wsrc = reinterpret_cast<const f32x4*>(d.whh_wp) + ((int64_t)ub * KC + c0) * 64 + lane;
    const int64_t gate_stride = (int64_t)KC * KC * 64;                   // f32x4 elements between the gates' tile rows
#pragma unroll
    for (int u = 0; u < CPW; ++u)
#pragma unroll
      for (int g = 0; g < 4; ++g) w[u][g] = wsrc[g * gate_stride + u * 64];
  }
  float c_reg = 0.f;
  if (cell && !a.zero_state) c_reg = d.c[row * H + unit];
  // The weights have to be IN their registers before the frame loop: a load still pending for the compiler at the loop's entry makes
  // it wait in front of the first products of every phase of every frame - and such a wait drains the whole memory pipe, the cell's
  // written-through stores and the next phase's fragments included (measured: the phase then starts ~1 us late).
#pragma unroll
  for (int u = 0; u < CPW; ++u)
#pragma unroll
    for (int g = 0; g < 4; ++g) asm volatile("" : "+v"(w[u][g]));
  asm volatile("" : "+v"(c_reg));

  // byte offsets of the two row blocks this lane reads the state of (MFMA A operand: row lane & 15, K offset 4 (lane >> 4)); clamped
  int hoff[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const int m = rb * 16 + (lane & 15);
    hoff[rb] = (int)(((int64_t)(m_first + (m < mg ? m : 0)) * H + c0 * 16 + 4 * (lane >> 4)) * sizeof(float));
  }
  auto image_of = [&](int s_src) {     // the state a step reads: the caller's before step 0, else the image step s_src - 1 wrote
    const float* base = (s_src == 0 || MMK_SQ_LOCALSRC) ? d.h : a.xch + (int64_t)((s_src - 1) * 2 + di) * image;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, -1, 0x00020000);
  };
  bool check = true;                                // false once this wave has given up: the launch only drains
  // diagnostic build: 10 ns ticks of [phase start, products done, barrier passed, cell done, re-requests] per (phase, wave) of one workgroup
  const bool stamping = STAMPS && a.stamps != nullptr && blockIdx.x == a.stamp_wg && blockIdx.y == 0 && blockIdx.z == 0 && lane == 0;
  auto stamp = [&](int ph, int k, unsigned long long v) { if (stamping && ph < 16) a.stamps[(ph * kSqWaves + wave) * 8 + k] = v; };
  unsigned rerequests = 0;
  float pooled = 0.f;                               // this thread's folded column summed over the frames the pooling takes

  // The state fragments of a phase are requested during the phase before (two register sets, RB == 2): block 1's fragments of step s
  // while block 0's products of step s run, block 0's of step s + 1 during block 1's.  With one block there is nothing to overlap:
  // the fragments are requested where the phase starts.
  constexpr bool kAhead = RB == 2 && MMK_SQ_AHEAD;
  constexpr int kAheadAt = (CPW * MMK_SQ_AHEAD_AT) / 8;     // chunk in front of whose products the next phase's requests go out
  u32x4s hv[RB][CPW];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int u = 0; u < CPW; ++u) hv[rb][u] = u32x4s{0u, 0u, 0u, 0u};
  if (kAhead && !a.zero_state) sq_request<CPW>(hv[0], image_of(0), hoff[0]);

  for (int s = 0; s < a.n_steps; ++s) {
    const int t = di == 0 ? s : a.n_steps - 1 - s;
    const bool product = s > 0 || !a.zero_state;
    const bool polled = s > 0;                      // the first step reads the caller's state: final before the launch
    const bool last = s + 1 == a.n_steps;
    const __amdgpu_buffer_rsrc_t src_now = image_of(s), src_next = image_of(last ? s : s + 1);
    float* hout = last ? d.h : a.xch + (int64_t)(s * 2 + di) * image;

    auto phase = [&](auto rbc) {
      constexpr int rb = decltype(rbc)::value;
      constexpr int nrb = RB == 2 ? rb ^ 1 : 0;
      const bool mine = has_pair && my_rb == rb;
      if (STAMPS) { rerequests = 0; stamp(s * RB + rb, 0, __builtin_amdgcn_s_memrealtime()); stamp(s * RB + rb, 5, __builtin_amdgcn_s_memtime()); }
      // partial-sum slot of this phase: one barrier per phase, so a slot must not be rewritten before the barrier after its readers
      const int slot = RB == 2 ? rb : (s & 1);
      // the additive gate terms of this thread's pair
      // (loaded where they are used - `mine && cell` - so that no path leaves them pending: the compiler would wait for the whole
      //  memory pipe, the written-through store included, where the registers are written next)
      float ga[4];          // (no initial value: it would be a write the compiler orders behind everything pending)
      if (mine && cell) {
        const float* g0 = d.gadd + row * a.gadd_ld + (int64_t)t * a.gadd_ts + unit;
#pragma unroll
        for (int g = 0; g < 4; ++g) ga[g] = g0[g * H];
      }
      __builtin_amdgcn_sched_barrier(0);
      // what the next phase reads: the other block of this step, or block 0 of the next one.  Every path through a phase defines
      // the other set anew (the launch's last phase asks for its own source once more): a set that could keep its old value is
      // alive around the loop for the compiler, which then moves it from register to register at the phase boundaries - and waits
      // for every fragment in flight first.
      auto request_next = [&]() {
        if constexpr (rb == 0 && RB == 2) sq_request<CPW>(hv[nrb], src_now, hoff[nrb]);
        else sq_request<CPW>(hv[nrb], src_next, hoff[nrb]);
      };
      // ... dealt out over the chunks kAheadAt .. CPW - 1 of this phase's products, in the order they will be used: eight requests
      // at once from every wave of the CU queue up in its address unit for ~1000 clocks, and a wave waiting there issues no MFMA
      auto request_next_part = [&](auto uc) {
        constexpr int u = decltype(uc)::value;
        constexpr int span = CPW - kAheadAt;
#pragma unroll
        for (int v = 0; v < CPW; ++v)
          if (kAheadAt + (v * span) / CPW == u) {
            if (MMK_SQ_NOLOAD) continue;
            if constexpr (rb == 0 && RB == 2) hv[nrb][v] = __builtin_amdgcn_raw_buffer_load_b128(src_now, hoff[nrb], v * 64, MMK_SQ_FIRST_SC1 ? 16 : 0);
            else hv[nrb][v] = __builtin_amdgcn_raw_buffer_load_b128(src_next, hoff[nrb], v * 64, MMK_SQ_FIRST_SC1 ? 16 : 0);
          }
      };
      f32x4 acc[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#if MMK_SQ_ACC2
      f32x4 acc2[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) acc2[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#endif
      if (product) {
        if (!kAhead) sq_request<CPW>(hv[rb], src_now, hoff[rb]);
        auto chunk = [&](auto uc) {
          constexpr int u = decltype(uc)::value;
          if constexpr (u < CPW) {
            if constexpr (kAhead && MMK_SQ_SPREAD && u >= kAheadAt) {
              request_next_part(uc);
              __builtin_amdgcn_sched_barrier(0);
            } else if constexpr (kAhead && !MMK_SQ_SPREAD && u == kAheadAt) {
              request_next();
              __builtin_amdgcn_sched_barrier(0);
            }
            constexpr int kGroup = MMK_SQ_CHECK_GROUP < CPW ? MMK_SQ_CHECK_GROUP : CPW;     // fragments checked together
            if (u % kGroup == 0 && polled && check && !MMK_SQ_NOCHECK && !MMK_SQ_NOLOAD) {
              unsigned spins = 0;
              auto group_poisoned = [&]() {
                unsigned m = 0;
#pragma unroll
                for (int v = u; v < u + kGroup && v < CPW; ++v) {
                  asm("v_max3_u32 %0, %1, %2, %3" : "=v"(m) : "v"(m), "v"(hv[rb][v].x), "v"(hv[rb][v].y));
                  asm("v_max3_u32 %0, %1, %2, %3" : "=v"(m) : "v"(m), "v"(hv[rb][v].z), "v"(hv[rb][v].w));
                }
                return m == kSqPoison;
              };
              while (__builtin_amdgcn_ballot_w64(kGroup == 1 ? sq_poisoned(hv[rb][u]) : group_poisoned()) != 0) {
                // not there yet: ask again for this fragment and the ones behind it (their producers are as late), past the L2
                sq_rerequest<CPW, u>(hv[rb], src_now, hoff[rb]);
                // (all of them landed before the check: the compiler's count of what is pending where the loop is left then is the
                //  straight path's, not "whatever this loop may have requested last")
#pragma unroll
                for (int v = u; v < CPW; ++v) asm volatile("" : "+v"(hv[rb][v]));
                if (STAMPS) ++rerequests;
                if (++spins > kSqSpinLimit || (MMK_WAIT_ERR_LOOK && (spins & 63u) == 0 && __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                  if (lane == 0) atomicOr(a.err, 1u);
                  check = false;
                  break;
                }
              }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
              for (int g = 0; g < 4; ++g) {
#if MMK_SQ_ACC2
                if (i & 1) acc2[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(hv[rb][u][i]), w[u][g][i], acc2[g], 0, 0, 0);
                else
#endif
                acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(hv[rb][u][i]), w[u][g][i], acc[g], 0, 0, 0);
#if MMK_SQ_BUBBLE
                if (g & 1) {
                  __builtin_amdgcn_sched_barrier(0);
                  asm volatile("s_nop 7");
                  __builtin_amdgcn_sched_barrier(0);
                }
#endif
              }
            }
            __builtin_amdgcn_sched_barrier(0);     // the next chunk's wait stays behind these MFMAs
          }
        };
        chunk(std::integral_constant<int, 0>{}); chunk(std::integral_constant<int, 1>{}); chunk(std::integral_constant<int, 2>{});
        chunk(std::integral_constant<int, 3>{}); chunk(std::integral_constant<int, 4>{}); chunk(std::integral_constant<int, 5>{});
        chunk(std::integral_constant<int, 6>{}); chunk(std::integral_constant<int, 7>{});
        static_assert(CPW <= 8, "H <= 1024");
#if MMK_SQ_ACC2
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] += acc2[g];
#endif
#pragma unroll
        for (int g = 0; g < 4; ++g) red[((slot * 4 + g) * kSqWaves + wave) * 64 + lane] = acc[g];
      } else if (kAhead) {
        // (the gate terms landed: on this path they would be the youngest requests, and the compiler's wait for them in the cell is
        //  the more cautious of the two paths' - behind the products they are OLDER than the next phase's fragments)
        if (mine && cell) asm volatile("" : "+v"(ga[0]), "+v"(ga[1]), "+v"(ga[2]), "+v"(ga[3]));
        if constexpr (rb == RB - 1) {
          request_next();   // zero state: the first products are block 0's of step 1
        } else {
#pragma unroll
          for (int u = 0; u < CPW; ++u) hv[nrb][u] = u32x4s{0u, 0u, 0u, 0u};
        }
      }
      if (STAMPS) stamp(s * RB + rb, 1, __builtin_amdgcn_s_memrealtime());
      __syncthreads();
      if (STAMPS) { stamp(s * RB + rb, 2, __builtin_amdgcn_s_memrealtime()); stamp(s * RB + rb, 4, rerequests); }
      if (mine) {
        const int r = e_m & 15;
        const int frag = ((r >> 2) * 16 + e_n) * 4 + (r & 3);          // (row r, col n) of a 16x16 accumulator image
        float sum[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float v = 0.f;
          if (product && !MMK_SQ_NOCELL) {
            const float* f = reinterpret_cast<const float*>(red + (slot * 4 + g) * kSqWaves * 64) + frag;
#pragma unroll
            for (int wv = 0; wv < kSqWaves; ++wv) v += f[wv * 256];
          }
          sum[g] = v;
        }
        if (cell) {
#pragma unroll
          for (int g = 0; g < 4; ++g) sum[g] += ga[g];
#if MMK_SQ_FAST_RCP
          const float ig = sq_sigmoid(sum[0]), fg = sq_sigmoid(sum[1]), cg = sq_tanh(sum[2]), og = sq_sigmoid(sum[3]);
          c_reg = fg * c_reg + ig * cg;
          const float hn = og * sq_tanh(c_reg);
#else
          const float ig = sigmoid_fast(sum[0]), fg = sigmoid_fast(sum[1]), cg = tanh_fast(sum[2]), og = sigmoid_fast(sum[3]);
          c_reg = fg * c_reg + ig * cg;
          const float hn = og * tanh_fast(c_reg);
#endif
          if (last) {
            hout[row * H + unit] = hn;
            d.c[row * H + unit] = c_reg;
          } else {
            __hip_atomic_store(hout + row * H + unit, hn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // written through: other XCDs poll it
          }
          if (d.y) d.y[row * a.y_ld + (int64_t)t * a.y_ts + unit] = hn;
          if (a.fold || a.pool) {
            // folded column (di H + unit) / 2 = this unit + its odd neighbour (the next lane; rows live or dead sixteen lanes at a time)
            const float pair = hn + __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, hn), 0xB1, 0xF, 0xF, true));
            if ((e_n & 1) == 0) {
              const int64_t at = row * a.y_ld + (int64_t)t * a.y_ts + ((di * H + unit) >> 1);
              const float v = a.res ? a.res[at] + pair : pair;
              if (a.fold) a.fold[at] = v;
              if (a.pool) {
                if (a.pool_mode >= 2 || t == 0 || t == a.n_steps - 1) pooled += v;
                if (last) {
                  const float scale = a.pool_mode == 1 ? 0.5f : (a.pool_mode == 3 ? 1.f / (float)a.n_steps : 1.f);
                  a.pool[row * H + ((di * H + unit) >> 1)] = a.pool_mode == 3 ? pooled / (float)a.n_steps : pooled * scale;
                }
              }
            }
          }
        }
      }
      if (STAMPS) stamp(s * RB + rb, 3, __builtin_amdgcn_s_memrealtime());
    };
    phase(std::integral_constant<int, 0>{});
    if constexpr (RB > 1) phase(std::integral_constant<int, 1>{});
  }
}

bool lstm_seq_supported(int H, int M, int n_steps, int n_cu) {
  if (!(H == 128 || H == 256 || H == 512 || H == 1024) || n_steps < 2 || M < 1) return false;
  const int rb = M > 16 ? 2 : 1;
  const int64_t wgs = (int64_t)(H / 16) * ((M + 16 * rb - 1) / (16 * rb)) * 2;
  return wgs <= n_cu;          // every workgroup waits for the others: all of them have to be on the chip at once
}

size_t lstm_seq_xch_floats(int H, int rows_pad, int n_steps) { return (size_t)(n_steps - 1) * 2 * rows_pad * H; }

int launch_lstm_seq(const LstmSeqArgs& a, hipStream_t stream) {
  const int rb = a.M > 16 ? 2 : 1;
  const size_t lds = (size_t)2 * 4 * kSqWaves * 64 * 16;          // two slots of split-K partials
  dim3 grid(a.H / 16, (a.M + 16 * rb - 1) / (16 * rb), 2), block(kSqThreads);
#ifdef MMK_DIAG
#define MMK_SQ_ST(CPW_, RB_)                                                                                          \
  if (a.stamps) hipLaunchKernelGGL((lstm_seq_kernel<CPW_, RB_, true>), grid, block, lds, stream, a);                  \
  else hipLaunchKernelGGL((lstm_seq_kernel<CPW_, RB_, false>), grid, block, lds, stream, a)
#else
#define MMK_SQ_ST(CPW_, RB_) hipLaunchKernelGGL((lstm_seq_kernel<CPW_, RB_, false>), grid, block, lds, stream, a)
#endif
#define MMK_SQ(CPW_)                   \
  if (rb == 2) { MMK_SQ_ST(CPW_, 2); } \
  else { MMK_SQ_ST(CPW_, 1); }
  switch (a.H) {
    case 128: MMK_SQ(1); break;
    case 256: MMK_SQ(2); break;
    case 512: MMK_SQ(4); break;
    case 1024: MMK_SQ(8); break;
    default: return fail(MMK_ERR_UNSUPPORTED, "lstm sequence kernel: H=%d", a.H);
  }
#undef MMK_SQ
#undef MMK_SQ_ST
  MMK_HIP(hipGetLastError());
  return MMK_OK;
}

}  // namespace mmk

// Probe: WaveNet cfg 2 (10 layers x 64 channels, kernel 2, gated, skips 64, embedding in, MLP head 128 -> 256 classes, 8 clips) as a
// PIPELINE OF WORKGROUPS THAT OWN WHOLE LAYERS - what DESIGN.md section 9 proposes instead of the chain kernel's one exchange per layer.
//   hipcc --offload-arch=gfx950 -O3 -o wn_layer_pipe wn_layer_pipe.hip && ./wn_layer_pipe
// A clip is served by 4 workgroups on one XCD (workgroup b: clip b % 8, stage b / 8): stage 0 owns layers 0-2, stage 1 layers 3-5,
// stage 2 layers 6-7, stage 3 layers 8-9 and the head.  Every stage keeps its layers' weights in registers for the whole launch
// (48 floats per thread and layer), a step travels through the four stages as two 64-float vectors (layer input, skip sum) in
// data-tagged 8-byte granules {step + 1, value}; the head's class goes back to stage 0 the same way.  The taps x_l[t - d_l] come from
// per-layer rings in global memory (L2), requested at the start of a stage's visit.  The arithmetic is the real one (so the time is
// honest); the first steps are checked against a host loop.  Prints us per step.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef unsigned long long u64;
constexpr int C = 64, L = 10, NCLIP = 8, HID = 128, Q = 256, NST = 4;
constexpr int kThreads = 512;

struct Params {
  const float* conv_w;   // [L][128 out][128 in]  in = [x(t - d) | x(t)], out = [f | g]
  const float* conv_b;   // [L][128]
  const float* rs_w;     // [L][128 out][64 in]   out = [res | skip]
  const float* rs_b;     // [L][128]
  const float* emb;      // [Q][64]
  const float* fc0_w;    // [HID][64]
  const float* fc0_b;    // [HID]
  const float* fc2_w;    // [Q][HID]
  const float* fc2_b;    // [Q]
  float* ring;           // [L][ring_len][NCLIP][64]
  u64* xg;               // [NST][NCLIP][128] granules: stage s reads xg[s] (x | skip) written by stage s - 1
  u64* cg;               // [NCLIP] class granule {step, class} for stage 0
  long long* out;        // [NCLIP][n_steps]
  int n_steps, ring_len;
  unsigned* err;
};

__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + __expf(-v)); }
__device__ __forceinline__ float mishf_(float v) { return v * tanhf(log1pf(__expf(v))); }
__device__ __forceinline__ float quad_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
  return v;
}
__device__ __forceinline__ u64 poll(const u64* p, unsigned epoch, unsigned* err) {
  u64 g = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  unsigned spins = 0;
  while ((unsigned)(g >> 32) != epoch) {
    if (++spins > (1u << 22)) { atomicExch(err, 1u); break; }
    g = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return g;
}

template <int STAGE>
__device__ void run_stage(const Params p, int clip) {
  constexpr int l0 = STAGE == 0 ? 0 : (STAGE == 1 ? 3 : (STAGE == 2 ? 6 : 8));
  constexpr int NL = STAGE < 2 ? 3 : 2;
  __shared__ float xs[C], taps[NL][C], fg[2 * C], zs[C], sk[C], hid[HID], lg[Q];
  __shared__ float embs[STAGE == 0 ? Q * C : 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int o = tid >> 2, kq = tid & 3;
  // ---- weights -> registers -------------------------------------------------------------------------------------------
  float wc[NL][32], wr[NL][16], bc[NL], br[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int l = l0 + i;
#pragma unroll
    for (int k = 0; k < 32; ++k) wc[i][k] = p.conv_w[((size_t)l * 128 + o) * 128 + kq * 32 + k];
#pragma unroll
    for (int k = 0; k < 16; ++k) wr[i][k] = p.rs_w[((size_t)l * 128 + o) * 64 + kq * 16 + k];
    bc[i] = p.conv_b[l * 128 + o];
    br[i] = p.rs_b[l * 128 + o];
  }
  float w0[STAGE == 3 ? 16 : 1], w2[STAGE == 3 ? 64 : 1], b0 = 0.f, b2 = 0.f;
  if (STAGE == 3) {
#pragma unroll
    for (int k = 0; k < 16; ++k) w0[k] = p.fc0_w[(size_t)o * 64 + kq * 16 + k];
    b0 = p.fc0_b[o];
#pragma unroll
    for (int k = 0; k < 64; ++k) w2[k] = p.fc2_w[(size_t)(tid >> 1) * HID + (tid & 1) * 64 + k];
    b2 = p.fc2_b[tid >> 1];
  }
  if (STAGE == 0) for (int i = tid; i < Q * C; i += kThreads) embs[i] = p.emb[i];
  __syncthreads();
  const u64* in_g = p.xg + ((size_t)STAGE * NCLIP + clip) * 128;
  u64* out_g = p.xg + ((size_t)((STAGE + 1) % NST) * NCLIP + clip) * 128;
  for (int t = 0; t < p.n_steps; ++t) {
    // ---- taps of my layers (addresses known): threads 0 .. 64 NL - 1 -----------------------------------------------------
    float tap = 0.f;
    if (tid < C * NL) {
      const int i = tid >> 6, l = l0 + i, d = 1 << l;
      // (read past this CU's L1: the slot was read d steps ago and rewritten since)
      tap = t >= d ? __hip_atomic_load(p.ring + (((size_t)l * p.ring_len + ((t - d) & (p.ring_len - 1))) * NCLIP + clip) * C + lane, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT)
                   : 0.f;
    }
    // ---- this step's input ---------------------------------------------------------------------------------------------------
    if (STAGE == 0) {
      if (wave == 0) {
        int cls = 128;                                                   // the prompt's last class
        if (t > 0) cls = (int)(unsigned)poll(p.cg + clip, (unsigned)t, p.err);
        xs[lane] = embs[cls * C + lane];
        sk[lane] = 0.f;
      }
    } else if (tid < 2 * C) {
      const u64 g = poll(in_g + tid, (unsigned)(t + 1), p.err);
      (tid < C ? xs : sk)[tid & (C - 1)] = __uint_as_float((unsigned)g);
    }
    if (tid < C * NL) taps[tid >> 6][lane] = tap;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int l = l0 + i;
      // the layer's input at t goes to its ring for later taps
      if (tid < C) p.ring[(((size_t)l * p.ring_len + (t & (p.ring_len - 1))) * NCLIP + clip) * C + tid] = xs[tid];
      // (f | g) = W [x(t - d) | x(t)] + b : thread (o, kq) takes 32 inputs
      const float* src = kq < 2 ? taps[i] + kq * 32 : xs + (kq - 2) * 32;
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < 32; ++k) acc = fmaf(wc[i][k], src[k], acc);
      acc = quad_sum(acc);
      if (kq == 0) fg[o] = acc + bc[i];
      __syncthreads();
      if (tid < C) zs[tid] = tanhf(fg[tid]) * sigmoidf_(fg[C + tid]);
      __syncthreads();
      float a2 = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) a2 = fmaf(wr[i][k], zs[kq * 16 + k], a2);
      a2 = quad_sum(a2);
      __syncthreads();                                                    // every thread has read xs / zs of this layer
      if (kq == 0) {
        if (o < C) xs[o] = xs[o] + (a2 + br[i]);
        else sk[o - C] += a2 + br[i];
      }
      __syncthreads();
    }
    if (STAGE < 3) {
      if (tid < 2 * C) {
        const float v = tid < C ? xs[tid] : sk[tid - C];
        __hip_atomic_store(out_g + tid, ((u64)(unsigned)(t + 1) << 32) | __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else {
      // ---- head: Linear(64 -> 128), Mish, Linear(128 -> 256), argmax ---------------------------------------------------------
      float h = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) h = fmaf(w0[k], sk[kq * 16 + k], h);
      h = quad_sum(h);
      if (kq == 0) hid[o] = mishf_(h + b0);
      __syncthreads();
      float q = 0.f;
      const float* hs = hid + (tid & 1) * 64;
#pragma unroll
      for (int k = 0; k < 64; ++k) q = fmaf(w2[k], hs[k], q);
      q += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(q), 0xB1, 0xf, 0xf, false));
      if ((tid & 1) == 0) lg[tid >> 1] = q + b2;
      __syncthreads();
      if (wave == 0) {
        float best = lg[lane * 4];
        int bi = lane * 4;
#pragma unroll
        for (int j = 1; j < 4; ++j)
          if (lg[lane * 4 + j] > best) { best = lg[lane * 4 + j]; bi = lane * 4 + j; }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
          const float ob = __shfl_xor(best, off);
          const int oi = __shfl_xor(bi, off);
          if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (lane == 0) {
          p.out[(size_t)clip * p.n_steps + t] = bi;
          __hip_atomic_store(p.cg + clip, ((u64)(unsigned)(t + 1) << 32) | (unsigned)bi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(kThreads) void wn_layer_pipe_kernel(const Params p) {
  const int clip = blockIdx.x % NCLIP, stage = blockIdx.x / NCLIP;
  switch (stage) {
    case 0: run_stage<0>(p, clip); break;
    case 1: run_stage<1>(p, clip); break;
    case 2: run_stage<2>(p, clip); break;
    default: run_stage<3>(p, clip); break;
  }
}

// ---- host reference of the same arithmetic (fp32, fma where the kernel uses it is not reproduced: classes are compared with a margin) ----
static void host_steps(const std::vector<float>& cw, const std::vector<float>& cb, const std::vector<float>& rw, const std::vector<float>& rb,
                       const std::vector<float>& emb, const std::vector<float>& f0w, const std::vector<float>& f0b, const std::vector<float>& f2w,
                       const std::vector<float>& f2b, int n, std::vector<int>& cls_out, std::vector<float>& gap_out) {
  std::vector<std::vector<float>> hist(L);   // x_l[t] rows
  int cls = 128;
  for (int t = 0; t < n; ++t) {
    float x[C], sk[C] = {0};
    for (int c = 0; c < C; ++c) x[c] = emb[cls * C + c];
    for (int l = 0; l < L; ++l) {
      const int d = 1 << l;
      hist[l].insert(hist[l].end(), x, x + C);
      float in[128];
      for (int c = 0; c < C; ++c) { in[c] = t >= d ? hist[l][(size_t)(t - d) * C + c] : 0.f; in[C + c] = x[c]; }
      float fgv[128];
      for (int o = 0; o < 128; ++o) {
        double a = 0;
        for (int k = 0; k < 128; ++k) a += (double)cw[((size_t)l * 128 + o) * 128 + k] * in[k];
        fgv[o] = (float)a + cb[l * 128 + o];
      }
      float z[C];
      for (int c = 0; c < C; ++c) z[c] = tanhf(fgv[c]) * (1.f / (1.f + expf(-fgv[C + c])));
      for (int o = 0; o < 128; ++o) {
        double a = 0;
        for (int k = 0; k < C; ++k) a += (double)rw[((size_t)l * 128 + o) * 64 + k] * z[k];
        const float v = (float)a + rb[l * 128 + o];
        if (o < C) x[o] += v; else sk[o - C] += v;
      }
    }
    float hid[HID];
    for (int o = 0; o < HID; ++o) {
      double a = 0;
      for (int k = 0; k < C; ++k) a += (double)f0w[(size_t)o * 64 + k] * sk[k];
      const float v = (float)a + f0b[o];
      hid[o] = v * tanhf(log1pf(expf(v)));
    }
    float best = -1e30f, second = -1e30f;
    int bi = 0;
    for (int o = 0; o < Q; ++o) {
      double a = 0;
      for (int k = 0; k < HID; ++k) a += (double)f2w[(size_t)o * HID + k] * hid[k];
      const float v = (float)a + f2b[o];
      if (v > best) { second = best; best = v; bi = o; } else if (v > second) second = v;
    }
    cls_out.push_back(bi);
    gap_out.push_back(best - second);
    cls = bi;
  }
}

int main() {
  const int n_steps = 4000, ring_len = 512;
  srand(7);
  auto fill = [](std::vector<float>& v, float s) { for (auto& x : v) x = (rand() / (float)RAND_MAX * 2.f - 1.f) * s; };
  std::vector<float> cw((size_t)L * 128 * 128), cb(L * 128), rw((size_t)L * 128 * 64), rb(L * 128), emb(Q * C), f0w(HID * 64), f0b(HID), f2w(Q * HID), f2b(Q);
  fill(cw, 2.f / sqrtf(128.f)); fill(cb, 0.1f); fill(rw, 2.f / 8.f); fill(rb, 0.1f); fill(emb, 1.f); fill(f0w, 2.f / 8.f); fill(f0b, 0.1f);
  fill(f2w, 2.f / sqrtf(128.f)); fill(f2b, 0.1f);
  Params p = {};
  auto up = [](const std::vector<float>& v) { float* d; hipMalloc(&d, v.size() * 4); hipMemcpy(d, v.data(), v.size() * 4, hipMemcpyHostToDevice); return d; };
  p.conv_w = up(cw); p.conv_b = up(cb); p.rs_w = up(rw); p.rs_b = up(rb); p.emb = up(emb); p.fc0_w = up(f0w); p.fc0_b = up(f0b); p.fc2_w = up(f2w); p.fc2_b = up(f2b);
  const size_t ring_bytes = (size_t)L * ring_len * NCLIP * C * 4;
  hipMalloc(&p.ring, ring_bytes); hipMemset(p.ring, 0, ring_bytes);
  hipMalloc(&p.xg, (size_t)NST * NCLIP * 128 * 8); hipMemset(p.xg, 0, (size_t)NST * NCLIP * 128 * 8);
  hipMalloc(&p.cg, NCLIP * 8); hipMemset(p.cg, 0, NCLIP * 8);
  hipMalloc(&p.out, (size_t)NCLIP * n_steps * 8);
  hipMalloc(&p.err, 4); hipMemset(p.err, 0, 4);
  p.n_steps = n_steps; p.ring_len = ring_len;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a);
  hipLaunchKernelGGL(wn_layer_pipe_kernel, dim3(NCLIP * NST), dim3(kThreads), 0, 0, p);
  hipEventRecord(b);
  hipError_t rc = hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  unsigned err = 0;
  hipMemcpy(&err, p.err, 4, hipMemcpyDeviceToHost);
  printf("launch: %s, hand-off timeouts: %u\n", hipGetErrorString(rc), err);
  printf("%d steps of %d clips: %.1f us total, %.2f us per step (%.0f samples/s over the %d clips)\n", n_steps, NCLIP, ms * 1e3, ms * 1e3 / n_steps,
         NCLIP * n_steps / (ms * 1e-3), NCLIP);
  std::vector<long long> out((size_t)NCLIP * n_steps);
  hipMemcpy(out.data(), p.out, out.size() * 8, hipMemcpyDeviceToHost);
  std::vector<int> want; std::vector<float> gap;
  host_steps(cw, cb, rw, rb, emb, f0w, f0b, f2w, f2b, 300, want, gap);
  int same = 0, checked = 0;
  for (int t = 0; t < 300; ++t) {
    if (out[t] != want[t]) {                         // (all clips run the same sequence: same prompt class, zero history)
      printf("step %d: device %lld, host %d (host top-1 / top-2 gap %.3g)%s\n", t, out[t], want[t], gap[t], gap[t] < 1e-4f ? " - a near tie, sequences part here" : "");
      break;
    }
    ++same; ++checked;
  }
  bool clips_agree = true;
  for (int c = 1; c < NCLIP; ++c)
    for (int t = 0; t < n_steps; ++t) clips_agree = clips_agree && out[(size_t)c * n_steps + t] == out[t];
  printf("first %d steps equal to the host loop; the %d clips agree with each other over all steps: %s\n", same, NCLIP, clips_agree ? "yes" : "NO");
  return 0;
}

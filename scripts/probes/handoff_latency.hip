// Probe: how long from a producer wave's store ISSUE to a polling consumer wave on another CU SEEING the value, by store flavour,
// poll flavour and placement (same XCD / other XCD)?  The payload is the producer's wall clock (s_memrealtime, 100 MHz, one counter
// for the chip), the consumer subtracts it from its own clock when the value shows up.
//   hipcc --offload-arch=gfx950 -O3 -o handoff_latency handoff_latency.hip && ./handoff_latency
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef unsigned long long u64;
constexpr int kRounds = 4000;

struct Params {
  unsigned* slot;        // [64] words on lines of their own: slot[0] same-XCD pair, slot[32] cross-XCD pair
  unsigned* count;       // [8] tickets per XCD
  unsigned* out;         // [2][kRounds] latencies in 10 ns ticks
  int store_kind;        // 0 plain, 1 sc1, 2 sc0 sc1, 3 nt, 4 atomic swap (no return), 5 atomic or agent, 6 plain + s_waitcnt vmcnt(0) right after
  int poll_kind;         // 0 one sc1 load at a time, 1 three in flight, 2 atomic-or-0 returning (poll at L2), 3 one sc0 sc1 load at a time
  int noise;             // other waves of the consumer CU spin on sc1 loads of an unrelated line
};

__device__ __forceinline__ unsigned now() { return (unsigned)__builtin_amdgcn_s_memrealtime(); }

__device__ __forceinline__ void do_store(unsigned* p, unsigned v, int kind) {
  switch (kind) {
    case 0: __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); break;
    case 1: __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break;
    case 2: __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break;
    case 3: __builtin_nontemporal_store(v, p); break;
    case 4: asm volatile("global_atomic_swap %0, %1, off" ::"v"(p), "v"(v) : "memory"); break;
    case 5: asm volatile("global_atomic_swap %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); break;
    case 6: __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

__device__ __forceinline__ unsigned do_load(const unsigned* p, int kind) {
  if (kind == 3) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (kind == 2) return atomicOr(const_cast<unsigned*>(p), 0u);
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(256) void probe(const Params p) {
  __shared__ int s_role;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    id &= 7;
    const unsigned t = atomicAdd(p.count + id, 1u);
    int role = -1;
    if (id == 0 && t == 0) role = 0;          // producer
    else if (id == 0 && t == 1) role = 1;     // consumer on the producer's XCD
    else if (id == 1 && t == 0) role = 2;     // consumer on another XCD
    s_role = role;
  }
  __syncthreads();
  const int role = s_role;
  if (role < 0) return;
  if (role == 0) {
    if (wave != 0) return;
    // wait until both consumers are up (they write 1 into their ready words)
    while (__hip_atomic_load(p.count + 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 2) __builtin_amdgcn_s_sleep(10);
    for (int r = 0; r < kRounds; ++r) {
      for (int k = 0; k < 6; ++k) __builtin_amdgcn_s_sleep(100);      // ~3 us between rounds
      unsigned t = now();
      t |= 1u;                                                       // never 0
      if (lane == 0) {
        do_store(p.slot + 0, t, p.store_kind);
        do_store(p.slot + 32, t, p.store_kind == 0 || p.store_kind == 6 || p.store_kind == 3 || p.store_kind == 4 ? 1 : p.store_kind);   // cross-XCD needs write-through
      }
    }
    return;
  }
  if (wave != 0) {
    if (!p.noise) return;
    // noise waves: spin on loads of an unrelated line until the consumer wave is done
    const unsigned* q = p.slot + 48;
    unsigned acc = 0;
    while (__hip_atomic_load(p.count + 9 + role, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) acc += __hip_atomic_load(q + (lane & 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (acc == 0x12345) p.out[0] = acc;
    return;
  }
  const unsigned* src = p.slot + (role == 1 ? 0 : 32);
  unsigned* out = p.out + (role - 1) * kRounds;
  if (lane == 0) atomicAdd(p.count + 8, 1u);
  unsigned last = 0;
  for (int r = 0; r < kRounds; ++r) {
    unsigned v;
    if (p.poll_kind == 1) {
      unsigned a = do_load(src, 0);
      __builtin_amdgcn_s_sleep(1);
      unsigned b = do_load(src, 0);
      __builtin_amdgcn_s_sleep(1);
      unsigned c = do_load(src, 0);
      for (;;) {
        v = a;
        if (v != last) break;
        a = b; b = c;
        __builtin_amdgcn_s_sleep(1);
        c = do_load(src, 0);
      }
    } else {
      do { v = do_load(src, p.poll_kind); } while (v == last);
    }
    const unsigned t = now();
    last = v;
    if (lane == 0) out[r] = t - v;
  }
  if (lane == 0) __hip_atomic_store(p.count + 9 + role, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

int main() {
  Params p;
  hipMalloc(&p.slot, 64 * 4 * 8);
  hipMalloc(&p.count, 64 * 4);
  hipMalloc(&p.out, 2 * kRounds * 4);
  const char* snames[] = {"plain", "sc1", "sc0 sc1", "nt", "atomic swap", "atomic swap sc1", "plain + vmcnt(0)"};
  const char* pnames[] = {"one sc1 load at a time", "three sc1 loads in flight", "returning atomic or 0", "one sc0 sc1 load at a time"};
  std::vector<unsigned> h(2 * kRounds);
  for (int noise = 0; noise < 2; ++noise)
    for (int pk = 0; pk < 4; ++pk)
      for (int sk = 0; sk < 7; ++sk) {
        hipMemset(p.slot, 0, 64 * 4 * 8);
        hipMemset(p.count, 0, 64 * 4);
        p.store_kind = sk; p.poll_kind = pk; p.noise = noise;
        hipLaunchKernelGGL(probe, dim3(256), dim3(256), 0, 0, p);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
        hipMemcpy(h.data(), p.out, h.size() * 4, hipMemcpyDeviceToHost);
        printf("noise %d | poll: %-28s | store: %-18s |", noise, pnames[pk], snames[sk]);
        for (int c = 0; c < 2; ++c) {
          std::vector<unsigned> v(h.begin() + c * kRounds + 100, h.begin() + (c + 1) * kRounds);
          std::sort(v.begin(), v.end());
          printf(" %s: min %.2f med %.2f p90 %.2f us |", c == 0 ? "same XCD" : "other XCD", v[0] * 0.01, v[v.size() / 2] * 0.01, v[v.size() * 9 / 10] * 0.01);
        }
        printf("\n");
      }
  return 0;
}

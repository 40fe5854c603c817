// Host-side helpers shared by the network plans: state_dict binding, workspace
// carving, packed-linear descriptors and the hipGraph step cache.
#pragma once
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "mmk_common.h"

namespace mmk {

// The execution switches of ONE plan: the `tuning` text of its config ("MMK_WN_CHAIN=0;MMK_WN_PIPE=1"), parsed when the plan is created.
// Nothing comes from the environment in the product library - a stray variable, or another thread's setenv between two plan creations,
// must not change which kernel a plan gets.  The diagnostic build (-DMMK_DIAG: stamps and timing experiments) falls back to the
// environment for names the text does not set; `diag_only` is the same fall-back for switches that exist in that build only.
inline const char* diag_only(const char* name) {
#ifdef MMK_DIAG
  return getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}
class Tuning {
 public:
  void parse(const char* text, size_t cap) {
    kv_.clear();
    std::string s(text, strnlen(text, cap));
    size_t at = 0;
    while (at < s.size()) {
      size_t end = s.find(';', at);
      if (end == std::string::npos) end = s.size();
      const std::string item = s.substr(at, end - at);
      const size_t eq = item.find('=');
      if (eq != std::string::npos && eq > 0) kv_[item.substr(0, eq)] = item.substr(eq + 1);
      at = end + 1;
    }
  }
  // the value's text, or nullptr when the switch is not set (the pointer lives as long as the plan)
  const char* get(const char* name) const {
    auto it = kv_.find(name);
    if (it != kv_.end()) return it->second.c_str();
    return diag_only(name);
  }

 private:
  std::map<std::string, std::string> kv_;
};

struct Bound {
  const float* ptr = nullptr;
  int64_t numel = 0;
};

class Binder {
 public:
  void bind(const std::string& key, const float* p, int64_t n) { map_[key] = Bound{p, n}; }
  // returns nullptr (and records the key) when missing or mis-sized
  const float* need(const std::string& key, int64_t numel) {
    auto it = map_.find(key);
    if (it == map_.end()) {
      if (missing_.empty()) missing_ = key + " (not bound)";
      return nullptr;
    }
    if (it->second.numel != numel) {
      if (missing_.empty())
        missing_ = key + " (expected " + std::to_string(numel) + " elements, got " + std::to_string(it->second.numel) + ")";
      return nullptr;
    }
    return it->second.ptr;
  }
  bool has(const std::string& key) const { return map_.count(key) != 0; }
  const std::string& missing() const { return missing_; }
  void clear_missing() { missing_.clear(); }

 private:
  std::map<std::string, Bound> map_;
  std::string missing_;
};

// Bump allocator over the caller's workspace; first pass (base == nullptr) sizes it.
class Carver {
 public:
  explicit Carver(void* base = nullptr) : base_((char*)base) {}
  template <typename T>
  T* take(int64_t count) {
    off_ = (size_t)round_up((int64_t)off_, 256);
    T* p = base_ ? reinterpret_cast<T*>(base_ + off_) : nullptr;
    off_ += (size_t)count * sizeof(T);
    return p;
  }
  size_t used() const { return (size_t)round_up((int64_t)off_, 256); }

 private:
  char* base_;
  size_t off_ = 0;
};

// A weight matrix in MFMA fragment order plus its K-segment geometry.
struct PackedLinear {
  float* Wp = nullptr;
  float* bias = nullptr;  // packed order, n_tiles*16 entries (zero padded) or nullptr
  int N = 0;              // packed rows in use
  int n_tiles = 0;
  int nseg = 0;
  int segK[kMaxSeg] = {0, 0, 0, 0};
  int seg_chunk0[kMaxSeg + 1] = {0, 0, 0, 0, 0};
  int k_chunks = 0;

  void set_geometry(int n_rows_packed, const std::vector<int>& ks) {
    N = n_rows_packed;
    n_tiles = (n_rows_packed + 15) / 16;
    nseg = (int)ks.size();
    k_chunks = 0;
    for (int s = 0; s < nseg; ++s) {
      segK[s] = ks[s];
      seg_chunk0[s] = k_chunks;
      k_chunks += (ks[s] + 15) / 16;
    }
    for (int s = nseg; s <= kMaxSeg; ++s) seg_chunk0[s] = k_chunks;
  }
  int64_t weight_floats() const { return (int64_t)n_tiles * 16 * k_chunks * 16; }
  int64_t bias_floats() const { return (int64_t)n_tiles * 16; }
  void carve(Carver& c, bool with_bias) {
    Wp = c.take<float>(weight_floats());
    bias = with_bias ? c.take<float>(bias_floats()) : nullptr;
  }
  int clear(hipStream_t st) const {
    MMK_HIP(hipMemsetAsync(Wp, 0, weight_floats() * sizeof(float), st));
    if (bias) MMK_HIP(hipMemsetAsync(bias, 0, bias_floats() * sizeof(float), st));
    return MMK_OK;
  }
  // fill the common LinearArgs fields; the caller sets seg[].x/ld/kind and the epilogue
  void fill(LinearArgs& a) const {
    a.nseg = nseg;
    for (int s = 0; s < nseg; ++s) {
      a.seg[s].K = segK[s];
      a.seg[s].kind = SEG_F32;
      a.seg[s].class_size = 1.f;
    }
    for (int s = 0; s <= kMaxSeg; ++s) a.seg_chunk0[s] = seg_chunk0[s];
    a.k_chunks = k_chunks;
    a.N = N;
    a.n_tiles = n_tiles;
    a.Wp = Wp;
    a.bias = bias;
  }
};

// Cache of one instantiated hipGraph holding `steps` consecutive steps.
struct GraphCache {
  hipGraphExec_t exec = nullptr;
  hipGraph_t graph = nullptr;
  std::vector<int64_t> key;
  int steps = 0;
  void reset() {
    if (exec) (void)hipGraphExecDestroy(exec);
    if (graph) (void)hipGraphDestroy(graph);
    exec = nullptr;
    graph = nullptr;
    key.clear();
    steps = 0;
  }
};

}  // namespace mmk

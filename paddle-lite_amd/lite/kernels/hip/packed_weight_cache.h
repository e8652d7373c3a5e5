// packed_weight_cache.h — one device copy of a layer's packed weights per PROCESS and device, shared by every predictor
// that runs the same model (the serving shape of lite/api/cxx_api.h:103-137: Predictor::Clone() shares the persistable
// variables of the scope; here the thing worth sharing is the pre-packed device copy the kernel object owns,
// conv_gemmlike.h:52-60's `weights_`).  Key = the bytes of the raw weights (two independent 64-bit hashes + length), the
// packing the descriptor selects and the device: predictors built from the same model share without being told to.
// The cache holds weak references: the copy dies with the last kernel object that uses it.  Packed weights are read-only
// after PrepareForRun, so sharing across predictors (threads, streams) needs no synchronisation beyond the one stream
// sync the first packer does before it publishes.
#pragma once
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <tuple>

#include "lite/core/tensor.h"

namespace paddle {
namespace lite {
namespace kernels {
namespace hip {

class PackedWeightCache {
 public:
  static PackedWeightCache& Global() {
    static PackedWeightCache c;
    return c;
  }
  // Returns the tensor that holds `packed_bytes` of device memory with the packed form of (raw, raw_bytes) under `layout`
  // (a string that names the packing: implementation + shape): an existing one, or a fresh one filled by `pack(dst)`,
  // which returns once the bytes are final (it synchronises its stream).  The caller keeps the shared_ptr for as long as
  // it launches kernels on the bytes; the cache itself holds a weak reference only.
  std::shared_ptr<Tensor> GetOrPack(int device, const std::string& layout, const void* raw_host, size_t raw_bytes, size_t packed_bytes,
                                    const std::function<void(void*)>& pack, bool* shared = nullptr) {
    const Key key{device, layout, raw_bytes, Hash(raw_host, raw_bytes, 0xcbf29ce484222325ull, 0x100000001b3ull),
                  Hash(raw_host, raw_bytes, 0x9e3779b97f4a7c15ull, 0xff51afd7ed558ccdull)};
    std::lock_guard<std::mutex> lk(mu_);
    auto it = map_.find(key);
    if (it != map_.end()) {
      if (auto sp = it->second.lock()) {
        ++hits_;
        if (shared) *shared = true;
        return sp;
      }
      map_.erase(it);
    }
    auto owner = std::make_shared<Tensor>();
    void* d = owner->mutable_data(TARGET(kHIP), packed_bytes);
    pack(d);
    map_[key] = owner;
    ++misses_;
    if (shared) *shared = false;
    return owner;
  }
  long hits() const { return hits_; }
  long misses() const { return misses_; }

 private:
  typedef std::tuple<int, std::string, size_t, uint64_t, uint64_t> Key;
  static uint64_t Hash(const void* p, size_t n, uint64_t seed, uint64_t mul) {
    const uint8_t* b = static_cast<const uint8_t*>(p);
    uint64_t h = seed;
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
      uint64_t v;
      __builtin_memcpy(&v, b + i, 8);
      h = (h ^ v) * mul;
      h ^= h >> 29;
    }
    for (; i < n; ++i) h = (h ^ b[i]) * mul;
    return h ^ (h >> 32);
  }
  std::mutex mu_;
  std::map<Key, std::weak_ptr<Tensor>> map_;
  long hits_{0}, misses_{0};
};

}  // namespace hip
}  // namespace kernels
}  // namespace lite
}  // namespace paddle

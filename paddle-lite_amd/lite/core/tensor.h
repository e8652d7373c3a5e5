// tensor.h — DDim / Buffer / TensorLite with the semantics of lite/core/tensor.h:104-234 and
// lite/core/memory.h:114-135: dims + shared Buffer + precision/target tags; mutable_data<T>(target) lazily
// (re)allocates through TargetMalloc only when the buffer must grow or the target changes.
#pragma once
#include <cstdint>
#include <memory>
#include <vector>

#include "lite/core/target_wrapper.h"

namespace paddle {
namespace lite {

class DDimLite {
 public:
  using value_type = int64_t;
  DDimLite() = default;
  DDimLite(const std::vector<value_type>& x) : data_(x) {}  // NOLINT (implicit like the reference)
  void ConstructFrom(const std::vector<value_type>& x) { data_ = x; }
  value_type operator[](int i) const { return data_[i]; }
  value_type& operator[](int i) { return data_[i]; }
  size_t size() const { return data_.size(); }
  bool empty() const { return data_.empty(); }
  value_type production() const {
    value_type r = 1;
    for (auto v : data_) r *= v;
    return data_.empty() ? 0 : r;
  }
  value_type count(int start, int end) const {
    value_type r = 1;
    for (int i = start; i < end && i < static_cast<int>(data_.size()); ++i) r *= data_[i];
    return r;
  }
  DDimLite Slice(int start, int end) const {
    return DDimLite(std::vector<value_type>(data_.begin() + start, data_.begin() + end));
  }
  const std::vector<value_type>& Vectorize() const { return data_; }
  friend bool operator==(const DDimLite& a, const DDimLite& b) { return a.data_ == b.data_; }
  friend bool operator!=(const DDimLite& a, const DDimLite& b) { return !(a == b); }
  friend std::ostream& operator<<(std::ostream& os, const DDimLite& d) {
    os << "{";
    for (size_t i = 0; i < d.size(); ++i) os << (i ? "," : "") << d[i];
    return os << "}";
  }

 private:
  std::vector<value_type> data_;
};
using DDim = DDimLite;

class Buffer {
 public:
  Buffer() = default;
  Buffer(const Buffer&) = delete;
  Buffer& operator=(const Buffer&) = delete;
  ~Buffer() { Free(); }
  void ResetLazy(TargetType target, size_t size) {
    if (target != target_ || space_ < size) {
      Free();
      data_ = TargetMalloc(target, size);
      CHECK(data_ != nullptr || size == 0) << "TargetMalloc(" << TargetToStr(target) << ", " << size << ") failed";
      target_ = target;
      space_ = size;
    }
  }
  void Free() {
    if (data_ && space_ > 0) TargetFree(target_, data_);
    data_ = nullptr;
    space_ = 0;
  }
  void* data() const { return data_; }
  size_t space() const { return space_; }
  TargetType target() const { return target_; }

 private:
  void* data_{nullptr};
  size_t space_{0};
  TargetType target_{TargetType::kHost};
};

class TensorLite {
 public:
  TensorLite() : buffer_(std::make_shared<Buffer>()) {}
  void Resize(const DDimLite& d) { dims_ = d; }
  void Resize(const std::vector<int64_t>& x) { dims_.ConstructFrom(x); }
  const DDimLite& dims() const { return dims_; }
  int64_t numel() const { return dims_.production(); }
  PrecisionType precision() const { return precision_; }
  void set_precision(PrecisionType p) { precision_ = p; }
  bool persistable() const { return persistable_; }
  void set_persistable(bool p) { persistable_ = p; }
  TargetType target() const { return target_; }
  size_t memory_size() const { return memory_size_; }
  bool IsInitialized() const { return buffer_->data() != nullptr; }

  template <typename T>
  const T* data() const {
    return reinterpret_cast<const T*>(static_cast<const char*>(buffer_->data()) + offset_);
  }
  template <typename T>
  T* mutable_data() {
    precision_ = lite_api::PrecisionTypeTrait<T>::Type();
    memory_size_ = static_cast<size_t>(dims_.production()) * sizeof(T);
    buffer_->ResetLazy(target_, memory_size_);
    return reinterpret_cast<T*>(static_cast<char*>(buffer_->data()) + offset_);
  }
  template <typename T>
  T* mutable_data(TargetType target) {
    target_ = target;
    return mutable_data<T>();
  }
  void* mutable_data(TargetType target, size_t memory_size) {
    target_ = target;
    memory_size_ = memory_size;
    buffer_->ResetLazy(target_, memory_size_);
    return static_cast<char*>(buffer_->data()) + offset_;
  }
  const void* raw_data() const { return static_cast<const char*>(buffer_->data()) + offset_; }
  void* raw_data() { return static_cast<char*>(buffer_->data()) + offset_; }
  void clear() {
    buffer_->Free();
    offset_ = 0;
  }
  void ShareDataWith(const TensorLite& o) {
    buffer_ = o.buffer_;
    dims_ = o.dims_;
    target_ = o.target_;
    precision_ = o.precision_;
    memory_size_ = o.memory_size_;
    offset_ = o.offset_;
  }
  void CopyDataFrom(const TensorLite& o) {
    dims_ = o.dims_;
    target_ = o.target_;
    precision_ = o.precision_;
    memory_size_ = o.memory_size_;
    buffer_->ResetLazy(target_, memory_size_);
    TargetCopy(target_, o.target_, buffer_->data(), o.raw_data(), memory_size_);
  }

 private:
  TargetType target_{TargetType::kHost};
  PrecisionType precision_{PrecisionType::kUnk};
  bool persistable_{false};
  DDimLite dims_;
  std::shared_ptr<Buffer> buffer_;
  size_t memory_size_{0};
  size_t offset_{0};
};
using Tensor = TensorLite;

}  // namespace lite
}  // namespace paddle

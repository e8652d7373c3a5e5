// any.h — the type-erased parameter holder used by KernelBase (lite/utils/any.h): param stored BY VALUE.
#pragma once
#include <memory>
#include <typeinfo>

#include "lite/utils/logging.h"

namespace paddle {
namespace lite {

class Any {
 public:
  template <typename T>
  void set(const T& v) {
    holder_ = std::make_shared<Holder<T>>(v);
  }
  template <typename T>
  T* get_mutable() const {
    auto* h = dynamic_cast<Holder<T>*>(holder_.get());
    CHECK(h != nullptr) << "Any: parameter holds a different type than " << typeid(T).name();
    return &h->v;
  }
  bool valid() const { return holder_ != nullptr; }

 private:
  struct HolderBase {
    virtual ~HolderBase() = default;
  };
  template <typename T>
  struct Holder : HolderBase {
    explicit Holder(const T& x) : v(x) {}
    T v;
  };
  std::shared_ptr<HolderBase> holder_;
};

}  // namespace lite
}  // namespace paddle

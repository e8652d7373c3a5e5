// type_system.h — the slice of lite/core/type_system.h the kernel registrations touch: a Type is
// (target, precision, layout); LiteType::GetTensorTy() interns them; ParamTypeRegistry records, per
// "op/alias" + place, the declared type of every input/output argument (BindInput / BindOutput / Finalize).
#pragma once
#include <map>
#include <memory>
#include <string>
#include <tuple>
#include <vector>

#include "lite/api/paddle_place.h"

namespace paddle {
namespace lite {

class Type {
 public:
  Type(TargetType t, PrecisionType p, DataLayoutType l, int device) : place_(t, p, l, static_cast<int16_t>(device)) {}
  TargetType target() const { return place_.target; }
  PrecisionType precision() const { return place_.precision; }
  DataLayoutType layout() const { return place_.layout; }
  const Place& place() const { return place_; }
  bool IsTensor() const { return true; }
  static const Type* GetTensorTy(TargetType target, PrecisionType precision = PRECISION(kFloat),
                                 DataLayoutType layout = DATALAYOUT(kNCHW), int device = 0) {
    static std::map<std::tuple<int, int, int, int>, std::unique_ptr<Type>> pool;
    auto key = std::make_tuple(static_cast<int>(target), static_cast<int>(precision), static_cast<int>(layout), device);
    auto it = pool.find(key);
    if (it == pool.end()) it = pool.emplace(key, std::unique_ptr<Type>(new Type(target, precision, layout, device))).first;
    return it->second.get();
  }

 private:
  Place place_;
};

// type_system.h:175-189: host-like targets exchange tensors freely; anything else needs an io_copy.
inline bool TargetCompatibleTo(TargetType a, TargetType b) {
  auto host_like = [](TargetType t) { return t == TARGET(kHost) || t == TARGET(kX86) || t == TARGET(kARM); };
  if (a == TARGET(kAny) || b == TARGET(kAny)) return true;
  if (host_like(a) && host_like(b)) return true;
  return a == b;
}

struct ParamTypeRecord {
  std::map<std::string, const Type*> inputs, outputs;
};

class ParamTypeRegistry {
 public:
  static ParamTypeRegistry& Global() {
    static ParamTypeRegistry* x = new ParamTypeRegistry;
    return *x;
  }
  class NewInstanceBuilder {
   public:
    NewInstanceBuilder(const std::string& key, const Place& place) : key_(key), place_(place) {}
    NewInstanceBuilder& BindInput(const std::string& arg, std::initializer_list<const Type*> tys) {
      rec_.inputs[arg] = *tys.begin();
      return *this;
    }
    NewInstanceBuilder& BindOutput(const std::string& arg, std::initializer_list<const Type*> tys) {
      rec_.outputs[arg] = *tys.begin();
      return *this;
    }
    bool Finalize() {
      ParamTypeRegistry::Global().records_[Key(key_, place_)] = rec_;
      return true;
    }

   private:
    std::string key_;
    Place place_;
    ParamTypeRecord rec_;
  };
  template <TargetType T, PrecisionType P, DataLayoutType L>
  static NewInstanceBuilder NewInstance(const std::string& kernel_type) {
    return NewInstanceBuilder(kernel_type, Place(T, P, L));
  }
  const ParamTypeRecord* Retrieve(const std::string& kernel_type, const Place& place) const {
    auto it = records_.find(Key(kernel_type, place));
    return it == records_.end() ? nullptr : &it->second;
  }

 private:
  static std::string Key(const std::string& k, const Place& p) { return k + ":" + p.DebugString(); }
  std::map<std::string, ParamTypeRecord> records_;
};

}  // namespace lite
}  // namespace paddle

using LiteType = paddle::lite::Type;  // registration sites spell it unqualified, as in the reference

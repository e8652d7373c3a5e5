// op_registry.h — KernelFactory / KernelRegistrar / REGISTER_LITE_KERNEL with the semantics of
// lite/core/op_registry.h:125-254: a static registrar inserts a creator keyed by
// (op_type, (target, precision, layout)); the macro also records the declared argument types and emits a
// touch_...() symbol that USE_LITE_KERNEL (lite/api/paddle_lite_factory_helper.h:34) references to force-link.
#pragma once
#include <functional>
#include <list>
#include <map>
#include <memory>
#include <string>
#include <tuple>

#include "lite/core/kernel.h"

namespace paddle {
namespace lite {

class KernelFactory {
 public:
  using creator_t = std::function<std::unique_ptr<KernelBase>()>;
  static KernelFactory& Global() {
    static KernelFactory* x = new KernelFactory;
    return *x;
  }
  void RegisterCreator(const std::string& op_type, TargetType t, PrecisionType p, DataLayoutType l, creator_t fun) {
    registry_[op_type][std::make_tuple(t, p, l)].push_back(std::move(fun));
  }
  std::list<std::unique_ptr<KernelBase>> Create(const std::string& op_type) {
    std::list<std::unique_ptr<KernelBase>> res;
    auto it = registry_.find(op_type);
    if (it == registry_.end()) return res;
    for (auto& kv : it->second)
      for (auto& f : kv.second) res.emplace_back(f());
    return res;
  }
  std::list<std::unique_ptr<KernelBase>> Create(const std::string& op_type, TargetType t, PrecisionType p,
                                                DataLayoutType l) {
    std::list<std::unique_ptr<KernelBase>> res;
    auto it = registry_.find(op_type);
    if (it == registry_.end()) return res;
    auto kt = it->second.find(std::make_tuple(t, p, l));
    if (kt == it->second.end()) return res;
    for (auto& f : kt->second) res.emplace_back(f());
    return res;
  }
  std::string DebugString() const {
    std::string s;
    for (auto& kv : registry_) s += " - " + kv.first + "\n";
    return s;
  }

 private:
  std::map<std::string, std::map<std::tuple<TargetType, PrecisionType, DataLayoutType>, std::list<creator_t>>> registry_;
};
using KernelRegistry = KernelFactory;

class KernelRegistrar {
 public:
  KernelRegistrar(const std::string& op_type, TargetType t, PrecisionType p, DataLayoutType l,
                  KernelFactory::creator_t fun) {
    KernelFactory::Global().RegisterCreator(op_type, t, p, l, std::move(fun));
  }
  void touch() {}
};

}  // namespace lite
}  // namespace paddle

#define REGISTER_LITE_KERNEL(op_type__, target__, precision__, layout__, KernelClass, alias__)                    \
  static paddle::lite::KernelRegistrar op_type__##target__##precision__##layout__##alias__##_kernel_registry(      \
      #op_type__, TARGET(target__), PRECISION(precision__), DATALAYOUT(layout__), []() {                           \
        std::unique_ptr<KernelClass> x(new KernelClass);                                                           \
        x->set_op_type(#op_type__);                                                                                \
        x->set_alias(#alias__);                                                                                    \
        return std::unique_ptr<paddle::lite::KernelBase>(std::move(x));                                            \
      });                                                                                                          \
  int touch_##op_type__##target__##precision__##layout__##alias__() {                                              \
    op_type__##target__##precision__##layout__##alias__##_kernel_registry.touch();                                 \
    return 0;                                                                                                      \
  }                                                                                                                \
  static auto op_type__##target__##precision__##layout__##alias__##param_register UNUSED =                         \
      paddle::lite::ParamTypeRegistry::NewInstance<TARGET(target__), PRECISION(precision__), DATALAYOUT(layout__)>( \
          #op_type__ "/" #alias__)

#define USE_LITE_KERNEL(op_type__, target__, precision__, layout__, alias__)            \
  extern int touch_##op_type__##target__##precision__##layout__##alias__();             \
  int op_type__##target__##precision__##layout__##alias__##__use_lite_kernel UNUSED =   \
      touch_##op_type__##target__##precision__##layout__##alias__();

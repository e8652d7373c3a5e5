// kernel.h — KernelBase / KernelLite with the contract of lite/core/kernel.h:42-248: PrepareForRun() once
// after SetContext + SetParam, ReInitWhenNeeded() every launch, then Run(); the parameter struct is stored
// BY VALUE in an Any (kernel.h:199) and read back with Param<T>().
#pragma once
#include <memory>
#include <string>

#include "lite/core/context.h"
#include "lite/core/profile/profiler.h"
#include "lite/core/type_system.h"
#include "lite/utils/any.h"

namespace paddle {
namespace lite {

class KernelBase {
 public:
  virtual void PrepareForRun() {}
  virtual void ReInitWhenNeeded() {}
  virtual void Run() = 0;

  // kernel.h:79-122
  void Launch() {
    if (is_first_epoch_) {
      PrepareForRun();
      is_first_epoch_ = false;
    }
    ReInitWhenNeeded();
    Run();
  }

  void SetContext(std::unique_ptr<KernelContext>&& ctx) { ctx_ = std::move(ctx); }
  template <typename T>
  void SetParam(T param) {
    param_.set<T>(param);
  }
  template <typename P>
  P& Param() const {
    return *param_.get_mutable<P>();
  }

  void set_op_type(const std::string& t) { op_type_ = t; }
  const std::string& op_type() const { return op_type_; }
  void set_alias(const std::string& a) { alias_ = a; }
  const std::string& alias() const { return alias_; }
  std::string key_with_alias() const { return op_type_ + "/" + alias_; }

  virtual Place place() const = 0;
  virtual TargetType target() const = 0;
  virtual PrecisionType precision() const = 0;
  virtual DataLayoutType layout() const = 0;
  virtual std::string name() const = 0;
  // lite/core/kernel.h:66-72 (LITE_WITH_PROFILE): the kernel names the device function it dispatches to, e.g.
  // GemmLikeConv sets "conv_im2col_gemm_int8" (conv_gemmlike.cc:393-394, 441-459)
  virtual void SetProfileRuntimeKernelInfo(paddle::lite::profile::OpCharacter* ch) { ch->kernel_func_name = std::string("NotImpl"); }
  std::string kernel_func_name() {  // convenience over the hook above
    profile::OpCharacter ch;
    SetProfileRuntimeKernelInfo(&ch);
    return ch.kernel_func_name;
  }

  const Type* GetInputDeclType(const std::string& arg) const {
    auto* r = ParamTypeRegistry::Global().Retrieve(key_with_alias(), place());
    CHECK(r) << "no param type record for " << key_with_alias();
    auto it = r->inputs.find(arg);
    CHECK(it != r->inputs.end()) << "no input " << arg << " declared for " << key_with_alias();
    return it->second;
  }
  const Type* GetOutputDeclType(const std::string& arg) const {
    auto* r = ParamTypeRegistry::Global().Retrieve(key_with_alias(), place());
    CHECK(r) << "no param type record for " << key_with_alias();
    auto it = r->outputs.find(arg);
    CHECK(it != r->outputs.end()) << "no output " << arg << " declared for " << key_with_alias();
    return it->second;
  }
  KernelContext* mutable_context() { return ctx_.get(); }
  virtual ~KernelBase() = default;

 protected:
  std::unique_ptr<KernelContext> ctx_{nullptr};
  mutable Any param_;
  std::string op_type_{};
  std::string alias_{};
  bool is_first_epoch_{true};
};

template <TargetType Target, PrecisionType Precision, DataLayoutType DataLayout = DataLayoutType::kNCHW>
class KernelLite : public KernelBase {
 public:
  void Run() override { CHECK(false) << "Not Implemented"; }
  TargetType target() const override { return Target; }
  PrecisionType precision() const override { return Precision; }
  DataLayoutType layout() const override { return DataLayout; }
  Place place() const override { return Place{Target, Precision, DataLayout}; }
  std::string name() const override {
    return op_type() + ":" + TargetToStr(Target) + "/" + PrecisionToStr(Precision) + "/" + DataLayoutToStr(DataLayout);
  }
};

}  // namespace lite
}  // namespace paddle

// context.h — KernelContext / Context<T> (lite/core/context.h:225,362,420-600).  Context<kHIP> has the shape of
// the CUDA one (lite/backends/cuda/context.h:35-140): device id, execution stream, and a grow-only device
// workspace that plays the role of ARMContext::workspace_data (device_info.h:144) for the im2col variant.
#pragma once
#include <memory>

#include "lite/core/tensor.h"

namespace paddle {
namespace lite {

template <TargetType Type>
class Context;

template <>
class Context<TARGET(kHost)> {
 public:
  void InitOnce() {}
  std::string name() const { return "HostContext"; }
};
using HostContext = Context<TARGET(kHost)>;

template <>
class Context<TARGET(kHIP)> {
 public:
  void InitOnce() {}
  // Binds the calling thread's default execution state (TargetWrapperHip::State()) unless one was given: the state
  // (stream + workspace) then belongs to this context, whichever thread later runs the kernel.
  void Init(int dev_id, int exec_stream_id = 0, std::shared_ptr<HipExecState> state = nullptr) {
    device_id_ = dev_id;
    (void)exec_stream_id;
    TargetWrapperHip::SetDevice(dev_id);
    state_ = state ? state : TargetWrapperHip::State();
  }
  int device_id() const { return device_id_; }
  const std::shared_ptr<HipExecState>& state() const { return state_; }
  plhip_ctx* ctx() const { return state_->ctx(); }
  void* exec_stream() const { return state_->stream(); }
  void Sync() const { state_->Sync(); }
  // Scratch shared by every kernel of this execution state (WorkSpace::Global_CUDA() analogue, kernel.h:91-100);
  // kernels run in stream order, so one grow-only arena is enough.
  void* workspace(size_t bytes) { return state_->Workspace(bytes); }
  void MemcpySync(void* dst, const void* src, size_t size, IoDirection dir) const { state_->MemcpySync(dst, src, size, dir); }
  std::string name() const { return "HIPContext"; }

 private:
  int device_id_{0};
  std::shared_ptr<HipExecState> state_;
};
using HIPContext = Context<TARGET(kHIP)>;

class KernelContext {
 public:
  template <typename ContextT>
  ContextT& As() {
    auto* p = dynamic_cast<Holder<ContextT>*>(holder_.get());
    if (!p) {
      CHECK(holder_ == nullptr) << "KernelContext already holds a different Context type";
      auto h = std::make_shared<Holder<ContextT>>();
      p = h.get();
      holder_ = h;
    }
    return p->ctx;
  }

 private:
  struct HolderBase {
    virtual ~HolderBase() = default;
  };
  template <typename T>
  struct Holder : HolderBase {
    T ctx;
  };
  std::shared_ptr<HolderBase> holder_;
};

// ContextScheduler::NewContext (context.h:426-470) for the targets present here.
inline std::unique_ptr<KernelContext> NewContext(TargetType target, int device_id = 0,
                                                 std::shared_ptr<HipExecState> state = nullptr) {
  std::unique_ptr<KernelContext> ctx(new KernelContext);
  if (target == TARGET(kHIP)) {
    ctx->As<HIPContext>().Init(device_id, 0, state);
  } else {
    ctx->As<HostContext>();
  }
  return ctx;
}

}  // namespace lite
}  // namespace paddle

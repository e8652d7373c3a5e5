// target_wrapper.cc — TargetWrapper<kHIP> and TargetMalloc/Free/Copy over the C ABI (include/plhip.h).
// Counterpart of lite/backends/cuda/target_wrapper.cc and the kCUDA arms of lite/core/memory.cc:29,73,121.
#include "lite/core/target_wrapper.h"

#include <map>

#include "plhip.h"

namespace paddle {
namespace lite {

HipExecState::HipExecState(int device, void* adopted_stream) : device_(device) {
  const int st = adopted_stream ? plhip_ctx_create_on_stream(device, adopted_stream, &ctx_) : plhip_ctx_create(device, &ctx_);
  if (st != 0) LOG(FATAL) << "plhip_ctx_create(" << device << ") -> " << st << ": " << plhip_last_error(nullptr);
}
HipExecState::~HipExecState() {
  if (ws_) plhip_free(ctx_, ws_);
  plhip_ctx_destroy(ctx_);
}
void* HipExecState::stream() const { return plhip_ctx_stream(ctx_); }
void HipExecState::Sync() const { HIP_CALL(ctx_, plhip_stream_sync(ctx_)); }
void* HipExecState::Workspace(size_t bytes) {
  if (ws_bytes_ < bytes) {
    if (ws_) {
      Sync();  // kernels still reading the old arena must finish before it is freed
      HIP_CALL(ctx_, plhip_free(ctx_, ws_));
      ws_ = nullptr;
    }
    HIP_CALL(ctx_, plhip_malloc(ctx_, bytes, &ws_));
    ws_bytes_ = bytes;
  }
  return ws_;
}
void HipExecState::MemcpyAsync(void* dst, const void* src, size_t size, IoDirection dir) const {
  switch (dir) {
    case IoDirection::HtoD: HIP_CALL(ctx_, plhip_memcpy_h2d(ctx_, dst, src, size)); break;
    case IoDirection::DtoH: HIP_CALL(ctx_, plhip_memcpy_d2h(ctx_, dst, src, size)); break;
    case IoDirection::DtoD: HIP_CALL(ctx_, plhip_memcpy_d2d(ctx_, dst, src, size)); break;
    default: std::memcpy(dst, src, size);
  }
}
void HipExecState::MemcpySync(void* dst, const void* src, size_t size, IoDirection dir) const {
  MemcpyAsync(dst, src, size, dir);
  if (dir == IoDirection::DtoD) Sync();  // h2d / d2h already complete on return (stream-ordered, then synchronised)
}

namespace {
struct ThreadHipState {
  int device{0};
  std::map<int, std::shared_ptr<HipExecState>> state;  // per device: the default state of this thread
};
thread_local ThreadHipState g_hip;
}  // namespace

size_t TargetWrapperHip::num_devices() { return static_cast<size_t>(plhip_device_count()); }
size_t TargetWrapperHip::GetCurDevice() { return static_cast<size_t>(g_hip.device); }
void TargetWrapperHip::SetDevice(int id) { g_hip.device = id; }

void TargetWrapperHip::AdoptStream(int device, stream_t stream) {
  // replaces this thread's default state for the device: contexts created from now on run on `stream`; predictors
  // built earlier keep the state they captured
  g_hip.state[device] = std::make_shared<HipExecState>(device, stream);
  g_hip.device = device;
}

std::shared_ptr<HipExecState> TargetWrapperHip::State() {
  auto it = g_hip.state.find(g_hip.device);
  if (it != g_hip.state.end()) return it->second;
  auto s = std::make_shared<HipExecState>(g_hip.device, nullptr);
  g_hip.state[g_hip.device] = s;
  return s;
}
plhip_ctx* TargetWrapperHip::Ctx() { return State()->ctx(); }

TargetWrapperHip::stream_t TargetWrapperHip::ExecStream() { return State()->stream(); }
void TargetWrapperHip::StreamSync() { State()->Sync(); }

void* TargetWrapperHip::Malloc(size_t size) {
  void* p = nullptr;
  HIP_CALL(Ctx(), plhip_malloc(Ctx(), size, &p));
  return p;
}
void TargetWrapperHip::Free(void* ptr) { HIP_CALL(Ctx(), plhip_free(Ctx(), ptr)); }

void TargetWrapperHip::MemcpySync(void* dst, const void* src, size_t size, IoDirection dir) { State()->MemcpySync(dst, src, size, dir); }
void TargetWrapperHip::MemcpyAsync(void* dst, const void* src, size_t size, IoDirection dir) { State()->MemcpyAsync(dst, src, size, dir); }
void TargetWrapperHip::MemsetAsync(void* dst, int value, size_t size) { HIP_CALL(Ctx(), plhip_memset(Ctx(), dst, value, size)); }
void* TargetWrapperHip::Workspace(size_t bytes) { return State()->Workspace(bytes); }

void* TargetMalloc(TargetType target, size_t size) {
  switch (target) {
    case TARGET(kHost):
    case TARGET(kX86):
    case TARGET(kARM): return TargetWrapper<TARGET(kHost)>::Malloc(size);
    case TARGET(kHIP): return TargetWrapperHip::Malloc(size);
    default: LOG(FATAL) << "TargetMalloc: unsupported target " << TargetToStr(target);
  }
  return nullptr;
}

void TargetFree(TargetType target, void* data) {
  switch (target) {
    case TARGET(kHost):
    case TARGET(kX86):
    case TARGET(kARM): TargetWrapper<TARGET(kHost)>::Free(data); break;
    case TARGET(kHIP): TargetWrapperHip::Free(data); break;
    default: LOG(FATAL) << "TargetFree: unsupported target " << TargetToStr(target);
  }
}

void TargetCopy(TargetType dst_target, TargetType src_target, void* dst, const void* src, size_t size) {
  const bool dh = dst_target == TARGET(kHIP), sh = src_target == TARGET(kHIP);
  if (dh && sh) TargetWrapperHip::MemcpySync(dst, src, size, IoDirection::DtoD);
  else if (dh) TargetWrapperHip::MemcpySync(dst, src, size, IoDirection::HtoD);
  else if (sh) TargetWrapperHip::MemcpySync(dst, src, size, IoDirection::DtoH);
  else std::memcpy(dst, src, size);
}

}  // namespace lite
}  // namespace paddle

// target_wrapper.cc — TargetWrapper<kHIP> and TargetMalloc/Free/Copy over the C ABI (include/plhip.h).
// Counterpart of lite/backends/cuda/target_wrapper.cc and the kCUDA arms of lite/core/memory.cc:29,73,121.
#include "lite/core/target_wrapper.h"

#include <map>

#include "plhip.h"

namespace paddle {
namespace lite {

namespace {
struct ThreadHipState {
  int device{0};
  std::map<int, plhip_ctx*> ctx;      // per device
  std::map<int, void*> ws;            // per device workspace
  std::map<int, size_t> ws_bytes;
  ~ThreadHipState() {
    for (auto& kv : ws)
      if (kv.second) plhip_free(ctx[kv.first], kv.second);
    for (auto& kv : ctx) plhip_ctx_destroy(kv.second);
  }
};
thread_local ThreadHipState g_hip;
}  // namespace

size_t TargetWrapperHip::num_devices() { return static_cast<size_t>(plhip_device_count()); }
size_t TargetWrapperHip::GetCurDevice() { return static_cast<size_t>(g_hip.device); }
void TargetWrapperHip::SetDevice(int id) { g_hip.device = id; }

void TargetWrapperHip::AdoptStream(int device, stream_t stream) {
  CHECK(g_hip.ctx.find(device) == g_hip.ctx.end()) << "AdoptStream must precede the first use of device " << device;
  plhip_ctx* c = nullptr;
  int st = plhip_ctx_create_on_stream(device, stream, &c);
  if (st != 0) LOG(FATAL) << "plhip_ctx_create_on_stream(" << device << ") -> " << st << ": " << plhip_last_error(nullptr);
  g_hip.ctx[device] = c;
  g_hip.device = device;
}

plhip_ctx* TargetWrapperHip::Ctx() {
  auto it = g_hip.ctx.find(g_hip.device);
  if (it != g_hip.ctx.end()) return it->second;
  plhip_ctx* c = nullptr;
  int st = plhip_ctx_create(g_hip.device, &c);
  if (st != 0) LOG(FATAL) << "plhip_ctx_create(" << g_hip.device << ") -> " << st << ": " << plhip_last_error(nullptr);
  g_hip.ctx[g_hip.device] = c;
  return c;
}

TargetWrapperHip::stream_t TargetWrapperHip::ExecStream() { return plhip_ctx_stream(Ctx()); }
void TargetWrapperHip::StreamSync() { HIP_CALL(Ctx(), plhip_stream_sync(Ctx())); }

void* TargetWrapperHip::Malloc(size_t size) {
  void* p = nullptr;
  HIP_CALL(Ctx(), plhip_malloc(Ctx(), size, &p));
  return p;
}
void TargetWrapperHip::Free(void* ptr) { HIP_CALL(Ctx(), plhip_free(Ctx(), ptr)); }

void TargetWrapperHip::MemcpySync(void* dst, const void* src, size_t size, IoDirection dir) {
  MemcpyAsync(dst, src, size, dir);
  if (dir == IoDirection::DtoD) StreamSync();  // h2d / d2h already complete on return
}
void TargetWrapperHip::MemcpyAsync(void* dst, const void* src, size_t size, IoDirection dir) {
  plhip_ctx* c = Ctx();
  switch (dir) {
    case IoDirection::HtoD: HIP_CALL(c, plhip_memcpy_h2d(c, dst, src, size)); break;
    case IoDirection::DtoH: HIP_CALL(c, plhip_memcpy_d2h(c, dst, src, size)); break;
    case IoDirection::DtoD: HIP_CALL(c, plhip_memcpy_d2d(c, dst, src, size)); break;
    default: std::memcpy(dst, src, size);
  }
}
void TargetWrapperHip::MemsetAsync(void* dst, int value, size_t size) { HIP_CALL(Ctx(), plhip_memset(Ctx(), dst, value, size)); }

void* TargetWrapperHip::Workspace(size_t bytes) {
  const int d = g_hip.device;
  if (g_hip.ws_bytes[d] < bytes) {
    if (g_hip.ws[d]) {
      StreamSync();  // kernels still reading the old arena must finish before it is freed
      Free(g_hip.ws[d]);
    }
    g_hip.ws[d] = Malloc(bytes);
    g_hip.ws_bytes[d] = bytes;
  }
  return g_hip.ws[d];
}

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

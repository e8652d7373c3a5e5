// target_wrapper.h — TargetWrapper<T> (lite/core/target_wrapper.h) for the two targets this build uses:
// kHost and the new kHIP.  TargetWrapper<kHIP> has the member set of TargetWrapper<kCUDA>
// (lite/backends/cuda/target_wrapper.h:27-85) and is implemented over the C ABI in include/plhip.h.
#pragma once
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <memory>

#include "lite/api/paddle_place.h"
#include "lite/utils/logging.h"

struct plhip_ctx;

namespace paddle {
namespace lite {

enum class IoDirection { HtoH = 0, HtoD, DtoH, DtoD };

template <TargetType Target>
class TargetWrapper;

template <>
class TargetWrapper<TARGET(kHost)> {
 public:
  static size_t num_devices() { return 1; }
  static void* Malloc(size_t size) {
    void* p = nullptr;
    if (posix_memalign(&p, 64, size ? size : 1) != 0) return nullptr;
    return p;
  }
  static void Free(void* ptr) { std::free(ptr); }
  static void MemcpySync(void* dst, const void* src, size_t size, IoDirection) { std::memcpy(dst, src, size); }
};

// plhip status -> LOG(FATAL), the CUDA_CALL convention (lite/backends/cuda/cuda_utils.h)
#define HIP_CALL(ctx__, expr__)                                                          \
  do {                                                                                   \
    int st__ = (expr__);                                                                 \
    if (st__ != 0) LOG(FATAL) << "HIP: " #expr__ " -> " << st__ << ": " << plhip_last_error(ctx__); \
  } while (0)

// Execution state of one predictor (or of a thread's default context): the plhip_ctx (device + HIP stream) and the
// grow-only scratch arena.  Owned through shared_ptr by every Context<kHIP> built from it, so a predictor created on
// thread A keeps running on ITS stream when Run() is called from thread B (the reference keeps the stream in the CUDA
// context object, lite/backends/cuda/context.h:46-73; a thread_local alone would silently switch streams).
class HipExecState {
 public:
  HipExecState(int device, void* adopted_stream);  // adopted_stream == nullptr: own stream
  ~HipExecState();
  HipExecState(const HipExecState&) = delete;
  HipExecState& operator=(const HipExecState&) = delete;
  int device() const { return device_; }
  plhip_ctx* ctx() const { return ctx_; }
  void* stream() const;
  void Sync() const;
  void* Workspace(size_t bytes);
  const void* workspace_ptr() const { return ws_; }   // identity of the arena (a recorded launch graph holds its address)
  size_t workspace_bytes() const { return ws_bytes_; }
  void MemcpySync(void* dst, const void* src, size_t size, IoDirection dir) const;
  void MemcpyAsync(void* dst, const void* src, size_t size, IoDirection dir) const;

 private:
  int device_;
  plhip_ctx* ctx_{nullptr};
  void* ws_{nullptr};
  size_t ws_bytes_{0};
};

template <>
class TargetWrapper<TARGET(kHIP)> {
 public:
  using stream_t = void*;  // hipStream_t
  using event_t = void*;   // hipEvent_t
  static size_t num_devices();
  static size_t maximum_stream() { return 0; }
  static size_t GetCurDevice();
  static void SetDevice(int id);
  // Adopt an externally owned hipStream_t as this thread's execution stream for `device`.
  static void AdoptStream(int device, stream_t stream);
  static plhip_ctx* Ctx();  // this thread's context on the current device (created on first use)
  static std::shared_ptr<HipExecState> State();  // ... and the state object that owns it
  static stream_t ExecStream();
  static void StreamSync();
  static void DeviceSync() { StreamSync(); }
  static void* Malloc(size_t size);
  static void Free(void* ptr);
  static void MemcpySync(void* dst, const void* src, size_t size, IoDirection dir);
  static void MemcpyAsync(void* dst, const void* src, size_t size, IoDirection dir);
  static void MemsetAsync(void* dst, int value, size_t size);
  static void* Workspace(size_t bytes);  // grow-only per-thread, per-device scratch
};
using TargetWrapperHip = TargetWrapper<TARGET(kHIP)>;

// lite/core/memory.{h,cc}: TargetMalloc / TargetFree / TargetCopy switch on the target.
void* TargetMalloc(TargetType target, size_t size);
void TargetFree(TargetType target, void* data);
void TargetCopy(TargetType dst_target, TargetType src_target, void* dst, const void* src, size_t size);

}  // namespace lite
}  // namespace paddle

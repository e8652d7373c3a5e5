// timer.h — lap timers with the member set of lite/core/profile/timer.h: TimeList<T> (:30-72), Timer (host clock,
// :74-121) and DeviceTimer<Target> (:123-158, the CUDA one brackets a stream with cudaEvents).  DeviceTimer<kHIP> does the
// same with hipEvents through the C ABI (plhip_event_*), on the execution stream of the kernel's Context<kHIP>.
#pragma once
#include <algorithm>
#include <chrono>  // NOLINT
#include <numeric>
#include <string>
#include <vector>

#include "lite/core/context.h"
#include "lite/core/profile/profiler.h"
#include "plhip.h"

namespace paddle {
namespace lite {
namespace profile {

template <typename T>
class TimeList {
 public:
  void Clear() { laps_.clear(); }
  void Add(T t) { laps_.push_back(t); }
  size_t Size(size_t offset = 0) const { return laps_.size() <= offset ? 0 : laps_.size() - offset; }
  T Last(size_t offset = 0) const { return Size(offset) ? laps_.back() : T(0); }
  T Max(size_t offset = 0) const { return Size(offset) ? *std::max_element(laps_.begin() + offset, laps_.end()) : T(0); }
  T Min(size_t offset = 0) const { return Size(offset) ? *std::min_element(laps_.begin() + offset, laps_.end()) : T(0); }
  T Sum(size_t offset = 0) const { return Size(offset) ? std::accumulate(laps_.begin() + offset, laps_.end(), T(0)) : T(0); }
  T Avg(size_t offset = 0) const { return Size(offset) ? Sum(offset) / static_cast<T>(Size(offset)) : T(0); }
  const std::vector<T>& Raw() const { return laps_; }

 private:
  std::vector<T> laps_;
};

class Timer {
 public:
  Timer() = default;
  virtual ~Timer() = default;
  void Reset() { laps_t_.Clear(); }
  void Start() { t_start_ = std::chrono::steady_clock::now(); }
  float Stop() {
    const auto us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_start_);
    const float ms = static_cast<float>(us.count()) * 1e-3f;
    laps_t_.Add(ms);
    return ms;
  }
  virtual void Start(KernelContext* ctx) { Start(); }
  virtual float Stop(KernelContext* ctx) { return Stop(); }
  float AvgLapTimeMs() const { return laps_t_.Avg(); }
  const TimeList<float>& LapTimes() const { return laps_t_; }

 protected:
  TimeList<float> laps_t_;

 private:
  std::chrono::time_point<std::chrono::steady_clock> t_start_;
};

template <TargetType Target>
class DeviceTimer final : public Timer {};

// Start / Stop record hipEvents on the kernel context's execution stream; Stop waits for the second event and adds the
// elapsed device time as a lap.  The events belong to the FIRST context the timer sees (one timer per instruction, as
// the reference's profiler keeps them).
template <>
class DeviceTimer<TargetType::kHIP> final : public Timer {
 public:
  DeviceTimer() = default;
  ~DeviceTimer() override {
    if (owner_) {
      plhip_event_destroy(owner_, e_start_);
      plhip_event_destroy(owner_, e_stop_);
    }
  }
  DeviceTimer(const DeviceTimer&) = delete;
  DeviceTimer& operator=(const DeviceTimer&) = delete;
  void Start(KernelContext* ctx) override {
    plhip_ctx* c = ctx->As<HIPContext>().ctx();
    if (!owner_) {
      HIP_CALL(c, plhip_event_create(c, &e_start_));
      HIP_CALL(c, plhip_event_create(c, &e_stop_));
      owner_ = c;
    }
    HIP_CALL(c, plhip_event_record(c, e_start_));
  }
  float Stop(KernelContext* ctx) override {
    plhip_ctx* c = ctx->As<HIPContext>().ctx();
    CHECK(owner_) << "DeviceTimer<kHIP>::Stop without Start";
    HIP_CALL(c, plhip_event_record(c, e_stop_));
    float elapse_ms = 1.f;
    HIP_CALL(c, plhip_event_elapsed_ms(c, e_start_, e_stop_, &elapse_ms));  // synchronises on e_stop_
    this->laps_t_.Add(elapse_ms);
    return elapse_ms;
  }

 private:
  plhip_ctx* owner_{nullptr};
  void* e_start_{nullptr};
  void* e_stop_{nullptr};
};

}  // namespace profile
}  // namespace lite
}  // namespace paddle

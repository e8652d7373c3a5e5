// profiler.h — the part of lite/core/profile/profiler.h the kernels touch: OpCharacter (:49-62), which a kernel fills in
// SetProfileRuntimeKernelInfo (lite/core/kernel.h:66-72).  The Profiler itself (timing tables, summaries) is outside
// the hot path; tools/ and bench.py time instructions with DeviceTimer<kHIP> / HIP events directly.
#pragma once
#include <string>

#include "lite/api/paddle_place.h"

namespace paddle {
namespace lite {
namespace profile {

// Characterisation record a kernel fills once for the profiler (lite/core/profile/profiler.h:49-62, the fields this
// build uses): kernels set kernel_func_name in SetProfileRuntimeKernelInfo (conv_gemmlike.cc:269-270, 384).
struct OpCharacter {
  TargetType target{TARGET(kUnk)};
  std::string op_type{"N/A"};
  std::string kernel_name{"N/A"};
  std::string kernel_attr{"N/A"};
  std::string kernel_func_name{"N/A"};
  std::string remark{"N/A"};
  std::string input_shape{"N/A"}, output_shape{"N/A"}, filter_shape{"N/A"};
  float macs{0}, macs_ps{0}, io_duration{0};
};

}  // namespace profile
}  // namespace lite
}  // namespace paddle

// paddle_place.h — source-compatible restatement of the enums of lite/api/paddle_place.h:46-136 with the one
// addition this backend needs: TargetType::kHIP = 16 (reference ends at kImaginationNNA = 15, NUM = 16).
// Enum values are part of the serialised-model ABI of the reference and are kept identical.
#pragma once
#include <string>

namespace paddle {
namespace lite_api {

enum class TargetType : int {
  kUnk = 0, kHost = 1, kX86 = 2, kCUDA = 3, kARM = 4, kOpenCL = 5, kAny = 6, kFPGA = 7, kNPU = 8, kXPU = 9,
  kBM = 10, kMLU = 11, kRKNPU = 12, kAPU = 13, kHuaweiAscendNPU = 14, kImaginationNNA = 15,
  kHIP = 16,  // NEW: AMD Instinct (gfx950) through libplhip.so
  NUM = 17,
};
enum class PrecisionType : int {
  kUnk = 0, kFloat = 1, kInt8 = 2, kInt32 = 3, kAny = 4, kFP16 = 5, kBool = 6, kInt64 = 7, kInt16 = 8,
  kUInt8 = 9, kFP64 = 10, NUM = 11,
};
enum class DataLayoutType : int {
  kUnk = 0, kNCHW = 1, kAny = 2, kNHWC = 3, kImageDefault = 4, kImageFolder = 5, kImageNW = 6, NUM = 7,
};
// lite/api/paddle_place.h:101-117
enum class ActivationType : int {
  kIndentity = 0, kRelu = 1, kRelu6 = 2, kPRelu = 3, kLeakyRelu = 4, kSigmoid = 5, kTanh = 6, kSwish = 7,
  kExp = 8, kAbs = 9, kHardSwish = 10, kReciprocal = 11, kThresholdedRelu = 12, kElu = 13, kHardSigmoid = 14,
  NUM = 15,
};

#define TARGET(item__) paddle::lite_api::TargetType::item__
#define PRECISION(item__) paddle::lite_api::PrecisionType::item__
#define DATALAYOUT(item__) paddle::lite_api::DataLayoutType::item__

inline const std::string& TargetToStr(TargetType t) {
  static const std::string names[] = {"unk", "host", "x86", "cuda", "arm", "opencl", "any", "fpga", "npu", "xpu",
                                      "bm", "mlu", "rknpu", "apu", "huawei_ascend_npu", "imagination_nna", "hip"};
  return names[static_cast<int>(t)];
}
inline const std::string& PrecisionToStr(PrecisionType p) {
  static const std::string names[] = {"unk", "float", "int8_t", "int32_t", "any", "float16", "bool", "int64_t",
                                      "int16_t", "uint8_t", "double"};
  return names[static_cast<int>(p)];
}
inline const std::string& DataLayoutToStr(DataLayoutType l) {
  static const std::string names[] = {"unk", "NCHW", "any", "NHWC", "ImageDefault", "ImageFolder", "ImageNW"};
  return names[static_cast<int>(l)];
}

template <typename T> struct PrecisionTypeTrait { static constexpr PrecisionType Type() { return PrecisionType::kUnk; } };
template <> struct PrecisionTypeTrait<float> { static constexpr PrecisionType Type() { return PrecisionType::kFloat; } };
template <> struct PrecisionTypeTrait<int8_t> { static constexpr PrecisionType Type() { return PrecisionType::kInt8; } };
template <> struct PrecisionTypeTrait<int32_t> { static constexpr PrecisionType Type() { return PrecisionType::kInt32; } };

// lite/api/paddle_place.h Place{target, precision, layout, device}
struct Place {
  TargetType target{TargetType::kUnk};
  PrecisionType precision{PrecisionType::kUnk};
  DataLayoutType layout{DataLayoutType::kUnk};
  int16_t device{0};
  Place() = default;
  Place(TargetType t, PrecisionType p = PrecisionType::kFloat, DataLayoutType l = DataLayoutType::kNCHW, int16_t d = 0)
      : target(t), precision(p), layout(l), device(d) {}
  bool operator==(const Place& o) const {
    return target == o.target && precision == o.precision && layout == o.layout && device == o.device;
  }
  std::string DebugString() const {
    return TargetToStr(target) + "/" + PrecisionToStr(precision) + "/" + DataLayoutToStr(layout);
  }
};

}  // namespace lite_api
namespace lite {
using lite_api::TargetType;
using lite_api::PrecisionType;
using lite_api::DataLayoutType;
using lite_api::Place;
using lite_api::TargetToStr;
using lite_api::PrecisionToStr;
using lite_api::DataLayoutToStr;
}  // namespace lite
}  // namespace paddle

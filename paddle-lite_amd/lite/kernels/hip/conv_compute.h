// conv_compute.h — ConvCompute<PRECISION(kInt8), OutType> for TARGET(kHIP): the drop-in for
// lite/kernels/arm/conv_compute.h:27-58 (+ conv_gemmlike / conv_depthwise / conv_direct / conv_winograd,
// which collapse into two device paths here: MFMA GEMM and depthwise).
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "lite/core/kernel.h"
#include "lite/kernels/hip/conv_fusion.h"
#include "lite/operators/op_params.h"
#include "plhip.h"

namespace paddle {
namespace lite {
namespace kernels {
namespace hip {

template <PrecisionType Ptype, PrecisionType OutType>
class ConvCompute : public KernelLite<TARGET(kHIP), Ptype>, public HipFusableKernel {
 public:
  using param_t = operators::ConvParam;
  void PrepareForRun() override;
  void ReInitWhenNeeded() override;
  void Run() override;
  void SetFusion(const HipConvFusion& f) override { fusion_ = f; }  // before the first Launch (conv_fusion.h)
  void SetProfileRuntimeKernelInfo(profile::OpCharacter* ch) override { ch->kernel_func_name = kernel_func_name_; }
  ~ConvCompute() override = default;

 private:
  void BuildDesc();
  HipConvFusion fusion_;  // default-constructed = the plain conv of the reference
  plhip_conv_desc desc_{};
  bool is_depthwise_{false};
  DDim last_shape_;
  Tensor weights_;   // packed (GEMM path) or raw OIHW (depthwise path), on device
  void PackWeights();
  std::string packed_impl_;   // the implementation the weights are packed for, and their size: checked on every reshape
  size_t packed_bytes_{0};
  std::shared_ptr<Tensor> packed_owner_;  // the process-wide shared copy weights_ aliases (packed_weight_cache.h), if any
  Tensor scale_;     // folded per-channel scale, device
  Tensor bias_;      // folded bias, device (only if param.bias)
  bool has_bias_{false};
  float act_alpha_{0.f};
  size_t workspace_bytes_{0};
  std::string kernel_func_name_{"NotImplForConv"};
  // fused 1x1 consumer of a depthwise conv (HipConvFusion::pw_*): its descriptor, packed weights, folded scale / bias; `mid_`
  // holds the depthwise result only when the shape is outside the fused kernel and the two kernels run instead
  void PreparePointwise();
  bool has_pw_{false}, pw_fused_{false}, pw_has_bias_{false};
  plhip_conv_desc pw_desc_{};
  Tensor pw_weights_, pw_scale_, pw_bias_, mid_, mid2_;
  Tensor xq_;                      // fused calib in front (HipConvFusion::calib_input_scale) on a shape without the one-launch form
  bool calib_in_fused_{false};
};

}  // namespace hip
}  // namespace kernels
}  // namespace lite
}  // namespace paddle

// conv_fusion.h — the kHIP-side state of this target's graph-level conv fusions.
//
// NOT part of operators::ConvParam: that struct stays a field-for-field subset of the reference's
// (lite/operators/op_params.h:446-502), so that lite/kernels/hip/conv_compute.cc compiles against the reference's own
// header unchanged.  The graph builder (lite/api/graph_builder.cc -> HipPredictor::AddConv) attaches a HipConvFusion to
// the picked kernel object through HipFusableKernel::SetFusion, after SetParam; a kernel that never receives one is the
// plain drop-in conv.  In a Paddle-Lite tree the same call is made by a kHIP mir pass after static_kernel_pick_pass
// (the shape to follow: lite/core/mir/fusion/conv_elementwise_fuse_pass.cc, which rewires the graph and then sets
// the op's `fuse_residual_connection`), see INTEGRATION.md 2.3.
//
// The residual operand itself uses the reference's own fields (ConvParam::residualData + fuse_residual_connection,
// lite/operators/conv_op.h:102); everything the reference has no word for lives here.
#pragma once
#include <vector>

#include "lite/core/tensor.h"
#include "lite/operators/op_params.h"

namespace paddle {
namespace lite {
namespace kernels {
namespace hip {

struct HipConvFusion {
  // conv2d[fp32_out] -> elementwise_add / fusion_elementwise_add_activation -> calib in ONE launch (every value rounded
  // exactly as the separate instructions round it): fuse_residual_relu = the add carried a relu; calib_output = the int8
  // tensor the following calib[fp32_to_int8] with scale calib_scale would produce; drop_fp32_output = `output` has no
  // other consumer and is never allocated.
  bool fuse_residual_relu{false};
  lite::Tensor* calib_output{nullptr};
  float calib_scale{1.f};
  bool drop_fp32_output{false};
  // opt-in (GraphBuilder::set_fuse_dwpw; measured slower than the two kernels): a depthwise_conv2d [int8_out]
  // takes its sole consumer, a plain conv2d 1x1, over.  `output` of the ConvParam is then the POINTWISE conv's output (int8
  // or fp32 by pw_int8_out); these are the pointwise op's filter / bias / scales / activation, its input scale is the
  // depthwise op's output_scale.
  lite::Tensor* pw_filter{nullptr};
  lite::Tensor* pw_bias{nullptr};
  std::vector<float> pw_weight_scale{};
  float pw_output_scale{1.f};
  bool pw_int8_out{true};
  operators::ActivationParam pw_activation_param;
  // ... and that conv's sole consumer, pool2d(avg, global_pooling) (graph_builder.cc, fusion E): `output` is then the POOL's
  // output [n, cout, 1, 1]; the fused kernel writes the plane average itself (PLHIP_OUT_F32_GAP), other shapes run the
  // instructions one by one inside the kernel object
  bool pw_global_avg_pool{false};
  // calib[fp32_to_int8] in FRONT of the conv taken over (graph_builder.cc, fusion F): `x` of the ConvParam is then the calib's
  // fp32 input and this its scale; the kernel quantises while it stages the rows (plhip_conv2d_calib_int8) where it has that
  // form (the 3x3 stride-2 stem), otherwise the kernel object runs the calib into a private tensor first.  0 = none.
  float calib_input_scale{0.f};
};

class HipFusableKernel {
 public:
  virtual void SetFusion(const HipConvFusion& f) = 0;
  virtual ~HipFusableKernel() = default;
};

}  // namespace hip
}  // namespace kernels
}  // namespace lite
}  // namespace paddle

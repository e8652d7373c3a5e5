// op_params.h — the parameter structs the hot-path kernels receive, field-for-field compatible (names, types,
// defaults) with the subset of lite/operators/op_params.h the ARM int8 kernels read:
//   WITH_INT8_CONFIG :49-54, IoCopyParam :70, CalibParam :82, FcParam :115-143, SoftmaxParam :319,
//   ActivationParam :395-419, ConvParam :446-502, PoolParam :539-, ElementwiseParam :643-651,
//   FusionElementwiseActivationParam :678-680.
// Tensors are NOT owned: params hold raw lite::Tensor* into the caller's scope (conv_op.h:72-74); bias may be null;
// paddings / dilations are shared_ptrs the op may mutate (UpdatePaddingAndDilation, conv_op.cc:55-81).
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "lite/core/tensor.h"

namespace paddle {
namespace lite {
namespace operators {

struct ParamBase {
  virtual ~ParamBase() = default;
};

#define WITH_INT8_CONFIG             \
  bool enable_int8{false};           \
  float input_scale{1.0f};           \
  std::vector<float> weight_scale{}; \
  float output_scale{1.0f};          \
  int bit_length{8};

struct IoCopyParam : ParamBase {
  const lite::Tensor* x{};
  lite::Tensor* y{};
  int process_type{0};
};

struct CalibParam : ParamBase {
  const lite::Tensor* input{};
  lite::Tensor* output{};
  float scale;
};

struct FcParam : ParamBase {
  lite::Tensor* input{nullptr};
  lite::Tensor* w{nullptr};
  lite::Tensor* bias{nullptr};
  lite::Tensor* output{nullptr};
  lite::DDim in_mat_dims;
  lite::DDim w_dims;
  int in_num_col_dims{1};
  std::string activation_type{""};
  bool padding_weights{false};
  WITH_INT8_CONFIG
};

struct SoftmaxParam : ParamBase {
  lite::Tensor* x{};
  lite::Tensor* output{};
  int axis{-1};
  bool use_cudnn{true};
};

struct ActivationParam : ParamBase {
  const lite::Tensor* X{};
  lite::Tensor* Out{};
  lite_api::ActivationType active_type{lite_api::ActivationType::kIndentity};
  bool has_active{false};
  float Leaky_relu_alpha{0};
  float Relu_clipped_coef{6};
  float threshold{6.0f};
};

struct ConvParam : ParamBase {
  lite::Tensor* x{};
  lite::Tensor* filter{};
  lite::Tensor* bias{nullptr};
  lite::Tensor* residualData{nullptr};
  lite::Tensor* output{};
  std::vector<int> strides{1, 1};
  std::shared_ptr<std::vector<int>> paddings;
  int groups{1};
  std::shared_ptr<std::vector<int>> dilations;
  bool fuse_relu_before_depthwise_conv{false};
  bool fuse_relu{false};
  bool fuse_residual_connection{false};
  std::string data_format{"Anylayout"};
  ActivationParam activation_param;
  bool var_length{false};
  std::vector<int> output_size;
  WITH_INT8_CONFIG
  // ---- kHIP graph-level fusion (not in the reference's struct; set only by lite/api/graph_builder.cc): the fp32_out
  // kernel can take over the instructions that follow it in the reference program.  residualData (above, the
  // reference's own field, lite/operators/conv_op.h:102) is the other operand of the fused elementwise_add;
  // fuse_residual_relu = the add was a fusion_elementwise_add_activation(relu); calib_output = the int8 tensor a following
  // calib[fp32_to_int8] with scale calib_scale would produce; drop_fp32_output = `output` has no other consumer.
  bool fuse_residual_relu{false};
  lite::Tensor* calib_output{nullptr};
  float calib_scale{1.f};
  bool drop_fp32_output{false};
  // ---- kHIP graph-level fusion, second kind (opt-in, GraphBuilder::set_fuse_dwpw): a depthwise_conv2d [int8_out] whose
  // only consumer is a plain conv2d 1x1 (stride 1, no padding, groups 1) takes that conv over.  `output` is then the
  // POINTWISE conv's output (int8 or fp32 by pw_int8_out); the depthwise result never exists as a tensor.  The pw_*
  // fields are the pointwise op's filter / bias / WITH_INT8_CONFIG scales / activation; its input scale is this op's
  // output_scale.  The kernel runs plhip_dwpw_fused_int8 when the shape is inside the fused path, else the two kernels.
  lite::Tensor* pw_filter{nullptr};
  lite::Tensor* pw_bias{nullptr};
  std::vector<float> pw_weight_scale{};
  float pw_output_scale{1.f};
  bool pw_int8_out{true};
  ActivationParam pw_activation_param;
};

struct PoolParam : ParamBase {
  lite::Tensor* x{};
  lite::Tensor* output{};
  std::string pooling_type{""};
  std::vector<int> ksize{};
  bool global_pooling{false};
  std::vector<int> strides{1, 1};
  std::shared_ptr<std::vector<int>> paddings;
  bool exclusive{true};
  bool adaptive{false};
  bool ceil_mode{false};
  bool use_quantizer{false};
  std::string data_format{"AnyLayout"};
  WITH_INT8_CONFIG
};

struct ElementwiseParam : ParamBase {
  const lite::Tensor* X{};
  const lite::Tensor* Y{};
  lite::Tensor* Out{};
  int axis{-1};  // for broadcasting.
  WITH_INT8_CONFIG
  float x_input_scale{1.0};
  float y_input_scale{1.0};
};

struct FusionElementwiseActivationParam : public ElementwiseParam {
  std::string act_type;
};

}  // namespace operators
}  // namespace lite
}  // namespace paddle

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

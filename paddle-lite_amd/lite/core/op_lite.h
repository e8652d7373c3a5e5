// op_lite.h — OpLite (lite/core/op_lite.h:54-) reduced to what Instruction::Run needs (program.cc:436-467):
// CheckShape() once, InferShape() before every launch, AttachKernel() to hand the op's parameter struct to the
// picked kernel.  The ops below restate the shape inference of lite/operators/{conv_op.cc:25-111, fc_op.cc,
// calib_op.cc, io_copy_op.cc, pool_op.cc, softmax_op.cc}; attributes are set on the param struct directly, as the
// reference's math tests do (conv_int8_compute_test.cc:90-117), because there is no model parser in this build.
#pragma once
#include <algorithm>
#include <memory>
#include <string>

#include "lite/core/kernel.h"
#include "lite/operators/op_params.h"

namespace paddle {
namespace lite {

class OpLite {
 public:
  explicit OpLite(const std::string& type) : op_type_(type) {}
  virtual ~OpLite() = default;
  virtual bool CheckShape() const { return true; }
  virtual bool InferShape() { return InferShapeImpl(); }
  virtual bool InferShapeImpl() const { return true; }
  virtual void AttachKernel(KernelBase* kernel) = 0;
  const std::string& Type() const { return op_type_; }

 protected:
  std::string op_type_;
};

namespace operators {

// conv_op.cc:25-52
inline int ConvOutputSize(int input_size, int filter_size, int dilation, int pad_left, int pad_right, int stride) {
  const int dkernel = dilation * (filter_size - 1) + 1;
  return (input_size + (pad_left + pad_right) - dkernel) / stride + 1;
}

// conv_op.cc:55-81
inline void UpdatePaddingAndDilation(std::vector<int>* paddings, std::vector<int>* dilations,
                                     const std::vector<int>& strides, const std::string& padding_algorithm,
                                     const DDim& data_dims, const DDim& ksize) {
  if (padding_algorithm == "SAME") {
    for (size_t i = 0; i < strides.size(); ++i) {
      const int out_size = static_cast<int>((data_dims[i + 2] + strides[i] - 1) / strides[i]);
      const int pad_sum = static_cast<int>(
          std::max<int64_t>((out_size - 1) * strides[i] + ksize[i + 2] - data_dims[i + 2], 0));
      const int pad_0 = pad_sum / 2;
      (*paddings)[i * 2] = pad_0;
      (*paddings)[i * 2 + 1] = pad_sum - pad_0;
      (*dilations)[i] = 1;
    }
  } else if (padding_algorithm == "VALID") {
    for (auto& p : *paddings) p = 0;
  }
}

class ConvOpLite : public OpLite {
 public:
  explicit ConvOpLite(const std::string& type = "conv2d") : OpLite(type) {}
  ConvParam& mutable_param() { return param_; }
  void set_padding_algorithm(const std::string& a) { padding_algorithm_ = a; }
  void set_output_channels(int64_t c) { out_channels_override_ = c; }  // kHIP fusions only (lite/kernels/hip/conv_fusion.h)
  void set_output_pooled() { out_pooled_ = true; }                     // ... with the global average pool behind it: [n, c, 1, 1]
  bool CheckShape() const override {
    CHECK(param_.x && param_.filter && param_.output) << "conv: x / filter / output must be set";
    const auto in = param_.x->dims(), f = param_.filter->dims();
    CHECK_EQ(in.size(), 4UL) << "conv input must be NCHW";
    CHECK_EQ(f.size(), 4UL);
    CHECK_EQ(in[1], f[1] * param_.groups) << "input channel must equal filter channel * groups";
    CHECK_EQ(f[0] % param_.groups, 0) << "filter number must be divisible by groups";
    // conv_op.h:149-161: 2-element paddings are expanded to {top, bottom, left, right}
    CHECK(param_.paddings && param_.dilations);
    if (param_.paddings->size() == 2UL) {
      const int ph = (*param_.paddings)[0], pw = (*param_.paddings)[1];
      *param_.paddings = {ph, ph, pw, pw};
    }
    CHECK_EQ(param_.paddings->size(), 4UL) << "paddings must have 2 or 4 entries";
    return true;
  }
  bool InferShapeImpl() const override {
    const auto in = param_.x->dims(), f = param_.filter->dims();
    UpdatePaddingAndDilation(param_.paddings.get(), param_.dilations.get(), param_.strides, padding_algorithm_, in, f);
    // kHIP dw -> pw fusion (opt-in): a depthwise conv that took its 1x1 consumer over writes THAT conv's output
    std::vector<int64_t> out{in[0], out_channels_override_ > 0 ? out_channels_override_ : f[0]};
    for (size_t i = 0; i < param_.strides.size(); ++i)
      out.push_back(ConvOutputSize(static_cast<int>(in[i + 2]), static_cast<int>(f[i + 2]), (*param_.dilations)[i],
                                   (*param_.paddings)[i * 2], (*param_.paddings)[i * 2 + 1], param_.strides[i]));
    if (out_pooled_)
      for (size_t i = 2; i < out.size(); ++i) out[i] = 1;
    param_.output->Resize(out);
    return true;
  }
  void AttachKernel(KernelBase* k) override { k->SetParam<ConvParam>(param_); }

 private:
  mutable ConvParam param_;
  std::string padding_algorithm_{""};
  int64_t out_channels_override_{0};
  bool out_pooled_{false};
};

class FcOpLite : public OpLite {
 public:
  FcOpLite() : OpLite("fc") {}
  FcParam& mutable_param() { return param_; }
  bool CheckShape() const override {
    CHECK(param_.input && param_.w && param_.output);
    CHECK_EQ(param_.w->dims().size(), 2UL);
    const auto in = param_.input->dims();
    CHECK_GT(static_cast<int>(in.size()), param_.in_num_col_dims);
    CHECK_EQ(in.count(param_.in_num_col_dims, static_cast<int>(in.size())), param_.w->dims()[0])
        << "fc: flattened input width must equal w.dims[0]";
    return true;
  }
  bool InferShapeImpl() const override {  // fc_op.cc: out = in.dims[:ncol] + {w.dims[1]}
    const auto in = param_.input->dims();
    std::vector<int64_t> out;
    for (int i = 0; i < param_.in_num_col_dims; ++i) out.push_back(in[i]);
    out.push_back(param_.w->dims()[1]);
    param_.output->Resize(out);
    return true;
  }
  void AttachKernel(KernelBase* k) override { k->SetParam<FcParam>(param_); }

 private:
  mutable FcParam param_;
};

class CalibOpLite : public OpLite {
 public:
  CalibOpLite() : OpLite("calib") {}
  CalibParam& mutable_param() { return param_; }
  bool InferShapeImpl() const override {
    param_.output->Resize(param_.input->dims());
    return true;
  }
  void AttachKernel(KernelBase* k) override { k->SetParam<CalibParam>(param_); }

 private:
  mutable CalibParam param_;
};

class IoCopyOp : public OpLite {
 public:
  IoCopyOp() : OpLite("io_copy") {}
  IoCopyParam& mutable_param() { return param_; }
  bool InferShapeImpl() const override {
    param_.y->Resize(param_.x->dims());
    return true;
  }
  void AttachKernel(KernelBase* k) override { k->SetParam<IoCopyParam>(param_); }

 private:
  mutable IoCopyParam param_;
};

// pool_op.cc:44-61
inline int PoolOutputSize(int input_size, int filter_size, int pad_left, int pad_right, int stride, bool ceil_mode) {
  if (!ceil_mode) return (input_size - filter_size + pad_left + pad_right) / stride + 1;
  return (input_size - filter_size + pad_left + pad_right + stride - 1) / stride + 1;
}

// pool_op.h:119-150
inline void UpdatePoolPadding(std::vector<int>* paddings, bool global_pooling, bool adaptive,
                              const std::string& padding_algorithm, const DDim& data_dims,
                              const std::vector<int>& strides, const std::vector<int>& ksize) {
  if (padding_algorithm == "SAME") {
    for (size_t i = 0; i < strides.size(); ++i) {
      const int out_size = static_cast<int>((data_dims[i + 2] + strides[i] - 1) / strides[i]);
      const int pad_sum =
          static_cast<int>(std::max<int64_t>((out_size - 1) * strides[i] + ksize[i] - data_dims[i + 2], 0));
      (*paddings)[i * 2] = pad_sum / 2;
      (*paddings)[i * 2 + 1] = pad_sum - pad_sum / 2;
    }
  } else if (padding_algorithm == "VALID") {
    for (auto& p : *paddings) p = 0;
  }
  if (global_pooling || adaptive)
    for (auto& p : *paddings) p = 0;
}

class PoolOpLite : public OpLite {
 public:
  PoolOpLite() : OpLite("pool2d") {}
  PoolParam& mutable_param() { return param_; }
  void set_padding_algorithm(const std::string& a) { padding_algorithm_ = a; }
  bool CheckShape() const override {
    CHECK(param_.x && param_.output && param_.paddings);
    CHECK_EQ(param_.x->dims().size(), 4UL) << "pool2d input must be NCHW";
    if (param_.paddings->size() == 2UL) {  // pool_op.h AttachKernel: 2-element paddings -> {top, bottom, left, right}
      const int ph = (*param_.paddings)[0], pw = (*param_.paddings)[1];
      *param_.paddings = {ph, ph, pw, pw};
    }
    CHECK_EQ(param_.paddings->size(), 4UL);
    if (!param_.global_pooling) {
      CHECK_EQ(param_.ksize.size(), 2UL);
      CHECK_EQ(param_.strides.size(), 2UL);
    }
    return true;
  }
  bool InferShapeImpl() const override {  // pool_op.cc:63-98
    const auto in = param_.x->dims();
    UpdatePoolPadding(param_.paddings.get(), param_.global_pooling, param_.adaptive, padding_algorithm_, in,
                      param_.strides, param_.ksize);
    if (param_.global_pooling) {
      param_.ksize.resize(2);
      param_.ksize[0] = static_cast<int>(in[2]);
      param_.ksize[1] = static_cast<int>(in[3]);
    }
    CHECK(!param_.adaptive) << "adaptive pooling is not on the int8 hot path";
    std::vector<int64_t> out{in[0], in[1]};
    for (size_t i = 0; i < 2; ++i)
      out.push_back(PoolOutputSize(static_cast<int>(in[i + 2]), param_.ksize[i], (*param_.paddings)[2 * i],
                                   (*param_.paddings)[2 * i + 1], param_.strides[i], param_.ceil_mode));
    param_.output->Resize(out);
    return true;
  }
  void AttachKernel(KernelBase* k) override { k->SetParam<PoolParam>(param_); }

 private:
  mutable PoolParam param_;
  std::string padding_algorithm_{""};
};

// elementwise_ops.cc: Out takes X's dims (same-shape operands on this path; Y broadcast along `axis` is not needed
// by the residual adds of ResNet50 / MobileNetV2)
class ElementwiseOp : public OpLite {
 public:
  explicit ElementwiseOp(const std::string& type = "elementwise_add") : OpLite(type) {}
  ElementwiseParam& mutable_param() { return param_; }
  bool CheckShape() const override {
    CHECK(param_.X && param_.Y && param_.Out);
    return true;
  }
  bool InferShapeImpl() const override {
    CHECK(param_.X->dims() == param_.Y->dims()) << op_type_ << ": operands must have the same shape on kHIP";
    param_.Out->Resize(param_.X->dims());
    return true;
  }
  void AttachKernel(KernelBase* k) override { k->SetParam<ElementwiseParam>(param_); }

 private:
  mutable ElementwiseParam param_;
};

// fusion_elementwise_activation_ops.cc
class FusionElementwiseActivationOp : public OpLite {
 public:
  explicit FusionElementwiseActivationOp(const std::string& type = "fusion_elementwise_add_activation") : OpLite(type) {}
  FusionElementwiseActivationParam& mutable_param() { return param_; }
  bool CheckShape() const override {
    CHECK(param_.X && param_.Y && param_.Out);
    return true;
  }
  bool InferShapeImpl() const override {
    CHECK(param_.X->dims() == param_.Y->dims()) << op_type_ << ": operands must have the same shape on kHIP";
    param_.Out->Resize(param_.X->dims());
    return true;
  }
  void AttachKernel(KernelBase* k) override { k->SetParam<FusionElementwiseActivationParam>(param_); }

 private:
  mutable FusionElementwiseActivationParam param_;
};

class SoftmaxOp : public OpLite {
 public:
  SoftmaxOp() : OpLite("softmax") {}
  SoftmaxParam& mutable_param() { return param_; }
  bool InferShapeImpl() const override {
    param_.output->Resize(param_.x->dims());
    return true;
  }
  void AttachKernel(KernelBase* k) override { k->SetParam<SoftmaxParam>(param_); }

 private:
  mutable SoftmaxParam param_;
};

}  // namespace operators
}  // namespace lite
}  // namespace paddle

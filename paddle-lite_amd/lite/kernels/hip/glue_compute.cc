// glue_compute.cc — the graph-edge kernels a kHIP whole-graph run needs besides conv / fc:
//   calib   fp32_to_int8 / int8_to_fp32   lite/kernels/arm/calib_compute.cc:25-57 (+ registrations :60-)
//   io_copy host_to_device / device_to_host  shape of lite/kernels/cuda/io_copy_compute.cc:45-112
//   pool2d  max / avg windows + global avg, fp32   lite/kernels/arm/pool_compute.cc:36-345
//   elementwise_add, fusion_elementwise_add_activation (relu), fp32   lite/kernels/arm/elementwise_compute.cc:182-207
//   softmax fp32                          lite/kernels/arm/softmax_compute.cc
#include "lite/core/op_registry.h"
#include "lite/operators/op_params.h"
#include "plhip.h"

namespace paddle {
namespace lite {
namespace kernels {
namespace hip {

class CalibComputeFp32ToInt8 : public KernelLite<TARGET(kHIP), PRECISION(kInt8)> {
 public:
  void Run() override {
    auto& param = this->Param<operators::CalibParam>();
    auto& ctx = this->ctx_->As<HIPContext>();
    CHECK(param.input->target() == TARGET(kHIP));
    HIP_CALL(ctx.ctx(), plhip_calib_f32_to_i8(ctx.ctx(), param.input->data<float>(),
                                              param.output->mutable_data<int8_t>(TARGET(kHIP)), param.scale,
                                              param.input->numel()));
  }
  void SetProfileRuntimeKernelInfo(profile::OpCharacter* ch) override { ch->kernel_func_name = "fp32_to_int8_hip"; }
};

class CalibComputeInt8ToFp32 : public KernelLite<TARGET(kHIP), PRECISION(kInt8)> {
 public:
  void Run() override {
    auto& param = this->Param<operators::CalibParam>();
    auto& ctx = this->ctx_->As<HIPContext>();
    CHECK(param.input->target() == TARGET(kHIP));
    HIP_CALL(ctx.ctx(), plhip_calib_i8_to_f32(ctx.ctx(), param.input->data<int8_t>(),
                                              param.output->mutable_data<float>(TARGET(kHIP)), param.scale,
                                              param.input->numel()));
  }
  void SetProfileRuntimeKernelInfo(profile::OpCharacter* ch) override { ch->kernel_func_name = "int8_to_fp32_hip"; }
};

// precision/layout kAny like the CUDA io_copy kernels: bytes are moved, whatever they mean.
class IoCopyHostToHipCompute : public KernelLite<TARGET(kHIP), PRECISION(kAny), DATALAYOUT(kAny)> {
 public:
  void Run() override {
    auto& param = this->Param<operators::IoCopyParam>();
    CHECK(param.x->target() == TARGET(kHost) || param.x->target() == TARGET(kX86) || param.x->target() == TARGET(kARM));
    const size_t bytes = param.x->memory_size();
    void* d = param.y->mutable_data(TARGET(kHIP), bytes);
    param.y->set_precision(param.x->precision());
    this->ctx_->As<HIPContext>().MemcpySync(d, param.x->raw_data(), bytes, IoDirection::HtoD);
  }
  void SetProfileRuntimeKernelInfo(profile::OpCharacter* ch) override { ch->kernel_func_name = "io_copy_host_to_hip"; }
};

class IoCopyHipToHostCompute : public KernelLite<TARGET(kHIP), PRECISION(kAny), DATALAYOUT(kAny)> {
 public:
  void Run() override {
    auto& param = this->Param<operators::IoCopyParam>();
    CHECK(param.x->target() == TARGET(kHIP));
    const size_t bytes = param.x->memory_size();
    void* d = param.y->mutable_data(TARGET(kHost), bytes);
    param.y->set_precision(param.x->precision());
    // on the context's own stream: ordered behind the kernels that produce x, complete on return
    this->ctx_->As<HIPContext>().MemcpySync(d, param.x->raw_data(), bytes, IoDirection::DtoH);
  }
  void SetProfileRuntimeKernelInfo(profile::OpCharacter* ch) override { ch->kernel_func_name = "io_copy_hip_to_host"; }
};

class PoolCompute : public KernelLite<TARGET(kHIP), PRECISION(kFloat)> {
 public:
  void Run() override {
    auto& param = this->Param<operators::PoolParam>();
    auto& ctx = this->ctx_->As<HIPContext>();
    CHECK(param.pooling_type == "avg" || param.pooling_type == "max") << "unsupported pooling type: " << param.pooling_type;
    CHECK(!param.adaptive) << "kHIP pool2d: adaptive pooling is not supported";
    const auto d = param.x->dims();
    const auto o = param.output->dims();
    const float* x = param.x->data<float>();
    float* y = param.output->mutable_data<float>(TARGET(kHIP));
    // pool_compute.cc:52-66: a window that covers the whole unpadded image is global pooling
    const auto& pads = *param.paddings;
    const bool whole = !param.global_pooling && pads[0] == 0 && pads[1] == 0 && pads[2] == 0 && pads[3] == 0 &&
                       param.ksize[0] == d[2] && param.ksize[1] == d[3];
    if ((param.global_pooling || whole) && param.pooling_type == "avg") {
      kernel_func_name_ = "pooling_global_avg_hip";
      HIP_CALL(ctx.ctx(), plhip_global_avg_pool_f32(ctx.ctx(), x, static_cast<int>(d[0] * d[1]),
                                                    static_cast<int>(d[2] * d[3]), y));
      return;
    }
    plhip_pool_desc pd;
    pd.planes = static_cast<int>(d[0] * d[1]);
    pd.h = static_cast<int>(d[2]);
    pd.w = static_cast<int>(d[3]);
    pd.oh = static_cast<int>(o[2]);
    pd.ow = static_cast<int>(o[3]);
    pd.kh = param.global_pooling ? pd.h : param.ksize[0];
    pd.kw = param.global_pooling ? pd.w : param.ksize[1];
    for (int i = 0; i < 4; ++i) pd.pad[i] = param.global_pooling ? 0 : pads[i];
    pd.stride[0] = param.strides[0];
    pd.stride[1] = param.strides[1];
    pd.is_max = param.pooling_type == "max";
    pd.exclusive = param.exclusive;
    kernel_func_name_ = std::string("pooling") + std::to_string(pd.kh) + "x" + std::to_string(pd.kw) + "s" +
                        std::to_string(pd.stride[0]) + "_" + param.pooling_type + "_hip";
    HIP_CALL(ctx.ctx(), plhip_pool2d_f32(ctx.ctx(), &pd, x, y));
  }
  void SetProfileRuntimeKernelInfo(profile::OpCharacter* ch) override { ch->kernel_func_name = kernel_func_name_; }

 private:
  std::string kernel_func_name_{"pooling_hip"};
};

// int8 max pool: exists only as the product of graph_builder.cc's fusion (C): conv -> pool2d(max) -> calib becomes
// conv + fused calib -> this kernel; bit-identical because max commutes with the monotonic quantiser.
class PoolMaxInt8Compute : public KernelLite<TARGET(kHIP), PRECISION(kInt8)> {
 public:
  void Run() override {
    auto& param = this->Param<operators::PoolParam>();
    auto& ctx = this->ctx_->As<HIPContext>();
    CHECK(param.pooling_type == "max" && !param.global_pooling && !param.adaptive) << "kHIP int8 pool2d: max windows only";
    const auto d = param.x->dims();
    const auto o = param.output->dims();
    const auto& pads = *param.paddings;
    plhip_pool_desc pd;
    pd.planes = static_cast<int>(d[0] * d[1]);
    pd.h = static_cast<int>(d[2]);
    pd.w = static_cast<int>(d[3]);
    pd.oh = static_cast<int>(o[2]);
    pd.ow = static_cast<int>(o[3]);
    pd.kh = param.ksize[0];
    pd.kw = param.ksize[1];
    for (int i = 0; i < 4; ++i) pd.pad[i] = pads[i];
    pd.stride[0] = param.strides[0];
    pd.stride[1] = param.strides[1];
    pd.is_max = 1;
    pd.exclusive = 1;
    HIP_CALL(ctx.ctx(), plhip_pool2d_max_i8(ctx.ctx(), &pd, param.x->data<int8_t>(),
                                            param.output->mutable_data<int8_t>(TARGET(kHIP))));
  }
  void SetProfileRuntimeKernelInfo(profile::OpCharacter* ch) override { ch->kernel_func_name = "pooling_max_int8_hip"; }
};

// elementwise_add, fp32, same-shape operands (elementwise_compute.cc:182-190)
class ElementwiseAddCompute : public KernelLite<TARGET(kHIP), PRECISION(kFloat)> {
 public:
  void Run() override {
    auto& param = this->Param<operators::ElementwiseParam>();
    auto& ctx = this->ctx_->As<HIPContext>();
    CHECK(param.X->target() == TARGET(kHIP) && param.Y->target() == TARGET(kHIP));
    CHECK(param.X->dims() == param.Y->dims()) << "kHIP elementwise_add: operands must have the same shape";
    HIP_CALL(ctx.ctx(), plhip_elementwise_add_f32(ctx.ctx(), param.X->data<float>(), param.Y->data<float>(),
                                                  param.Out->mutable_data<float>(TARGET(kHIP)), param.X->numel(), 0));
  }
  void SetProfileRuntimeKernelInfo(profile::OpCharacter* ch) override { ch->kernel_func_name = "elementwise_add_hip"; }
};

// fusion_elementwise_add_activation, act_type relu (elementwise_compute.cc:192-207)
class ElementwiseAddActivationCompute : public KernelLite<TARGET(kHIP), PRECISION(kFloat)> {
 public:
  void Run() override {
    auto& param = this->Param<operators::FusionElementwiseActivationParam>();
    auto& ctx = this->ctx_->As<HIPContext>();
    if (param.act_type != "relu") LOG(FATAL) << "fusion_elementwise_add_activation: unsupported activation " << param.act_type;
    CHECK(param.X->target() == TARGET(kHIP) && param.Y->target() == TARGET(kHIP));
    CHECK(param.X->dims() == param.Y->dims()) << "kHIP elementwise_add: operands must have the same shape";
    HIP_CALL(ctx.ctx(), plhip_elementwise_add_f32(ctx.ctx(), param.X->data<float>(), param.Y->data<float>(),
                                                  param.Out->mutable_data<float>(TARGET(kHIP)), param.X->numel(), 1));
  }
  void SetProfileRuntimeKernelInfo(profile::OpCharacter* ch) override { ch->kernel_func_name = "elementwise_add_relu_hip"; }
};

class SoftmaxCompute : public KernelLite<TARGET(kHIP), PRECISION(kFloat)> {
 public:
  void Run() override {
    auto& param = this->Param<operators::SoftmaxParam>();
    auto& ctx = this->ctx_->As<HIPContext>();
    const auto d = param.x->dims();
    const int nd = static_cast<int>(d.size());
    const int axis = param.axis < 0 ? param.axis + nd : param.axis;
    CHECK_EQ(d.count(axis + 1, nd), 1) << "kHIP softmax: reduction axis must be innermost";
    HIP_CALL(ctx.ctx(), plhip_softmax_f32(ctx.ctx(), param.x->data<float>(), static_cast<int>(d.count(0, axis)),
                                          static_cast<int>(d[axis]), param.output->mutable_data<float>(TARGET(kHIP))));
  }
  void SetProfileRuntimeKernelInfo(profile::OpCharacter* ch) override { ch->kernel_func_name = "softmax_inner1_hip"; }
};

}  // namespace hip
}  // namespace kernels
}  // namespace lite
}  // namespace paddle

REGISTER_LITE_KERNEL(calib, kHIP, kInt8, kNCHW, paddle::lite::kernels::hip::CalibComputeFp32ToInt8, fp32_to_int8)
    .BindInput("Input", {LiteType::GetTensorTy(TARGET(kHIP), PRECISION(kFloat))})
    .BindOutput("Out", {LiteType::GetTensorTy(TARGET(kHIP), PRECISION(kInt8))})
    .Finalize();
REGISTER_LITE_KERNEL(calib, kHIP, kInt8, kNCHW, paddle::lite::kernels::hip::CalibComputeInt8ToFp32, int8_to_fp32)
    .BindInput("Input", {LiteType::GetTensorTy(TARGET(kHIP), PRECISION(kInt8))})
    .BindOutput("Out", {LiteType::GetTensorTy(TARGET(kHIP), PRECISION(kFloat))})
    .Finalize();
REGISTER_LITE_KERNEL(io_copy, kHIP, kAny, kAny, paddle::lite::kernels::hip::IoCopyHostToHipCompute, host_to_device)
    .BindInput("Input", {LiteType::GetTensorTy(TARGET(kHost), PRECISION(kAny), DATALAYOUT(kAny))})
    .BindOutput("Out", {LiteType::GetTensorTy(TARGET(kHIP), PRECISION(kAny), DATALAYOUT(kAny))})
    .Finalize();
REGISTER_LITE_KERNEL(io_copy, kHIP, kAny, kAny, paddle::lite::kernels::hip::IoCopyHipToHostCompute, device_to_host)
    .BindInput("Input", {LiteType::GetTensorTy(TARGET(kHIP), PRECISION(kAny), DATALAYOUT(kAny))})
    .BindOutput("Out", {LiteType::GetTensorTy(TARGET(kHost), PRECISION(kAny), DATALAYOUT(kAny))})
    .Finalize();
REGISTER_LITE_KERNEL(pool2d, kHIP, kFloat, kNCHW, paddle::lite::kernels::hip::PoolCompute, def)
    .BindInput("X", {LiteType::GetTensorTy(TARGET(kHIP))})
    .BindOutput("Out", {LiteType::GetTensorTy(TARGET(kHIP))})
    .Finalize();
REGISTER_LITE_KERNEL(pool2d, kHIP, kInt8, kNCHW, paddle::lite::kernels::hip::PoolMaxInt8Compute, def)
    .BindInput("X", {LiteType::GetTensorTy(TARGET(kHIP), PRECISION(kInt8))})
    .BindOutput("Out", {LiteType::GetTensorTy(TARGET(kHIP), PRECISION(kInt8))})
    .Finalize();
REGISTER_LITE_KERNEL(elementwise_add, kHIP, kFloat, kNCHW, paddle::lite::kernels::hip::ElementwiseAddCompute, def)
    .BindInput("X", {LiteType::GetTensorTy(TARGET(kHIP))})
    .BindInput("Y", {LiteType::GetTensorTy(TARGET(kHIP))})
    .BindOutput("Out", {LiteType::GetTensorTy(TARGET(kHIP))})
    .Finalize();
REGISTER_LITE_KERNEL(fusion_elementwise_add_activation, kHIP, kFloat, kNCHW,
                     paddle::lite::kernels::hip::ElementwiseAddActivationCompute, def)
    .BindInput("X", {LiteType::GetTensorTy(TARGET(kHIP))})
    .BindInput("Y", {LiteType::GetTensorTy(TARGET(kHIP))})
    .BindOutput("Out", {LiteType::GetTensorTy(TARGET(kHIP))})
    .Finalize();
REGISTER_LITE_KERNEL(softmax, kHIP, kFloat, kNCHW, paddle::lite::kernels::hip::SoftmaxCompute, def)
    .BindInput("X", {LiteType::GetTensorTy(TARGET(kHIP))})
    .BindOutput("Out", {LiteType::GetTensorTy(TARGET(kHIP))})
    .Finalize();

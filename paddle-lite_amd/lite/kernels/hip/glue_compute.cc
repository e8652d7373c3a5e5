// glue_compute.cc — the graph-edge kernels a kHIP whole-graph run needs besides conv / fc:
//   calib   fp32_to_int8 / int8_to_fp32   lite/kernels/arm/calib_compute.cc:25-57 (+ registrations :60-)
//   io_copy host_to_device / device_to_host  shape of lite/kernels/cuda/io_copy_compute.cc:45-112
//   pool2d  global average, fp32          lite/kernels/arm/pool_compute.cc (global_pooling branch)
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
    TargetWrapperHip::MemcpySync(d, param.x->raw_data(), bytes, IoDirection::HtoD);
  }
};

class IoCopyHipToHostCompute : public KernelLite<TARGET(kHIP), PRECISION(kAny), DATALAYOUT(kAny)> {
 public:
  void Run() override {
    auto& param = this->Param<operators::IoCopyParam>();
    CHECK(param.x->target() == TARGET(kHIP));
    const size_t bytes = param.x->memory_size();
    void* d = param.y->mutable_data(TARGET(kHost), bytes);
    param.y->set_precision(param.x->precision());
    TargetWrapperHip::MemcpySync(d, param.x->raw_data(), bytes, IoDirection::DtoH);
  }
};

class PoolCompute : public KernelLite<TARGET(kHIP), PRECISION(kFloat)> {
 public:
  void Run() override {
    auto& param = this->Param<operators::PoolParam>();
    auto& ctx = this->ctx_->As<HIPContext>();
    CHECK(param.global_pooling && param.pooling_type == "avg") << "kHIP pool2d: global average only";
    const auto d = param.x->dims();
    HIP_CALL(ctx.ctx(), plhip_global_avg_pool_f32(ctx.ctx(), param.x->data<float>(), static_cast<int>(d[0] * d[1]),
                                                  static_cast<int>(d[2] * d[3]),
                                                  param.output->mutable_data<float>(TARGET(kHIP))));
  }
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
REGISTER_LITE_KERNEL(softmax, kHIP, kFloat, kNCHW, paddle::lite::kernels::hip::SoftmaxCompute, def)
    .BindInput("X", {LiteType::GetTensorTy(TARGET(kHIP))})
    .BindOutput("Out", {LiteType::GetTensorTy(TARGET(kHIP))})
    .Finalize();

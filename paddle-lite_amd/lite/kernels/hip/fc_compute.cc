// fc_compute.cc — TARGET(kHIP) / PRECISION(kInt8) fc kernels (aliases int8out / fp32out, the reference's
// spelling: lite/kernels/arm/fc_compute.cc:368-380).  Mirrors FcCompute<kInt8,*> :84-164, 229-344:
//   m = prod(x.dims[:in_num_col_dims]), k = w.dims[0], n = w.dims[1]; scale_j = w_scale[j or 0] * in (/ out);
//   int8-out bias_j = bias[j] / out (read from param.bias — the reference reads its own uninitialised buffer,
//   fc_compute.cc:154-163, which is a bug and is not reproduced); activation_type == "relu" only.
// One device kernel serves every m (no gemv/gemm split) with a per-column scale.
#include "lite/kernels/hip/packed_weight_cache.h"
#include <string>
#include <vector>

#include "lite/core/op_registry.h"
#include "lite/operators/op_params.h"
#include "plhip.h"

namespace paddle {
namespace lite {
namespace kernels {
namespace hip {

template <PrecisionType Ptype, PrecisionType OutType>
class FcCompute : public KernelLite<TARGET(kHIP), Ptype> {
 public:
  using param_t = operators::FcParam;

  void ReInitWhenNeeded() override {
    auto& param = this->template Param<param_t>();
    const auto x = param.input->dims();
    if (last_shape_ == x) return;
    last_shape_ = x;
    m_ = static_cast<int>(x.count(0, param.in_num_col_dims));
    k_ = static_cast<int>(x.count(param.in_num_col_dims, static_cast<int>(x.size())));
    CHECK_EQ(k_, static_cast<int>(param.w->dims()[0]));
    n_ = static_cast<int>(param.w->dims()[1]);
  }

  void PrepareForRun() override {
    auto& param = this->template Param<param_t>();
    auto& ctx = this->ctx_->template As<HIPContext>();
    ReInitWhenNeeded();
    constexpr bool kInt8Out = OutType == PRECISION(kInt8);
    CHECK(param.weight_scale.size() == 1 || param.weight_scale.size() == static_cast<size_t>(n_))
        << "fc weight_scale must have 1 or n entries";
    CHECK(param.activation_type.empty() || param.activation_type == "relu")
        << "fc: only relu can be fused (fc_compute.cc:229-)";
    relu_ = param.activation_type == "relu";
    // check_fc_use_gemm (fc_compute.cc:66-81): m > 1 and one weight scale -> gemm_s8, whose fp32 output gets its bias
    // from fill_bias_fc with a second rounding; otherwise gemv_int8 per row with one fused multiply-add
    single_scale_ = param.weight_scale.size() == 1;
    std::vector<float> s(n_);
    for (int j = 0; j < n_; ++j) {
      const float ws = param.weight_scale[param.weight_scale.size() == 1 ? 0 : j];
      s[j] = kInt8Out ? ws * param.input_scale / param.output_scale : ws * param.input_scale;
    }
    scale_.Resize({n_});
    TargetWrapperHip::MemcpySync(scale_.mutable_data<float>(TARGET(kHIP)), s.data(), n_ * sizeof(float), IoDirection::HtoD);
    has_bias_ = param.bias != nullptr;
    if (has_bias_) {
      CHECK_EQ(param.bias->numel(), n_);
      std::vector<float> b(n_);
      TargetCopy(TARGET(kHost), param.bias->target(), b.data(), param.bias->raw_data(), n_ * sizeof(float));
      if (kInt8Out)
        for (auto& v : b) v = v / param.output_scale;
      bias_.Resize({n_});
      TargetWrapperHip::MemcpySync(bias_.mutable_data<float>(TARGET(kHIP)), b.data(), n_ * sizeof(float), IoDirection::HtoD);
    }
    // weight pre-pack [k,n] -> [k/4][n][4] (replaces the transpose of fc_compute.cc:53-62)
    // (one packed device copy per process and device, shared by the predictors that run the same model: packed_weight_cache.h)
    const size_t wb = static_cast<size_t>(k_) * n_;
    auto pack_into = [&](void* wp) {
      Tensor staged;
      const void* w_dev = param.w->raw_data();
      if (param.w->target() != TARGET(kHIP)) {
        void* d = staged.mutable_data(TARGET(kHIP), wb);
        TargetWrapperHip::MemcpySync(d, param.w->raw_data(), wb, IoDirection::HtoD);
        w_dev = d;
      }
      HIP_CALL(ctx.ctx(), plhip_pack_fc_weights(ctx.ctx(), k_, n_, static_cast<const int8_t*>(w_dev), wp));
      ctx.Sync();
    };
    const size_t packed = plhip_fc_packed_weight_bytes(k_, n_);
    if (param.w->target() == TARGET(kHost)) {
      packed_owner_ = PackedWeightCache::Global().GetOrPack(static_cast<int>(TargetWrapperHip::GetCurDevice()),
                                                            "fc_" + std::to_string(k_) + "_" + std::to_string(n_), param.w->raw_data(), wb,
                                                            packed, pack_into);
      weights_.ShareDataWith(*packed_owner_);
    } else {
      pack_into(weights_.mutable_data(TARGET(kHIP), packed));
    }
  }

  void Run() override {
    auto& param = this->template Param<param_t>();
    auto& ctx = this->ctx_->template As<HIPContext>();
    CHECK(param.input->target() == TARGET(kHIP)) << "fc input must live on the HIP device";
    void* y;
    plhip_out_kind kind;
    if (OutType == PRECISION(kInt8)) {
      y = param.output->template mutable_data<int8_t>(TARGET(kHIP));
      kind = PLHIP_OUT_I8;
    } else {
      y = param.output->template mutable_data<float>(TARGET(kHIP));
      kind = PLHIP_OUT_F32;
    }
    const bool gemm_route = OutType == PRECISION(kFloat) && m_ > 1 && single_scale_;
    HIP_CALL(ctx.ctx(), plhip_fc_int8(ctx.ctx(), m_, k_, n_, param.input->template data<int8_t>(), weights_.raw_data(),
                                      scale_.data<float>(), has_bias_ ? bias_.data<float>() : nullptr,
                                      (relu_ ? 1 : 0) | (gemm_route ? 2 : 0), y, kind));
  }
  // the library runs the LDS-staged dot4 kernel when k % 16 == 0 (csrc/misc_ops.hip launch_fc), the MFMA form for the
  // k % 32 == 0 shapes that one cannot take (or under PLHIP_FC_MFMA=1), the plain dot4 kernel otherwise
  void SetProfileRuntimeKernelInfo(profile::OpCharacter* ch) override {
    ch->kernel_func_name = (k_ % 16 == 0 && k_ <= 4096) ? "fc_int8_dot4_lds_hip" : ((k_ % 32 == 0) ? "fc_int8_mfma32x32x32_hip" : "fc_int8_dot4_hip");
  }

 private:
  DDim last_shape_;
  int m_{0}, k_{0}, n_{0};
  bool relu_{false}, has_bias_{false}, single_scale_{false};
  Tensor weights_, scale_, bias_;
  std::shared_ptr<Tensor> packed_owner_;  // the shared copy weights_ aliases, if any
};

}  // namespace hip
}  // namespace kernels
}  // namespace lite
}  // namespace paddle

typedef paddle::lite::kernels::hip::FcCompute<PRECISION(kInt8), PRECISION(kInt8)> FcCompute_int8_int8;
typedef paddle::lite::kernels::hip::FcCompute<PRECISION(kInt8), PRECISION(kFloat)> FcCompute_int8_fp32;

REGISTER_LITE_KERNEL(fc, kHIP, kInt8, kNCHW, FcCompute_int8_int8, int8out)
    .BindInput("Input", {LiteType::GetTensorTy(TARGET(kHIP), PRECISION(kInt8))})
    .BindInput("Bias", {LiteType::GetTensorTy(TARGET(kHIP), PRECISION(kFloat))})
    .BindInput("W", {LiteType::GetTensorTy(TARGET(kHIP), PRECISION(kInt8))})
    .BindOutput("Out", {LiteType::GetTensorTy(TARGET(kHIP), PRECISION(kInt8))})
    .Finalize();

REGISTER_LITE_KERNEL(fc, kHIP, kInt8, kNCHW, FcCompute_int8_fp32, fp32out)
    .BindInput("Input", {LiteType::GetTensorTy(TARGET(kHIP), PRECISION(kInt8))})
    .BindInput("Bias", {LiteType::GetTensorTy(TARGET(kHIP), PRECISION(kFloat))})
    .BindInput("W", {LiteType::GetTensorTy(TARGET(kHIP), PRECISION(kInt8))})
    .BindOutput("Out", {LiteType::GetTensorTy(TARGET(kHIP), PRECISION(kFloat))})
    .Finalize();

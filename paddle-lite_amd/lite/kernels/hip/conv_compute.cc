// conv_compute.cc — TARGET(kHIP) / PRECISION(kInt8) conv2d + depthwise_conv2d kernels.
//
// Mirrors lite/kernels/arm/conv_compute.cc:87-185 (impl selection), conv_gemmlike.cc:85-264 and
// conv_depthwise.cc:138-296 (shape-change re-init, weight pre-pack, scale / bias / relu6 folding) and the six
// registrations of conv_compute.cc:216-252.  Differences by design: 3x3s1 goes through the
// direct accumulator (im2col GEMM), not the integer Winograd transform, and 3x3s2 needs no special DirectConv.
#include "lite/kernels/hip/conv_compute.h"
#include "lite/kernels/hip/packed_weight_cache.h"

#include "lite/core/op_registry.h"

namespace paddle {
namespace lite {
namespace kernels {
namespace hip {

namespace {
const void* DeviceCopyOf(const Tensor* src, Tensor* holder, size_t bytes) {
  // Persistable params may still live on the host (the reference moves them with an io_copy_once
  // instruction inserted by type_target_cast_pass); upload once if so.
  if (src->target() == TARGET(kHIP)) return src->raw_data();
  void* d = holder->mutable_data(TARGET(kHIP), bytes);
  TargetWrapperHip::MemcpySync(d, src->raw_data(), bytes, IoDirection::HtoD);
  return d;
}
}  // namespace

template <PrecisionType Ptype, PrecisionType OutType>
void ConvCompute<Ptype, OutType>::BuildDesc() {
  auto& param = this->template Param<param_t>();
  const auto x = param.x->dims(), w = param.filter->dims();
  CHECK_EQ(x.size(), 4UL);
  CHECK(param.paddings && param.paddings->size() == 4UL) << "paddings must be {top, bottom, left, right}";
  CHECK(param.dilations && param.dilations->size() == 2UL);
  desc_.n = static_cast<int>(x[0]);
  desc_.cin = static_cast<int>(x[1]);
  desc_.h = static_cast<int>(x[2]);
  desc_.w = static_cast<int>(x[3]);
  desc_.cout = static_cast<int>(w[0]);
  desc_.kh = static_cast<int>(w[2]);
  desc_.kw = static_cast<int>(w[3]);
  for (int i = 0; i < 4; ++i) desc_.pad[i] = (*param.paddings)[i];
  desc_.stride[0] = param.strides[0];
  desc_.stride[1] = param.strides[1];
  desc_.dil[0] = (*param.dilations)[0];
  desc_.dil[1] = (*param.dilations)[1];
  desc_.groups = param.groups;
}

// Weights: pre-pack once per IMPLEMENTATION (trans_gemm_weights<kInt8> -> prepackA_int8 analogue), or keep OIHW for depthwise.
// The implementation plhip's conv_geom picks — and with it the packed layout — depends on the input shape (the patch kernels
// need a row pitch of 8..64, the 7x7 stem OW % 4 == 0 ...): ReInitWhenNeeded calls this again when a resized feed changes it.
template <PrecisionType Ptype, PrecisionType OutType>
void ConvCompute<Ptype, OutType>::PackWeights() {
  auto& param = this->template Param<param_t>();
  auto& ctx = this->ctx_->template As<HIPContext>();
  // One packed device copy per process and device (packed_weight_cache.h): predictors that run the same model — the three
  // in flight of bench.py, a serving process with a predictor per thread (cxx_api.h:103-137) — share it; host-resident
  // weights are the key (persistable params of a model; weights already on the device are packed privately).
  const size_t w_bytes = static_cast<size_t>(param.filter->numel());
  const size_t packed = is_depthwise_ ? w_bytes : plhip_conv_packed_weight_bytes(&desc_);
  CHECK_GT(packed, 0UL) << "invalid conv configuration";
  auto pack_into = [&](void* d) {
    Tensor staged;
    const int8_t* w_dev = static_cast<const int8_t*>(DeviceCopyOf(param.filter, &staged, w_bytes));
    if (is_depthwise_) {
      ctx.MemcpySync(d, w_dev, w_bytes, IoDirection::DtoD);
    } else {
      HIP_CALL(ctx.ctx(), plhip_pack_conv_weights(ctx.ctx(), &desc_, w_dev, d));
    }
    ctx.Sync();  // the bytes are final (and `staged` may die) before anybody else sees them
  };
  if (param.filter->target() == TARGET(kHost)) {
    const auto wd = param.filter->dims();
    std::string layout = is_depthwise_ ? std::string("dw_oihw") : std::string(plhip_conv_impl_name(&desc_));
    for (size_t i = 0; i < wd.size(); ++i) layout += "_" + std::to_string(wd[i]);
    layout += "_g" + std::to_string(desc_.groups) + "_w" + std::to_string(desc_.w) + "_p" + std::to_string(desc_.pad[2]) + "_" + std::to_string(desc_.pad[3]);
    packed_owner_ = PackedWeightCache::Global().GetOrPack(static_cast<int>(TargetWrapperHip::GetCurDevice()), layout, param.filter->raw_data(), w_bytes, packed, pack_into);
    weights_.ShareDataWith(*packed_owner_);
  } else {
    pack_into(weights_.mutable_data(TARGET(kHIP), packed));
  }
  packed_impl_ = is_depthwise_ ? std::string("dw_oihw") : std::string(plhip_conv_impl_name(&desc_));
  packed_bytes_ = packed;
}

template <PrecisionType Ptype, PrecisionType OutType>
void ConvCompute<Ptype, OutType>::ReInitWhenNeeded() {
  auto& param = this->template Param<param_t>();
  if (last_shape_ == param.x->dims()) return;  // conv_gemmlike.cc:92 idiom
  BuildDesc();
  if (!is_depthwise_ && packed_bytes_ != 0) {
    // a resized feed may cross an implementation boundary (3x3 64 -> 64 from W = 56 to W = 112: patch kernel -> implicit GEMM;
    // the ResNet stem from 224 to 226): the packed bytes belong to ONE implementation, so pack again for the new one
    // (through the shared cache: a layout seen before is reused) instead of running it on the old layout
    if (packed_impl_ != plhip_conv_impl_name(&desc_) || packed_bytes_ != plhip_conv_packed_weight_bytes(&desc_)) {
      PackWeights();
      kernel_func_name_ = std::string(plhip_conv_impl_name(&desc_));
    }
  }
  workspace_bytes_ = is_depthwise_ ? 0 : plhip_conv_workspace_bytes(&desc_);
  calib_in_fused_ = false;
  if (fusion_.calib_input_scale > 0.f) {
    CHECK(!is_depthwise_ && !has_pw_ && !param.fuse_residual_connection && fusion_.calib_output == nullptr)
        << "kHIP: a conv that took the calib in front of it over has no other fusion";
    calib_in_fused_ = plhip_conv2d_calib_supported(&desc_) != 0;
    if (calib_in_fused_) kernel_func_name_ = "calib_fp32_to_int8+" + std::string(plhip_conv_impl_name(&desc_));
  }
  if (has_pw_) {  // the pointwise conv sees the depthwise conv's output plane (`output` may be the pooled one: from the descriptor)
    pw_desc_.n = desc_.n;
    pw_desc_.h = (desc_.h + desc_.pad[0] + desc_.pad[1] - (desc_.dil[0] * (desc_.kh - 1) + 1)) / desc_.stride[0] + 1;
    pw_desc_.w = (desc_.w + desc_.pad[2] + desc_.pad[3] - (desc_.dil[1] * (desc_.kw - 1) + 1)) / desc_.stride[1] + 1;
    const bool gap = fusion_.pw_global_avg_pool;
    pw_fused_ = plhip_dwpw_fused_supported(&desc_, pw_desc_.cout, gap ? PLHIP_OUT_F32_GAP : (fusion_.pw_int8_out ? PLHIP_OUT_I8 : PLHIP_OUT_F32)) != 0;
    kernel_func_name_ = pw_fused_ ? "conv_depthwise_3x3_pointwise_1x1_fused_int8_hip" : "conv_depthwise_int8_hip+conv1x1s1_gemm_int8_mfma32x32x32";
    if (gap) kernel_func_name_ += "+pooling_global_avg";
  }
  last_shape_ = param.x->dims();
}

// The 1x1 consumer taken over by a depthwise conv (graph_builder.cc, fusion D): folded exactly as ConvCompute folds a
// stand-alone conv2d (conv_gemmlike.cc:208-263) with the depthwise output scale as its input scale.
template <PrecisionType Ptype, PrecisionType OutType>
void ConvCompute<Ptype, OutType>::PreparePointwise() {
  auto& param = this->template Param<param_t>();
  auto& ctx = this->ctx_->template As<HIPContext>();
  CHECK(is_depthwise_ && OutType == PRECISION(kInt8)) << "kHIP: only a depthwise conv with int8 output takes a 1x1 consumer over";
  const auto wd = fusion_.pw_filter->dims();
  CHECK(wd.size() == 4UL && wd[2] == 1 && wd[3] == 1 && wd[1] == desc_.cout) << "fused consumer must be a 1x1 conv over the depthwise channels";
  const int m = static_cast<int>(wd[0]);
  pw_desc_ = plhip_conv_desc{};
  pw_desc_.n = desc_.n; pw_desc_.cin = desc_.cout; pw_desc_.h = 1; pw_desc_.w = 1; pw_desc_.cout = m;
  pw_desc_.kh = pw_desc_.kw = 1;
  pw_desc_.stride[0] = pw_desc_.stride[1] = 1;
  pw_desc_.dil[0] = pw_desc_.dil[1] = 1;
  pw_desc_.groups = 1;
  const auto& act = fusion_.pw_activation_param;
  float alpha = 0.f;
  pw_desc_.act = PLHIP_ACT_NONE;
  if (act.has_active) {
    switch (act.active_type) {
      case lite_api::ActivationType::kRelu: pw_desc_.act = PLHIP_ACT_RELU; break;
      case lite_api::ActivationType::kRelu6: pw_desc_.act = PLHIP_ACT_RELU6; alpha = act.Relu_clipped_coef; break;
      case lite_api::ActivationType::kLeakyRelu: pw_desc_.act = PLHIP_ACT_LEAKY_RELU; alpha = act.Leaky_relu_alpha; break;
      default: LOG(FATAL) << "this act_type: " << static_cast<int>(act.active_type) << " fuse not support";
    }
  }
  std::vector<float> ws = fusion_.pw_weight_scale;
  if (ws.size() != 1 && ws.size() != static_cast<size_t>(m)) LOG(FATAL) << "weights scale size must equal to filter size";
  if (ws.size() == 1) ws.resize(m, ws[0]);
  const float in_scale = param.output_scale, out_scale = fusion_.pw_output_scale;  // dw output scale = pw input scale
  for (auto& v : ws) v = fusion_.pw_int8_out ? v * in_scale / out_scale : v * in_scale;
  pw_scale_.Resize({m});
  TargetWrapperHip::MemcpySync(pw_scale_.mutable_data<float>(TARGET(kHIP)), ws.data(), m * sizeof(float), IoDirection::HtoD);
  pw_has_bias_ = fusion_.pw_bias != nullptr;
  if (pw_has_bias_) {
    CHECK_EQ(fusion_.pw_bias->numel(), m) << "bias size must equal to filter number";
    std::vector<float> b(m);
    TargetCopy(TARGET(kHost), fusion_.pw_bias->target(), b.data(), fusion_.pw_bias->raw_data(), m * sizeof(float));
    if (fusion_.pw_int8_out)
      for (auto& v : b) v = v / out_scale;
    pw_bias_.Resize({m});
    TargetWrapperHip::MemcpySync(pw_bias_.mutable_data<float>(TARGET(kHIP)), b.data(), m * sizeof(float), IoDirection::HtoD);
  }
  if (fusion_.pw_int8_out && pw_desc_.act == PLHIP_ACT_RELU6) alpha = alpha / out_scale;
  pw_desc_.act_alpha = alpha;
  Tensor staged;
  const size_t w_bytes = static_cast<size_t>(fusion_.pw_filter->numel());
  const int8_t* w_dev = static_cast<const int8_t*>(DeviceCopyOf(fusion_.pw_filter, &staged, w_bytes));
  const size_t packed = plhip_conv_packed_weight_bytes(&pw_desc_);
  CHECK_GT(packed, 0UL) << "invalid fused pointwise configuration";
  void* d = pw_weights_.mutable_data(TARGET(kHIP), packed);
  HIP_CALL(ctx.ctx(), plhip_pack_conv_weights(ctx.ctx(), &pw_desc_, w_dev, d));
  ctx.Sync();  // `staged` dies at scope exit
  has_pw_ = true;
}

template <PrecisionType Ptype, PrecisionType OutType>
void ConvCompute<Ptype, OutType>::PrepareForRun() {
  auto& param = this->template Param<param_t>();
  CHECK(this->ctx_) << "SetContext must precede PrepareForRun";
  auto& ctx = this->ctx_->template As<HIPContext>();
  const auto w_dims = param.filter->dims();
  const int oc = static_cast<int>(w_dims[0]);
  const int ic = static_cast<int>(w_dims[1]) * param.groups;
  BuildDesc();

  // ---- impl selection (conv_compute.cc:87-134): depthwise iff groups == ic == oc; everything else is GEMM-like
  is_depthwise_ = param.groups == ic && ic == oc && param.groups > 1;

  // ---- activation (conv_gemmlike.cc:325-345 reads activation_param; fuse_relu is the legacy flag)
  const auto& act = param.activation_param;
  desc_.act = PLHIP_ACT_NONE;
  act_alpha_ = 0.f;
  if (act.has_active) {
    switch (act.active_type) {
      case lite_api::ActivationType::kRelu: desc_.act = PLHIP_ACT_RELU; break;
      case lite_api::ActivationType::kRelu6:
        desc_.act = PLHIP_ACT_RELU6;
        act_alpha_ = act.Relu_clipped_coef;
        break;
      case lite_api::ActivationType::kLeakyRelu:
        desc_.act = PLHIP_ACT_LEAKY_RELU;
        act_alpha_ = act.Leaky_relu_alpha;
        break;
      default: LOG(FATAL) << "this act_type: " << static_cast<int>(act.active_type) << " fuse not support";
    }
  } else if (param.fuse_relu) {
    desc_.act = PLHIP_ACT_RELU;
  }

  // ---- scale / bias folding: conv_gemmlike.cc:208-263, conv_depthwise.cc:146-158,242-271 (fp32, as written)
  std::vector<float> w_scale = param.weight_scale;
  if (w_scale.size() != 1 && w_scale.size() != static_cast<size_t>(oc)) {
    LOG(FATAL) << "weights scale size must equal to filter size";
  }
  if (w_scale.size() == 1) w_scale.resize(oc, w_scale[0]);
  const float in_scale = param.input_scale, out_scale = param.output_scale;
  constexpr bool kInt8Out = OutType == PRECISION(kInt8);
  for (auto& ws : w_scale) {
    if (kInt8Out) ws = ws * in_scale / out_scale;
    else ws = ws * in_scale;
  }
  float* ds = scale_.mutable_data<float>(TARGET(kHIP));  // sized below
  (void)ds;
  scale_.Resize({oc});
  TargetWrapperHip::MemcpySync(scale_.mutable_data<float>(TARGET(kHIP)), w_scale.data(), oc * sizeof(float),
                               IoDirection::HtoD);
  has_bias_ = param.bias != nullptr;
  if (has_bias_) {
    CHECK_EQ(param.bias->numel(), oc) << "bias size must equal to filter number";
    std::vector<float> b(oc);
    TargetCopy(TARGET(kHost), param.bias->target(), b.data(), param.bias->raw_data(), oc * sizeof(float));
    if (kInt8Out)
      for (auto& v : b) v = v / out_scale;
    bias_.Resize({oc});
    TargetWrapperHip::MemcpySync(bias_.mutable_data<float>(TARGET(kHIP)), b.data(), oc * sizeof(float), IoDirection::HtoD);
  }
  if (kInt8Out && desc_.act == PLHIP_ACT_RELU6) act_alpha_ = act_alpha_ / out_scale;  // conv_gemmlike.cc:259-263
  desc_.act_alpha = act_alpha_;

  PackWeights();
  kernel_func_name_ = is_depthwise_ ? std::string("conv_depthwise_") + std::to_string(desc_.kh) + "x" + std::to_string(desc_.kw) +
                                          (kInt8Out ? "_int8_int8_hip" : "_int8_fp32_hip")
                                    : std::string(plhip_conv_impl_name(&desc_));
  if (fusion_.pw_filter) PreparePointwise();
  last_shape_ = DDim();
  ReInitWhenNeeded();
}

template <PrecisionType Ptype, PrecisionType OutType>
void ConvCompute<Ptype, OutType>::Run() {
  auto& param = this->template Param<param_t>();
  auto& ctx = this->ctx_->template As<HIPContext>();
  CHECK(param.x->target() == TARGET(kHIP)) << "conv input must live on the HIP device (io_copy missing?)";
  const float* sc = scale_.data<float>();
  const float* bi = has_bias_ ? bias_.data<float>() : nullptr;
  const int8_t* x;
  if (fusion_.calib_input_scale > 0.f) {  // `x` is the fp32 input of the calib this conv took over (fusion F)
    const float* xf = param.x->template data<float>();
    if (calib_in_fused_) {
      void* yo = OutType == PRECISION(kInt8) ? static_cast<void*>(param.output->template mutable_data<int8_t>(TARGET(kHIP)))
                                             : static_cast<void*>(param.output->template mutable_data<float>(TARGET(kHIP)));
      HIP_CALL(ctx.ctx(), plhip_conv2d_calib_int8(ctx.ctx(), &desc_, xf, fusion_.calib_input_scale, weights_.raw_data(), sc, bi, yo,
                                                  OutType == PRECISION(kInt8) ? PLHIP_OUT_I8 : PLHIP_OUT_F32));
      return;
    }
    xq_.Resize(param.x->dims());  // no one-launch form for this shape: the calib into a private tensor, then the conv
    int8_t* q = xq_.mutable_data<int8_t>(TARGET(kHIP));
    HIP_CALL(ctx.ctx(), plhip_calib_f32_to_i8(ctx.ctx(), xf, q, fusion_.calib_input_scale, static_cast<int64_t>(param.x->dims().production())));
    x = q;
  } else {
    x = param.x->template data<int8_t>();
  }
  void* y;
  plhip_out_kind kind;
  if (OutType == PRECISION(kInt8)) {
    y = has_pw_ ? nullptr : param.output->template mutable_data<int8_t>(TARGET(kHIP));  // has_pw_: allocated below, by the pointwise conv's precision
    kind = PLHIP_OUT_I8;
  } else {
    // a fused tail may leave the fp32 tensor without consumers (drop_fp32_output): then it is never allocated
    y = fusion_.drop_fp32_output ? nullptr : param.output->template mutable_data<float>(TARGET(kHIP));
    kind = PLHIP_OUT_F32;
  }
  const bool fused_tail = OutType == PRECISION(kFloat) && (param.fuse_residual_connection || fusion_.calib_output != nullptr);
  if (fused_tail) {
    CHECK(!is_depthwise_) << "kHIP: the fused conv tail exists on the GEMM-like convs only";
    const float* res = nullptr;
    if (param.fuse_residual_connection) {
      CHECK(param.residualData && param.residualData->target() == TARGET(kHIP)) << "fused residual operand must live on the device";
      CHECK(param.residualData->dims() == param.output->dims()) << "fused residual operand must have the output's shape";
      res = param.residualData->template data<float>();
    }
    int8_t* q = nullptr;
    if (fusion_.calib_output) {
      fusion_.calib_output->Resize(param.output->dims());
      q = fusion_.calib_output->template mutable_data<int8_t>(TARGET(kHIP));
    }
    void* ws = workspace_bytes_ ? ctx.workspace(workspace_bytes_) : nullptr;
    HIP_CALL(ctx.ctx(), plhip_conv2d_int8_fused(ctx.ctx(), &desc_, x, weights_.raw_data(), sc, bi,
                                                fusion_.drop_fp32_output ? nullptr : static_cast<float*>(y), res,
                                                fusion_.fuse_residual_relu ? 1 : 0, q, fusion_.calib_scale, ws, workspace_bytes_));
  } else if (is_depthwise_ && has_pw_) {
    // `output` is the pointwise conv's tensor (HipConvFusion::pw_*); y above was allocated as int8: redo it for fp32
    void* yo = fusion_.pw_int8_out ? static_cast<void*>(param.output->template mutable_data<int8_t>(TARGET(kHIP)))
                                 : static_cast<void*>(param.output->template mutable_data<float>(TARGET(kHIP)));
    const bool gap = fusion_.pw_global_avg_pool;  // `output` is the pool's [n, cout, 1, 1]
    const plhip_out_kind ko = gap ? PLHIP_OUT_F32_GAP : (fusion_.pw_int8_out ? PLHIP_OUT_I8 : PLHIP_OUT_F32);
    const float* psc = pw_scale_.data<float>();
    const float* pbi = pw_has_bias_ ? pw_bias_.data<float>() : nullptr;
    if (pw_fused_) {
      HIP_CALL(ctx.ctx(), plhip_dwpw_fused_int8(ctx.ctx(), &desc_, x, weights_.data<int8_t>(), sc, bi, pw_desc_.cout,
                                                pw_weights_.raw_data(), psc, pbi, pw_desc_.act, pw_desc_.act_alpha, yo, ko));
    } else {  // shape outside the fused kernel: the kernels one by one, the intermediate results in private tensors
      mid_.Resize({desc_.n, desc_.cout, pw_desc_.h, pw_desc_.w});
      int8_t* mid = mid_.mutable_data<int8_t>(TARGET(kHIP));
      HIP_CALL(ctx.ctx(), plhip_depthwise_conv_int8(ctx.ctx(), &desc_, x, weights_.data<int8_t>(), sc, bi, mid, PLHIP_OUT_I8));
      if (gap) {
        mid2_.Resize({desc_.n, pw_desc_.cout, pw_desc_.h, pw_desc_.w});
        float* m2 = mid2_.mutable_data<float>(TARGET(kHIP));
        HIP_CALL(ctx.ctx(), plhip_conv2d_int8(ctx.ctx(), &pw_desc_, mid, pw_weights_.raw_data(), psc, pbi, m2, PLHIP_OUT_F32, nullptr, 0));
        HIP_CALL(ctx.ctx(), plhip_global_avg_pool_f32(ctx.ctx(), m2, desc_.n * pw_desc_.cout, pw_desc_.h * pw_desc_.w, static_cast<float*>(yo)));
      } else {
        HIP_CALL(ctx.ctx(), plhip_conv2d_int8(ctx.ctx(), &pw_desc_, mid, pw_weights_.raw_data(), psc, pbi, yo, ko, nullptr, 0));
      }
    }
  } else if (is_depthwise_) {
    HIP_CALL(ctx.ctx(), plhip_depthwise_conv_int8(ctx.ctx(), &desc_, x, weights_.data<int8_t>(), sc, bi, y, kind));
  } else {
    void* ws = workspace_bytes_ ? ctx.workspace(workspace_bytes_) : nullptr;
    HIP_CALL(ctx.ctx(), plhip_conv2d_int8(ctx.ctx(), &desc_, x, weights_.raw_data(), sc, bi, y, kind, ws, workspace_bytes_));
  }
}

template class ConvCompute<PRECISION(kInt8), PRECISION(kInt8)>;
template class ConvCompute<PRECISION(kInt8), PRECISION(kFloat)>;

}  // namespace hip
}  // namespace kernels
}  // namespace lite
}  // namespace paddle

typedef paddle::lite::kernels::hip::ConvCompute<PRECISION(kInt8), PRECISION(kFloat)> ConvInt8_Fp32;
typedef paddle::lite::kernels::hip::ConvCompute<PRECISION(kInt8), PRECISION(kInt8)> ConvInt8_Int8;

// Same argument names, precisions and aliases as lite/kernels/arm/conv_compute.cc:216-252, target kHIP.
REGISTER_LITE_KERNEL(conv2d, kHIP, kInt8, kNCHW, ConvInt8_Int8, int8_out)
    .BindInput("Input", {LiteType::GetTensorTy(TARGET(kHIP), PRECISION(kInt8))})
    .BindInput("Bias", {LiteType::GetTensorTy(TARGET(kHIP), PRECISION(kFloat))})
    .BindInput("Filter", {LiteType::GetTensorTy(TARGET(kHIP), PRECISION(kInt8))})
    .BindOutput("Output", {LiteType::GetTensorTy(TARGET(kHIP), PRECISION(kInt8))})
    .Finalize();

REGISTER_LITE_KERNEL(conv2d, kHIP, kInt8, kNCHW, ConvInt8_Fp32, fp32_out)
    .BindInput("Input", {LiteType::GetTensorTy(TARGET(kHIP), PRECISION(kInt8))})
    .BindInput("Bias", {LiteType::GetTensorTy(TARGET(kHIP), PRECISION(kFloat))})
    .BindInput("Filter", {LiteType::GetTensorTy(TARGET(kHIP), PRECISION(kInt8))})
    .BindOutput("Output", {LiteType::GetTensorTy(TARGET(kHIP), PRECISION(kFloat))})
    .Finalize();

REGISTER_LITE_KERNEL(depthwise_conv2d, kHIP, kInt8, kNCHW, ConvInt8_Int8, int8_out)
    .BindInput("Input", {LiteType::GetTensorTy(TARGET(kHIP), PRECISION(kInt8))})
    .BindInput("Bias", {LiteType::GetTensorTy(TARGET(kHIP), PRECISION(kFloat))})
    .BindInput("Filter", {LiteType::GetTensorTy(TARGET(kHIP), PRECISION(kInt8))})
    .BindOutput("Output", {LiteType::GetTensorTy(TARGET(kHIP), PRECISION(kInt8))})
    .Finalize();

REGISTER_LITE_KERNEL(depthwise_conv2d, kHIP, kInt8, kNCHW, ConvInt8_Fp32, fp32_out)
    .BindInput("Input", {LiteType::GetTensorTy(TARGET(kHIP), PRECISION(kInt8))})
    .BindInput("Bias", {LiteType::GetTensorTy(TARGET(kHIP), PRECISION(kFloat))})
    .BindInput("Filter", {LiteType::GetTensorTy(TARGET(kHIP), PRECISION(kInt8))})
    .BindOutput("Output", {LiteType::GetTensorTy(TARGET(kHIP), PRECISION(kFloat))})
    .Finalize();

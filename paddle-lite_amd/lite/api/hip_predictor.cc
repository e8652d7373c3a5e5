// hip_predictor.cc — see hip_predictor.h.
#include "lite/api/hip_predictor.h"

#include "lite/kernels/hip/conv_fusion.h"
#include "plhip.h"

#include <cstring>

namespace paddle {
namespace lite {

std::unique_ptr<KernelBase> PickKernel(const std::string& op_type, const Place& place, const std::string& alias) {
  auto ks = KernelFactory::Global().Create(op_type, place.target, place.precision, place.layout);
  for (auto& k : ks)
    if (k->alias() == alias) return std::move(k);
  LOG(FATAL) << "no kernel registered for " << op_type << "/" << alias << " at " << place.DebugString()
             << "; registered ops:\n" << KernelFactory::Global().DebugString();
  return nullptr;
}

Tensor* HipPredictor::Var(const std::string& name) {
  auto it = vars_.find(name);
  if (it == vars_.end()) it = vars_.emplace(name, std::unique_ptr<Tensor>(new Tensor)).first;
  return it->second.get();
}

Tensor* HipPredictor::NewParam(const void* host, size_t bytes, const std::vector<int64_t>& dims, PrecisionType prec) {
  params_.emplace_back(new Tensor);
  Tensor* t = params_.back().get();
  t->Resize(dims);
  t->set_persistable(true);
  void* p = t->mutable_data(TARGET(kHost), bytes);
  t->set_precision(prec);
  std::memcpy(p, host, bytes);
  return t;
}

const std::shared_ptr<HipExecState>& HipPredictor::state() {
  if (!state_) {
    TargetWrapperHip::SetDevice(device_);
    state_ = TargetWrapperHip::State();
  }
  return state_;
}

void HipPredictor::Emit(std::shared_ptr<OpLite> op, std::unique_ptr<KernelBase> kernel) {
  kernel->SetContext(NewContext(TARGET(kHIP), device_, state()));
  op->AttachKernel(kernel.get());
  program_.Add(Instruction(std::move(op), std::move(kernel)));
}

Tensor* HipPredictor::AddFeed(const std::string& name, const std::vector<int64_t>& dims, PrecisionType prec) {
  Tensor* t = Var(name);
  t->Resize(dims);
  const size_t esz = prec == PRECISION(kInt8) ? 1 : 4;
  t->mutable_data(TARGET(kHost), static_cast<size_t>(t->numel()) * esz);
  t->set_precision(prec);
  return t;
}

void HipPredictor::AddIoCopy(const std::string& in, const std::string& out, bool h2d) {
  auto op = std::make_shared<operators::IoCopyOp>();
  op->mutable_param().x = Var(in);
  op->mutable_param().y = Var(out);
  Emit(op, PickKernel("io_copy", Place(TARGET(kHIP), PRECISION(kAny), DATALAYOUT(kAny)),
                      h2d ? "host_to_device" : "device_to_host"));
}

void HipPredictor::AddCalib(const std::string& in, const std::string& out, float scale, bool f2i) {
  auto op = std::make_shared<operators::CalibOpLite>();
  op->mutable_param().input = Var(in);
  op->mutable_param().output = Var(out);
  op->mutable_param().scale = scale;
  Emit(op, PickKernel("calib", Place(TARGET(kHIP), PRECISION(kInt8)), f2i ? "fp32_to_int8" : "int8_to_fp32"));
}

void HipPredictor::AddConv(const std::string& op_type, const std::string& in, const std::string& out, const int8_t* w,
                           const std::vector<int64_t>& w_dims, const float* bias, const ConvAttrs& a) {
  auto op = std::make_shared<operators::ConvOpLite>(op_type);
  auto& p = op->mutable_param();
  kernels::hip::HipConvFusion fz;  // this target's graph-level fusion state (lite/kernels/hip/conv_fusion.h)
  const bool fused = !a.calib_out.empty() || !a.residual.empty() || a.pw_w != nullptr || a.in_calib_scale > 0.f;
  size_t wn = 1;
  for (auto d : w_dims) wn *= static_cast<size_t>(d);
  p.x = Var(in);
  p.output = Var(out);
  p.filter = NewParam(w, wn, w_dims, PRECISION(kInt8));
  p.bias = bias ? NewParam(bias, static_cast<size_t>(w_dims[0]) * 4, {w_dims[0]}, PRECISION(kFloat)) : nullptr;
  p.strides = a.strides;
  p.paddings = std::make_shared<std::vector<int>>(a.paddings);
  p.dilations = std::make_shared<std::vector<int>>(a.dilations);
  p.groups = a.groups;
  p.enable_int8 = true;
  p.input_scale = a.input_scale;
  p.output_scale = a.output_scale;
  p.weight_scale = a.weight_scale;
  if (a.act != 0) {
    p.activation_param.has_active = true;
    p.activation_param.active_type = static_cast<lite_api::ActivationType>(a.act);
    if (a.act == 1) p.fuse_relu = true;
    if (a.act == 2) p.activation_param.Relu_clipped_coef = a.act_coef;
    if (a.act == 4) p.activation_param.Leaky_relu_alpha = a.act_coef;
  }
  if (!a.residual.empty()) {
    CHECK(!a.int8_out) << "the fused residual add belongs to the fp32_out kernel";
    p.fuse_residual_connection = true;
    p.residualData = Var(a.residual);
    fz.fuse_residual_relu = a.residual_relu;
  }
  if (!a.calib_out.empty()) {
    CHECK(!a.int8_out) << "the fused calib belongs to the fp32_out kernel";
    fz.calib_output = Var(a.calib_out);
    fz.calib_output->set_precision(PRECISION(kInt8));
    fz.calib_scale = a.calib_scale;
    fz.drop_fp32_output = a.drop_fp32;
  }
  if (a.pw_w) {
    CHECK(a.int8_out && op_type == "depthwise_conv2d") << "only a depthwise conv with int8 output takes a 1x1 consumer over";
    size_t pn = 1;
    for (auto d : a.pw_w_dims) pn *= static_cast<size_t>(d);
    fz.pw_filter = NewParam(a.pw_w, pn, a.pw_w_dims, PRECISION(kInt8));
    op->set_output_channels(a.pw_w_dims[0]);
    fz.pw_bias = a.pw_bias ? NewParam(a.pw_bias, static_cast<size_t>(a.pw_w_dims[0]) * 4, {a.pw_w_dims[0]}, PRECISION(kFloat)) : nullptr;
    fz.pw_weight_scale = a.pw_weight_scale;
    fz.pw_output_scale = a.pw_output_scale;
    fz.pw_int8_out = a.pw_int8_out;
    if (a.pw_act != 0) {
      fz.pw_activation_param.has_active = true;
      fz.pw_activation_param.active_type = static_cast<lite_api::ActivationType>(a.pw_act);
      if (a.pw_act == 2) fz.pw_activation_param.Relu_clipped_coef = a.pw_act_coef;
      if (a.pw_act == 4) fz.pw_activation_param.Leaky_relu_alpha = a.pw_act_coef;
    }
    p.output->set_precision(a.pw_int8_out ? PRECISION(kInt8) : PRECISION(kFloat));
    if (a.pw_pool) {
      CHECK(!a.pw_int8_out) << "the fused global average pool reads the 1x1 conv's fp32 output";
      fz.pw_global_avg_pool = true;
      op->set_output_pooled();
    }
  }
  if (a.in_calib_scale > 0.f) fz.calib_input_scale = a.in_calib_scale;
  op->set_padding_algorithm(a.padding_algorithm);
  auto kernel = PickKernel(op_type, Place(TARGET(kHIP), PRECISION(kInt8)), a.int8_out ? "int8_out" : "fp32_out");
  if (fused) {  // this target's fusion state goes to the kernel object, not into the reference's ConvParam (conv_fusion.h)
    auto* fk = dynamic_cast<kernels::hip::HipFusableKernel*>(kernel.get());
    CHECK(fk) << "the picked conv kernel does not take a kHIP fusion";
    fk->SetFusion(fz);
  }
  Emit(op, std::move(kernel));
}

void HipPredictor::AddFc(const std::string& in, const std::string& out, const int8_t* w, int k, int n, const float* bias,
                         float input_scale, const std::vector<float>& weight_scale, float output_scale, bool int8_out,
                         bool relu) {
  auto op = std::make_shared<operators::FcOpLite>();
  auto& p = op->mutable_param();
  p.input = Var(in);
  p.output = Var(out);
  p.w = NewParam(w, static_cast<size_t>(k) * n, {k, n}, PRECISION(kInt8));
  p.bias = bias ? NewParam(bias, static_cast<size_t>(n) * 4, {n}, PRECISION(kFloat)) : nullptr;
  p.in_num_col_dims = 1;
  p.enable_int8 = true;
  p.input_scale = input_scale;
  p.weight_scale = weight_scale;
  p.output_scale = output_scale;
  if (relu) p.activation_type = "relu";
  Emit(op, PickKernel("fc", Place(TARGET(kHIP), PRECISION(kInt8)), int8_out ? "int8out" : "fp32out"));
}

void HipPredictor::AddGlobalAvgPool(const std::string& in, const std::string& out) {
  auto op = std::make_shared<operators::PoolOpLite>();
  auto& p = op->mutable_param();
  p.x = Var(in);
  p.output = Var(out);
  p.pooling_type = "avg";
  p.global_pooling = true;
  p.paddings = std::make_shared<std::vector<int>>(std::vector<int>{0, 0, 0, 0});
  Emit(op, PickKernel("pool2d", Place(TARGET(kHIP), PRECISION(kFloat)), "def"));
}

void HipPredictor::AddPool(const std::string& in, const std::string& out, const std::string& pooling_type,
                           const std::vector<int>& ksize, const std::vector<int>& strides, const std::vector<int>& paddings,
                           bool global_pooling, bool exclusive, bool ceil_mode, bool int8) {
  auto op = std::make_shared<operators::PoolOpLite>();
  auto& p = op->mutable_param();
  p.x = Var(in);
  p.output = Var(out);
  p.pooling_type = pooling_type;
  p.ksize = ksize;
  p.strides = strides;
  p.global_pooling = global_pooling;
  p.exclusive = exclusive;
  p.ceil_mode = ceil_mode;
  p.paddings = std::make_shared<std::vector<int>>(paddings);
  Emit(op, PickKernel("pool2d", Place(TARGET(kHIP), int8 ? PRECISION(kInt8) : PRECISION(kFloat)), "def"));
}

void HipPredictor::AddElementwiseAdd(const std::string& x, const std::string& y, const std::string& out,
                                     const std::string& act_type) {
  if (act_type.empty()) {
    auto op = std::make_shared<operators::ElementwiseOp>("elementwise_add");
    auto& p = op->mutable_param();
    p.X = Var(x);
    p.Y = Var(y);
    p.Out = Var(out);
    Emit(op, PickKernel("elementwise_add", Place(TARGET(kHIP), PRECISION(kFloat)), "def"));
  } else {
    auto op = std::make_shared<operators::FusionElementwiseActivationOp>("fusion_elementwise_add_activation");
    auto& p = op->mutable_param();
    p.X = Var(x);
    p.Y = Var(y);
    p.Out = Var(out);
    p.act_type = act_type;
    Emit(op, PickKernel("fusion_elementwise_add_activation", Place(TARGET(kHIP), PRECISION(kFloat)), "def"));
  }
}

void HipPredictor::AddSoftmax(const std::string& in, const std::string& out) {
  auto op = std::make_shared<operators::SoftmaxOp>();
  op->mutable_param().x = Var(in);
  op->mutable_param().output = Var(out);
  op->mutable_param().axis = -1;
  Emit(op, PickKernel("softmax", Place(TARGET(kHIP), PRECISION(kFloat)), "def"));
}

std::vector<std::string> HipPredictor::KernelNames() {
  std::vector<std::string> r;
  for (auto& i : program_.instructions())
    r.push_back(i.kernel()->name() + "/" + i.kernel()->alias() + " -> " + i.kernel()->kernel_func_name());
  return r;
}

// Everything a recorded launch graph has baked in: the shape of every variable (grid sizes, strides), the instruction list
// and the workspace arena (its address is a kernel argument; HipExecState::Workspace frees the arena when it grows).
size_t HipPredictor::GraphKey() const {
  size_t h = 1469598103934665603ULL;
  auto mix = [&h](size_t v) { h = (h ^ v) * 1099511628211ULL; };
  mix(const_cast<HipPredictor*>(this)->program_.instructions().size());
  mix(reinterpret_cast<size_t>(state_->workspace_ptr()));
  mix(state_->workspace_bytes());
  for (auto& kv : vars_) {
    const auto d = kv.second->dims();
    mix(d.size());
    for (size_t i = 0; i < d.size(); ++i) mix(static_cast<size_t>(d[i]));
  }
  return h;
}

void HipPredictor::RunGraph() {
  TargetWrapperHip::SetDevice(device_);
  plhip_ctx* ctx = state()->ctx();
  // a feed resized, an instruction added or the arena re-allocated since the recording: the graph is stale (it would read
  // freed memory / use old shapes): drop it and record again
  if (graph_exec_ && graph_key_ != GraphKey()) {
    (void)plhip_graph_destroy(ctx, graph_exec_);
    graph_exec_ = nullptr;
  }
  if (!graph_exec_) {
    // one ordinary run first: PrepareForRun (allocations, weight packing, host syncs) must not happen inside a capture,
    // and it brings the workspace arena to its final size
    program_.Run(/*skip_io_copy=*/true);
    state()->Sync();
    HIP_CALL(ctx, plhip_graph_begin(ctx));
    try {
      program_.Run(/*skip_io_copy=*/true);  // InferShape() + Launch() per instruction, recorded instead of executed
    } catch (...) {
      void* dead = nullptr;  // never leave the stream in capture mode: end it, discard whatever was recorded, re-throw
      if (plhip_graph_end(ctx, &dead) == PLHIP_OK && dead) (void)plhip_graph_destroy(ctx, dead);
      throw;
    }
    HIP_CALL(ctx, plhip_graph_end(ctx, &graph_exec_));
    graph_key_ = GraphKey();
  }
  HIP_CALL(ctx, plhip_graph_launch(ctx, graph_exec_));
}

HipPredictor::~HipPredictor() {
  if (graph_exec_ && state_) (void)plhip_graph_destroy(state_->ctx(), graph_exec_);
}

}  // namespace lite
}  // namespace paddle

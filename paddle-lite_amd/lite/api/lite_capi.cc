// lite_capi.cc — see lite_capi.h.
#include "lite/kernels/hip/packed_weight_cache.h"
#include "lite/api/lite_capi.h"

#include <cstring>
#include <string>

#include "lite/api/graph_builder.h"
#include "lite/api/hip_predictor.h"
#include "lite/core/profile/timer.h"
#include "lite/model_parser/hip_model.h"

using paddle::lite::HipPredictor;
using paddle::lite::Tensor;

struct pllite_predictor {
  explicit pllite_predictor(int dev) : pred(dev) {}
  HipPredictor pred;
  paddle::lite::GraphBuilder graph;
};

namespace {
thread_local std::string g_err;
template <typename F>
int guarded(F&& f) {
  try {
    f();
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}
}  // namespace

extern "C" {

const char* pllite_last_error(void) { return g_err.c_str(); }

int pllite_registered_kernels(const char* op_type, int precision, int layout) {
  auto ks = paddle::lite::KernelFactory::Global().Create(op_type, TARGET(kHIP),
                                                         static_cast<paddle::lite::PrecisionType>(precision),
                                                         static_cast<paddle::lite::DataLayoutType>(layout));
  return static_cast<int>(ks.size());
}

// how often a kernel object found its packed weights already on the device / packed them itself (packed_weight_cache.h)
void pllite_packed_weight_cache_stats(long* hits, long* misses) {
  auto& c = paddle::lite::kernels::hip::PackedWeightCache::Global();
  if (hits) *hits = c.hits();
  if (misses) *misses = c.misses();
}

int pllite_adopt_stream(int device, void* stream) {
  return guarded([&] { paddle::lite::TargetWrapperHip::AdoptStream(device, stream); });
}

pllite_predictor* pllite_predictor_create(int device) {
  pllite_predictor* p = nullptr;
  if (guarded([&] {
        p = new pllite_predictor(device);
        (void)p->pred.state();  // fail here, loudly, when there is no gfx950 device
      }) != 0) {
    delete p;
    return nullptr;
  }
  return p;
}
void pllite_predictor_destroy(pllite_predictor* p) { delete p; }

int pllite_add_feed(pllite_predictor* p, const char* name, const int64_t* dims, int ndims, int precision) {
  return guarded([&] {
    p->pred.AddFeed(name, std::vector<int64_t>(dims, dims + ndims), static_cast<paddle::lite::PrecisionType>(precision));
  });
}
int pllite_add_io_copy(pllite_predictor* p, const char* in, const char* out, int h2d) {
  return guarded([&] { p->pred.AddIoCopy(in, out, h2d != 0); });
}
int pllite_add_calib(pllite_predictor* p, const char* in, const char* out, float scale, int f2i) {
  return guarded([&] { p->pred.AddCalib(in, out, scale, f2i != 0); });
}
int pllite_add_conv(pllite_predictor* p, const char* op_type, const char* in, const char* out, const int8_t* w,
                    const int64_t* w_dims, const float* bias, const int* strides, const int* paddings, int n_paddings,
                    const int* dilations, int groups, int act, float act_coef, float input_scale,
                    const float* weight_scale, int n_weight_scale, float output_scale, int int8_out,
                    const char* padding_algorithm) {
  return guarded([&] {
    paddle::lite::ConvAttrs a;
    a.strides = {strides[0], strides[1]};
    a.paddings.assign(paddings, paddings + n_paddings);
    a.dilations = {dilations[0], dilations[1]};
    a.groups = groups;
    a.act = act;
    a.act_coef = act_coef;
    a.input_scale = input_scale;
    a.output_scale = output_scale;
    a.weight_scale.assign(weight_scale, weight_scale + n_weight_scale);
    a.int8_out = int8_out != 0;
    a.padding_algorithm = padding_algorithm ? padding_algorithm : "";
    p->pred.AddConv(op_type, in, out, w, std::vector<int64_t>(w_dims, w_dims + 4), bias, a);
  });
}
int pllite_add_fc(pllite_predictor* p, const char* in, const char* out, const int8_t* w, int k, int n, const float* bias,
                  float input_scale, const float* weight_scale, int n_ws, float output_scale, int int8_out, int relu) {
  return guarded([&] {
    p->pred.AddFc(in, out, w, k, n, bias, input_scale, std::vector<float>(weight_scale, weight_scale + n_ws), output_scale,
                  int8_out != 0, relu != 0);
  });
}
int pllite_add_global_avg_pool(pllite_predictor* p, const char* in, const char* out) {
  return guarded([&] { p->pred.AddGlobalAvgPool(in, out); });
}
int pllite_add_softmax(pllite_predictor* p, const char* in, const char* out) {
  return guarded([&] { p->pred.AddSoftmax(in, out); });
}
int pllite_add_pool(pllite_predictor* p, const char* in, const char* out, const char* pooling_type, const int* ksize,
                    const int* strides, const int* pads, int global_pooling, int exclusive, int ceil_mode) {
  return guarded([&] {
    p->pred.AddPool(in, out, pooling_type, {ksize[0], ksize[1]}, {strides[0], strides[1]},
                    {pads[0], pads[1], pads[2], pads[3]}, global_pooling != 0, exclusive != 0, ceil_mode != 0);
  });
}
int pllite_add_elementwise_add(pllite_predictor* p, const char* x, const char* y, const char* out, const char* act_type) {
  return guarded([&] { p->pred.AddElementwiseAdd(x, y, out, act_type ? act_type : ""); });
}

// ---- graph mode
pllite_predictor* pllite_predictor_create_planner(void) {
  pllite_predictor* p = nullptr;
  if (guarded([&] { p = new pllite_predictor(0); }) != 0) return nullptr;
  return p;
}
int pllite_graph_feed(pllite_predictor* p, const char* name, const int64_t* dims, int ndims, int precision) {
  return guarded([&] {
    p->graph.Feed(name, std::vector<int64_t>(dims, dims + ndims), static_cast<paddle::lite::PrecisionType>(precision));
  });
}
int pllite_graph_conv(pllite_predictor* p, const char* op_type, const char* in, const char* out, const int8_t* w,
                      const int64_t* w_dims, const float* bias, const int* strides, const int* paddings, int n_paddings,
                      const int* dilations, int groups, int act, float act_coef, float input_scale,
                      const float* weight_scale, int n_weight_scale, const char* padding_algorithm) {
  return guarded([&] {
    auto& op = p->graph.Add(op_type, {in}, out);
    op.enable_int8 = true;
    op.w_dims.assign(w_dims, w_dims + 4);
    size_t wn = 1;
    for (auto d : op.w_dims) wn *= static_cast<size_t>(d);
    op.w.assign(w, w + wn);
    op.has_bias = bias != nullptr;
    if (bias) op.bias.assign(bias, bias + w_dims[0]);
    auto& a = op.conv;
    a.strides = {strides[0], strides[1]};
    a.paddings.assign(paddings, paddings + n_paddings);
    a.dilations = {dilations[0], dilations[1]};
    a.groups = groups;
    a.act = act;
    a.act_coef = act_coef;
    a.input_scale = input_scale;
    a.weight_scale.assign(weight_scale, weight_scale + n_weight_scale);
    a.padding_algorithm = padding_algorithm ? padding_algorithm : "";
  });
}
int pllite_graph_fc(pllite_predictor* p, const char* in, const char* out, const int8_t* w, int k, int n, const float* bias,
                    float input_scale, const float* weight_scale, int n_ws, int relu) {
  return guarded([&] {
    auto& op = p->graph.Add("fc", {in}, out);
    op.enable_int8 = true;
    op.w_dims = {k, n};
    op.w.assign(w, w + static_cast<size_t>(k) * n);
    op.has_bias = bias != nullptr;
    if (bias) op.bias.assign(bias, bias + n);
    op.conv.input_scale = input_scale;
    op.conv.weight_scale.assign(weight_scale, weight_scale + n_ws);
    op.fc_relu = relu != 0;
  });
}
int pllite_graph_pool(pllite_predictor* p, const char* in, const char* out, const char* pooling_type, const int* ksize,
                      const int* strides, const int* pads, int global_pooling, int exclusive, int ceil_mode) {
  return guarded([&] {
    auto& op = p->graph.Add("pool2d", {in}, out);
    op.pooling_type = pooling_type;
    op.ksize = {ksize[0], ksize[1]};
    op.pool_strides = {strides[0], strides[1]};
    op.pool_paddings = {pads[0], pads[1], pads[2], pads[3]};
    op.global_pooling = global_pooling != 0;
    op.exclusive = exclusive != 0;
    op.ceil_mode = ceil_mode != 0;
  });
}
int pllite_graph_elementwise_add(pllite_predictor* p, const char* x, const char* y, const char* out, const char* act_type) {
  return guarded([&] {
    const std::string act = act_type ? act_type : "";
    auto& op = p->graph.Add(act.empty() ? "elementwise_add" : "fusion_elementwise_add_activation", {x, y}, out);
    op.act_type = act;
  });
}
int pllite_graph_softmax(pllite_predictor* p, const char* in, const char* out) {
  return guarded([&] { p->graph.Add("softmax", {in}, out); });
}
int pllite_graph_set_fuse(pllite_predictor* p, int on) {
  return guarded([&] { p->graph.set_fuse(on != 0); });
}
int pllite_graph_set_fuse_dwpw(pllite_predictor* p, int on) {
  return guarded([&] { p->graph.set_fuse_dwpw(on != 0); });
}
int pllite_graph_fetch(pllite_predictor* p, const char* name) {
  return guarded([&] { p->graph.Fetch(name); });
}
static void join_lines(const std::vector<std::string>& v, char* buf, int cap) {
  std::string s;
  for (auto& n : v) s += n + "\n";
  CHECK_LT(static_cast<int>(s.size()), cap) << "buffer too small";
  std::memcpy(buf, s.c_str(), s.size() + 1);
}
int pllite_graph_plan(pllite_predictor* p, char* buf, int cap) {
  return guarded([&] { join_lines(p->graph.Plan(), buf, cap); });
}
int pllite_graph_lower(pllite_predictor* p, char* buf, int cap) {
  return guarded([&] { join_lines(p->graph.Lower(&p->pred), buf, cap); });
}

int pllite_load_model(pllite_predictor* p, const void* bytes, int64_t nbytes, int batch) {
  return guarded([&] {
    CHECK(bytes && nbytes > 16 && batch > 0) << "pllite_load_model: bad argument";
    const uint8_t* b = static_cast<const uint8_t*>(bytes);
    std::vector<uint8_t> v(b, b + nbytes);
    auto model = paddle::lite::model_parser::ParseContainer(v);
    paddle::lite::model_parser::BuildGraph(&model, batch, &p->graph);
  });
}
int pllite_graph_num_ops(pllite_predictor* p) { return static_cast<int>(p->graph.ops().size()); }
int pllite_graph_op_params(pllite_predictor* p, int index, char* type, int type_cap, int8_t* w, int64_t* n_w, float* bias,
                           int* n_bias, float* weight_scale, int* n_scale, float* input_scale, int* act, float* act_coef) {
  return guarded([&] {
    const auto& ops = p->graph.ops();
    CHECK(index >= 0 && index < static_cast<int>(ops.size())) << "op index out of range";
    const auto& o = ops[index];
    if (type && type_cap > 0) {
      std::strncpy(type, o.type.c_str(), static_cast<size_t>(type_cap) - 1);
      type[type_cap - 1] = 0;
    }
    if (n_w) *n_w = static_cast<int64_t>(o.w.size());
    if (n_bias) *n_bias = o.has_bias ? static_cast<int>(o.bias.size()) : 0;
    if (n_scale) *n_scale = static_cast<int>(o.conv.weight_scale.size());
    if (w) std::memcpy(w, o.w.data(), o.w.size());
    if (bias && o.has_bias) std::memcpy(bias, o.bias.data(), o.bias.size() * sizeof(float));
    if (weight_scale) std::memcpy(weight_scale, o.conv.weight_scale.data(), o.conv.weight_scale.size() * sizeof(float));
    if (input_scale) *input_scale = o.conv.input_scale;
    if (act) *act = o.conv.act;
    if (act_coef) *act_coef = o.conv.act_coef;
  });
}

int pllite_set_input(pllite_predictor* p, const char* name, const void* host, int64_t bytes) {
  return guarded([&] {
    CHECK(p->pred.HasVar(name)) << "unknown variable " << name;
    Tensor* t = p->pred.Var(name);
    CHECK_EQ(static_cast<int64_t>(t->memory_size()), bytes) << "input size mismatch for " << name;
    paddle::lite::TargetCopy(t->target(), TARGET(kHost), t->raw_data(), host, static_cast<size_t>(bytes));
  });
}
int pllite_run(pllite_predictor* p, int skip_io_copy) {
  return guarded([&] { p->pred.Run(skip_io_copy != 0); });
}
int pllite_run_graph(pllite_predictor* p) {
  return guarded([&] { p->pred.RunGraph(); });
}
int pllite_num_instructions(pllite_predictor* p) { return static_cast<int>(p->pred.program().instructions().size()); }
int pllite_run_instruction(pllite_predictor* p, int index) {
  return guarded([&] {
    auto& insts = p->pred.program().instructions();
    CHECK(index >= 0 && index < static_cast<int>(insts.size())) << "instruction index out of range";
    insts[index].Run();
  });
}
int pllite_sync(pllite_predictor* p) {
  return guarded([&] { p->pred.Sync(); });
}
int pllite_get_var(pllite_predictor* p, const char* name, void* host, int64_t capacity, int64_t* bytes, int64_t* dims4,
                   int* ndims) {
  return guarded([&] {
    CHECK(p->pred.HasVar(name)) << "unknown variable " << name;
    Tensor* t = p->pred.Var(name);
    const int64_t n = static_cast<int64_t>(t->memory_size());
    CHECK_LE(n, capacity) << "host buffer too small for " << name;
    p->pred.Sync();
    paddle::lite::TargetCopy(TARGET(kHost), t->target(), host, t->raw_data(), static_cast<size_t>(n));
    if (bytes) *bytes = n;
    if (ndims) *ndims = static_cast<int>(t->dims().size());
    if (dims4)
      for (size_t i = 0; i < t->dims().size() && i < 4; ++i) dims4[i] = t->dims()[static_cast<int>(i)];
  });
}
void* pllite_var_device_ptr(pllite_predictor* p, const char* name) {
  if (!p->pred.HasVar(name)) return nullptr;
  Tensor* t = p->pred.Var(name);
  return t->target() == TARGET(kHIP) ? t->raw_data() : nullptr;
}
int pllite_copy_var_to_device(pllite_predictor* p, const char* name, void* dst_dev, int64_t bytes) {
  return guarded([&] {
    CHECK(p->pred.HasVar(name)) << "unknown variable " << name;
    Tensor* t = p->pred.Var(name);
    CHECK(t->target() == TARGET(kHIP)) << name << " is not device resident";
    CHECK_EQ(static_cast<int64_t>(t->memory_size()), bytes);
    p->pred.state()->MemcpyAsync(dst_dev, t->raw_data(), static_cast<size_t>(bytes), paddle::lite::IoDirection::DtoD);
  });
}
int pllite_time_instruction(pllite_predictor* p, int index, int reps, float* avg_ms, float* min_ms, char* func_name, int cap) {
  return guarded([&] {
    auto& insts = p->pred.program().instructions();
    CHECK(index >= 0 && index < static_cast<int>(insts.size())) << "instruction index out of range";
    CHECK_GT(reps, 0);
    auto& inst = insts[index];
    paddle::lite::profile::DeviceTimer<paddle::lite::TargetType::kHIP> timer;
    inst.Run();  // PrepareForRun / warm-up outside the laps
    for (int r = 0; r < reps; ++r) {
      timer.Start(inst.kernel()->mutable_context());
      inst.Run();
      timer.Stop(inst.kernel()->mutable_context());
    }
    if (avg_ms) *avg_ms = timer.LapTimes().Avg();
    if (min_ms) *min_ms = timer.LapTimes().Min();
    paddle::lite::profile::OpCharacter ch;
    inst.kernel()->SetProfileRuntimeKernelInfo(&ch);
    if (func_name && cap > 0) {
      std::strncpy(func_name, ch.kernel_func_name.c_str(), static_cast<size_t>(cap) - 1);
      func_name[cap - 1] = 0;
    }
  });
}
int pllite_kernel_names(pllite_predictor* p, char* buf, int cap) {
  return guarded([&] {
    std::string s;
    for (auto& n : p->pred.KernelNames()) s += n + "\n";
    CHECK_LT(static_cast<int>(s.size()), cap);
    std::memcpy(buf, s.c_str(), s.size() + 1);
  });
}

}  // extern "C"

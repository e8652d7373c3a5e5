// hip_predictor.h — a mini CxxPredictor for the kHIP target: a name->Tensor scope plus a RuntimeProgram of
// {OpLite, KernelBase} instructions built in code (the model parser and MIR optimiser of the reference are out of
// scope — SURVEY.md §2).  What it does restate from the reference's build pipeline:
//   * kernel choice by (op_type, Place{kHIP, precision, layout}, alias) through KernelFactory, with the
//     static_kernel_pick rule for enable_int8 ops: alias int8_out iff the consumer is int8, else fp32_out
//     (lite/core/mir/static_kernel_pick_pass.cc:92-165) — the caller states which;
//   * io_copy between host and device tensors (type_target_cast_pass) and calib on precision edges
//     (type_precision_cast_pass);
//   * one KernelContext per instruction from NewContext(target) (runtime_context_assign_pass);
//   * Run() = for inst: InferShape(); Launch()   (program.cc:265-315, 436-467).
#pragma once
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "lite/core/op_registry.h"
#include "lite/core/program.h"

namespace paddle {
namespace lite {

std::unique_ptr<KernelBase> PickKernel(const std::string& op_type, const Place& place, const std::string& alias);

struct ConvAttrs {
  std::vector<int> strides{1, 1}, paddings{0, 0, 0, 0}, dilations{1, 1};
  int groups{1};
  int act{0};            // lite_api::ActivationType value: 0, 1 relu, 2 relu6, 4 leaky
  float act_coef{0.f};   // Relu_clipped_coef or Leaky_relu_alpha
  float input_scale{1.f}, output_scale{1.f};
  std::vector<float> weight_scale;
  bool int8_out{true};
  std::string padding_algorithm{""};
  // kHIP fused tail of an fp32_out conv (op_params.h ConvParam, graph_builder.h): variable names, "" = none
  std::string residual, calib_out;
  bool residual_relu{false}, drop_fp32{false};
  float calib_scale{1.f};
  // kHIP fused 1x1 consumer of a depthwise conv (ConvParam::pw_*): `out` of AddConv is then the pointwise conv's output
  const int8_t* pw_w{nullptr};
  std::vector<int64_t> pw_w_dims;
  const float* pw_bias{nullptr};
  std::vector<float> pw_weight_scale;
  float pw_output_scale{1.f};
  bool pw_int8_out{true};
  int pw_act{0};
  float pw_act_coef{0.f};
  bool pw_pool{false};  // ... and the global average pool2d behind it: `out` is the pool's output, [n, cout, 1, 1] fp32
  float in_calib_scale{0.f};  // kHIP: the calib[fp32_to_int8] in front taken over: `in` of AddConv is the calib's fp32 input (0 = none)
};

class HipPredictor {
 public:
  explicit HipPredictor(int device_id) : device_(device_id) { TargetWrapperHip::SetDevice(device_id); }
  Tensor* Var(const std::string& name);
  bool HasVar(const std::string& name) const { return vars_.count(name) != 0; }

  // host tensor that the caller fills (feed) — fp32 or int8
  Tensor* AddFeed(const std::string& name, const std::vector<int64_t>& dims, PrecisionType prec);
  void AddIoCopy(const std::string& in, const std::string& out, bool host_to_device);
  void AddCalib(const std::string& in, const std::string& out, float scale, bool fp32_to_int8);
  void AddConv(const std::string& op_type, const std::string& in, const std::string& out, const int8_t* w,
               const std::vector<int64_t>& w_dims, const float* bias, const ConvAttrs& attrs);
  void AddFc(const std::string& in, const std::string& out, const int8_t* w, int k, int n, const float* bias,
             float input_scale, const std::vector<float>& weight_scale, float output_scale, bool int8_out, bool relu);
  void AddGlobalAvgPool(const std::string& in, const std::string& out);
  void AddPool(const std::string& in, const std::string& out, const std::string& pooling_type, const std::vector<int>& ksize,
               const std::vector<int>& strides, const std::vector<int>& paddings, bool global_pooling, bool exclusive,
               bool ceil_mode, bool int8 = false);
  // act_type "" -> elementwise_add, "relu" -> fusion_elementwise_add_activation
  void AddElementwiseAdd(const std::string& x, const std::string& y, const std::string& out, const std::string& act_type);
  void AddSoftmax(const std::string& in, const std::string& out);

  void Run(bool skip_io_copy = false) {
    TargetWrapperHip::SetDevice(device_);
    program_.Run(skip_io_copy);
  }
  // Replays the device part of the program (everything but the io_copy instructions) as ONE launch graph: recorded on
  // the first call (the program must have run once before: PrepareForRun, workspace), replayed afterwards.  Feeds and
  // fetches keep their device addresses, so new input is a plain copy into the feed's device tensor before the call.
  void RunGraph();
  ~HipPredictor();
  // The predictor's execution state (device stream + workspace): the creating thread's default state at the first
  // instruction (after pllite_adopt_stream: the adopted stream), kept for life — Run() from any thread uses it.
  const std::shared_ptr<HipExecState>& state();
  void Sync() { state()->Sync(); }
  RuntimeProgram& program() { return program_; }
  std::vector<std::string> KernelNames();

 private:
  void Emit(std::shared_ptr<OpLite> op, std::unique_ptr<KernelBase> kernel);
  Tensor* NewParam(const void* host, size_t bytes, const std::vector<int64_t>& dims, PrecisionType prec);
  int device_;
  std::shared_ptr<HipExecState> state_;
  void* graph_exec_{nullptr};  // plhip launch graph of the program (RunGraph)
  size_t graph_key_{0};        // what the recorded graph depends on: shapes of every variable, instruction count, workspace arena
  size_t GraphKey() const;
  std::map<std::string, std::unique_ptr<Tensor>> vars_;
  std::vector<std::unique_ptr<Tensor>> params_;
  RuntimeProgram program_;
};

}  // namespace lite
}  // namespace paddle

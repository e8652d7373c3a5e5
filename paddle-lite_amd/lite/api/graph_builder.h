// graph_builder.h — from an op list "as the optimiser sees it after the fusion passes" to the kHIP RuntimeProgram.
//
// The reference builds its program with ~60 MIR passes; the model parser and the generic optimiser are
// out of scope.  What decides WHICH int8 kernels run and WHERE precision casts sit is restated here, in the order the
// reference applies it (lite/api/cxx_api.cc / lite/core/optimizer.h pass list):
//   1. static_kernel_pick_pass (lite/core/mir/static_kernel_pick_pass.cc:92-165): an enable_int8 op takes the
//      int8-output kernel iff EVERY consumer of its output is enable_int8, and then inherits the first consumer's
//      input scale as its output scale; otherwise the fp32-output kernel.
//   2. type_target_cast_pass: io_copy host->device behind every feed, device->host in front of every fetch.
//   3. type_precision_cast_pass (lite/core/mir/type_precision_cast_pass.cc:60-100, 130-260): where a consumer's declared
//      input precision differs from the tensor's, a calib op is inserted; one calib per source tensor, shared by later
//      consumers (`cast_nodes`); its scale is the consumer's input scale (fp32->int8) or the producer's output scale
//      (int8->fp32); its output is named "<var>/precision_trans".
// Ops arrive in topological order with the conv+bn, conv+activation, fc and elementwise_add+activation fusions already
// applied (the model loader, lite/model_parser of this repo, does the weight-side part of those).
#pragma once
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "lite/api/hip_predictor.h"

namespace paddle {
namespace lite {

struct GraphOp {
  std::string type;  // conv2d | depthwise_conv2d | fc | pool2d | elementwise_add | fusion_elementwise_add_activation | softmax
  std::vector<std::string> inputs;
  std::string output;
  bool enable_int8{false};
  // conv2d / depthwise_conv2d / fc
  std::vector<int8_t> w;
  std::vector<int64_t> w_dims;
  std::vector<float> bias;
  bool has_bias{false};
  ConvAttrs conv;  // input_scale, weight_scale, act, strides ...; output_scale / int8_out are set by Lower()
  bool fc_relu{false};
  // pool2d
  std::string pooling_type{"max"};
  std::vector<int> ksize{1, 1}, pool_strides{1, 1}, pool_paddings{0, 0, 0, 0};
  bool global_pooling{false}, exclusive{true}, ceil_mode{false};
  // fusion_elementwise_add_activation
  std::string act_type;
};

class GraphBuilder {
 public:
  void Feed(const std::string& name, const std::vector<int64_t>& dims, PrecisionType prec);
  void Fetch(const std::string& name) { fetches_.push_back(name); }
  // Graph-level fusions of the kHIP target on top of the reference's program (default on; results are bit-identical to
  // the unfused program, every fused value is rounded as the separate instructions round it):
  //   conv2d[fp32_out] -> elementwise_add | fusion_elementwise_add_activation(relu) -> calib   => ONE conv launch
  //   conv2d[fp32_out] -> pool2d(max) -> calib                                  => conv + fused calib, int8 max pool
  void set_fuse(bool on) { fuse_ = on; }
  //   depthwise_conv2d[int8_out] -> conv2d 1x1 (stride 1, no padding, groups 1, no fused tail), sole consumer  => ONE instruction
  // Default (mode 2): only the pairs the fused kernel takes (plhip_dwpw_fused_supported on the shapes propagated from the
  // feeds: one launch, the int8 tensor between the two convs never leaves the CU — DESIGN.md 8); set_fuse_dwpw(true) = mode 1
  // takes every eligible pair over (shapes outside the kernel run as two launches inside the one instruction), false = off.
  void set_fuse_dwpw(bool on) { fuse_dwpw_ = on ? 1 : 0; }
  void set_fuse_dwpw_mode(int mode) { fuse_dwpw_ = mode; }
  GraphOp& Add(const std::string& type, const std::vector<std::string>& inputs, const std::string& output);
  // Emits the program into `pred`; returns the host-side names of the fetched variables ("<name>/host").
  std::vector<std::string> Lower(HipPredictor* pred);
  // The decisions of passes 1-3 as text, one instruction per line (CPU-testable without a device):
  //   "conv2d/int8_out in=a out=b oscale=0.031496"   "calib/fp32_to_int8 in=x out=x/precision_trans scale=..."
  std::vector<std::string> Plan();
  const std::vector<GraphOp>& ops() const { return ops_; }

 private:
  struct Step {
    int op{-1};              // index into ops_, or -1 for an inserted instruction
    std::string kind;        // "op", "io_copy_h2d", "io_copy_d2h", "calib_f2i", "calib_i2f"
    std::string in, out;
    float scale{0.f};
    bool int8_out{false};
    float out_scale{1.f};
    std::vector<std::string> op_inputs;  // op inputs after cast renaming
    // kHIP fusions (FuseSteps): tail taken over by an fp32_out conv, int8 max pool behind a fused calib
    std::string res;          // residual operand of the fused elementwise_add ("" = none)
    bool res_relu{false};
    std::string calib_out;    // int8 tensor of the fused calib ("" = none)
    float calib_scale{1.f};
    bool drop_f32{false};     // the fp32 output has no consumer left
    bool pool_int8{false};    // pool2d(max) moved behind the calib: max commutes with the monotonic quantiser
    int pw_op{-1};            // depthwise conv that took its 1x1 consumer over: index of that conv in ops_
    bool pw_int8_out{true};
    float pw_out_scale{1.f};
    std::string via;          // name the depthwise result would have had
    bool pw_pool{false};      // ... and that conv's sole consumer, a global average pool2d, too (E): `out` is the pool's output
    std::string via_pw;       // name the 1x1 conv's result would have had
    float in_calib_scale{0.f};  // conv that took the calib[fp32_to_int8] in front of it over (F): its input is the calib's fp32 input
    std::string via_in;       // name the calib's int8 result would have had
  };
  std::vector<Step> Schedule();
  void FuseSteps(std::vector<Step>* steps);
  struct FeedDesc {
    std::string name;
    std::vector<int64_t> dims;
    PrecisionType prec;
  };
  bool fuse_{true};
  int fuse_dwpw_{2};
  std::vector<FeedDesc> feeds_;
  std::vector<std::string> fetches_;
  std::vector<GraphOp> ops_;
};

}  // namespace lite
}  // namespace paddle

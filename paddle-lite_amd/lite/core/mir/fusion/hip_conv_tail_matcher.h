// hip_conv_tail_matcher.h — the pattern matcher of the kHIP conv-tail fusions, graph-representation agnostic.
//
// ONE algorithm, two users: this repository's GraphBuilder::FuseSteps (lite/api/graph_builder.cc) and, inside a Paddle-Lite
// tree, mir::HipConvTailFusePass (patches/0006: lite/core/mir/fusion/hip_conv_tail_fuse_pass.cc carries this file verbatim and
// adapts it to the SSAGraph).  It runs AFTER static_kernel_pick_pass and type_precision_cast_pass — the calib instructions
// exist and every conv knows its output precision — in the place of the pass list where the reference's target-specific
// post-passes sit, and follows the shape of lite/core/mir/fusion/conv_elementwise_fuse_pass.cc + conv_elementwise_fuser.cc
// (match, rewire, mark the surviving op), attaching the residual operand through the reference's own
// ConvParam::residualData / fuse_residual_connection (lite/operators/conv_op.h:102).
//
// Patterns (every fused value is rounded exactly as the separate instructions round it: results are bit-identical):
//   (A) conv2d[fp32_out] -> elementwise_add | fusion_elementwise_add_activation(relu)
//         the add goes into the LATER (in program order) of its fp32 conv producers whose output feeds only the add;
//   (C) conv2d[fp32_out] -> pool2d(max) -> calib[fp32_to_int8], each with one consumer
//         the conv quantises (its fp32 output is never written), the pool runs on int8: max commutes with the monotonic
//         quantiser of type_trans.cc:183-184;
//   (B) conv2d[fp32_out] (possibly already carrying an add) -> calib[fp32_to_int8]
//         the calib's int8 tensor becomes a second output of the conv; the fp32 one is dropped when nobody else reads it.
#pragma once
#include <string>
#include <vector>

namespace paddle {
namespace lite {
namespace mir {
namespace fusion {

struct TailInst {
  enum Kind { kOther = 0, kConvF32, kAdd, kAddRelu, kMaxPool, kCalibF2I };
  Kind kind{kOther};
  std::vector<std::string> inputs;  // data inputs in order (add: X, Y; every other kind reads inputs[0]); kOther: all of them
  std::string output;
  float calib_scale{0.f};           // kCalibF2I
  // ---- filled by MatchConvTails
  bool dead{false};                 // the instruction disappears
  std::string residual;             // kConvF32: fp32 operand of the fused add ("" = none)
  bool residual_relu{false};
  std::string calib_out;            // kConvF32: int8 tensor of the fused calib ("" = none)
  float fused_calib_scale{1.f};
  bool drop_f32{false};             // kConvF32: the fp32 output has no reader left
  bool pool_int8{false};            // kMaxPool: runs on the int8 tensor now (inputs[0] / output were renamed)
};

// Rewrites `prog` (topological order) in place: sets the fused fields, renames outputs / inputs, marks dead instructions.
inline void MatchConvTails(std::vector<TailInst>* prog_io) {
  std::vector<TailInst>& p = *prog_io;
  auto uses = [&](const std::string& v) {
    int n = 0;
    for (const TailInst& t : p) {
      if (t.dead) continue;
      for (const std::string& in : t.inputs) n += in == v;
      n += t.residual == v;
    }
    return n;
  };
  auto producer = [&](const std::string& v) {
    for (size_t i = 0; i < p.size(); ++i)
      if (!p[i].dead && (p[i].output == v || (!p[i].calib_out.empty() && p[i].calib_out == v))) return static_cast<int>(i);
    return -1;
  };
  auto open_conv = [&](int i) { return i >= 0 && p[i].kind == TailInst::kConvF32 && p[i].calib_out.empty() && !p[i].drop_f32; };
  // (A)
  for (size_t i = 0; i < p.size(); ++i) {
    if (p[i].dead || (p[i].kind != TailInst::kAdd && p[i].kind != TailInst::kAddRelu) || p[i].inputs.size() != 2) continue;
    const int pa = producer(p[i].inputs[0]), pb = producer(p[i].inputs[1]);
    int conv = -1, other = -1;
    if (open_conv(pb) && pb > pa && p[pb].residual.empty() && uses(p[pb].output) == 1) conv = pb, other = 0;
    else if (open_conv(pa) && pa > pb && p[pa].residual.empty() && uses(p[pa].output) == 1) conv = pa, other = 1;
    if (conv < 0) continue;
    p[conv].residual = p[i].inputs[other];
    p[conv].residual_relu = p[i].kind == TailInst::kAddRelu;
    p[conv].output = p[i].output;  // the conv now writes the sum
    p[i].dead = true;
  }
  // (C)
  for (size_t i = 0; i < p.size(); ++i) {
    if (p[i].dead || p[i].kind != TailInst::kCalibF2I) continue;
    const int pp = producer(p[i].inputs[0]);
    if (pp < 0 || p[pp].kind != TailInst::kMaxPool) continue;
    const int pc = producer(p[pp].inputs[0]);
    if (!open_conv(pc) || uses(p[pc].output) != 1 || uses(p[pp].output) != 1) continue;
    p[pc].calib_out = p[pc].output + "/precision_trans";
    p[pc].fused_calib_scale = p[i].calib_scale;
    p[pc].drop_f32 = true;
    p[pp].inputs[0] = p[pc].calib_out;
    p[pp].output = p[i].output;
    p[pp].pool_int8 = true;
    p[i].dead = true;
  }
  // (B)
  for (size_t i = 0; i < p.size(); ++i) {
    if (p[i].dead || p[i].kind != TailInst::kCalibF2I) continue;
    const int pc = producer(p[i].inputs[0]);
    if (!open_conv(pc) || p[pc].output != p[i].inputs[0]) continue;
    p[pc].calib_out = p[i].output;
    p[pc].fused_calib_scale = p[i].calib_scale;
    p[i].dead = true;
    p[pc].drop_f32 = uses(p[pc].output) == 0;
  }
}

}  // namespace fusion
}  // namespace mir
}  // namespace lite
}  // namespace paddle

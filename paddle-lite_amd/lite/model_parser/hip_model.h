// hip_model.h — quantised-model ingestion for the kHIP target.
//
// The reference loads a PaddleSlim QAT model (protobuf / naive buffer; lite/model_parser, out of scope here) and turns
// it into int8 kernels' parameters with MIR passes.  This loader reads the SAME information from a small self-contained
// container and restates the weight-side semantics of those passes, in their order (lite/core/optimizer.h):
//   1. DeleteQuantOpFuser   lite/core/mir/fusion/quant_dequant_op_fuser.cc:58-92   fake_quantize_*: the consumers get
//      input_scale = OutScale[0] / 127 and read the un-quantised tensor;
//   2. DequantOpFuser       :132-203   conv2d | depthwise_conv2d | mul + fake_dequantize_max_abs: enable_int8,
//      weight_scale[i] = (127*127) / max_range / 127 for every output channel (conv: dims[0], mul: dims[1]), weights cast
//      float -> int8 (they are stored as floats on the integer grid);
//   3. ConvBNFuser          lite/core/mir/fusion/conv_bn_fuser.cc:100-245   alpha = scale / sqrt(var + eps),
//      beta = -mean * alpha; int8: weight_scale[i] *= |alpha[i]|, int8 filter row i negated when alpha[i] < 0;
//      bias = bn_bias (+ alpha * conv_bias) + beta;
//   4. ConvActivationFuser  conv_activation_fuse_pass.cc: relu / relu6 (threshold) / leaky_relu (alpha) into the conv;
//   5. FcFuser              fc_fuser.cc: mul + elementwise_add(persistable bias) -> fc;
//   6. ElementwiseActivationFuser  elementwise_add + relu -> fusion_elementwise_add_activation.
// The result is the op list lite/api/graph_builder.h lowers (kernel pick, calib / io_copy placement, kHIP fusions).
//
// Container (little endian): "PLHIPM01", u32 n_tensors, u32 n_ops, tensors, ops.
//   tensor: str name, u8 dtype (0 = fp32, 1 = int8), u8 ndim, i64 dims[ndim], u64 nbytes, data, zero pad to 8 bytes
//   op:     str type, u16 n_in { str arg, str var }, u16 n_out { str arg, str var }, u16 n_attr { str name, u8 kind, .. }
//           kind 0: i32   1: f32   2: u16 n, i32[n]   3: str   4: u16 n, f32[n]
//   str = u16 length + bytes.   Op and argument names are Paddle's (conv2d: Input / Filter / Output, batch_norm: X / Scale /
//   Bias / Mean / Variance / Y, mul: X / Y / Out, fake_quantize_*: X / OutScale / Out, ...).
#pragma once
#include <map>
#include <string>
#include <vector>

#include "lite/api/graph_builder.h"

namespace paddle {
namespace lite {
namespace model_parser {

struct RawTensor {
  int dtype{0};  // 0 fp32, 1 int8
  std::vector<int64_t> dims;
  std::vector<uint8_t> data;
  int64_t numel() const {
    int64_t n = 1;
    for (auto d : dims) n *= d;
    return n;
  }
  const float* f32() const { return reinterpret_cast<const float*>(data.data()); }
};

struct RawOp {
  std::string type;
  std::map<std::string, std::string> in, out;  // argument -> variable
  std::map<std::string, int> iattr;
  std::map<std::string, float> fattr;
  std::map<std::string, std::vector<int>> ivattr;
  std::map<std::string, std::string> sattr;
  bool dead{false};
};

struct RawModel {
  std::map<std::string, RawTensor> tensors;  // persistable variables
  std::vector<RawOp> ops;                    // topological order
};

RawModel ParseContainer(const std::vector<uint8_t>& bytes);
// Runs passes 1-6 and appends the resulting ops (plus feeds / fetches) to `g`.  `batch` sizes the feed.
void BuildGraph(RawModel* m, int batch, GraphBuilder* g);

}  // namespace model_parser
}  // namespace lite
}  // namespace paddle

// graph_builder.cc — see graph_builder.h.
#include "lite/api/graph_builder.h"

#include <cstdio>
#include <set>

namespace paddle {
namespace lite {

void GraphBuilder::Feed(const std::string& name, const std::vector<int64_t>& dims, PrecisionType prec) {
  feeds_.push_back({name, dims, prec});
}

GraphOp& GraphBuilder::Add(const std::string& type, const std::vector<std::string>& inputs, const std::string& output) {
  ops_.emplace_back();
  GraphOp& op = ops_.back();
  op.type = type;
  op.inputs = inputs;
  op.output = output;
  return op;
}

std::vector<GraphBuilder::Step> GraphBuilder::Schedule() {
  // ---- who consumes what (fetch counts as a consumer that is not enable_int8)
  std::map<std::string, std::vector<int>> consumers;
  std::map<std::string, int> producer;
  std::set<std::string> known;
  for (auto& f : feeds_) known.insert(f.name);
  for (size_t i = 0; i < ops_.size(); ++i) {
    for (auto& in : ops_[i].inputs) {
      CHECK(known.count(in)) << ops_[i].type << ": input " << in << " is not produced by an earlier op or feed";
      consumers[in].push_back(static_cast<int>(i));
    }
    CHECK(!known.count(ops_[i].output)) << "variable " << ops_[i].output << " is written twice";
    known.insert(ops_[i].output);
    producer[ops_[i].output] = static_cast<int>(i);
  }
  std::set<std::string> fetched(fetches_.begin(), fetches_.end());
  for (auto& f : fetches_) CHECK(known.count(f)) << "fetch of unknown variable " << f;

  // ---- pass 1: static_kernel_pick_pass.cc:92-165
  std::vector<bool> int8_out(ops_.size(), false);
  std::vector<float> out_scale(ops_.size(), 1.f);
  for (size_t i = 0; i < ops_.size(); ++i) {
    if (!ops_[i].enable_int8) continue;
    const auto it = consumers.find(ops_[i].output);
    bool all_int8 = it != consumers.end() && !it->second.empty() && !fetched.count(ops_[i].output);
    if (all_int8)
      for (int c : it->second) all_int8 = all_int8 && ops_[c].enable_int8;
    int8_out[i] = all_int8;
    if (all_int8) out_scale[i] = ops_[it->second.front()].conv.input_scale;  // :118-121 (first adjacent op)
  }

  // ---- passes 2 + 3 while walking the ops in order
  std::map<std::string, PrecisionType> prec;   // precision of every device variable
  std::map<std::string, std::string> cast_of;  // type_precision_cast_pass's cast_nodes
  std::vector<Step> steps;
  for (auto& f : feeds_) {
    Step s;
    s.kind = "io_copy_h2d";
    s.in = f.name;
    s.out = f.name + "/target_trans";
    steps.push_back(s);
    prec[f.name] = f.prec;
  }
  auto dev_name = [&](const std::string& v) {
    for (auto& f : feeds_)
      if (f.name == v) return v + "/target_trans";
    return v;
  };
  for (size_t i = 0; i < ops_.size(); ++i) {
    const GraphOp& op = ops_[i];
    const PrecisionType want = op.enable_int8 ? PRECISION(kInt8) : PRECISION(kFloat);
    Step s;
    s.op = static_cast<int>(i);
    s.kind = "op";
    for (auto& in : op.inputs) {
      std::string use = dev_name(in);
      if (prec[in] != want) {
        auto c = cast_of.find(in);
        if (c == cast_of.end()) {
          Step cs;
          cs.in = use;
          cs.out = in + "/precision_trans";
          if (want == PRECISION(kInt8)) {
            cs.kind = "calib_f2i";
            cs.scale = op.conv.input_scale;  // InferScale case 1
          } else {
            cs.kind = "calib_i2f";
            const auto p = producer.find(in);
            CHECK(p != producer.end()) << "int8 feed " << in << " consumed by an fp32 op: no scale to dequantise with";
            cs.scale = out_scale[p->second];  // InferScale case 2
          }
          steps.push_back(cs);
          c = cast_of.emplace(in, cs.out).first;
        }
        use = c->second;
      }
      s.op_inputs.push_back(use);
    }
    s.out = op.output;
    s.int8_out = int8_out[i];
    s.out_scale = out_scale[i];
    steps.push_back(s);
    prec[op.output] = (op.enable_int8 && int8_out[i]) ? PRECISION(kInt8) : PRECISION(kFloat);
  }
  for (auto& f : fetches_) {
    Step s;
    s.kind = "io_copy_d2h";
    s.in = dev_name(f);
    s.out = f + "/host";
    steps.push_back(s);
  }
  return steps;
}

std::vector<std::string> GraphBuilder::Plan() {
  std::vector<std::string> lines;
  char buf[64];
  for (auto& s : Schedule()) {
    std::string l;
    if (s.kind == "op") {
      const GraphOp& op = ops_[s.op];
      l = op.type;
      if (op.enable_int8) {
        const bool fc = op.type == "fc";
        l += s.int8_out ? (fc ? "/int8out" : "/int8_out") : (fc ? "/fp32out" : "/fp32_out");
      } else {
        l += "/def";
      }
      l += " in=";
      for (size_t i = 0; i < s.op_inputs.size(); ++i) l += (i ? "," : "") + s.op_inputs[i];
      l += " out=" + s.out;
      if (op.enable_int8 && s.int8_out) {
        snprintf(buf, sizeof buf, " oscale=%.9g", s.out_scale);
        l += buf;
      }
    } else {
      l = s.kind == "io_copy_h2d" ? "io_copy/host_to_device"
          : s.kind == "io_copy_d2h" ? "io_copy/device_to_host"
          : s.kind == "calib_f2i" ? "calib/fp32_to_int8" : "calib/int8_to_fp32";
      l += " in=" + s.in + " out=" + s.out;
      if (s.kind[0] == 'c') {
        snprintf(buf, sizeof buf, " scale=%.9g", s.scale);
        l += buf;
      }
    }
    lines.push_back(l);
  }
  return lines;
}

std::vector<std::string> GraphBuilder::Lower(HipPredictor* pred) {
  for (auto& f : feeds_) pred->AddFeed(f.name, f.dims, f.prec);
  std::vector<std::string> outs;
  for (auto& s : Schedule()) {
    if (s.kind == "io_copy_h2d") {
      pred->AddIoCopy(s.in, s.out, true);
    } else if (s.kind == "io_copy_d2h") {
      pred->AddIoCopy(s.in, s.out, false);
      outs.push_back(s.out);
    } else if (s.kind == "calib_f2i" || s.kind == "calib_i2f") {
      pred->AddCalib(s.in, s.out, s.scale, s.kind == "calib_f2i");
    } else {
      GraphOp& op = ops_[s.op];
      if (op.type == "conv2d" || op.type == "depthwise_conv2d") {
        CHECK(op.enable_int8) << "kHIP has int8 conv kernels only";
        ConvAttrs a = op.conv;
        a.int8_out = s.int8_out;
        a.output_scale = s.int8_out ? s.out_scale : 1.f;
        pred->AddConv(op.type, s.op_inputs[0], s.out, op.w.data(), op.w_dims, op.has_bias ? op.bias.data() : nullptr, a);
      } else if (op.type == "fc") {
        CHECK(op.enable_int8) << "kHIP has int8 fc kernels only";
        pred->AddFc(s.op_inputs[0], s.out, op.w.data(), static_cast<int>(op.w_dims[0]), static_cast<int>(op.w_dims[1]),
                    op.has_bias ? op.bias.data() : nullptr, op.conv.input_scale, op.conv.weight_scale,
                    s.int8_out ? s.out_scale : 1.f, s.int8_out, op.fc_relu);
      } else if (op.type == "pool2d") {
        pred->AddPool(s.op_inputs[0], s.out, op.pooling_type, op.ksize, op.pool_strides, op.pool_paddings,
                      op.global_pooling, op.exclusive, op.ceil_mode);
      } else if (op.type == "elementwise_add") {
        pred->AddElementwiseAdd(s.op_inputs[0], s.op_inputs[1], s.out, "");
      } else if (op.type == "fusion_elementwise_add_activation") {
        pred->AddElementwiseAdd(s.op_inputs[0], s.op_inputs[1], s.out, op.act_type);
      } else if (op.type == "softmax") {
        pred->AddSoftmax(s.op_inputs[0], s.out);
      } else {
        LOG(FATAL) << "GraphBuilder: no kHIP kernel for op type " << op.type;
      }
    }
  }
  return outs;
}

}  // namespace lite
}  // namespace paddle

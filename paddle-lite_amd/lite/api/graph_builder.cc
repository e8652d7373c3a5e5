// graph_builder.cc — see graph_builder.h.
#include "lite/api/graph_builder.h"

#include <string.h>

#include "lite/core/mir/fusion/hip_conv_tail_matcher.h"
#include "plhip.h"

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

void GraphBuilder::FuseSteps(std::vector<Step>* steps_io) {
  std::vector<Step>& st = *steps_io;
  std::vector<bool> dead(st.size(), false);
  // conv_op.h:149-161: 2-element paddings mean {top = bottom, left = right}
  auto pad4 = [](const std::vector<int>& p) {
    return p.size() == 2 ? std::vector<int>{p[0], p[0], p[1], p[1]} : p;
  };
  auto uses = [&](const std::string& v) {
    int n = 0;
    for (size_t i = 0; i < st.size(); ++i) {
      if (dead[i]) continue;
      if (st[i].kind == "op") {
        for (auto& in : st[i].op_inputs) n += in == v;
        n += st[i].res == v;
      } else {
        n += st[i].in == v;
      }
    }
    return n;
  };
  // (A) (C) (B): the conv-tail patterns, matched by the SAME code a Paddle-Lite tree runs as a mir pass
  // (lite/core/mir/fusion/hip_conv_tail_matcher.h; patches/0006 carries it with its SSAGraph adapter)
  {
    using mir::fusion::TailInst;
    std::vector<TailInst> prog(st.size());
    for (size_t i = 0; i < st.size(); ++i) {
      TailInst& t = prog[i];
      t.output = st[i].out;
      if (st[i].kind == "op") {
        const GraphOp& op = ops_[st[i].op];
        t.inputs = st[i].op_inputs;
        if (op.type == "conv2d" && op.enable_int8 && !st[i].int8_out) t.kind = TailInst::kConvF32;
        else if (op.type == "elementwise_add") t.kind = TailInst::kAdd;
        else if (op.type == "fusion_elementwise_add_activation" && op.act_type == "relu") t.kind = TailInst::kAddRelu;
        else if (op.type == "pool2d" && op.pooling_type == "max") t.kind = TailInst::kMaxPool;
      } else {
        t.inputs = {st[i].in};
        if (st[i].kind == "calib_f2i") {
          t.kind = TailInst::kCalibF2I;
          t.calib_scale = st[i].scale;
        }
      }
    }
    mir::fusion::MatchConvTails(&prog);
    for (size_t i = 0; i < st.size(); ++i) {
      const TailInst& t = prog[i];
      dead[i] = t.dead;
      st[i].out = t.output;
      if (st[i].kind == "op") st[i].op_inputs = t.inputs;
      st[i].res = t.residual;
      st[i].res_relu = t.residual_relu;
      st[i].calib_out = t.calib_out;
      if (!t.calib_out.empty()) st[i].calib_scale = t.fused_calib_scale;
      st[i].drop_f32 = t.drop_f32;
      st[i].pool_int8 = t.pool_int8;
    }
  }
  // (D) depthwise_conv2d[int8_out] whose only consumer is a plain 1x1 conv (no tail of its own) takes it over.  Mode 2 (default):
  // only where the fused kernel takes the pair, which needs the depthwise conv's input shape: propagated from the feeds through
  // conv / calib / elementwise ops (anything else: shape unknown, no fusion)
  std::map<std::string, std::vector<int64_t>> shape;
  if (fuse_dwpw_ == 2) {
    for (auto& f : feeds_) shape[f.name] = f.dims;
    for (size_t i = 0; i < st.size(); ++i) {
      if (dead[i]) continue;
      if (st[i].kind != "op") {  // io_copy / calib: same shape
        auto it = shape.find(st[i].in);
        if (it != shape.end()) shape[st[i].out] = it->second;
        continue;
      }
      const GraphOp& op = ops_[st[i].op];
      if (st[i].op_inputs.empty()) continue;
      auto it = shape.find(st[i].op_inputs[0]);
      if (it == shape.end() || it->second.size() != 4) continue;
      const std::vector<int64_t> in = it->second;
      std::vector<int64_t> o;
      const std::vector<int> pd = pad4(op.conv.paddings);
      if ((op.type == "conv2d" || op.type == "depthwise_conv2d") && op.w_dims.size() == 4 && pd.size() == 4 &&
          op.conv.strides.size() == 2 && op.conv.dilations.size() == 2 && op.conv.padding_algorithm.empty()) {
        const int64_t keh = op.conv.dilations[0] * (op.w_dims[2] - 1) + 1, kew = op.conv.dilations[1] * (op.w_dims[3] - 1) + 1;
        o = {in[0], op.w_dims[0], (in[2] + pd[0] + pd[1] - keh) / op.conv.strides[0] + 1,
             (in[3] + pd[2] + pd[3] - kew) / op.conv.strides[1] + 1};
      } else if (op.type == "elementwise_add" || op.type == "fusion_elementwise_add_activation") {
        o = in;
      }
      if (o.empty()) continue;
      shape[st[i].out] = o;
      if (!st[i].calib_out.empty()) shape[st[i].calib_out] = o;
    }
  }
  if (fuse_dwpw_) {
    for (size_t i = 0; i < st.size(); ++i) {
      if (dead[i] || st[i].kind != "op" || ops_[st[i].op].type != "depthwise_conv2d" || !st[i].int8_out || st[i].pw_op >= 0) continue;
      if (uses(st[i].out) != 1) continue;
      {  // a TRUE depthwise conv only (channel multiplier 1): anything else stays two instructions instead of failing a CHECK later
        const GraphOp& dwo = ops_[st[i].op];
        if (dwo.w_dims.size() != 4 || dwo.w_dims[1] != 1 || dwo.w_dims[0] != dwo.conv.groups) continue;
      }
      int j = -1;
      for (size_t t = 0; t < st.size(); ++t)
        if (!dead[t] && st[t].kind == "op" && !st[t].op_inputs.empty() && st[t].op_inputs[0] == st[i].out) j = static_cast<int>(t);
      if (j < 0) continue;
      const GraphOp& c = ops_[st[j].op];
      if (c.type != "conv2d" || !c.enable_int8 || c.w_dims.size() != 4 || c.w_dims[2] != 1 || c.w_dims[3] != 1 || c.conv.groups != 1 ||
          c.conv.strides != std::vector<int>({1, 1}) || c.conv.dilations != std::vector<int>({1, 1}))
        continue;
      bool pad0 = true;
      for (int v : c.conv.paddings) pad0 = pad0 && v == 0;
      if (!pad0 || !st[j].res.empty() || !st[j].calib_out.empty() || st[j].drop_f32) continue;
      plhip_conv_desc d;
      memset(&d, 0, sizeof(d));
      if (fuse_dwpw_ == 2) {
        const GraphOp& dwo = ops_[st[i].op];
        auto it = st[i].op_inputs.empty() ? shape.end() : shape.find(st[i].op_inputs[0]);
        const std::vector<int> dpd = pad4(dwo.conv.paddings);
        if (it == shape.end() || it->second.size() != 4 || dpd.size() != 4 || !dwo.conv.padding_algorithm.empty()) continue;
        d.n = static_cast<int>(it->second[0]); d.cin = static_cast<int>(it->second[1]);
        d.h = static_cast<int>(it->second[2]); d.w = static_cast<int>(it->second[3]);
        d.cout = static_cast<int>(dwo.w_dims[0]); d.kh = static_cast<int>(dwo.w_dims[2]); d.kw = static_cast<int>(dwo.w_dims[3]);
        for (int q = 0; q < 4; ++q) d.pad[q] = dpd[q];
        d.stride[0] = dwo.conv.strides[0]; d.stride[1] = dwo.conv.strides[1];
        d.dil[0] = dwo.conv.dilations[0]; d.dil[1] = dwo.conv.dilations[1];
        d.groups = dwo.conv.groups;
        if (!plhip_dwpw_fused_supported(&d, static_cast<int>(c.w_dims[0]), st[j].int8_out ? PLHIP_OUT_I8 : PLHIP_OUT_F32)) continue;
      }
      st[i].pw_op = st[j].op;
      st[i].pw_int8_out = st[j].int8_out;
      st[i].pw_out_scale = st[j].out_scale;
      st[i].via = st[i].out;
      st[i].out = st[j].out;
      dead[j] = true;
      // (E) ... and the global average pool2d that is the only reader of that conv's fp32 output, where the fused kernel writes
      // the plane average itself (PLHIP_OUT_F32_GAP): MobileNetV1's pw14 -> pool
      if (fuse_dwpw_ == 2 && !st[i].pw_int8_out && uses(st[i].out) == 1) {
        int pj = -1;
        for (size_t t = 0; t < st.size(); ++t)
          if (!dead[t] && st[t].kind == "op" && !st[t].op_inputs.empty() && st[t].op_inputs[0] == st[i].out) pj = static_cast<int>(t);
        if (pj >= 0) {
          const GraphOp& po = ops_[st[pj].op];
          if (po.type == "pool2d" && po.pooling_type == "avg" && po.global_pooling && !st[pj].pool_int8 &&
              plhip_dwpw_fused_supported(&d, static_cast<int>(c.w_dims[0]), PLHIP_OUT_F32_GAP)) {
            st[i].pw_pool = true;
            st[i].via_pw = st[i].out;
            st[i].out = st[pj].out;
            dead[pj] = true;
          }
        }
      }
    }
  }
  // (F) a calib[fp32_to_int8] whose only reader is a conv2d that quantises while it stages its rows (plhip_conv2d_calib_supported:
  // the 3x3 stride-2 stem) is taken over by that conv: the head of the MobileNet programs, the int8 image is never written
  if (fuse_dwpw_ == 2) {
    for (size_t i = 0; i < st.size(); ++i) {
      if (dead[i] || st[i].kind != "calib_f2i" || uses(st[i].out) != 1) continue;
      int j = -1;
      for (size_t t = 0; t < st.size(); ++t)
        if (!dead[t] && st[t].kind == "op" && !st[t].op_inputs.empty() && st[t].op_inputs[0] == st[i].out) j = static_cast<int>(t);
      if (j < 0) continue;
      const GraphOp& c = ops_[st[j].op];
      const std::vector<int> cpd = pad4(c.conv.paddings);
      if (c.type != "conv2d" || !c.enable_int8 || c.w_dims.size() != 4 || cpd.size() != 4 || c.conv.strides.size() != 2 ||
          c.conv.dilations.size() != 2 || !c.conv.padding_algorithm.empty() || !st[j].res.empty() || !st[j].calib_out.empty() ||
          st[j].pw_op >= 0)
        continue;
      auto it = shape.find(st[i].in);
      if (it == shape.end() || it->second.size() != 4) continue;
      plhip_conv_desc d;
      memset(&d, 0, sizeof(d));
      d.n = static_cast<int>(it->second[0]); d.cin = static_cast<int>(it->second[1]);
      d.h = static_cast<int>(it->second[2]); d.w = static_cast<int>(it->second[3]);
      d.cout = static_cast<int>(c.w_dims[0]); d.kh = static_cast<int>(c.w_dims[2]); d.kw = static_cast<int>(c.w_dims[3]);
      for (int q = 0; q < 4; ++q) d.pad[q] = cpd[q];
      d.stride[0] = c.conv.strides[0]; d.stride[1] = c.conv.strides[1];
      d.dil[0] = c.conv.dilations[0]; d.dil[1] = c.conv.dilations[1];
      d.groups = c.conv.groups;
      if (!plhip_conv2d_calib_supported(&d)) continue;
      st[j].in_calib_scale = st[i].scale;
      st[j].via_in = st[i].out;
      st[j].op_inputs[0] = st[i].in;
      dead[i] = true;
    }
  }
  std::vector<Step> kept;
  for (size_t i = 0; i < st.size(); ++i)
    if (!dead[i]) kept.push_back(st[i]);
  st.swap(kept);
}

std::vector<std::string> GraphBuilder::Plan() {
  std::vector<std::string> lines;
  char buf[64];
  auto steps = Schedule();
  if (fuse_) FuseSteps(&steps);
  for (auto& s : steps) {
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
      if (s.in_calib_scale > 0.f) {
        snprintf(buf, sizeof buf, " in_scale=%.9g", s.in_calib_scale);
        l += " +calib_in=" + s.via_in + buf;
      }
      if (!s.res.empty()) l += std::string(" +add=") + s.res + (s.res_relu ? " +relu" : "");
      if (!s.calib_out.empty()) {
        snprintf(buf, sizeof buf, " scale=%.9g", s.calib_scale);
        l += " +calib=" + s.calib_out + buf;
      }
      if (s.drop_f32) l += " -f32";
      if (s.pool_int8) l += " int8";
      if (s.pw_op >= 0) {
        l += std::string(" +pw=conv2d/") + (s.pw_int8_out ? "int8_out" : "fp32_out") + " via=" + s.via;
        if (s.pw_int8_out) {
          snprintf(buf, sizeof buf, " pw_oscale=%.9g", s.pw_out_scale);
          l += buf;
        }
        if (s.pw_pool) l += " +pool=avg/global pw_out=" + s.via_pw;
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
  auto steps = Schedule();
  if (fuse_) FuseSteps(&steps);
  for (auto& s : steps) {
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
        a.residual = s.res;
        a.residual_relu = s.res_relu;
        a.calib_out = s.calib_out;
        a.calib_scale = s.calib_scale;
        a.drop_fp32 = s.drop_f32;
        a.in_calib_scale = s.in_calib_scale;
        if (s.pw_op >= 0) {
          const GraphOp& c = ops_[s.pw_op];
          a.pw_w = c.w.data();
          a.pw_w_dims = c.w_dims;
          a.pw_bias = c.has_bias ? c.bias.data() : nullptr;
          a.pw_weight_scale = c.conv.weight_scale;
          a.pw_output_scale = s.pw_int8_out ? s.pw_out_scale : 1.f;
          a.pw_int8_out = s.pw_int8_out;
          a.pw_act = c.conv.act;
          a.pw_act_coef = c.conv.act_coef;
          a.pw_pool = s.pw_pool;
        }
        pred->AddConv(op.type, s.op_inputs[0], s.out, op.w.data(), op.w_dims, op.has_bias ? op.bias.data() : nullptr, a);
      } else if (op.type == "fc") {
        CHECK(op.enable_int8) << "kHIP has int8 fc kernels only";
        pred->AddFc(s.op_inputs[0], s.out, op.w.data(), static_cast<int>(op.w_dims[0]), static_cast<int>(op.w_dims[1]),
                    op.has_bias ? op.bias.data() : nullptr, op.conv.input_scale, op.conv.weight_scale,
                    s.int8_out ? s.out_scale : 1.f, s.int8_out, op.fc_relu);
      } else if (op.type == "pool2d") {
        pred->AddPool(s.op_inputs[0], s.out, op.pooling_type, op.ksize, op.pool_strides, op.pool_paddings,
                      op.global_pooling, op.exclusive, op.ceil_mode, s.pool_int8);
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

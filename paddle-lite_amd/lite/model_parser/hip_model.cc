// hip_model.cc — see hip_model.h.
#include "lite/model_parser/hip_model.h"

#include <cmath>
#include <cstring>

namespace paddle {
namespace lite {
namespace model_parser {

namespace {
struct Reader {
  const std::vector<uint8_t>& b;
  size_t p{0};
  explicit Reader(const std::vector<uint8_t>& bytes) : b(bytes) {}
  // n <= size - p (p <= size always): no wrap-around for sizes read from the file
  void need(size_t n) { CHECK(p <= b.size() && n <= b.size() - p) << "model container truncated at byte " << p; }
  template <typename T>
  T get() {
    need(sizeof(T));
    T v;
    std::memcpy(&v, b.data() + p, sizeof(T));
    p += sizeof(T);
    return v;
  }
  std::string str() {
    const uint16_t n = get<uint16_t>();
    need(n);
    std::string s(reinterpret_cast<const char*>(b.data() + p), n);
    p += n;
    return s;
  }
};

const std::string& arg(const std::map<std::string, std::string>& m, const std::string& k, const RawOp& op) {
  auto it = m.find(k);
  CHECK(it != m.end()) << op.type << ": missing argument " << k;
  return it->second;
}
bool is_conv(const std::string& t) { return t == "conv2d" || t == "depthwise_conv2d"; }
// an fp32 tensor of the model with at least `n` elements (n < 0: exactly one or more), by name: every read of f32() below
// goes through this, so a malformed model (wrong dtype, too few elements, missing tensor) is an error, not a stray read
const RawTensor& f32_tensor(const RawModel& m, const std::string& name, int64_t n, const char* what) {
  const auto it = m.tensors.find(name);
  CHECK(it != m.tensors.end()) << what << ": tensor " << name << " missing";
  CHECK_EQ(it->second.dtype, 0) << what << ": tensor " << name << " must be fp32";
  CHECK(n < 0 ? it->second.numel() >= 1 : it->second.numel() == n) << what << ": tensor " << name << " has " << it->second.numel()
                                                                    << " elements, expected " << n;
  return it->second;
}
}  // namespace

RawModel ParseContainer(const std::vector<uint8_t>& bytes) {
  Reader r(bytes);
  r.need(8);
  CHECK(std::memcmp(bytes.data(), "PLHIPM01", 8) == 0) << "not a PLHIPM01 model container";
  r.p = 8;
  const uint32_t nt = r.get<uint32_t>(), no = r.get<uint32_t>();
  RawModel m;
  for (uint32_t i = 0; i < nt; ++i) {
    const std::string name = r.str();
    RawTensor t;
    t.dtype = r.get<uint8_t>();
    CHECK(t.dtype == 0 || t.dtype == 1) << "tensor " << name << ": unknown dtype " << t.dtype;
    const int nd = r.get<uint8_t>();
    CHECK(nd >= 1 && nd <= 6) << "tensor " << name << ": rank " << nd << " outside 1..6";
    uint64_t numel = 1;
    for (int d = 0; d < nd; ++d) {
      const int64_t dim = r.get<int64_t>();
      CHECK(dim > 0 && dim <= (int64_t(1) << 31)) << "tensor " << name << ": dim " << d << " = " << dim << " is not a positive 32-bit size";
      CHECK(numel <= (uint64_t(1) << 40) / static_cast<uint64_t>(dim)) << "tensor " << name << ": element count overflows";
      numel *= static_cast<uint64_t>(dim);
      t.dims.push_back(dim);
    }
    const uint64_t nb = r.get<uint64_t>();
    CHECK_EQ(nb, numel * (t.dtype == 0 ? 4 : 1)) << "tensor " << name << ": byte count does not match dims";
    r.need(nb);
    t.data.assign(bytes.begin() + r.p, bytes.begin() + r.p + nb);
    r.p += nb;
    r.p = (r.p + 7) & ~size_t(7);
    CHECK(!m.tensors.count(name)) << "tensor " << name << " appears twice";
    m.tensors.emplace(name, std::move(t));
  }
  for (uint32_t i = 0; i < no; ++i) {
    RawOp op;
    op.type = r.str();
    const int ni = r.get<uint16_t>();
    for (int k = 0; k < ni; ++k) {
      const std::string a = r.str();
      op.in[a] = r.str();
    }
    const int nout = r.get<uint16_t>();
    for (int k = 0; k < nout; ++k) {
      const std::string a = r.str();
      op.out[a] = r.str();
    }
    const int na = r.get<uint16_t>();
    for (int k = 0; k < na; ++k) {
      const std::string name = r.str();
      const int kind = r.get<uint8_t>();
      if (kind == 0) op.iattr[name] = r.get<int32_t>();
      else if (kind == 1) op.fattr[name] = r.get<float>();
      else if (kind == 2) {
        const int n = r.get<uint16_t>();
        auto& v = op.ivattr[name];
        for (int j = 0; j < n; ++j) v.push_back(r.get<int32_t>());
      } else if (kind == 3) op.sattr[name] = r.str();
      else if (kind == 4) {
        const int n = r.get<uint16_t>();
        for (int j = 0; j < n; ++j) (void)r.get<float>();
      } else LOG(FATAL) << "op " << op.type << ": unknown attribute kind " << kind;
    }
    m.ops.push_back(std::move(op));
  }
  return m;
}

void BuildGraph(RawModel* mp, int batch, GraphBuilder* g) {
  RawModel& m = *mp;
  auto& ops = m.ops;
  const int range = 127;  // bit_length 8: (1 << 7) - 1
  // consumers of a variable (live ops only)
  auto consumers = [&](const std::string& v) {
    std::vector<int> c;
    for (size_t i = 0; i < ops.size(); ++i) {
      if (ops[i].dead) continue;
      for (auto& kv : ops[i].in)
        if (kv.second == v) c.push_back(static_cast<int>(i));
    }
    return c;
  };
  auto rename_input = [&](const std::string& from, const std::string& to) {
    for (auto& o : ops)
      if (!o.dead)
        for (auto& kv : o.in)
          if (kv.second == from) kv.second = to;
  };
  std::map<std::string, float> act_scale;  // variable -> input scale of its quantised consumers

  // ---- 1. DeleteQuantOpFuser (quant_dequant_op_fuser.cc:58-92)
  for (auto& o : ops) {
    if (o.dead || o.type.rfind("fake_quantize", 0) != 0) continue;
    const int bits = o.iattr.count("bit_length") ? o.iattr["bit_length"] : 8;
    CHECK_EQ(bits, 8) << "only 8-bit quantisation is supported";
    const float scale_value = f32_tensor(m, arg(o.out, "OutScale", o), -1, "fake_quantize OutScale").f32()[0] / range;
    const std::string x = arg(o.in, "X", o), out = arg(o.out, "Out", o);
    act_scale[x] = scale_value;
    rename_input(out, x);
    o.dead = true;
  }
  // ---- 2. DequantOpFuser (:132-203)
  struct Q {
    std::vector<float> weight_scale;
    float input_scale{1.f};
    int act{0};
    float act_coef{0.f};
    std::vector<float> bias;
    bool has_bias{false};
    std::vector<int8_t> w;
  };
  std::map<int, Q> q;  // op index -> quantised state
  for (size_t i = 0; i < ops.size(); ++i) {
    auto& o = ops[i];
    if (o.dead || !(is_conv(o.type) || o.type == "mul")) continue;
    const bool conv = is_conv(o.type);
    const std::string outv = arg(o.out, conv ? "Output" : "Out", o);
    const auto cs = consumers(outv);
    if (cs.size() != 1 || ops[cs[0]].type != "fake_dequantize_max_abs") continue;  // not a quantised op
    RawOp& dq = ops[cs[0]];
    const float max_range = dq.fattr.count("max_range") ? dq.fattr["max_range"] : 0.f;
    const float whole_weight_scale = static_cast<float>(range * range) / max_range / range;  // :146-147, as written
    const std::string wname = arg(o.in, conv ? "Filter" : "Y", o);
    const auto wt = m.tensors.find(wname);
    CHECK(wt != m.tensors.end() && wt->second.dtype == 0) << o.type << ": fp32 weight tensor " << wname << " missing";
    CHECK_EQ(wt->second.dims.size(), conv ? 4UL : 2UL) << o.type << ": weight tensor " << wname << " must have rank " << (conv ? 4 : 2);
    CHECK(dq.fattr.count("max_range") && max_range > 0.f) << "fake_dequantize_max_abs: max_range missing or not positive";
    Q st;
    const int n_scale = static_cast<int>(conv ? wt->second.dims[0] : wt->second.dims[1]);  // :159-174
    st.weight_scale.assign(n_scale, whole_weight_scale);
    st.w.resize(static_cast<size_t>(wt->second.numel()));
    for (size_t k = 0; k < st.w.size(); ++k) st.w[k] = static_cast<int8_t>(wt->second.f32()[k]);  // :184-186
    const std::string xin = arg(o.in, conv ? "Input" : "X", o);
    const auto as = act_scale.find(xin);
    CHECK(as != act_scale.end()) << o.type << ": input " << xin << " has no fake_quantize in front (input scale unknown)";
    st.input_scale = as->second;
    o.out[conv ? "Output" : "Out"] = arg(dq.out, "Out", dq);
    dq.dead = true;
    q.emplace(static_cast<int>(i), std::move(st));
  }
  // ---- 3. ConvBNFuser (conv_bn_fuser.cc:100-245)
  for (auto& kv : q) {
    RawOp& o = ops[kv.first];
    if (!is_conv(o.type)) continue;
    Q& st = kv.second;
    const auto cs = consumers(arg(o.out, "Output", o));
    if (cs.size() != 1 || ops[cs[0]].type != "batch_norm") continue;
    RawOp& bn = ops[cs[0]];
    const int h = static_cast<int>(st.weight_scale.size());  // = the conv's output channels
    // "The BN bias's size should be equal to the size of the first dim size of the conv weights" (conv_bn_fuser.cc) - and so
    // must Scale, Mean and Variance: all four are read h floats deep below
    const RawTensor &sc = f32_tensor(m, arg(bn.in, "Scale", bn), h, "batch_norm Scale"), &bi = f32_tensor(m, arg(bn.in, "Bias", bn), h, "batch_norm Bias"),
                    &mean = f32_tensor(m, arg(bn.in, "Mean", bn), h, "batch_norm Mean"), &var = f32_tensor(m, arg(bn.in, "Variance", bn), h, "batch_norm Variance");
    const float eps = bn.fattr.count("epsilon") ? bn.fattr["epsilon"] : 1e-5f;
    const int w = static_cast<int>(st.w.size()) / h;
    std::vector<float> bias(bi.f32(), bi.f32() + h);
    for (int i = 0; i < h; ++i) {
      const float alpha = sc.f32()[i] / std::sqrt(var.f32()[i] + eps);  // conv_bn_fuser.h:44-46
      const float beta = (-mean.f32()[i]) * alpha;                       // :47-49
      st.weight_scale[i] *= std::fabs(alpha);                            // .cc:180
      if (alpha < 0.f)
        for (int j = 0; j < w; ++j) st.w[static_cast<size_t>(i) * w + j] *= -1;  // :181-186
      if (st.has_bias) bias[i] += alpha * st.bias[i];                    // :232-234 (conv bias first)
      bias[i] += beta;                                                   // :236-238
    }
    st.bias = bias;
    st.has_bias = true;
    o.out["Output"] = arg(bn.out, "Y", bn);
    bn.dead = true;
  }
  // ---- 4. ConvActivationFuser
  for (auto& kv : q) {
    RawOp& o = ops[kv.first];
    if (!is_conv(o.type)) continue;
    const auto cs = consumers(arg(o.out, "Output", o));
    if (cs.size() != 1) continue;
    RawOp& a = ops[cs[0]];
    if (a.type == "relu") kv.second.act = 1;
    else if (a.type == "relu6") {
      kv.second.act = 2;
      kv.second.act_coef = a.fattr.count("threshold") ? a.fattr["threshold"] : 6.f;
    } else if (a.type == "leaky_relu") {
      kv.second.act = 4;
      kv.second.act_coef = a.fattr.count("alpha") ? a.fattr["alpha"] : 0.02f;
    } else continue;
    o.out["Output"] = arg(a.out, "Out", a);
    a.dead = true;
  }
  // ---- 5. FcFuser: mul + elementwise_add with a persistable Y
  for (auto& kv : q) {
    RawOp& o = ops[kv.first];
    if (o.type != "mul") continue;
    const auto cs = consumers(arg(o.out, "Out", o));
    if (cs.size() != 1 || ops[cs[0]].type != "elementwise_add") continue;
    RawOp& add = ops[cs[0]];
    const auto bt = m.tensors.find(arg(add.in, "Y", add));
    if (bt == m.tensors.end()) continue;
    CHECK_EQ(bt->second.dtype, 0) << "fc bias " << arg(add.in, "Y", add) << " must be fp32";
    CHECK_EQ(bt->second.numel(), static_cast<int64_t>(kv.second.weight_scale.size())) << "fc bias size must equal the output width";
    kv.second.bias.assign(bt->second.f32(), bt->second.f32() + bt->second.numel());
    kv.second.has_bias = true;
    o.out["Out"] = arg(add.out, "Out", add);
    add.dead = true;
  }
  // ---- emit
  for (size_t i = 0; i < ops.size(); ++i) {
    RawOp& o = ops[i];
    if (o.dead) continue;
    if (o.type == "feed") {
      CHECK(o.ivattr.count("shape") && o.ivattr["shape"].size() == 3UL) << "feed: attribute shape must be {c, h, w}";
      const auto& d = o.ivattr.at("shape");  // {c, h, w}
      g->Feed(arg(o.out, "Out", o), {batch, d[0], d[1], d[2]}, PRECISION(kFloat));
    } else if (o.type == "fetch") {
      g->Fetch(arg(o.in, "X", o));
    } else if (is_conv(o.type)) {
      CHECK(q.count(static_cast<int>(i))) << "kHIP has int8 conv kernels only: " << o.type << " without quantisation info";
      Q& st = q[static_cast<int>(i)];
      GraphOp& op = g->Add(o.type, {arg(o.in, "Input", o)}, arg(o.out, "Output", o));
      op.enable_int8 = true;
      const RawTensor& wt = m.tensors.at(arg(o.in, "Filter", o));
      op.w_dims = wt.dims;
      op.w = st.w;
      op.has_bias = st.has_bias;
      op.bias = st.bias;
      CHECK(o.ivattr.count("strides") && o.ivattr.count("paddings") && o.ivattr.count("dilations") && o.iattr.count("groups"))
          << o.type << ": strides / paddings / dilations / groups attributes are required";
      op.conv.strides = o.ivattr.at("strides");
      op.conv.paddings = o.ivattr.at("paddings");
      op.conv.dilations = o.ivattr.at("dilations");
      op.conv.groups = o.iattr.at("groups");
      op.conv.act = st.act;
      op.conv.act_coef = st.act_coef;
      op.conv.input_scale = st.input_scale;
      op.conv.weight_scale = st.weight_scale;
      if (o.sattr.count("padding_algorithm")) op.conv.padding_algorithm = o.sattr["padding_algorithm"];
    } else if (o.type == "mul") {
      CHECK(q.count(static_cast<int>(i))) << "kHIP has int8 fc kernels only";
      Q& st = q[static_cast<int>(i)];
      GraphOp& op = g->Add("fc", {arg(o.in, "X", o)}, arg(o.out, "Out", o));
      op.enable_int8 = true;
      op.w_dims = m.tensors.at(arg(o.in, "Y", o)).dims;
      op.w = st.w;
      op.has_bias = st.has_bias;
      op.bias = st.bias;
      op.conv.input_scale = st.input_scale;
      op.conv.weight_scale = st.weight_scale;
    } else if (o.type == "pool2d") {
      GraphOp& op = g->Add("pool2d", {arg(o.in, "X", o)}, arg(o.out, "Out", o));
      CHECK(o.sattr.count("pooling_type") && o.ivattr.count("ksize") && o.ivattr.count("strides") && o.ivattr.count("paddings"))
          << "pool2d: pooling_type / ksize / strides / paddings attributes are required";
      op.pooling_type = o.sattr.at("pooling_type");
      op.ksize = o.ivattr.at("ksize");
      op.pool_strides = o.ivattr.at("strides");
      op.pool_paddings = o.ivattr.at("paddings");
      if (op.pool_paddings.size() == 2) op.pool_paddings = {op.pool_paddings[0], op.pool_paddings[0], op.pool_paddings[1], op.pool_paddings[1]};
      op.global_pooling = o.iattr.count("global_pooling") && o.iattr["global_pooling"];
      op.exclusive = !o.iattr.count("exclusive") || o.iattr["exclusive"];
      op.ceil_mode = o.iattr.count("ceil_mode") && o.iattr["ceil_mode"];
    } else if (o.type == "elementwise_add") {
      // 6. ElementwiseActivationFuser: + relu when the relu is the only consumer
      const std::string outv = arg(o.out, "Out", o);
      const auto cs = consumers(outv);
      if (cs.size() == 1 && ops[cs[0]].type == "relu") {
        RawOp& r = ops[cs[0]];
        GraphOp& op = g->Add("fusion_elementwise_add_activation", {arg(o.in, "X", o), arg(o.in, "Y", o)}, arg(r.out, "Out", r));
        op.act_type = "relu";
        r.dead = true;
      } else {
        g->Add("elementwise_add", {arg(o.in, "X", o), arg(o.in, "Y", o)}, outv);
      }
    } else if (o.type == "softmax") {
      g->Add("softmax", {arg(o.in, "X", o)}, arg(o.out, "Out", o));
    } else {
      LOG(FATAL) << "model op " << o.type << " has no kHIP lowering (left over after the fusion passes)";
    }
  }
}

}  // namespace model_parser
}  // namespace lite
}  // namespace paddle

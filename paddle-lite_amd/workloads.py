"""Synthetic INT8 workloads of BASELINE.json: the MobileNetV1 graph exactly as it reaches the kernel boundary after
the reference's passes (SURVEY.md Appendix B layer table + Appendix D program shape), with random-init weights.

  feed(fp32) -> io_copy h2d -> calib fp32->int8 -> conv2d 3x3s2 [int8_out, relu]
     -> 13 x { depthwise_conv2d 3x3 [int8_out, relu] -> conv2d 1x1 [int8_out, relu] }   (last 1x1: fp32_out)
     -> pool2d global avg (fp32) -> calib fp32->int8 -> fc [fp32out] -> softmax -> io_copy d2h

Layer shapes: lite/tests/benchmark/src/convolution_configs.h:355-379.  Scales follow the reference tests' convention
(in = 1/127 at the input, per-channel-varying weight scales as produced by conv_bn fusion — SURVEY.md A.9) and are
chosen so that every int8 activation tensor keeps a healthy spread (neither all-zero nor saturated).
"""
import numpy as np

# (cin, cout, stride) of the 13 depthwise-separable blocks
MBV1_BLOCKS = [(32, 64, 1), (64, 128, 2), (128, 128, 1), (128, 256, 2), (256, 256, 1), (256, 512, 2),
               (512, 512, 1), (512, 512, 1), (512, 512, 1), (512, 512, 1), (512, 512, 1), (512, 1024, 2), (1024, 1024, 1)]
NUM_CLASSES = 1000


def mobilenet_v1_layers(res=224):
    """[(name, op_type, cin, cout, k, stride, pad, groups, hin)] for the 27 convs."""
    layers = [("conv1", "conv2d", 3, 32, 3, 2, 1, 1, res)]
    h = (res + 2 - 3) // 2 + 1
    for i, (cin, cout, s) in enumerate(MBV1_BLOCKS):
        layers.append(("dw%d" % (i + 2), "depthwise_conv2d", cin, cin, 3, s, 1, cin, h))
        h = (h + 2 - 3) // s + 1
        layers.append(("pw%d" % (i + 2), "conv2d", cin, cout, 1, 1, 0, 1, h))
    return layers


def mobilenet_v1_macs(res=224):
    tot = {"pointwise": 0, "depthwise": 0, "first": 0}
    act_bytes = 0
    for (name, op, cin, cout, k, s, p, g, hin) in mobilenet_v1_layers(res):
        ho = (hin + 2 * p - k) // s + 1
        macs = ho * ho * cout * (cin // g) * k * k
        key = "first" if name == "conv1" else ("depthwise" if g > 1 else "pointwise")
        tot[key] += macs
        act_bytes += cin * hin * hin + cout * ho * ho
    tot["fc"] = 1024 * NUM_CLASSES
    tot["act_bytes"] = act_bytes
    return tot


def make_mobilenet_v1_weights(seed=1234, res=224):
    """Seeded random-init int8 weights, fp32 biases and scales for every layer."""
    rng = np.random.default_rng(seed)
    W = {}
    in_scale = np.float32(1.0 / 127)  # network input in [-1, 1]
    W["input_scale"] = in_scale
    sig_x = 73.0  # std of a uniform int8 input
    for (name, op, cin, cout, k, s, p, g, hin) in mobilenet_v1_layers(res):
        kk = (cin // g) * k * k
        w = rng.integers(-127, 128, (cout, cin // g, k, k)).astype(np.int8)
        # every activation tensor covers a real range of about +-4: out_scale = 4/127; the per-channel weight scales
        # (varying, as conv_bn fusion leaves them — SURVEY.md A.9) are sized so that the requantised int8 output has
        # a std of ~45 before relu: acc_std * in_scale * w_scale / out_scale = 45
        out_scale = np.float32(4.0 / 127)
        acc_std = np.sqrt(kk) * sig_x * 73.0
        var = (1.0 + (np.arange(cout) % 7) / 8.0) / 1.375
        w_scale = (var * 45.0 * float(out_scale) / (acc_std * float(in_scale))).astype(np.float32)
        bias = (rng.uniform(-0.5, 0.5, cout) * 45.0 * float(out_scale)).astype(np.float32)
        W[name] = dict(w=w, bias=bias, w_scale=w_scale, in_scale=in_scale, out_scale=out_scale)
        in_scale = out_scale
        sig_x = 30.0  # post-relu int8 activations: half-normal with the std above
    # fc: input = calib(pool(fp32 output of the last pointwise)); pooled relu outputs are positive, O(real_std)
    pool_scale = np.float32(W["pw14"]["out_scale"] * 60.0 / 127.0)
    W["pool_scale"] = pool_scale
    wf = rng.integers(-127, 128, (1024, NUM_CLASSES)).astype(np.int8)
    W["fc"] = dict(w=wf, bias=rng.uniform(-1, 1, NUM_CLASSES).astype(np.float32),
                   w_scale=((1.0 + (np.arange(NUM_CLASSES) % 5) / 8.0) / 127.0 / 32.0).astype(np.float32),
                   in_scale=pool_scale, out_scale=np.float32(1.0))
    return W


def build_mobilenet_v1(pred, W, batch, res=224):
    """Emit the Appendix-D program into a liteapi.Predictor.  Returns the output variable name."""
    from . import liteapi
    pred.add_feed("image", (batch, 3, res, res), liteapi.PREC_FLOAT)
    pred.add_io_copy("image", "image_dev", True)
    pred.add_calib("image_dev", "x0", float(W["input_scale"]), True)
    cur = "x0"
    layers = mobilenet_v1_layers(res)
    for i, (name, op, cin, cout, k, s, p, g, hin) in enumerate(layers):
        L = W[name]
        last = i == len(layers) - 1  # consumer pool2d is not enable_int8 -> fp32_out (static_kernel_pick_pass.cc:93-106)
        pred.add_conv(op, cur, name, L["w"], L["bias"], (s, s), (p, p, p, p), (1, 1), g, 1, 0.0, float(L["in_scale"]),
                      L["w_scale"], float(L["out_scale"]), not last)
        cur = name
    pred.add_global_avg_pool(cur, "pool")
    pred.add_calib("pool", "pool_i8", float(W["pool_scale"]), True)
    F = W["fc"]
    pred.add_fc("pool_i8", "logits", F["w"], F["bias"], float(F["in_scale"]), F["w_scale"], float(F["out_scale"]), False, False)
    pred.add_softmax("logits", "prob")
    pred.add_io_copy("prob", "prob_host", False)
    return "prob_host"


# =====================================================================================================================
# Generic op-list networks ("as the optimiser sees them after its fusion passes") for graph mode
# (lite/api/graph_builder.h): ResNet50 and MobileNetV2 of BASELINE.json configs C4 / C5, MobileNetV1 again in this form.
# Layer shapes: lite/tests/benchmark/src/convolution_configs.h:839-891 (ResNet50: stride 2 sits on the 3x3 conv of a
# stage's first block, shortcut = 1x1 stride-2 conv) and :381-466 (MobileNetV2, t/c/n/s table); PH/PW there are totals.
# An op is a dict:
#   conv2d / depthwise_conv2d: name=out, src, w [cout, cin/g, k, k] int8, bias, stride, pad, groups, act (0 none, 1 relu,
#                              2 relu6), act_coef, in_scale (the activation scale of its input tensor), w_scale [cout]
#   fc: src, w [k, n], bias, in_scale, w_scale [n]
#   pool2d: src, pooling_type, ksize, stride, pad, global_pooling
#   add: x, y, act ("" | "relu")      softmax: src
# On the reference's ARM target pool2d and elementwise_add exist in fp32 only, so the kernel-pick
# rule gives the convs in front of them the fp32_out kernel and the consumers behind them a calib.
# =====================================================================================================================
class _NetGen:
    def __init__(self, seed):
        self.rng = np.random.default_rng(seed)
        self.ops = []
        self.act_scale = {}   # tensor -> quantisation scale its int8 consumers use (Input0_scale)
        self.sig_i8 = {}      # tensor -> rough std of its int8 image (sizes the synthetic weight scales)
        self.shape = {}       # tensor -> (c, h, w)

    def tensor(self, name, c, h, w, act_scale, sig_i8):
        self.shape[name] = (c, h, w)
        self.act_scale[name] = np.float32(act_scale)
        self.sig_i8[name] = sig_i8

    def conv(self, name, src, cout, k, stride, pad, groups=1, act=1, act_coef=0.0, out_range=4.0, op=None):
        cin, h, w = self.shape[src]
        kk = (cin // groups) * k * k
        wt = self.rng.integers(-127, 128, (cout, cin // groups, k, k)).astype(np.int8)
        in_scale = self.act_scale[src]
        out_scale = np.float32(out_range / 127.0)
        acc_std = np.sqrt(kk) * self.sig_i8[src] * 73.0
        var = (1.0 + (np.arange(cout) % 7) / 8.0) / 1.375
        w_scale = (var * 45.0 * float(out_scale) / (acc_std * float(in_scale))).astype(np.float32)
        bias = (self.rng.uniform(-0.5, 0.5, cout) * 45.0 * float(out_scale)).astype(np.float32)
        if op is None:
            op = "depthwise_conv2d" if (groups == cin and groups == cout and groups > 1) else "conv2d"
        self.ops.append(dict(op=op, name=name, src=src, w=wt, bias=bias, stride=stride, pad=pad, groups=groups, act=act,
                             act_coef=float(act_coef), in_scale=in_scale, w_scale=w_scale))
        ho = (h + 2 * pad - k) // stride + 1
        wo = (w + 2 * pad - k) // stride + 1
        self.tensor(name, cout, ho, wo, out_scale, 30.0 if act else 45.0)
        return name

    def pool(self, name, src, pooling_type, k, stride, pad, global_pooling=False):
        c, h, w = self.shape[src]
        self.ops.append(dict(op="pool2d", name=name, src=src, pooling_type=pooling_type, ksize=k, stride=stride, pad=pad,
                             global_pooling=global_pooling))
        if global_pooling:
            self.tensor(name, c, 1, 1, np.float32(self.act_scale[src] * 100.0 / 127.0), 40.0)
        else:
            self.tensor(name, c, (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1, self.act_scale[src],
                        self.sig_i8[src] * 1.3)
        return name

    def add(self, name, x, y, act=""):
        c, h, w = self.shape[x]
        self.ops.append(dict(op="add", name=name, x=x, y=y, act=act))
        # real-valued std of the sum = hypot of the operands' (int8 std x scale); quantise it so that the int8 image
        # has a std of ~45 before the relu, like every conv output
        real = float(np.hypot(self.sig_i8[x] * float(self.act_scale[x]), self.sig_i8[y] * float(self.act_scale[y])))
        self.tensor(name, c, h, w, np.float32(real / (28.0 if act else 45.0)), 42.0 if act else 45.0)
        return name

    def fc(self, name, src, n):
        c, h, w = self.shape[src]
        k = c * h * w
        wf = self.rng.integers(-127, 128, (k, n)).astype(np.int8)
        self.ops.append(dict(op="fc", name=name, src=src, w=wf, bias=self.rng.uniform(-1, 1, n).astype(np.float32),
                             in_scale=self.act_scale[src],
                             w_scale=((1.0 + (np.arange(n) % 5) / 8.0) / 127.0 / 32.0).astype(np.float32)))
        self.tensor(name, n, 1, 1, np.float32(1.0), 30.0)
        return name

    def softmax(self, name, src):
        self.ops.append(dict(op="softmax", name=name, src=src))
        self.shape[name] = self.shape[src]
        return name


def _finish(g, res, out):
    return dict(ops=g.ops, input="image", input_shape=(3, res, res), output=out, shapes=dict(g.shape))


def resnet50_net(seed=50, res=224, num_classes=NUM_CLASSES):
    g = _NetGen(seed)
    g.tensor("image", 3, res, res, 1.0 / 127, 73.0)
    x = g.conv("conv1", "image", 64, 7, 2, 3, act=1)
    x = g.pool("pool1", x, "max", 3, 2, 1)
    for si, (width, blocks, stride) in enumerate([(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)]):
        for b in range(blocks):
            p = "res%d%s" % (si + 2, "abcdef"[b])
            s = stride if b == 0 else 1
            y = g.conv(p + "_branch2a", x, width, 1, 1, 0, act=1)
            y = g.conv(p + "_branch2b", y, width, 3, s, 1, act=1)
            y = g.conv(p + "_branch2c", y, 4 * width, 1, 1, 0, act=0)
            sc = g.conv(p + "_branch1", x, 4 * width, 1, s, 0, act=0) if b == 0 else x
            x = g.add(p, sc, y, act="relu")
    x = g.pool("pool5", x, "avg", 7, 1, 0, global_pooling=True)
    x = g.fc("fc", x, num_classes)
    x = g.softmax("prob", x)
    return _finish(g, res, x)


MBV2_SETTING = [(1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1)]


def mobilenet_v2_net(seed=52, res=224, num_classes=NUM_CLASSES):
    g = _NetGen(seed)
    g.tensor("image", 3, res, res, 1.0 / 127, 73.0)
    R6 = dict(act=2, act_coef=6.0, out_range=8.0)  # relu6 tensors quantised over [-8, 8]: the clip at 6 is visible in int8
    x = g.conv("conv1", "image", 32, 3, 2, 1, **R6)
    cin, bi = 32, 0
    for (t, c, n, s) in MBV2_SETTING:
        for i in range(n):
            bi += 1
            p = "b%d" % bi
            stride = s if i == 0 else 1
            y = x
            if t != 1:
                y = g.conv(p + "_expand", y, cin * t, 1, 1, 0, **R6)
            y = g.conv(p + "_dw", y, cin * t, 3, stride, 1, groups=cin * t, **R6)
            y = g.conv(p + "_project", y, c, 1, 1, 0, act=0)
            x = g.add(p + "_add", x, y) if (stride == 1 and cin == c) else y
            cin = c
    x = g.conv("conv_last", x, 1280, 1, 1, 0, **R6)
    x = g.pool("pool", x, "avg", g.shape[x][1], 1, 0, global_pooling=True)
    x = g.fc("fc", x, num_classes)
    x = g.softmax("prob", x)
    return _finish(g, res, x)


def mobilenet_v1_net(seed=1234, res=224):
    """MobileNetV1 in op-list form, same weights as make_mobilenet_v1_weights(seed): graph mode must arrive at exactly
    the Appendix-D program that build_mobilenet_v1 writes out by hand."""
    W = make_mobilenet_v1_weights(seed, res)
    ops, shapes = [], {}
    cur = "image"
    for (name, op, cin, cout, k, s, p, g, hin) in mobilenet_v1_layers(res):
        L = W[name]
        ops.append(dict(op=op, name=name, src=cur, w=L["w"], bias=L["bias"], stride=s, pad=p, groups=g, act=1, act_coef=0.0,
                        in_scale=L["in_scale"], w_scale=L["w_scale"]))
        ho = (hin + 2 * p - k) // s + 1
        shapes[name] = (cout, ho, ho)
        cur = name
    ops.append(dict(op="pool2d", name="pool", src=cur, pooling_type="avg", ksize=shapes[cur][1], stride=1, pad=0,
                    global_pooling=True))
    shapes["pool"] = (shapes[cur][0], 1, 1)
    F = W["fc"]
    ops.append(dict(op="fc", name="logits", src="pool", w=F["w"], bias=F["bias"], in_scale=F["in_scale"], w_scale=F["w_scale"]))
    ops.append(dict(op="softmax", name="prob", src="logits"))
    shapes["logits"] = shapes["prob"] = (NUM_CLASSES, 1, 1)
    return dict(ops=ops, input="image", input_shape=(3, res, res), output="prob", shapes=shapes)


def net_stats(net):
    """MACs and algorithmic activation bytes per image by op class (int8 tensors 1 B/elt, fp32 tensors 4 B/elt are
    decided by the lowering; here: conv MACs and element counts only)."""
    macs = {"conv1x1": 0, "conv_kxk": 0, "depthwise": 0, "fc": 0}
    shapes = net["shapes"]
    for o in net["ops"]:
        if o["op"] in ("conv2d", "depthwise_conv2d"):
            cout, cg, k, _ = o["w"].shape
            c, h, w = shapes[o["name"]]
            m = h * w * cout * cg * k * k
            key = "depthwise" if o["op"] == "depthwise_conv2d" else ("conv1x1" if k == 1 else "conv_kxk")
            macs[key] += m
        elif o["op"] == "fc":
            macs["fc"] += o["w"].shape[0] * o["w"].shape[1]
    return macs


def emit_graph(pred, net, batch, fuse=True, fuse_dwpw=None):
    """Feed the op list to the predictor's graph mode and lower it.  Returns the host name of the output variable.
    fuse=False: the reference program instruction for instruction (no kHIP graph-level fusion).
    fuse_dwpw: None = the builder's default (depthwise -> pointwise pairs the fused kernel takes become one instruction),
    True = every eligible pair (shapes outside the kernel run as two launches inside the instruction), False = none."""
    from . import liteapi
    pred.graph_set_fuse(fuse)
    if fuse_dwpw is not None:
        pred.graph_set_fuse_dwpw(fuse_dwpw)
    c, h, w = net["input_shape"]
    pred.graph_feed(net["input"], (batch, c, h, w), liteapi.PREC_FLOAT)
    for o in net["ops"]:
        t = o["op"]
        if t in ("conv2d", "depthwise_conv2d"):
            p = o["pad"]
            pred.graph_conv(t, o["src"], o["name"], o["w"], o["bias"], (o["stride"],) * 2, (p, p, p, p), (1, 1), o["groups"],
                            o["act"], o["act_coef"], float(o["in_scale"]), o["w_scale"])
        elif t == "fc":
            pred.graph_fc(o["src"], o["name"], o["w"], o["bias"], float(o["in_scale"]), o["w_scale"], False)
        elif t == "pool2d":
            p = o["pad"]
            pred.graph_pool(o["src"], o["name"], o["pooling_type"], (o["ksize"],) * 2, (o["stride"],) * 2, (p, p, p, p),
                            o["global_pooling"], True, False)
        elif t == "add":
            pred.graph_elementwise_add(o["x"], o["y"], o["name"], o["act"])
        elif t == "softmax":
            pred.graph_softmax(o["src"], o["name"])
        else:
            raise ValueError(t)
    pred.graph_fetch(net["output"])
    return net["output"] + "/host"


def program_costs(net, batch, plan_lines):
    """Algorithmic work of every instruction of the lowered program (unique input + weights + output once,
    no im2col expansion, no re-reads), aligned with `plan_lines` (GraphBuilder::Plan / Predictor.graph_plan()).
    Returns [dict(name, family, ops, bytes)]; io_copy lines get family "io_copy" and zero cost."""
    shapes = dict(net["shapes"])
    shapes[net["input"]] = net["input_shape"]
    esz = {}  # variable -> bytes per element

    def numel(v):
        base = v.replace("/target_trans", "").replace("/precision_trans", "")
        c, h, w = shapes[base]
        return batch * c * h * w

    int8_ops = iter([o for o in net["ops"] if o["op"] in ("conv2d", "depthwise_conv2d", "fc")])  # never reordered
    out = []
    esz[net["input"]] = 4
    for line in plan_lines:
        head, rest = line.split(" ", 1)
        toks = rest.split(" ")
        kv = dict(f.split("=", 1) for f in toks if "=" in f)
        flags = {f for f in toks if "=" not in f}
        ins, dst = kv["in"].split(","), kv["out"]
        op, alias = head.split("/")
        if op == "io_copy":
            esz[dst] = esz.get(ins[0], 4)
            out.append(dict(name=dst, family="io_copy", ops=0, bytes=0))
            continue
        if op == "calib":
            esz[dst] = 1 if alias == "fp32_to_int8" else 4
            out.append(dict(name=dst, family="calib", ops=0, bytes=numel(ins[0]) * esz[ins[0]] + numel(dst) * esz[dst]))
            continue
        if op in ("conv2d", "depthwise_conv2d", "fc"):
            o = next(int8_ops)
            assert o["op"] == op, (o["op"], line)
            esz[dst] = 1 if alias in ("int8_out", "int8out") else 4
            wbytes = int(o["w"].size)
            if op == "fc":
                macs = batch * o["w"].shape[0] * o["w"].shape[1]
                fam = "fc"
            else:
                cout, cg, k, _ = o["w"].shape
                c, h, w = shapes[o["name"]]
                macs = batch * h * w * cout * cg * k * k
                cin = shapes[o["src"]][0]
                if op == "depthwise_conv2d":
                    fam = "depthwise%dx%d" % (k, k)
                elif k == 1:
                    fam = "pointwise1x1" if o["stride"] == 1 else "conv1x1s2"
                else:
                    fam = "stem_conv" if cin <= 4 else "conv%dx%d" % (k, k)
            byts = sum(numel(i) * esz[i] for i in ins) + wbytes
            if "+pw" in kv:       # opt-in fusion: this depthwise conv took its 1x1 consumer over: `dst` is that conv's output
                o2 = next(int8_ops)
                pw_name = kv.get("pw_out", dst)  # (+pool: `dst` is the global average pool's output, the conv's plane is never written)
                assert o2["op"] == "conv2d" and o2["name"] == pw_name, (o2["name"], line)
                esz[dst] = 1 if kv["+pw"].endswith("int8_out") else 4
                m2, c2 = o2["w"].shape[0], o2["w"].shape[1]
                _, h2, w2 = shapes[pw_name]
                macs += batch * h2 * w2 * m2 * c2
                wbytes += int(o2["w"].size)
                byts += int(o2["w"].size)
                fam = "dwpw_fused"
            if "-f32" not in flags:
                byts += numel(dst) * esz[dst]
            if "+add" in kv:      # fused residual operand (fp32)
                byts += numel(kv["+add"]) * 4
            if "+calib" in kv:    # fused int8 copy
                esz[kv["+calib"]] = 1
                byts += numel(kv["+calib"])
            out.append(dict(name=dst, family=fam, ops=2 * macs, bytes=byts))
        else:
            esz[dst] = 1 if "int8" in flags else 4
            fam = {"pool2d": "pool2d", "elementwise_add": "elementwise_add", "fusion_elementwise_add_activation": "elementwise_add",
                   "softmax": "softmax"}[op]
            out.append(dict(name=dst, family=fam, ops=0, bytes=sum(numel(i) * esz[i] for i in ins) + numel(dst) * esz[dst]))
    return out

"""Writer of the PLHIPM01 model container (lite/model_parser/hip_model.h) and a synthetic PaddleSlim-style quantised
model to put in it.

The container holds a graph the way a PaddleSlim QAT model arrives at the reference's optimiser — fake_quantize_* in
front of every quantised op, fp32 weights on the integer grid + fake_dequantize_max_abs(max_range) behind it, separate
batch_norm / relu / elementwise_add ops — so that the C++ loader has the reference's fusion semantics to apply
(quant_dequant_op_fuser.cc, conv_bn_fuser.cc, ...).  `fuse_reference()` is a numpy restatement of those passes used by
the tests to check the loader value for value; it is NOT used by the product path (the loader is C++)."""
import struct

import numpy as np


def _s(b):
    b = b.encode() if isinstance(b, str) else b
    return struct.pack("<H", len(b)) + b


def write_container(path, tensors, ops):
    """tensors: {name: np.ndarray (float32 | int8)}; ops: [dict(type, inputs={arg: var}, outputs={arg: var}, attrs={})]."""
    out = [b"PLHIPM01", struct.pack("<II", len(tensors), len(ops))]
    for name, a in tensors.items():
        a = np.ascontiguousarray(a)
        assert a.dtype in (np.float32, np.int8), (name, a.dtype)
        raw = a.tobytes()
        out.append(_s(name) + struct.pack("<BB", 0 if a.dtype == np.float32 else 1, a.ndim) +
                   struct.pack("<%dq" % a.ndim, *a.shape) + struct.pack("<Q", len(raw)) + raw)
        pos = sum(len(x) for x in out)
        out.append(b"\0" * ((-pos) % 8))
    for o in ops:
        rec = [_s(o["type"]), struct.pack("<H", len(o.get("inputs", {})))]
        for k, v in o.get("inputs", {}).items():
            rec += [_s(k), _s(v)]
        rec.append(struct.pack("<H", len(o.get("outputs", {}))))
        for k, v in o.get("outputs", {}).items():
            rec += [_s(k), _s(v)]
        attrs = o.get("attrs", {})
        rec.append(struct.pack("<H", len(attrs)))
        for k, v in attrs.items():
            if isinstance(v, (bool, int, np.integer)):
                rec.append(_s(k) + struct.pack("<Bi", 0, int(v)))
            elif isinstance(v, (float, np.floating)):
                rec.append(_s(k) + struct.pack("<Bf", 1, float(v)))
            elif isinstance(v, str):
                rec.append(_s(k) + struct.pack("<B", 3) + _s(v))
            else:
                v = list(v)
                if v and isinstance(v[0], (float, np.floating)):
                    rec.append(_s(k) + struct.pack("<BH", 4, len(v)) + struct.pack("<%df" % len(v), *v))
                else:
                    rec.append(_s(k) + struct.pack("<BH", 2, len(v)) + struct.pack("<%di" % len(v), *[int(x) for x in v]))
        out.append(b"".join(rec))
    blob = b"".join(out)
    if path is not None:
        with open(path, "wb") as f:
            f.write(blob)
    return blob


class SlimBuilder:
    """Emits the un-fused QAT op sequence layer by layer."""

    def __init__(self, seed):
        self.rng = np.random.default_rng(seed)
        self.tensors, self.ops = {}, []
        self.n = 0

    def feed(self, name, chw):
        self.ops.append(dict(type="feed", outputs={"Out": name}, attrs={"shape": list(chw)}))
        return name

    def _quant(self, x, abs_max):
        q, sc = x + ".quantized", x + ".scale"
        if sc not in self.tensors:
            self.tensors[sc] = np.array([abs_max], np.float32)
        q = "%s.q%d" % (x, self.n)
        self.n += 1
        self.ops.append(dict(type="fake_quantize_moving_average_abs_max", inputs={"X": x, "InScale": sc},
                             outputs={"Out": q, "OutScale": sc}, attrs={"bit_length": 8}))
        return q

    def conv_bn(self, name, x, cin, cout, k, stride, pad, groups=1, act="relu", x_abs_max=4.0, threshold=6.0):
        rng = self.rng
        kk = (cin // groups) * k * k
        w_int = rng.integers(-127, 128, (cout, cin // groups, k, k))
        self.tensors[name + "_weights"] = w_int.astype(np.float32)  # fp32 values on the int8 grid
        whole = 1.0 / (73.0 * np.sqrt(kk))                          # real weight = w_int * whole
        max_range = np.float32(127.0 * 127.0 / (127.0 * whole))
        q = self._quant(x, x_abs_max)
        op = "depthwise_conv2d" if (groups == cin and groups == cout and groups > 1) else "conv2d"
        self.ops.append(dict(type=op, inputs={"Input": q, "Filter": name + "_weights"}, outputs={"Output": name + ".conv"},
                             attrs={"strides": [stride, stride], "paddings": [pad, pad], "dilations": [1, 1], "groups": groups}))
        self.ops.append(dict(type="fake_dequantize_max_abs", inputs={"X": name + ".conv"}, outputs={"Out": name + ".deq"},
                             attrs={"max_range": float(max_range)}))
        sign = np.where(np.arange(cout) % 5 == 3, -1.0, 1.0)        # some negative BN scales: the int8 rows get negated
        self.tensors[name + "_bn_scale"] = (sign * rng.uniform(0.6, 1.4, cout)).astype(np.float32)
        self.tensors[name + "_bn_offset"] = rng.uniform(-0.5, 0.5, cout).astype(np.float32)
        self.tensors[name + "_bn_mean"] = rng.uniform(-0.2, 0.2, cout).astype(np.float32)
        self.tensors[name + "_bn_variance"] = rng.uniform(0.5, 2.0, cout).astype(np.float32)
        self.ops.append(dict(type="batch_norm", inputs={"X": name + ".deq", "Scale": name + "_bn_scale", "Bias": name + "_bn_offset",
                                                        "Mean": name + "_bn_mean", "Variance": name + "_bn_variance"},
                             outputs={"Y": name + ".bn"}, attrs={"epsilon": 1e-5}))
        if act == "relu":
            self.ops.append(dict(type="relu", inputs={"X": name + ".bn"}, outputs={"Out": name}))
        elif act == "relu6":
            self.ops.append(dict(type="relu6", inputs={"X": name + ".bn"}, outputs={"Out": name}, attrs={"threshold": threshold}))
        else:
            self.ops[-1]["outputs"]["Y"] = name
        return name

    def pool(self, name, x, pooling_type, k, stride, pad, global_pooling=False):
        self.ops.append(dict(type="pool2d", inputs={"X": x}, outputs={"Out": name},
                             attrs={"pooling_type": pooling_type, "ksize": [k, k], "strides": [stride, stride], "paddings": [pad, pad],
                                    "global_pooling": int(global_pooling), "exclusive": 1, "ceil_mode": 0}))
        return name

    def fc(self, name, x, k, n, x_abs_max):
        w_int = self.rng.integers(-127, 128, (k, n))
        self.tensors[name + "_weights"] = w_int.astype(np.float32)
        whole = 1.0 / (73.0 * np.sqrt(k))
        self.tensors[name + "_offset"] = self.rng.uniform(-1, 1, n).astype(np.float32)
        q = self._quant(x, x_abs_max)
        self.ops.append(dict(type="mul", inputs={"X": q, "Y": name + "_weights"}, outputs={"Out": name + ".mul"},
                             attrs={"x_num_col_dims": 1, "y_num_col_dims": 1}))
        self.ops.append(dict(type="fake_dequantize_max_abs", inputs={"X": name + ".mul"}, outputs={"Out": name + ".deq"},
                             attrs={"max_range": float(np.float32(127.0 * 127.0 / (127.0 * whole)))}))
        self.ops.append(dict(type="elementwise_add", inputs={"X": name + ".deq", "Y": name + "_offset"}, outputs={"Out": name},
                             attrs={"axis": 1}))
        return name

    def add(self, name, x, y, relu=False):
        self.ops.append(dict(type="elementwise_add", inputs={"X": x, "Y": y}, outputs={"Out": name + (".sum" if relu else "")},
                             attrs={"axis": -1}))
        if relu:
            self.ops.append(dict(type="relu", inputs={"X": name + ".sum"}, outputs={"Out": name}))
        return name

    def softmax(self, name, x):
        self.ops.append(dict(type="softmax", inputs={"X": x}, outputs={"Out": name}, attrs={"axis": -1}))
        return name

    def fetch(self, x):
        self.ops.append(dict(type="fetch", inputs={"X": x}))


def slim_mobilenet_v1(seed=4242, res=224, classes=1000):
    from .workloads import MBV1_BLOCKS
    b = SlimBuilder(seed)
    x = b.feed("image", (3, res, res))
    x = b.conv_bn("conv1", x, 3, 32, 3, 2, 1, x_abs_max=1.0)
    for i, (cin, cout, s) in enumerate(MBV1_BLOCKS):
        x = b.conv_bn("dw%d" % (i + 2), x, cin, cin, 3, s, 1, groups=cin)
        x = b.conv_bn("pw%d" % (i + 2), x, cin, cout, 1, 1, 0)
    x = b.pool("pool", x, "avg", (res // 32), 1, 0, global_pooling=True)
    x = b.fc("logits", x, 1024, classes, x_abs_max=2.0)
    x = b.softmax("prob", x)
    b.fetch(x)
    return b.tensors, b.ops


def slim_residual_toy(seed=4243, res=32):
    """A small graph with every pattern of the loader: relu6, a linear conv, elementwise_add with and without relu."""
    b = SlimBuilder(seed)
    x = b.feed("image", (3, res, res))
    x = b.conv_bn("stem", x, 3, 16, 3, 2, 1, act="relu6", x_abs_max=1.0)
    y = b.conv_bn("a_expand", x, 16, 48, 1, 1, 0, act="relu6", x_abs_max=6.0)
    y = b.conv_bn("a_dw", y, 48, 48, 3, 1, 1, groups=48, act="relu6", x_abs_max=6.0)
    y = b.conv_bn("a_project", y, 48, 16, 1, 1, 0, act=None, x_abs_max=6.0)
    x = b.add("a_add", x, y)
    y = b.conv_bn("b_conv", x, 16, 16, 3, 1, 1, act=None, x_abs_max=8.0)
    x = b.add("b_add", x, y, relu=True)
    x = b.pool("pool", x, "max", 2, 2, 0)
    x = b.pool("gap", x, "avg", res // 4, 1, 0, global_pooling=True)
    x = b.fc("logits", x, 16, 10, x_abs_max=4.0)
    x = b.softmax("prob", x)
    b.fetch(x)
    return b.tensors, b.ops


def fuse_reference(tensors, ops):
    """numpy restatement (fp32, operation for operation) of what the loader must produce: an op-list network in the form
    workloads.emit_graph / oracle.graph_oracle take.  Handles exactly the patterns SlimBuilder emits."""
    f32 = np.float32
    net_ops, shapes = [], {}
    scale_of = {}
    alias = {}
    i = 0
    inp = None
    out = None

    def real(v):
        while v in alias:
            v = alias[v]
        return v

    while i < len(ops):
        o = ops[i]
        t = o["type"]
        if t == "feed":
            inp = o["outputs"]["Out"]
            in_shape = tuple(o["attrs"]["shape"])
            shapes[inp] = in_shape
            i += 1
        elif t.startswith("fake_quantize"):
            scale_of[o["outputs"]["Out"]] = f32(tensors[o["outputs"]["OutScale"]][0]) / f32(127)
            alias[o["outputs"]["Out"]] = o["inputs"]["X"]
            i += 1
        elif t in ("conv2d", "depthwise_conv2d"):
            deq, bn = ops[i + 1], ops[i + 2]
            assert deq["type"] == "fake_dequantize_max_abs" and bn["type"] == "batch_norm"
            wf = tensors[o["inputs"]["Filter"]]
            w = wf.astype(np.int8).copy()
            whole = f32(127 * 127) / f32(deq["attrs"]["max_range"]) / f32(127)
            ws = np.full(w.shape[0], whole, f32)
            sc, bi = tensors[bn["inputs"]["Scale"]], tensors[bn["inputs"]["Bias"]]
            mean, var = tensors[bn["inputs"]["Mean"]], tensors[bn["inputs"]["Variance"]]
            alpha = (sc / np.sqrt(var + f32(bn["attrs"]["epsilon"]))).astype(f32)
            beta = ((-mean) * alpha).astype(f32)
            ws = (ws * np.abs(alpha)).astype(f32)
            w[alpha < 0] *= -1
            bias = (bi + beta).astype(f32)
            name = bn["outputs"]["Y"]
            act, coef, step = 0, 0.0, 3
            if i + 3 < len(ops) and ops[i + 3]["type"] in ("relu", "relu6") and ops[i + 3]["inputs"]["X"] == name:
                a = ops[i + 3]
                act, coef = (1, 0.0) if a["type"] == "relu" else (2, float(a["attrs"]["threshold"]))
                name, step = a["outputs"]["Out"], 4
            src = real(o["inputs"]["Input"])
            at = o["attrs"]
            net_ops.append(dict(op=t, name=name, src=src, w=w, bias=bias, stride=at["strides"][0], pad=at["paddings"][0],
                                groups=at["groups"], act=act, act_coef=coef, in_scale=scale_of[o["inputs"]["Input"]], w_scale=ws))
            c, h, wd = shapes[src]
            k = w.shape[2]
            shapes[name] = (w.shape[0], (h + 2 * at["paddings"][0] - k) // at["strides"][0] + 1, (wd + 2 * at["paddings"][0] - k) // at["strides"][0] + 1)
            i += step
        elif t == "mul":
            deq, add = ops[i + 1], ops[i + 2]
            wf = tensors[o["inputs"]["Y"]]
            whole = f32(127 * 127) / f32(deq["attrs"]["max_range"]) / f32(127)
            name = add["outputs"]["Out"]
            src = real(o["inputs"]["X"])
            net_ops.append(dict(op="fc", name=name, src=src, w=wf.astype(np.int8), bias=tensors[add["inputs"]["Y"]],
                                in_scale=scale_of[o["inputs"]["X"]], w_scale=np.full(wf.shape[1], whole, f32)))
            shapes[name] = (wf.shape[1], 1, 1)
            i += 3
        elif t == "pool2d":
            at = o["attrs"]
            name, src = o["outputs"]["Out"], real(o["inputs"]["X"])
            net_ops.append(dict(op="pool2d", name=name, src=src, pooling_type=at["pooling_type"], ksize=at["ksize"][0],
                                stride=at["strides"][0], pad=at["paddings"][0], global_pooling=bool(at["global_pooling"])))
            c, h, wd = shapes[src]
            shapes[name] = (c, 1, 1) if at["global_pooling"] else (c, (h + 2 * at["paddings"][0] - at["ksize"][0]) // at["strides"][0] + 1,
                                                                    (wd + 2 * at["paddings"][0] - at["ksize"][0]) // at["strides"][0] + 1)
            i += 1
        elif t == "elementwise_add":
            name = o["outputs"]["Out"]
            relu = i + 1 < len(ops) and ops[i + 1]["type"] == "relu" and ops[i + 1]["inputs"]["X"] == name
            if relu:
                name = ops[i + 1]["outputs"]["Out"]
            x, y = real(o["inputs"]["X"]), real(o["inputs"]["Y"])
            net_ops.append(dict(op="add", name=name, x=x, y=y, act="relu" if relu else ""))
            shapes[name] = shapes[x]
            i += 2 if relu else 1
        elif t == "softmax":
            net_ops.append(dict(op="softmax", name=o["outputs"]["Out"], src=real(o["inputs"]["X"])))
            shapes[o["outputs"]["Out"]] = shapes[real(o["inputs"]["X"])]
            i += 1
        elif t == "fetch":
            out = real(o["inputs"]["X"])
            i += 1
        else:
            raise ValueError(t)
    return dict(ops=net_ops, input=inp, input_shape=in_shape, output=out, shapes=shapes)

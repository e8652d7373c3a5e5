"""BASELINE config C1 — MobileNetV1 fp32, 1x3x224x224, through the reference's x86 CPU path, restated (TEST / BASELINE
INFRASTRUCTURE ONLY: imported by tests/ and by bench.py's cpu_baseline leg; the product path never imports it).

What the reference runs for this config (lite/api/test_mobilenetv1_lite_x86.cc:30-80): CxxPredictor with
Place{kX86, kFloat}; input all ones; per conv layer `conv2d` / `depthwise_conv2d` = im2col + cblas_sgemm per (image, group)
with NO bias / activation in the kernel (lite/kernels/x86/conv_compute.h:48-150), then `batch_norm` and `relu` as separate
fp32 instructions; `pool2d` global average; `mul` + `elementwise_add` (fc); `softmax`.  The reference binary cannot be built
here (MKLML, gflags, protobuf ... are network downloads); the GEMM below is a plain k-ascending loop, so results
agree with the reference's only to fp32 summation-order tolerance: parity UNPINNED beyond 1e-5 relative (stated in DESIGN).

The fp32 model is the synthetic int8 network de-quantised: w_f32 = w_i8 * w_scale[c]; batch_norm carries the conv bias
(scale 1, mean 0, variance 1)."""
import ctypes as C
import time

import numpy as np

from . import plref

EPS = 1e-5


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def conv2d_f32(x, w, stride, pad, groups):
    """x [n,cin,h,w] fp32, w [cout,cin/g,k,k] fp32 -> [n,cout,oh,ow]; the x86 kernel: no bias, no activation."""
    L = plref.lib()
    n, cin, h, wd = x.shape
    cout, cg, k, _ = w.shape
    s = plref.shape(n, cin, h, wd, cout, k, k, (pad, pad, pad, pad), (stride, stride), (1, 1), groups)
    oh, ow = plref.out_dims(s)
    y = np.empty((n, cout, oh, ow), np.float32)
    col = np.empty(max(1, cg * k * k * oh * ow), np.float32)
    x = np.ascontiguousarray(x, np.float32)
    w = np.ascontiguousarray(w, np.float32)
    L.plref_conv2d_f32_x86(C.byref(s), _f(x), _f(w), _f(y), _f(col))
    return y


def batch_norm_relu(x, bias, relu=True):
    L = plref.lib()
    n, c, h, w = x.shape
    y = np.empty_like(x)
    one, zero = np.ones(c, np.float32), np.zeros(c, np.float32)
    bias = np.ascontiguousarray(bias if bias is not None else zero, np.float32)
    L.plref_batch_norm_f32(_f(x), _f(y), n, c, h * w, _f(one), _f(bias), _f(zero), _f(one), C.c_float(EPS), 1 if relu else 0)
    return y


def fp32_model(net):
    """De-quantise the op list once (not timed)."""
    ops = []
    for o in net["ops"]:
        if o["op"] in ("conv2d", "depthwise_conv2d"):
            ws = np.asarray(o["w_scale"], np.float32).reshape(-1)
            wf = o["w"].astype(np.float32) * (ws if ws.size > 1 else ws[0]).reshape(-1, 1, 1, 1)
            ops.append(dict(op="conv", w=np.ascontiguousarray(wf), bias=o["bias"], stride=o["stride"], pad=o["pad"], groups=o["groups"],
                            relu=o["act"] == 1))
        elif o["op"] == "pool2d":
            ops.append(dict(op="gap"))
        elif o["op"] == "fc":
            ws = np.asarray(o["w_scale"], np.float32).reshape(-1)
            ops.append(dict(op="fc", w=np.ascontiguousarray(o["w"].astype(np.float32) * (ws if ws.size > 1 else ws[0])), bias=o["bias"]))
        elif o["op"] == "softmax":
            ops.append(dict(op="softmax"))
        else:
            raise ValueError("C1 covers MobileNetV1's op set; got %s" % o["op"])
    return ops


def forward(model, image):
    L = plref.lib()
    t = np.ascontiguousarray(image, np.float32)
    for o in model:
        if o["op"] == "conv":
            t = batch_norm_relu(conv2d_f32(t, o["w"], o["stride"], o["pad"], o["groups"]), o["bias"], o["relu"])
        elif o["op"] == "gap":
            t = t.mean(axis=(2, 3), dtype=np.float32).reshape(t.shape[0], -1)
        elif o["op"] == "fc":
            x2 = t.reshape(t.shape[0], -1)
            y = np.empty((x2.shape[0], o["w"].shape[1]), np.float32)
            L.plref_sgemm_f32(x2.shape[0], o["w"].shape[1], x2.shape[1], _f(np.ascontiguousarray(x2)), _f(o["w"]), _f(y))
            t = y + (o["bias"] if o["bias"] is not None else 0)
        elif o["op"] == "softmax":
            e = np.exp(t - t.max(axis=1, keepdims=True))
            t = (e / e.sum(axis=1, keepdims=True)).astype(np.float32)
    return t


def time_c1(net, seconds=4.0, warmup=2):
    """The reference's own protocol (all-ones 1x3xHxW input, warm-up then repeats), bounded by `seconds`."""
    c, h, w = net["input_shape"]
    model = fp32_model(net)
    img = np.ones((1, c, h, w), np.float32)
    for _ in range(warmup):
        out = forward(model, img)
    ts = []
    t_end = time.perf_counter() + seconds
    while time.perf_counter() < t_end or len(ts) < 3:
        t0 = time.perf_counter()
        out = forward(model, img)
        ts.append(time.perf_counter() - t0)
    return {"avg_ms": round(1e3 * float(np.mean(ts)), 2), "min_ms": round(1e3 * float(np.min(ts)), 2), "repeats": len(ts),
            "prob_sum": float(out.sum()), "top1": int(out.argmax())}

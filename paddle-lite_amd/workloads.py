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

"""Oracle (CPU) forward of the synthetic MobileNetV1-INT8 graph, op by op, with the reference semantics."""
import numpy as np


def forward(plref, wl, W, image, res=224, upto=None, via_gemm=False):
    """Returns dict name -> tensor for every variable of the Appendix-D program."""
    out = {}
    x = plref.calib_f32_to_i8(image, float(W["input_scale"]))
    out["x0"] = x
    layers = wl.mobilenet_v1_layers(res)
    for i, (name, op, cin, cout, k, s, p, g, hin) in enumerate(layers):
        L = W[name]
        last = i == len(layers) - 1
        sh = plref.shape(x.shape[0], cin, x.shape[2], x.shape[3], cout, k, k, (p, p, p, p), (s, s), (1, 1), g)
        y, _ = plref.conv2d(sh, x, L["w"], L["bias"], float(L["in_scale"]), L["w_scale"], float(L["out_scale"]), 1, 0.0,
                            not last, via_gemm=(via_gemm and g == 1))
        out[name] = y
        x = y
        if upto == name:
            return out
    pool = plref.global_avg_pool(x)
    out["pool"] = pool
    q = plref.calib_f32_to_i8(pool, float(W["pool_scale"]))
    out["pool_i8"] = q
    F = W["fc"]
    sc = (F["w_scale"] * np.float32(F["in_scale"])).astype(np.float32)
    logits, _ = plref.fc(q.reshape(q.shape[0], -1), F["w"], F["bias"], sc, False, False)
    out["logits"] = logits
    out["prob"] = plref.softmax(logits)
    return out

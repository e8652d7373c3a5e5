#!/usr/bin/env python3
"""Mint the golden vectors under tests/golden/ (run in the authoring container only).

Provenance of each array in a fixture:
  x, w, bias, scales      seeded numpy RNG (seed stored), int8 in [-127, 127] as the reference's
                          tests use (lite/tests/utils/tensor_utils.h:143-151)
  acc_ref                 the REFERENCE ITSELF: conv_basic<int8_t,int> / basic_gemm<int8_t,int> of
                          lite/tests/utils/naive_math_impl.h compiled in place (oracle/_ref)
  f32_baseline            the reference's float baseline conv_basic<float,float> on de-quantised
                          inputs (methodology of conv_int8_compute_test.cc:298-327)
  y_f32, y_i8             our epilogue restatement (oracle/plref.c) applied to acc_ref — the ARM
                          epilogue (NEON asm) cannot run on this x86 host, so these two are pinned
                          only through the reference's own +-1 LSB rule against f32_baseline, which
                          tests/test_oracle.py asserts.
The reference has no committed golden vectors for the int8 path, hence this script.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import plref  # noqa: E402

# name: (n, cin, h, w, cout, kh, kw, pads(t,b,l,r), stride, dil, groups, act, alpha, has_bias)
CONV_CASES = {
    "dw3x3_s1_p1_relu":   (2, 8, 15, 15, 8, 3, 3, (1, 1, 1, 1), 1, 1, 8, 1, 0.0, True),
    "dw3x3_s2_p0_none":   (1, 5, 33, 33, 5, 3, 3, (0, 0, 0, 0), 2, 1, 5, 0, 0.0, False),
    "dw3x3_s2_p1_relu6":  (2, 16, 14, 14, 16, 3, 3, (1, 1, 1, 1), 2, 1, 16, 2, 6.0, True),
    "dw5x5_s1_p2_relu6":  (1, 5, 15, 15, 5, 5, 5, (2, 2, 2, 2), 1, 1, 5, 2, 6.0, True),
    "dw5x5_s2_p1_leaky":  (2, 3, 33, 33, 3, 5, 5, (1, 1, 1, 1), 2, 1, 3, 4, 0.25, True),
    "pw1x1_leaky":        (2, 33, 16, 16, 17, 1, 1, (0, 0, 0, 0), 1, 1, 1, 4, 0.5, True),
    "pw1x1_g2_none":      (2, 8, 9, 9, 6, 1, 1, (0, 0, 0, 0), 1, 1, 2, 0, 0.0, False),
    "pw1x1_hw49_relu":    (3, 64, 7, 7, 40, 1, 1, (0, 0, 0, 0), 1, 1, 1, 1, 0.0, True),
    "c3x3_s1_asym_relu":  (2, 8, 17, 17, 5, 3, 3, (1, 2, 2, 1), 1, 1, 1, 1, 0.0, True),
    "c3x3_s2_asym_relu6": (1, 3, 19, 19, 33, 3, 3, (1, 2, 1, 2), 2, 1, 1, 2, 6.0, True),
    "c3x3_s2_first":      (1, 3, 32, 32, 32, 3, 3, (1, 1, 1, 1), 2, 1, 1, 1, 0.0, True),
    "rand_k2x3_s2_d2":    (2, 17, 19, 19, 8, 2, 3, (0, 1, 2, 0), 2, 2, 1, 4, 1.5, True),
    "rand_g2_k3_d2":      (1, 8, 13, 11, 6, 3, 3, (2, 0, 1, 2), 1, 2, 2, 1, 0.0, False),
    "c7x7_s2":            (1, 3, 29, 29, 8, 7, 7, (3, 3, 3, 3), 2, 1, 1, 1, 0.0, True),
}


def make_conv(name, spec, seed):
    n, cin, h, w_, cout, kh, kw, pads, st, dl, g, act, alpha, has_bias = spec
    rng = np.random.default_rng(seed)
    s = plref.shape(n, cin, h, w_, cout, kh, kw, pads, (st, st), (dl, dl), g)
    x = rng.integers(-127, 128, (n, cin, h, w_)).astype(np.int8)
    w = rng.integers(-127, 128, (cout, cin // g, kh, kw)).astype(np.int8)
    bias = rng.uniform(-1, 1, cout).astype(np.float32) if has_bias else None
    kk = (cin // g) * kh * kw
    in_scale = np.float32(1.0 / 127)
    # per-channel-varying weight scale — real models arrive like this (A.9)
    w_scale = ((1 + np.arange(cout) % 7) / 127.0 / 4.0).astype(np.float32)
    out_scale = np.float32(kk / 127.0) if act != 2 else np.float32(alpha / 127.0)
    if act == 4 and abs(alpha) > 1:
        out_scale = np.float32(out_scale * abs(alpha))
    acc_ref = plref.ref_conv_acc(s, x, w)
    outs = {}
    for int8_out in (0, 1):
        sc, bi, al = plref.fold_scales(int8_out, in_scale, w_scale, out_scale, bias, cout, act, alpha)
        outs[int8_out] = plref.epilogue(acc_ref, sc, bi, act, al, bool(int8_out))
    # the reference's float baseline on dequantised tensors
    xf = x.astype(np.float32) * in_scale
    wf = w.astype(np.float32) * w_scale[:, None, None, None]
    base = plref.ref_conv_f32(s, xf, wf, bias, act, alpha if act == 2 else 6.0, alpha)
    d = dict(x=x, w=w, pads=np.array(pads, np.int32), stride=np.int32(st), dil=np.int32(dl),
             groups=np.int32(g), act=np.int32(act), alpha=np.float32(alpha),
             in_scale=in_scale, w_scale=w_scale, out_scale=out_scale,
             acc_ref=acc_ref, y_f32=outs[0], y_i8=outs[1], f32_baseline=base, seed=np.int32(seed))
    if bias is not None:
        d["bias"] = bias
    np.savez_compressed(os.path.join(HERE, "conv_%s.npz" % name), **d)
    return acc_ref.size


def make_gemm(seed):
    rng = np.random.default_rng(seed)
    for (m, n, k) in [(35, 141, 61), (1, 13, 3), (33, 512, 71), (397, 3, 8)]:
        a = rng.integers(-127, 128, (m, k)).astype(np.int8)
        b = rng.integers(-127, 128, (k, n)).astype(np.int8)
        c = plref.ref_gemm_acc(a, b)
        np.savez_compressed(os.path.join(HERE, "gemm_m%d_n%d_k%d.npz" % (m, n, k)), a=a, b=b, acc_ref=c,
                            seed=np.int32(seed))


def make_fc(seed):
    rng = np.random.default_rng(seed)
    m, k, n = 3, 67, 13
    x = rng.integers(-127, 128, (m, k)).astype(np.int8)
    w = rng.integers(-127, 128, (k, n)).astype(np.int8)
    bias = rng.uniform(-1, 1, n).astype(np.float32)
    # FC accumulators == basic_gemm (reference) on the same operands
    acc = plref.ref_gemm_acc(x, w)
    scale = ((1 + np.arange(n) % 5) / 127.0 / 127.0).astype(np.float32)
    y, _ = plref.fc(x, w, bias, scale, True, False)
    scale8 = (scale / np.float32(k / 127.0)).astype(np.float32)
    y8, _ = plref.fc(x, w, bias / np.float32(k / 127.0), scale8, True, True)
    np.savez_compressed(os.path.join(HERE, "fc_m3_k67_n13.npz"), x=x, w=w, bias=bias, scale=scale,
                        scale8=scale8, bias8=(bias / np.float32(k / 127.0)).astype(np.float32),
                        acc_ref=acc, y_f32=y, y_i8=y8, seed=np.int32(seed))


def make_calib(seed):
    rng = np.random.default_rng(seed)
    x = np.concatenate([rng.uniform(-3, 3, 1000).astype(np.float32),
                        np.array([0.5, -0.5, 1.5, 2.5, -2.5, 0.49999997, -0.49999997, 126.5, 127.5,
                                  -127.5, -128.5, 1e9, -1e9, 0.0], np.float32)])
    scale = np.float32(1.0 / 63.5)
    q = plref.calib_f32_to_i8(x, scale)
    xf = plref.calib_i8_to_f32(q, scale)
    np.savez_compressed(os.path.join(HERE, "calib.npz"), x=x, scale=scale, q=q, deq=xf, seed=np.int32(seed))


if __name__ == "__main__":
    assert plref.ref_lib() is not None, "build oracle/_ref first (make -C oracle ref)"
    tot = 0
    for i, (name, spec) in enumerate(sorted(CONV_CASES.items())):
        tot += make_conv(name, spec, 1000 + i)
    make_gemm(2000)
    make_fc(3000)
    make_calib(4000)
    print("golden fixtures written to", HERE, "conv output elements:", tot)

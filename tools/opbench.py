#!/usr/bin/env python3
"""Per-op micro-benchmark through the C ABI (device-resident buffers, HIP events on the ctx stream).
Usage: python tools/opbench.py [pw|dw|conv1|s2|all|<layer name>] [--batch 128] [--reps 30] [--net mobilenet_v1|dw5x5|resnet50_3x3]
Prints one line per layer: time, algorithmic GB/s, TOP/s.  --net dw5x5: depthwise 5x5 stride 1 / 2 planes (the shapes of
lite/tests/math/conv_int8_compute_test.cc's 5x5 depthwise sweep at network sizes); --net resnet50_3x3: BASELINE config #2
(at its own batch 32) and ResNet50's dense 3x3 layers."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.import_package()
capi = pkg.capi
import importlib  # noqa: E402

wl = importlib.import_module("paddle_lite_amd.workloads")


def time_op(ctx, fn, reps):
    L = ctx.L
    for _ in range(3):
        fn()
    e0, e1 = C.c_void_p(), C.c_void_p()
    L.plhip_event_create(ctx.h, C.byref(e0))
    L.plhip_event_create(ctx.h, C.byref(e1))
    L.plhip_event_record(ctx.h, e0)
    for _ in range(reps):
        fn()
    L.plhip_event_record(ctx.h, e1)
    ms = C.c_float()
    ctx.check(L.plhip_event_elapsed_ms(ctx.h, e0, e1, C.byref(ms)), "elapsed")
    return ms.value / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="?", default="all")
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--f32", action="store_true", help="fp32 output instead of int8")
    ap.add_argument("--net", default="mobilenet_v1")
    args = ap.parse_args()
    rng = np.random.default_rng(0)
    B = args.batch
    tot = 0.0
    with capi.Context(0) as ctx:
        L = ctx.L
        if args.net == "dw5x5":
            layers = [("dw5_%d_%d_s%d" % (c, h, st), "depthwise_conv2d", c, c, 5, st, 2, c, h)
                      for (c, h, st) in [(32, 112, 1), (64, 112, 2), (128, 56, 1), (128, 56, 2), (256, 28, 1), (256, 28, 2), (512, 14, 1),
                                         (512, 14, 2), (1024, 7, 1)]]
        elif args.net == "resnet50_3x3":
            layers = [("c2", "conv2d", 64, 128, 3, 1, 1, 1, 56), ("res2", "conv2d", 64, 64, 3, 1, 1, 1, 56), ("res3", "conv2d", 128, 128, 3, 1, 1, 1, 28),
                      ("res4", "conv2d", 256, 256, 3, 1, 1, 1, 14), ("res5", "conv2d", 512, 512, 3, 1, 1, 1, 7), ("res3a", "conv2d", 128, 128, 3, 2, 1, 1, 56),
                      ("res4a", "conv2d", 256, 256, 3, 2, 1, 1, 28), ("res5a", "conv2d", 512, 512, 3, 2, 1, 1, 14),
                      ("stem7", "conv2d", 3, 64, 7, 2, 3, 1, 224)]
        else:
            layers = wl.mobilenet_v1_layers()
        for (name, op, cin, cout, k, s, p, g, hin) in layers:
            kind = "conv1" if name == "conv1" else ("dw" if g > 1 else ("pw" if k == 1 else "conv"))
            if args.what not in ("all", kind, name) and not (args.what == "s2" and kind == "conv" and s == 2):
                continue
            B = 32 if name == "c2" else args.batch
            ho = (hin + 2 * p - k) // s + 1
            d = capi.conv_desc(B, cin, hin, hin, cout, k, k, (p, p, p, p), (s, s), (1, 1), g, capi.ACT_RELU, 0.0)
            x = rng.integers(-127, 128, (B, cin, hin, hin), dtype=np.int8)
            w = rng.integers(-127, 128, (cout, cin // g, k, k), dtype=np.int8)
            dx, dw = ctx.to_device(x), ctx.to_device(w)
            ds = ctx.to_device(np.full(cout, 1e-4, np.float32))
            db = ctx.to_device(np.zeros(cout, np.float32))
            out_kind = capi.OUT_F32 if (args.f32 or name == "pw14") else capi.OUT_I8
            esz = 4 if out_kind == capi.OUT_F32 else 1
            dy = ctx.malloc(B * cout * ho * ho * esz)
            if g > 1:
                fn = lambda: ctx.check(L.plhip_depthwise_conv_int8(ctx.h, C.byref(d), dx, dw, ds, db, dy, out_kind), "dw")
            else:
                dwp = ctx.malloc(L.plhip_conv_packed_weight_bytes(C.byref(d)))
                ctx.check(L.plhip_pack_conv_weights(ctx.h, C.byref(d), dw, dwp), "pack")
                wsb = L.plhip_conv_workspace_bytes(C.byref(d))
                dws = ctx.malloc(wsb) if wsb else C.c_void_p()
                fn = lambda: ctx.check(L.plhip_conv2d_int8(ctx.h, C.byref(d), dx, dwp, ds, db, dy, out_kind, dws, wsb), "conv")
            ms = time_op(ctx, fn, args.reps)
            macs = B * ho * ho * cout * (cin // g) * k * k
            byts = B * (cin * hin * hin + cout * ho * ho * esz) + w.size
            tot += ms
            print("%-6s %-4s %4d->%4d %3dx%-3d s%d  %8.2f us  %7.1f GB/s  %7.1f TOP/s  %s" % (
                name, kind, cin, cout, hin, hin, s, ms * 1e3, byts / ms / 1e6, 2 * macs / ms / 1e9,
                L.plhip_conv_impl_name(C.byref(d)).decode() if g == 1 else "depthwise"), flush=True)
            for q in list(ctx._allocs):
                ctx.free(q)
    if args.what in ("fused", "all") and args.net == "mobilenet_v1":
        ftot = 0.0
        with capi.Context(0) as ctx:
            L = ctx.L
            layers = wl.mobilenet_v1_layers()
            for i in range(1, len(layers), 2):
                (dn, _, c, _, k, s, p, g, hin) = layers[i]
                (pn, _, _, m, _, _, _, _, hmid) = layers[i + 1]
                d = capi.conv_desc(B, c, hin, hin, c, 3, 3, (p, p, p, p), (s, s), (1, 1), c, capi.ACT_RELU, 0.0)
                dp = capi.conv_desc(B, c, hmid, hmid, m, 1, 1, act=capi.ACT_RELU)
                dx = ctx.to_device(rng.integers(-127, 128, (B, c, hin, hin), dtype=np.int8))
                dwd = ctx.to_device(rng.integers(-127, 128, (c, 1, 3, 3), dtype=np.int8))
                dsd = ctx.to_device(np.full(c, 1e-2, np.float32))
                dwr = ctx.to_device(rng.integers(-127, 128, (m, c, 1, 1), dtype=np.int8))
                dwp = ctx.malloc(L.plhip_conv_packed_weight_bytes(C.byref(dp)))
                ctx.check(L.plhip_pack_conv_weights(ctx.h, C.byref(dp), dwr, dwp), "pack")
                dsp = ctx.to_device(np.full(m, 1e-4, np.float32))
                out_kind = capi.OUT_F32 if pn == "pw14" else capi.OUT_I8
                esz = 4 if out_kind == capi.OUT_F32 else 1
                dy = ctx.malloc(B * m * hmid * hmid * esz)
                how = "fused"
                if L.plhip_dwpw_fused_supported(C.byref(d), m, out_kind):
                    fn = lambda: ctx.check(L.plhip_dwpw_fused_int8(ctx.h, C.byref(d), dx, dwd, dsd, None, m, dwp, dsp, None,
                                                                   capi.ACT_RELU, 0.0, dy, out_kind), "fused")
                else:  # outside the fused path: what the predictor runs instead, the two kernels
                    how = "2-krn"
                    dmid = ctx.malloc(B * c * hmid * hmid)

                    def fn():
                        ctx.check(L.plhip_depthwise_conv_int8(ctx.h, C.byref(d), dx, dwd, dsd, None, dmid, capi.OUT_I8), "dw")
                        ctx.check(L.plhip_conv2d_int8(ctx.h, C.byref(dp), dmid, dwp, dsp, None, dy, out_kind, None, 0), "pw")
                ms = time_op(ctx, fn, args.reps)
                byts = B * (c * hin * hin + m * hmid * hmid * esz)
                macs = B * hmid * hmid * (m * c + 9 * c)
                ftot += ms
                print("%-4s+%-5s %s %4d->%4d %3dx%-3d s%d  %8.2f us  %7.1f GB/s  %7.1f TOP/s" % (
                    dn, pn, how, c, m, hin, hin, s, ms * 1e3, byts / ms / 1e6, 2 * macs / ms / 1e9), flush=True)
                for q in list(ctx._allocs):
                    ctx.free(q)
        print("fused total %.2f us" % (ftot * 1e3))
    print("total %.2f us" % (tot * 1e3))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Timeline of one fused depthwise->pointwise launch from in-kernel s_memtime stamps (PLHIP_FUSED_DEBUG & 32).
Usage: PLHIP_FUSED_DEBUG=32 python tools/fused_timeline.py [dw layer=dw8] [--batch 128]"""
import argparse
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.import_package()
capi = pkg.capi
wl = importlib.import_module("paddle_lite_amd.workloads")
SLOTS = 32


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("layer", nargs="?", default="dw8")
    ap.add_argument("--batch", type=int, default=128)
    args = ap.parse_args()
    assert int(os.environ.get("PLHIP_FUSED_DEBUG", "0")) & 32, "run with PLHIP_FUSED_DEBUG=32 (+ experiment bits)"
    rng = np.random.default_rng(0)
    B = args.batch
    layers = wl.mobilenet_v1_layers()
    with capi.Context(0) as ctx:
        L = ctx.L
        for i, (name, op, cin, cout, k, s, p, g, hin) in enumerate(layers):
            if name != args.layer:
                continue
            m = layers[i + 1][3]
            ho = (hin + 2 * p - k) // s + 1
            d = capi.conv_desc(B, cin, hin, hin, cin, 3, 3, (p, p, p, p), (s, s), (1, 1), cin, capi.ACT_RELU, 0.0)
            dpw = capi.conv_desc(B, cin, ho, ho, m, 1, 1)
            x = rng.integers(-127, 128, (B, cin, hin, hin), dtype=np.int8)
            dx = ctx.to_device(x)
            dwd = ctx.to_device(rng.integers(-127, 128, (cin, 1, 3, 3), dtype=np.int8))
            dsd = ctx.to_device(np.full(cin, 1e-3, np.float32))
            wpw = ctx.to_device(rng.integers(-127, 128, (m, cin, 1, 1), dtype=np.int8))
            dwp = ctx.malloc(L.plhip_conv_packed_weight_bytes(C.byref(dpw)))
            ctx.check(L.plhip_pack_conv_weights(ctx.h, C.byref(dpw), wpw, dwp), "pack")
            dsp = ctx.to_device(np.full(m, 1e-4, np.float32))
            dy = ctx.malloc(B * m * ho * ho)
            for _ in range(20):
                ctx.check(L.plhip_dwpw_fused_int8(ctx.h, C.byref(d), dx, dwd, dsd, None, m, dwp, dsp, None, capi.ACT_RELU, 0.0, dy,
                                                  capi.OUT_I8), "fused")
            ctx.sync()
            buf = np.zeros(1024 * 8 * SLOTS, np.uint64)
            rd = L.plhip_debug_read_fz_stamps
            rd.argtypes = [C.c_void_p, C.c_size_t]
            rd.restype = C.c_int
            assert rd(buf.ctypes.data, buf.nbytes) == 0
            st = buf.reshape(1024, 8, SLOTS).astype(np.int64)
            st = st[st[:, 0, 1] != 0]
            print("%s + pw: C=%d M=%d %dx%d s%d batch %d; blocks with stamps: %d" % (name, cin, m, ho, ho, s, B, st.shape[0]))
            rt0, rt1 = st[:, 0, 0], st[:, :, 31].max(axis=1)
            print("kernel span %.2f us; block start offsets us p50 %.2f max %.2f; block lifetime us p10 %.2f p50 %.2f p90 %.2f" % (
                (rt1.max() - rt0.min()) / 100.0, np.median(rt0 - rt0.min()) / 100.0, (rt0.max() - rt0.min()) / 100.0,
                *np.percentile((rt1 - rt0) / 100.0, [10, 50, 90])))
            t = st.reshape(-1, SLOTS)

            def show(label, a):
                print("  %-36s cyc p10 %7.0f  p50 %7.0f  p90 %7.0f" % ((label,) + tuple(np.percentile(a, [10, 50, 90]))))

            ks = (cin + 31) // 32
            show("entry -> dw parameters visible", t[:, 2] - t[:, 1])
            show("first K-step produced", t[:, 3] - t[:, 2])
            nk = min(ks, 16)
            for i in range(nk - 1):
                show("K-step %d" % i, t[:, 5 + i] - t[:, 4 + i])
            if int(os.environ.get("PLHIP_FUSED_DEBUG", "0")) & 64 and ks > 7:
                show("  K-step 6: top -> counted vmcnt wait done", t[:, 20] - t[:, 10])
                show("  K-step 6: lgkm wait + barrier", t[:, 21] - t[:, 20])
                show("  K-step 6: tr reads, DMA, weight loads issued", t[:, 22] - t[:, 21])
                show("  K-step 6: 8 MFMAs issued", t[:, 23] - t[:, 22])
                show("  K-step 6: produce (reads, arithmetic, writes)", t[:, 24] - t[:, 23])
                show("  K-step 6: -> next top", t[:, 11] - t[:, 24])
            show("last stamped K-step top -> loop end", t[:, 26] - t[:, 4 + nk - 1])
            show("whole loop", t[:, 26] - t[:, 4])
            show("scale/bias + requantise + stage", t[:, 27] - t[:, 26])
            show("read back + stores issued", t[:, 28] - t[:, 27])
            show("store drain", t[:, 29] - t[:, 28])
            show("wave total", t[:, 29] - t[:, 1])
            clk = (t[:, 29] - t[:, 1]).astype(np.float64) / np.maximum(1, (t[:, 31] - t[:, 0])) / 10.0
            print("  shader clock over wave lifetime: median %.2f GHz" % np.median(clk))


if __name__ == "__main__":
    main()

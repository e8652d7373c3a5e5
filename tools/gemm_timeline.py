#!/usr/bin/env python3
"""Timeline of one LDS-DMA GEMM launch from in-kernel s_memtime stamps (diagnostic build path PLHIP_GEMM_DEBUG=32).
Usage: PLHIP_GEMM_DEBUG=32 python tools/gemm_timeline.py [layer=pw8] [--batch 128]
Prints, per phase, the median / p10 / p90 over waves, and the block start-time spread (dispatch ramp)."""
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
SLOTS = 32


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("layer", nargs="?", default="pw8")
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--tr", action="store_true", help="the transposed-read ring kernel (gemm_tr_i8.hip): 8 waves per block")
    ap.add_argument("--conv", default=None, help="cin,cout,hw,k,stride: a dense conv of that shape instead of a MobileNetV1 layer (implies --tr)")
    args = ap.parse_args()
    if args.conv:
        args.tr = True
    WPB = 8 if args.tr else 4
    assert int(os.environ.get("PLHIP_GEMM_DEBUG", "0")) & 32, "run with PLHIP_GEMM_DEBUG=32 (or 33, 34 ...)"
    rng = np.random.default_rng(0)
    B = args.batch
    with capi.Context(0) as ctx:
        L = ctx.L
        layers = wl.mobilenet_v1_layers()
        if args.conv:
            cin, cout, hin, k, s = [int(v) for v in args.conv.split(",")]
            layers = [("conv", "conv2d", cin, cout, k, s, k // 2, 1, hin)]
            args.layer = "conv"
        for (name, op, cin, cout, k, s, p, g, hin) in layers:
            if name != args.layer:
                continue
            ho = (hin + 2 * p - k) // s + 1
            d = capi.conv_desc(B, cin, hin, hin, cout, k, k, (p, p, p, p), (s, s), (1, 1), g, capi.ACT_RELU, 0.0)
            x = rng.integers(-127, 128, (B, cin, hin, hin), dtype=np.int8)
            w = rng.integers(-127, 128, (cout, cin // g, k, k), dtype=np.int8)
            dx, dw = ctx.to_device(x), ctx.to_device(w)
            ds = ctx.to_device(np.full(cout, 1e-4, np.float32))
            db = ctx.to_device(np.zeros(cout, np.float32))
            dy = ctx.malloc(B * cout * ho * ho)
            dwp = ctx.malloc(L.plhip_conv_packed_weight_bytes(C.byref(d)))
            ctx.check(L.plhip_pack_conv_weights(ctx.h, C.byref(d), dw, dwp), "pack")
            wsb = L.plhip_conv_workspace_bytes(C.byref(d))
            dws = ctx.malloc(wsb) if wsb else None
            for _ in range(20):  # warm clocks and caches; the stamps of the last launch stay
                ctx.check(L.plhip_conv2d_int8(ctx.h, C.byref(d), dx, dwp, ds, db, dy, capi.OUT_I8, dws, wsb), "conv")
            ctx.sync()
            hwp = (ho * ho + 15) // 16 * 16
            nblk = 1024 if args.conv else min(1024, ((cout + 255) // 256) * ((B * hwp + 127) // 128 + 7) // 8 * 8)
            buf = np.zeros(1024 * WPB * SLOTS, np.uint64)
            rd = L.plhip_debug_read_tr_stamps if args.tr else L.plhip_debug_read_stamps
            rd.argtypes = [C.c_void_p, C.c_size_t]
            rc = rd(buf.ctypes.data, buf.nbytes)
            assert rc == 0, rc
            st = buf.reshape(1024, WPB, SLOTS)[:nblk].astype(np.int64)
            live = st[:, :, 1] != 0
            st = st[live[:, 0]]
            print("blocks with stamps:", st.shape[0], "of", nblk)
            ks = (cin * k * k // g + 31) // 32
            rt0 = st[:, 0, 0]
            rt1 = st[:, :, SLOTS - 1].max(axis=1)
            print("realtime (100 MHz ticks): first start %d, last start +%d, last end +%d  => kernel span %.2f us" % (
                0, rt0.max() - rt0.min(), rt1.max() - rt0.min(), (rt1.max() - rt0.min()) / 100.0))
            order = np.argsort(rt0)
            starts = (rt0[order] - rt0.min()) / 100.0
            print("block start offsets us: p10 %.2f p50 %.2f p90 %.2f max %.2f" % tuple(np.percentile(starts, [10, 50, 90, 100])))
            dur = (rt1 - rt0) / 100.0
            print("block lifetime us: p10 %.2f p50 %.2f p90 %.2f max %.2f" % tuple(np.percentile(dur, [10, 50, 90, 100])))
            t = st[:, :, :].reshape(-1, SLOTS)

            def show(label, a):
                print("  %-34s cyc p10 %7.0f  p50 %7.0f  p90 %7.0f" % ((label,) + tuple(np.percentile(a, [10, 50, 90]))))

            show("entry -> prologue issued", t[:, 3] - t[:, 1])
            show("prologue -> loop top (first data)", t[:, 4] - t[:, 3])
            if args.tr:
                nk = min(ks, SLOTS - 9)
                show("loop top -> K-step 0 top (first reads)", t[:, 5] - t[:, 4])
                for i in range(nk - 1):
                    show("K-step %d" % i, t[:, 6 + i] - t[:, 5 + i])
                lastk = 5 + nk - 1
            else:
                for i in range(min(ks, SLOTS - 8) - 1):
                    show("K-step %d" % i, t[:, 5 + i] - t[:, 4 + i])
                lastk = 4 + min(ks, SLOTS - 8) - 1
            if args.tr and int(os.environ.get("PLHIP_GEMM_DEBUG", "0")) & 64:
                show("  K-step 6: top -> vmcnt wait done", t[:, 22] - t[:, 11])
                show("  K-step 6: barrier", t[:, 23] - t[:, 22])
                show("  K-step 6: 10 LDS reads issued", t[:, 24] - t[:, 23])
                show("  K-step 6: DMA pieces issued", t[:, 25] - t[:, 24])
                show("  K-step 6: 8 MFMAs issued", t[:, 21] - t[:, 25])
                show("  K-step 6: -> next top (lgkmcnt wait)", t[:, 12] - t[:, 21])
            elif int(os.environ.get("PLHIP_GEMM_DEBUG", "0")) & 64:
                show("  K-step 6: top -> vmcnt wait done", t[:, 20] - t[:, 10])
                show("  K-step 6: barrier", t[:, 21] - t[:, 20])
                show("  K-step 6: LDS reads issued + returned", t[:, 22] - t[:, 21])
                show("  K-step 6: DMA issue + MFMAs + perms", t[:, 23] - t[:, 22])
                show("  K-step 6: end -> next top", t[:, 11] - t[:, 23])
            show("last K-step -> loop end", t[:, SLOTS - 4] - t[:, lastk])
            show("whole loop", t[:, SLOTS - 4] - t[:, 4])
            if args.tr:
                show("  loop end -> barrier passed", t[:, SLOTS - 6] - t[:, SLOTS - 4])
                show("  requantise + stage in LDS", t[:, SLOTS - 5] - t[:, SLOTS - 6])
                show("  read back + global stores issued", t[:, SLOTS - 3] - t[:, SLOTS - 5])
            show("epilogue issue", t[:, SLOTS - 3] - t[:, SLOTS - 4])
            show("store drain (vmcnt 0)", t[:, SLOTS - 2] - t[:, SLOTS - 3])
            show("wave total", t[:, SLOTS - 2] - t[:, 1])
            clk = (t[:, SLOTS - 2] - t[:, 1]).astype(np.float64) / np.maximum(1, (t[:, SLOTS - 1] - t[:, 0])) / 10.0
            print("  shader clock over wave lifetime: median %.2f GHz" % np.median(clk * 1.0))
            hw = st[:, 0, 2]
            xcc = (hw >> 32) & 0xF
            cu = ((hw & 0xFFFFFFFF) >> 8) & 0xF
            se = ((hw & 0xFFFFFFFF) >> 13) & 0x7
            print("  blocks per XCC:", np.bincount(xcc, minlength=8).tolist())
            key = xcc * 1000 + se * 16 + cu
            cnt = np.bincount(np.unique(key, return_inverse=True)[1])
            print("  distinct (xcc,se,cu): %d; blocks per CU histogram: %s" % (len(cnt), np.bincount(cnt).tolist()))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Timeline of one wide-tile GEMM launch (gemm_wide_i8.hip) from in-kernel s_memtime stamps (PLHIP_GEMM_DEBUG=32).
Usage: PLHIP_GEMM_DEBUG=32 python tools/wide_timeline.py [layer=pw8] [--batch 128]"""
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
SLOTS, WPB, NBLK = 48, 8, 512


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("layer", nargs="?", default="pw8")
    ap.add_argument("--batch", type=int, default=128)
    args = ap.parse_args()
    assert int(os.environ.get("PLHIP_GEMM_DEBUG", "0")) & 32, "run with PLHIP_GEMM_DEBUG=32"
    rng = np.random.default_rng(0)
    B = args.batch
    with capi.Context(0) as ctx:
        L = ctx.L
        for (name, op, cin, cout, k, s, p, g, hin) in wl.mobilenet_v1_layers():
            if name != args.layer:
                continue
            ho = (hin + 2 * p - k) // s + 1
            d = capi.conv_desc(B, cin, hin, hin, cout, k, k, (p, p, p, p), (s, s), (1, 1), g, capi.ACT_RELU, 0.0)
            x = rng.integers(-127, 128, (B, cin, hin, hin), dtype=np.int8)
            w = rng.integers(-127, 128, (cout, cin // g, k, k), dtype=np.int8)
            dx, dw = ctx.to_device(x), ctx.to_device(w)
            ds = ctx.to_device(np.full(cout, 1e-4, np.float32))
            db = ctx.to_device(np.zeros(cout, np.float32))
            dy = ctx.malloc(B * cout * ho * ho * (4 if name == "pw14" else 1))
            dwp = ctx.malloc(L.plhip_conv_packed_weight_bytes(C.byref(d)))
            ctx.check(L.plhip_pack_conv_weights(ctx.h, C.byref(d), dw, dwp), "pack")
            for _ in range(20):  # warm clocks and caches; the stamps of the last launch stay
                ctx.check(L.plhip_conv2d_int8(ctx.h, C.byref(d), dx, dwp, ds, db, dy, (capi.OUT_F32 if name == "pw14" else capi.OUT_I8), None, 0), "conv")
            ctx.sync()
            buf = np.zeros(NBLK * WPB * SLOTS, np.uint64)
            rd = L.plhip_debug_read_wide_stamps
            rd.argtypes = [C.c_void_p, C.c_size_t]
            assert rd(buf.ctypes.data, buf.nbytes) == 0
            st = buf.reshape(NBLK, WPB, SLOTS).astype(np.int64)
            st = st[st[:, 0, 1] != 0]
            print("blocks with stamps:", st.shape[0])
            ks = cin // 32
            rt0, rt1 = st[:, 0, 0], st[:, :, 9].max(axis=1)
            print("kernel span %.2f us; block starts p50 %.2f max %.2f us; block lifetime p10 %.2f p50 %.2f p90 %.2f max %.2f us" % (
                (rt1.max() - rt0.min()) / 100.0, np.median(rt0 - rt0.min()) / 100.0, (rt0.max() - rt0.min()) / 100.0,
                *np.percentile((rt1 - rt0) / 100.0, [10, 50, 90, 100])))
            t = st.reshape(-1, SLOTS)

            def show(label, a):
                print("  %-40s cyc p10 %7.0f  p50 %7.0f  p90 %7.0f" % ((label,) + tuple(np.percentile(a, [10, 50, 90]))))

            ntt = int(os.environ.get("PLHIP_WIDE_NTT", "0")) or (4 if cin >= 1024 or ho < 14 else 7)
            deep = ks >= 16
            tk = ntt if deep else (2 if ntt >= 6 else 1)
            s1 = min(ks, (ks - 4 + 1) // 2 + 2) if deep else ks
            show("entry -> first loads issued", t[:, 3] - t[:, 1])
            show("-> K-step 0 landed everywhere", t[:, 4] - t[:, 3])
            for i in range(min(s1, 20) - 1):
                show("phase 1 (tiles 0-%d) K-step %d" % (tk - 1, i), t[:, 11 + i] - t[:, 10 + i])
            show("whole phase 1 (%d K-steps, %d tiles K-outer)" % (s1, tk), t[:, 5] - t[:, 4])
            prev = t[:, 5]
            for i in range(0 if s1 < ks else tk, ntt):
                show("phase 2 tile %d: %d MFMAs + slices" % (i, ks - (s1 if i < tk else 0)), t[:, 30 + i] - prev)
                prev = t[:, 30 + i]
            show("whole phase 2", t[:, 7] - t[:, 5])
            show("remaining epilogue + store drain", t[:, 8] - t[:, 7])
            show("wave total", t[:, 8] - t[:, 1])
            clk = (t[:, 8] - t[:, 1]).astype(np.float64) / np.maximum(1, (t[:, 9] - t[:, 0])) / 10.0
            print("  shader clock over wave lifetime: median %.2f GHz" % np.median(clk))
            hw = st[:, 0, 2]
            xcc = (hw >> 32) & 0xF
            cu = ((hw & 0xFFFFFFFF) >> 8) & 0xF
            se = ((hw & 0xFFFFFFFF) >> 13) & 0x7
            key = xcc * 1000 + se * 16 + cu
            cnt = np.bincount(np.unique(key, return_inverse=True)[1])
            print("  blocks per XCC %s; distinct CUs %d; blocks per CU histogram %s" % (
                np.bincount(xcc, minlength=8).tolist(), len(cnt), np.bincount(cnt).tolist()))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Timeline of one fused depthwise -> pointwise launch from in-kernel s_memtime stamps (plhip_debug_set("fused_stamps", 1)).
Usage: python tools/fused_timeline.py [--batch 128] [--c 512] [--m 512]"""
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
NAMES = ["rt0", "entry", "operands requested", "round 0 produced", "round 1 done", "barrier", "round 2 done", "barrier", "round 3 done", "barrier",
         "round 4 done", "barrier", "last K-steps multiplied", "requantised + staged", "stores issued", "stores acknowledged"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--c", type=int, default=512)
    ap.add_argument("--m", type=int, default=512)
    ap.add_argument("--exp", type=int, default=0, help="timing experiment: 1 = rounds without MFMAs, 2 = rounds without depthwise arithmetic")
    args = ap.parse_args()
    B, c, m = args.batch, args.c, args.m
    rng = np.random.default_rng(0)
    with capi.Context(0) as ctx:
        L = ctx.L
        L.plhip_debug_set.argtypes = [C.c_char_p, C.c_int]
        L.plhip_debug_read_fw_stamps.argtypes = [C.c_void_p, C.c_size_t]
        d = capi.conv_desc(B, c, 14, 14, c, 3, 3, (1, 1, 1, 1), (1, 1), (1, 1), c, capi.ACT_RELU, 0.0)
        dp = capi.conv_desc(B, c, 14, 14, m, 1, 1, act=capi.ACT_RELU)
        dx = ctx.to_device(rng.integers(-127, 128, (B, c, 14, 14), dtype=np.int8))
        dwd = ctx.to_device(rng.integers(-127, 128, (c, 1, 3, 3), dtype=np.int8))
        dsd = ctx.to_device(np.full(c, 1e-2, np.float32))
        dwr = ctx.to_device(rng.integers(-127, 128, (m, c, 1, 1), dtype=np.int8))
        dwp = ctx.malloc(L.plhip_conv_packed_weight_bytes(C.byref(dp)))
        ctx.check(L.plhip_pack_conv_weights(ctx.h, C.byref(dp), dwr, dwp), "pack")
        dsp = ctx.to_device(np.full(m, 1e-4, np.float32))
        dy = ctx.malloc(B * m * 196)
        fn = lambda: ctx.check(L.plhip_dwpw_fused_int8(ctx.h, C.byref(d), dx, dwd, dsd, None, m, dwp, dsp, None, capi.ACT_RELU, 0.0, dy, capi.OUT_I8), "fused")
        for _ in range(5):
            fn()
        ctx.sync()
        assert L.plhip_debug_set(b"fused_stamps", 1) == 0
        if args.exp:
            assert L.plhip_debug_set(b"fused_exp", args.exp) == 0
        for _ in range(3):
            fn()
        ctx.sync()
        nt = min(2 * B, 1024)
        st = np.zeros((nt, 8, 16), np.uint64)
        assert L.plhip_debug_read_fw_stamps(st.ctypes.data_as(C.c_void_p), st.nbytes) == 0
        L.plhip_debug_set(b"fused_stamps", 0)
        L.plhip_debug_set(b"fused_exp", 0)
    R = c // 128
    rel = st[:, :, 1:].astype(np.int64) - st[:, :, 1:2].astype(np.int64)  # cycles since the wave's entry
    rt = st[:, :, 0].astype(np.int64)
    print("fused dw3x3 + pw1x1  %d -> %d @14x14, batch %d: %d tiles, %d rounds" % (c, m, B, nt, R))
    print("blocks start within %.2f us (s_memrealtime, 100 MHz)" % ((rt.max() - rt.min()) / 100.0))
    used = [0, 1, 2] + [3 + i for i in range(2 * (R - 1))] + [11, 12, 13, 14]
    prev = None
    for k in used:
        v = rel[:, :, k]
        med = int(np.median(v))
        print("  %-26s median %7d  (min %7d max %7d)%s" % (NAMES[k + 1], med, v.min(), v.max(), "" if prev is None else "   +%d" % (med - prev)))
        prev = med
    # per-wave spread inside a block at the barriers
    for k in used:
        if NAMES[k + 1] == "barrier":
            continue
        sp = (rel[:, :, k].max(axis=1) - rel[:, :, k].min(axis=1))
        print("  spread of the 8 waves at '%s': median %d cycles" % (NAMES[k + 1], int(np.median(sp))))


if __name__ == "__main__":
    main()

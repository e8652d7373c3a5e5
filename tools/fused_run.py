#!/usr/bin/env python3
"""Runs the fused depthwise -> pointwise op N times (for rocprofv3 passes).  Usage: python tools/fused_run.py [--exp E] [--reps N]"""
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
ap = argparse.ArgumentParser()
ap.add_argument("--exp", type=int, default=0)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--c", type=int, default=512)
ap.add_argument("--m", type=int, default=512)
ap.add_argument("--hw", type=int, default=14)
ap.add_argument("--stride", type=int, default=1)
ap.add_argument("--two", action="store_true", help="also time the two separate kernels (depthwise, then 1x1)")
args = ap.parse_args()
B, c, m, hw = args.batch, args.c, args.m, args.hw
rng = np.random.default_rng(0)
with capi.Context(0) as ctx:
    L = ctx.L
    L.plhip_debug_set.argtypes = [C.c_char_p, C.c_int]
    st = args.stride
    oh = hw // st
    d = capi.conv_desc(B, c, hw, hw, c, 3, 3, (1, 1, 1, 1), (st, st), (1, 1), c, capi.ACT_RELU, 0.0)
    dp = capi.conv_desc(B, c, oh, oh, m, 1, 1, act=capi.ACT_RELU)
    dx = ctx.to_device(rng.integers(-127, 128, (B, c, hw, hw), dtype=np.int8))
    dwd = ctx.to_device(rng.integers(-127, 128, (c, 1, 3, 3), dtype=np.int8))
    dsd = ctx.to_device(np.full(c, 1e-2, np.float32))
    dwr = ctx.to_device(rng.integers(-127, 128, (m, c, 1, 1), dtype=np.int8))
    dwp = ctx.malloc(L.plhip_conv_packed_weight_bytes(C.byref(dp)))
    ctx.check(L.plhip_pack_conv_weights(ctx.h, C.byref(dp), dwr, dwp), "pack")
    dsp = ctx.to_device(np.full(m, 1e-4, np.float32))
    dy = ctx.malloc(B * m * oh * oh)
    if args.exp:
        assert L.plhip_debug_set(b"fused_exp", args.exp) == 0
    for _ in range(args.reps):
        ctx.check(L.plhip_dwpw_fused_int8(ctx.h, C.byref(d), dx, dwd, dsd, None, m, dwp, dsp, None, capi.ACT_RELU, 0.0, dy, capi.OUT_I8), "fused")
    ctx.sync()
    import time
    t0 = time.perf_counter()
    for _ in range(args.reps):
        ctx.check(L.plhip_dwpw_fused_int8(ctx.h, C.byref(d), dx, dwd, dsd, None, m, dwp, dsp, None, capi.ACT_RELU, 0.0, dy, capi.OUT_I8), "fused")
    ctx.sync()
    print("fused: %.1f us per pair (wall clock over %d launches)" % ((time.perf_counter() - t0) / args.reps * 1e6, args.reps))
    if args.two:
        dmid = ctx.malloc(B * c * oh * oh)
        wsb = L.plhip_conv_workspace_bytes(C.byref(dp))
        dws = ctx.malloc(wsb) if wsb else C.c_void_p()
        for rep in range(2):
            t0 = time.perf_counter()
            for _ in range(args.reps):
                ctx.check(L.plhip_depthwise_conv_int8(ctx.h, C.byref(d), dx, dwd, dsd, None, dmid, capi.OUT_I8), "dw")
                ctx.check(L.plhip_conv2d_int8(ctx.h, C.byref(dp), dmid, dwp, dsp, None, dy, capi.OUT_I8, dws, wsb), "pw")
            ctx.sync()
        print("two kernels: %.1f us per pair" % ((time.perf_counter() - t0) / args.reps * 1e6))
print("done")

#!/usr/bin/env python3
"""Timeline of the streaming fused depthwise -> pointwise kernel (fused_dwpw_stream.hip) from in-kernel stamps.
Usage: python tools/stream_timeline.py [--c 128 --m 128 --hw 56] [--stride 1] [--batch 128] [--f32]
(7 x 7 output planes: the small-plane kernel, fused_dwpw_small.hip)"""
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
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--c", type=int, default=128)
ap.add_argument("--m", type=int, default=128)
ap.add_argument("--hw", type=int, default=56)
ap.add_argument("--stride", type=int, default=1)
ap.add_argument("--f32", action="store_true")
args = ap.parse_args()
B, c, m, hw, st_ = args.batch, args.c, args.m, args.hw, args.stride
oh = hw // st_
small = oh == 7
okind = capi.OUT_F32 if args.f32 else capi.OUT_I8
rng = np.random.default_rng(0)
with capi.Context(0) as ctx:
    L = ctx.L
    L.plhip_debug_read_fs_stamps.argtypes = [C.c_void_p, C.c_size_t]
    L.plhip_debug_read_f7_stamps.argtypes = [C.c_void_p, C.c_size_t]
    d = capi.conv_desc(B, c, hw, hw, c, 3, 3, (1, 1, 1, 1), (st_, st_), (1, 1), c, capi.ACT_RELU, 0.0)
    dp = capi.conv_desc(B, c, oh, oh, m, 1, 1, act=capi.ACT_RELU)
    dx = ctx.to_device(rng.integers(-127, 128, (B, c, hw, hw), dtype=np.int8))
    dwd = ctx.to_device(rng.integers(-127, 128, (c, 1, 3, 3), dtype=np.int8))
    dsd = ctx.to_device(np.full(c, 1e-2, np.float32))
    dwr = ctx.to_device(rng.integers(-127, 128, (m, c, 1, 1), dtype=np.int8))
    dwp = ctx.malloc(L.plhip_conv_packed_weight_bytes(C.byref(dp)))
    ctx.check(L.plhip_pack_conv_weights(ctx.h, C.byref(dp), dwr, dwp), "pack")
    dsp = ctx.to_device(np.full(m, 1e-4, np.float32))
    dy = ctx.malloc(B * m * oh * oh * 4)
    fn = lambda: ctx.check(L.plhip_dwpw_fused_int8(ctx.h, C.byref(d), dx, dwd, dsd, None, m, dwp, dsp, None, capi.ACT_RELU, 0.0, dy, okind), "fused")
    for _ in range(5):
        fn()
    ctx.sync()
    assert L.plhip_debug_set(b"fused_stamps", 1) == 0
    for _ in range(2):
        fn()
    ctx.sync()
    if small:
        st = np.zeros((1024, 8, 8), np.uint64)
        assert L.plhip_debug_read_f7_stamps(st.ctypes.data_as(C.c_void_p), st.nbytes) == 0
    else:
        st = np.zeros((2048, 4, 8), np.uint64)
        assert L.plhip_debug_read_fs_stamps(st.ctypes.data_as(C.c_void_p), st.nbytes) == 0
    L.plhip_debug_set(b"fused_stamps", 0)
if small:
    nt = min(1024, 2 * B)
    names = ["entry", "first operands requested", "parameters staged", "produced", "behind the barrier", "multiplied"]
else:
    tp = 448 if oh == 112 else (128 if oh == 14 else 224)
    tr = 7 if oh == 14 else tp // oh
    nt = min(2048, B * ((oh + tr - 1) // tr))
    names = ["entry", "first operands requested", "produced", "behind the barrier", "multiplied", "stores issued"]
st = st[:nt]
rel = st[:, :, 1:7].astype(np.int64) - st[:, :, 1:2].astype(np.int64)
rt = st[:, :, 0].astype(np.int64)
life = (st[:, :, 7].astype(np.int64) - rt) / 100.0
print("%s fused dw3x3 s%d + pw1x1  %d -> %d @%dx%d, batch %d (first %d blocks)" % ("small-plane" if small else "streaming", st_, c, m, hw, hw, B, nt))
print("block start times span %.1f us; a block lives %.2f us (median), %.2f .. %.2f" % ((rt.max() - rt.min()) / 100.0, np.median(life), life.min(), life.max()))
prev = 0
for k, nme in enumerate(names):
    med = int(np.median(rel[:, :, k]))
    print("  %-26s median %7d cycles  (+%d)   min %d max %d" % (nme, med, med - prev, rel[:, :, k].min(), rel[:, :, k].max()))
    prev = med
# generations: blocks sorted by start time
order = np.argsort(rt[:, 0])
t0 = rt[order, 0]
print("  start time percentiles (us): " + " ".join("%.1f" % ((np.percentile(t0, p) - t0[0]) / 100.0) for p in (0, 10, 25, 50, 75, 90, 100)))

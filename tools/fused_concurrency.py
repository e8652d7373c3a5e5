#!/usr/bin/env python3
"""Is the fused depthwise -> pointwise kernel a win when several predictors share the GPU (the bench's 3 steps in flight)?
P host threads, one plhip context (stream) each, loop over the five 512 -> 512 14x14 pairs of MobileNetV1 (batch 128),
either as two kernels per pair or as the fused kernel; reports pairs per second over all threads.
Usage: python tools/fused_concurrency.py [--threads 3] [--reps 200]"""
import argparse
import ctypes as C
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.import_package()
capi = pkg.capi


def setup(ctx, B, c, hw, m, rng):
    L = ctx.L
    d = capi.conv_desc(B, c, hw, hw, c, 3, 3, (1, 1, 1, 1), (1, 1), (1, 1), c, capi.ACT_RELU, 0.0)
    dp = capi.conv_desc(B, c, hw, hw, m, 1, 1, act=capi.ACT_RELU)
    dx = ctx.to_device(rng.integers(-127, 128, (B, c, hw, hw), dtype=np.int8))
    dwd = ctx.to_device(rng.integers(-127, 128, (c, 1, 3, 3), dtype=np.int8))
    dsd = ctx.to_device(np.full(c, 1e-2, np.float32))
    dwr = ctx.to_device(rng.integers(-127, 128, (m, c, 1, 1), dtype=np.int8))
    dwp = ctx.malloc(L.plhip_conv_packed_weight_bytes(C.byref(dp)))
    ctx.check(L.plhip_pack_conv_weights(ctx.h, C.byref(dp), dwr, dwp), "pack")
    dsp = ctx.to_device(np.full(m, 1e-4, np.float32))
    dmid = ctx.malloc(B * c * hw * hw)
    dy = ctx.malloc(B * m * hw * hw)

    def two():
        ctx.check(L.plhip_depthwise_conv_int8(ctx.h, C.byref(d), dx, dwd, dsd, None, dmid, capi.OUT_I8), "dw")
        ctx.check(L.plhip_conv2d_int8(ctx.h, C.byref(dp), dmid, dwp, dsp, None, dy, capi.OUT_I8, None, 0), "pw")

    def fused():
        ctx.check(L.plhip_dwpw_fused_int8(ctx.h, C.byref(d), dx, dwd, dsd, None, m, dwp, dsp, None, capi.ACT_RELU, 0.0, dy, capi.OUT_I8), "fused")
    return two, fused


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=3)
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--batch", type=int, default=128)
    a = ap.parse_args()
    ctxs = [capi.Context(0) for _ in range(a.threads)]
    fns = [setup(cx, a.batch, 512, 14, 512, np.random.default_rng(i)) for i, cx in enumerate(ctxs)]
    for mode in (0, 1, 0, 1):
        for cx, f in zip(ctxs, fns):
            for _ in range(10):
                f[mode]()
            cx.sync()
        bar = threading.Barrier(a.threads + 1)

        def work(cx, f):
            bar.wait()
            for _ in range(a.reps):
                f()
            cx.sync()
            bar.wait()
        ths = [threading.Thread(target=work, args=(cx, f[mode])) for cx, f in zip(ctxs, fns)]
        for t in ths:
            t.start()
        bar.wait()
        t0 = time.perf_counter()
        bar.wait()
        dt = time.perf_counter() - t0
        for t in ths:
            t.join()
        print("%-12s %d threads: %8.2f us per pair (aggregate), %.0f pairs/s" % (
            "fused" if mode else "two kernels", a.threads, dt / (a.reps * a.threads) * 1e6, a.reps * a.threads / dt), flush=True)


if __name__ == "__main__":
    main()

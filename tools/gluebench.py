#!/usr/bin/env python3
"""Micro-benchmark of the glue ops of the MobileNet tail / head through the C ABI (device-resident, HIP events).
Usage: python tools/gluebench.py [--batch 128] [--reps 30]"""
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
sys.path.insert(0, os.path.join(ROOT, "tools"))
from opbench import time_op  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--reps", type=int, default=30)
    a = ap.parse_args()
    B = a.batch
    rng = np.random.default_rng(0)
    with capi.Context(0) as ctx:
        L = ctx.L
        img = ctx.to_device(rng.uniform(-1, 1, (B, 3, 224, 224)).astype(np.float32))
        q = ctx.malloc(B * 3 * 224 * 224)
        n_in = B * 3 * 224 * 224
        t = time_op(ctx, lambda: ctx.check(L.plhip_calib_f32_to_i8(ctx.h, img, q, C.c_float(1 / 127.0), C.c_int64(n_in)), "calib"), a.reps)
        print("calib_in   %8.2f us  %7.1f GB/s" % (t * 1e3, n_in * 5 / t / 1e6))
        feat = ctx.to_device(rng.uniform(0, 4, (B, 1024, 7, 7)).astype(np.float32))
        pooled = ctx.malloc(B * 1024 * 4)
        t = time_op(ctx, lambda: ctx.check(L.plhip_global_avg_pool_f32(ctx.h, feat, B * 1024, 49, pooled), "pool"), a.reps)
        print("pool       %8.2f us  %7.1f GB/s" % (t * 1e3, B * 1024 * 49 * 4 / t / 1e6))
        pq = ctx.malloc(B * 1024)
        t = time_op(ctx, lambda: ctx.check(L.plhip_calib_f32_to_i8(ctx.h, pooled, pq, C.c_float(4 / 127.0), C.c_int64(B * 1024)), "calib2"), a.reps)
        print("calib_fc   %8.2f us" % (t * 1e3))
        w = ctx.to_device(rng.integers(-127, 128, (1024, 1000), dtype=np.int8))
        wp = ctx.malloc(L.plhip_fc_packed_weight_bytes(1024, 1000))
        ctx.check(L.plhip_pack_fc_weights(ctx.h, 1024, 1000, w, wp), "pack")
        sc = ctx.to_device(np.full(1000, 1e-4, np.float32))
        bi = ctx.to_device(np.zeros(1000, np.float32))
        logits = ctx.malloc(B * 1000 * 4)
        t = time_op(ctx, lambda: ctx.check(L.plhip_fc_int8(ctx.h, B, 1024, 1000, pq, wp, sc, bi, 0, logits, capi.OUT_F32), "fc"), a.reps)
        print("fc         %8.2f us  %7.1f GOP/s" % (t * 1e3, 2 * B * 1024 * 1000 / t / 1e6))
        prob = ctx.malloc(B * 1000 * 4)
        t = time_op(ctx, lambda: ctx.check(L.plhip_softmax_f32(ctx.h, logits, B, 1000, prob), "softmax"), a.reps)
        print("softmax    %8.2f us" % (t * 1e3))


if __name__ == "__main__":
    main()

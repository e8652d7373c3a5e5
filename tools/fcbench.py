#!/usr/bin/env python3
"""fc micro-benchmark through the C ABI (device-resident buffers, HIP events on the ctx stream): MobileNetV1 / ResNet50 tails.
Usage: python tools/fcbench.py [--reps 50]"""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=50)
    args = ap.parse_args()
    rng = np.random.default_rng(0)
    with capi.Context(0) as ctx:
        L = ctx.L
        for (m, k, n) in [(128, 1024, 1000), (256, 1024, 1000), (256, 2048, 1000), (1024, 1280, 1000)]:
            dx = ctx.to_device(rng.integers(-127, 128, (m, k), dtype=np.int8))
            dw = ctx.to_device(rng.integers(-127, 128, (k, n), dtype=np.int8))
            dwp = ctx.malloc(L.plhip_fc_packed_weight_bytes(k, n))
            ctx.check(L.plhip_pack_fc_weights(ctx.h, k, n, dw, dwp), "pack_fc")
            ds = ctx.to_device(np.full(n, 1e-4, np.float32))
            db = ctx.to_device(np.zeros(n, np.float32))
            dy = ctx.malloc(m * n * 4)
            fn = lambda: ctx.check(L.plhip_fc_int8(ctx.h, m, k, n, dx, dwp, ds, db, 0, dy, capi.OUT_F32), "fc")
            for _ in range(3):
                fn()
            e0, e1 = C.c_void_p(), C.c_void_p()
            L.plhip_event_create(ctx.h, C.byref(e0))
            L.plhip_event_create(ctx.h, C.byref(e1))
            L.plhip_event_record(ctx.h, e0)
            for _ in range(args.reps):
                fn()
            L.plhip_event_record(ctx.h, e1)
            ms = C.c_float()
            ctx.check(L.plhip_event_elapsed_ms(ctx.h, e0, e1, C.byref(ms)), "elapsed")
            us = ms.value / args.reps * 1e3
            print("fc m=%4d k=%4d n=%4d  %7.2f us  %6.1f TOP/s  %6.1f GB/s" % (
                m, k, n, us, 2.0 * m * k * n / us / 1e6, (m * k + k * n + 4.0 * m * n) / us / 1e3), flush=True)
            for q in list(ctx._allocs):
                ctx.free(q)


if __name__ == "__main__":
    main()

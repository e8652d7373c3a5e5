#!/usr/bin/env python3
"""BASELINE config #2 (single conv2d_int8: N=32 Cin=64 Cout=128 HW=56 k=3 s=1 p=1) through the C ABI: time, TOP/s and
fraction of the dense int8 MFMA peak, for int8 and int32-accumulator outputs.  Usage: python tools/c2bench.py [--reps 30]"""
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

PEAK = 256 * 4 * 2048 * 2.4e9 / 1e12


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--n", type=int, default=32)
    ap.add_argument("--cin", type=int, default=64)
    ap.add_argument("--cout", type=int, default=128)
    ap.add_argument("--hw", type=int, default=56)
    a = ap.parse_args()
    n, cin, hw, cout = a.n, a.cin, a.hw, a.cout
    rng = np.random.default_rng(0)
    with capi.Context(0) as ctx:
        L = ctx.L
        d = capi.conv_desc(n, cin, hw, hw, cout, 3, 3, (1, 1, 1, 1), (1, 1), (1, 1), 1, capi.ACT_RELU, 0.0)
        dx = ctx.to_device(rng.integers(-127, 128, (n, cin, hw, hw), dtype=np.int8))
        dw = ctx.to_device(rng.integers(-127, 128, (cout, cin, 3, 3), dtype=np.int8))
        ds = ctx.to_device(np.full(cout, 1e-4, np.float32))
        db = ctx.to_device(np.zeros(cout, np.float32))
        dwp = ctx.malloc(L.plhip_conv_packed_weight_bytes(C.byref(d)))
        ctx.check(L.plhip_pack_conv_weights(ctx.h, C.byref(d), dw, dwp), "pack")
        wsb = L.plhip_conv_workspace_bytes(C.byref(d))
        dws = ctx.malloc(wsb) if wsb else C.c_void_p()
        ops = 2.0 * n * cout * hw * hw * cin * 9
        for kind, name, esz in ((capi.OUT_I8, "int8 out", 1), (capi.OUT_I32, "int32 acc", 4)):
            dy = ctx.malloc(n * cout * hw * hw * esz)
            ms = time_op(ctx, lambda: ctx.check(L.plhip_conv2d_int8(ctx.h, C.byref(d), dx, dwp, ds, db, dy, kind, dws, wsb), "conv"), a.reps)
            print("conv3x3 n%d %d->%d @%d %-9s %8.2f us  %7.1f TOP/s  %.1f %% of dense i8 MFMA peak  (workspace %.1f MB, %s)" % (
                n, cin, cout, hw, name, ms * 1e3, ops / ms / 1e9, 100 * ops / ms / 1e9 / PEAK, wsb / 1e6, L.plhip_conv_impl_name(C.byref(d)).decode()))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Timeline of one patch-kernel launch (conv_patch_i8.hip) from in-kernel s_memtime stamps (PLHIP_GEMM_DEBUG=32).
Usage: PLHIP_GEMM_DEBUG=32 python tools/patch_timeline.py [--n 32 --cin 64 --cout 128 --hw 56]"""
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
SLOTS, WPB, NBLK = 32, 8, 512


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=32)
    ap.add_argument("--cin", type=int, default=64)
    ap.add_argument("--cout", type=int, default=128)
    ap.add_argument("--hw", type=int, default=56)
    a = ap.parse_args()
    assert int(os.environ.get("PLHIP_GEMM_DEBUG", "0")) & 32, "run with PLHIP_GEMM_DEBUG=32"
    rng = np.random.default_rng(0)
    n, cin, cout, hw = a.n, a.cin, a.cout, a.hw
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
        dws = ctx.malloc(wsb)
        dy = ctx.malloc(n * cout * hw * hw)
        for _ in range(20):
            ctx.check(L.plhip_conv2d_int8(ctx.h, C.byref(d), dx, dwp, ds, db, dy, capi.OUT_I8, dws, wsb), "conv")
        ctx.sync()
        buf = np.zeros(NBLK * WPB * SLOTS, np.uint64)
        rd = L.plhip_debug_read_patch_stamps
        rd.argtypes = [C.c_void_p, C.c_size_t]
        assert rd(buf.ctypes.data, buf.nbytes) == 0
        st = buf.reshape(NBLK, WPB, SLOTS).astype(np.int64)
        st = st[st[:, 0, 1] != 0]
        print("%s n%d %d->%d @%d: blocks with stamps: %d" % (L.plhip_conv_impl_name(C.byref(d)).decode(), n, cin, cout, hw, st.shape[0]))
        rt0, rt1 = st[:, 0, 0], st[:, :, 31].max(axis=1)
        print("kernel span %.2f us; block starts p50 %.2f max %.2f us; block lifetime p10 %.2f p50 %.2f p90 %.2f max %.2f us" % (
            (rt1.max() - rt0.min()) / 100.0, np.median(rt0 - rt0.min()) / 100.0, (rt0.max() - rt0.min()) / 100.0,
            *np.percentile((rt1 - rt0) / 100.0, [10, 50, 90, 100])))
        t = st.reshape(-1, SLOTS)

        def show(label, v):
            print("  %-44s cyc p10 %7.0f  p50 %7.0f  p90 %7.0f" % ((label,) + tuple(np.percentile(v, [10, 50, 90]))))

        show("entry -> prologue issued", t[:, 3] - t[:, 1])
        show("-> barrier of step 0 passed", t[:, 5] - t[:, 3])
        for i in range(6):
            show("step %d: barrier -> MFMAs issued" % i, t[:, 11 + i] - t[:, 5 + i])
            if i < 5:
                show("step %d end -> barrier of step %d passed" % (i, i + 1), t[:, 6 + i] - t[:, 11 + i])
        show("whole wave", t[:, 30] - t[:, 1])
        rounds = [i for i in range(17, 30) if (t[:, i] != 0).all()]
        prev = t[:, 1]
        for i in rounds:
            show("-> end of round %d" % (i - 17), t[:, i] - prev)
            prev = t[:, i]
        show("last round end -> exit (drain)", t[:, 30] - prev)
        clk = (t[:, 30] - t[:, 1]) / np.maximum(1, (t[:, 31] - t[:, 0])) * 100.0
        print("  shader clock p50 %.0f MHz" % np.median(clk))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Timeline of one patch-kernel launch (conv_patch_i8.hip) from in-kernel s_memtime stamps (PLHIP_PATCH_DEBUG=32).
Usage: PLHIP_PATCH_DEBUG=32 python tools/patch_timeline.py [--n 32 --cin 64 --cout 128 --hw 56]"""
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
    ap.add_argument("--stride", type=int, default=1, help="2: the phase-plane form (6 steps of 2 / 1 tap rows per 32 channels)")
    a = ap.parse_args()
    assert int(os.environ.get("PLHIP_PATCH_DEBUG", "0")) & 32, "run with PLHIP_PATCH_DEBUG=32"
    rng = np.random.default_rng(0)
    n, cin, cout, hw = a.n, a.cin, a.cout, a.hw
    with capi.Context(0) as ctx:
        L = ctx.L
        d = capi.conv_desc(n, cin, hw, hw, cout, 3, 3, (1, 1, 1, 1), (a.stride, a.stride), (1, 1), 1, capi.ACT_RELU, 0.0)
        ho = (hw + 2 - 3) // a.stride + 1
        dx = ctx.to_device(rng.integers(-127, 128, (n, cin, hw, hw), dtype=np.int8))
        dw = ctx.to_device(rng.integers(-127, 128, (cout, cin, 3, 3), dtype=np.int8))
        ds = ctx.to_device(np.full(cout, 1e-4, np.float32))
        db = ctx.to_device(np.zeros(cout, np.float32))
        dwp = ctx.malloc(L.plhip_conv_packed_weight_bytes(C.byref(d)))
        ctx.check(L.plhip_pack_conv_weights(ctx.h, C.byref(d), dw, dwp), "pack")
        wsb = L.plhip_conv_workspace_bytes(C.byref(d))
        dws = ctx.malloc(wsb)
        dy = ctx.malloc(n * cout * ho * ho)
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
        print('blocks x waves:', st.shape[0], int((st[0, :, 1] != 0).sum()))
        print("kernel span %.2f us; block starts p50 %.2f max %.2f us; block lifetime p10 %.2f p50 %.2f p90 %.2f max %.2f us" % (
            (rt1.max() - rt0.min()) / 100.0, np.median(rt0 - rt0.min()) / 100.0, (rt0.max() - rt0.min()) / 100.0,
            *np.percentile((rt1 - rt0) / 100.0, [10, 50, 90, 100])))
        t = st.reshape(-1, SLOTS)
        t = t[t[:, 1] != 0]  # (4-wave blocks leave the rows of waves 4-7 empty)

        def show(label, v):
            print("  %-44s cyc p10 %7.0f  p50 %7.0f  p90 %7.0f" % ((label,) + tuple(np.percentile(v, [10, 50, 90]))))

        show("entry -> prologue issued", t[:, 3] - t[:, 1])
        show("-> barrier of step 0 passed", t[:, 5] - t[:, 3])
        for i in range(6):
            show("step %d: barrier -> MFMAs issued" % i, t[:, 11 + i] - t[:, 5 + i])
            if i < 5:
                show("step %d end -> barrier of step %d passed" % (i, i + 1), t[:, 6 + i] - t[:, 11 + i])
        show("step 5 MFMAs -> epilogue group 0 staged", t[:, 18] - t[:, 16])
        show("  -> its stores issued", t[:, 19] - t[:, 18])
        show("  -> group 1 staged", t[:, 20] - t[:, 19])
        show("  -> its stores issued", t[:, 21] - t[:, 20])
        show("whole wave", t[:, 30] - t[:, 1])
        nr = int(os.environ.get("ROUNDS", "1"))
        rounds = list(range(22, min(30, 22 + nr)))
        prev = t[:, 1]
        for i in rounds:
            show("-> end of round %d" % (i - 22), t[:, i] - prev)
            prev = t[:, i]
        show("last round end -> exit (drain)", t[:, 30] - prev)
        # blocks that shared a CU (HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]; XCC_ID): start offsets and the offset of
        # their first epilogues, in cycles
        w0 = st[:, 0, :]
        hw = w0[:, 2] & 0xffffffff
        key = ((w0[:, 2] >> 32) & 0xf) * 4096 + ((hw >> 8) & 0xff)
        from collections import defaultdict
        groups = defaultdict(list)
        for i, k in enumerate(key):
            groups[int(k)].append(i)
        sizes = np.array([len(v) for v in groups.values()])
        print("  CUs used %d, blocks per CU min %d max %d" % (len(groups), sizes.min(), sizes.max()))
        offs, eoffs, ids = [], [], []
        for v in groups.values():
            if len(v) == 2:
                a0, b0 = v
                offs.append(abs(int(w0[a0, 0]) - int(w0[b0, 0])) * (np.median((t[:, 30] - t[:, 1]) / np.maximum(1, (t[:, 31] - t[:, 0])))))
                eoffs.append(abs((int(w0[a0, 18]) - int(w0[a0, 1])) - (int(w0[b0, 18]) - int(w0[b0, 1]))))
                ids.append(abs(a0 - b0))
        if offs:
            print("  CU partners: block index distance p50 %d; start offset p50 %.0f cycles; first-epilogue offset (rel. to own start) p10 %.0f p50 %.0f p90 %.0f cycles" % (
                np.median(ids), np.median(offs), *np.percentile(eoffs, [10, 50, 90])))
        clk = (t[:, 30] - t[:, 1]) / np.maximum(1, (t[:, 31] - t[:, 0])) * 100.0
        print("  shader clock p50 %.0f MHz" % np.median(clk))


if __name__ == "__main__":
    main()

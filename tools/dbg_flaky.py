"""Stress one 1x1 conv shape through the C ABI: N runs, every run compared with the first (and the first with the oracle)."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.import_package()
capi = pkg.capi
import ctypes as C
from oracle import plref
n, cin, hw, cout = [int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (256, 64, 56, 256))]
kind = {"f32": capi.OUT_F32, "i8": capi.OUT_I8, "i32": capi.OUT_I32}[sys.argv[5] if len(sys.argv) > 5 else "f32"]
runs = int(sys.argv[6]) if len(sys.argv) > 6 else 40
rng = np.random.default_rng(5)
x = rng.integers(-127, 128, (n, cin, hw, hw)).astype(np.int8)
w = rng.integers(-127, 128, (cout, cin, 1, 1)).astype(np.int8)
sc = np.full(cout, 1e-4, np.float32); bi = rng.uniform(-1, 1, cout).astype(np.float32)
with capi.Context(0) as ctx:
    L = ctx.L
    d = capi.conv_desc(n, cin, hw, hw, cout, 1, 1, act=capi.ACT_NONE)
    dx, dw = ctx.to_device(x), ctx.to_device(w)
    ds, db = ctx.to_device(sc), ctx.to_device(bi)
    esz = 1 if kind == capi.OUT_I8 else 4
    dy = ctx.malloc(n * cout * hw * hw * esz)
    dwp = ctx.malloc(L.plhip_conv_packed_weight_bytes(C.byref(d)))
    ctx.check(L.plhip_pack_conv_weights(ctx.h, C.byref(d), dw, dwp), "pack")
    outs = []
    dt = {capi.OUT_F32: np.float32, capi.OUT_I8: np.int8, capi.OUT_I32: np.int32}[kind]
    first = None; bad_runs = 0
    for r in range(runs):
        for _ in range(3):
            ctx.check(L.plhip_conv2d_int8(ctx.h, C.byref(d), dx, dwp, ds, db, dy, kind, None, 0), "conv")
        ctx.sync()
        y = np.empty(n * cout * hw * hw, dt)
        ctx.check(L.plhip_memcpy_d2h(ctx.h, y.ctypes.data_as(C.c_void_p), dy, y.nbytes), "d2h"); ctx.sync()
        if first is None:
            first = y.copy()
        else:
            nb = int((y != first).sum())
            if nb:
                bad_runs += 1
                idx = np.flatnonzero(y != first)
                i0 = idx[0]
                print("run", r, "differs from run 0 in", nb, "elements; first at flat", i0, "-> img", i0 // (cout*hw*hw), "ch", (i0 // (hw*hw)) % cout, "hw", i0 % (hw*hw))
    print(sys.argv[1:], "impl", L.plhip_conv_impl_name(C.byref(d)).decode(), "runs differing from run 0:", bad_runs, "of", runs - 1)
    if kind == capi.OUT_F32:
        s_ = plref.shape(n, cin, hw, hw, cout, 1, 1, (0, 0, 0, 0), (1, 1), (1, 1), 1)
        acc = plref.conv2d_acc(s_, x, w, via_gemm=True).reshape(n, cout, hw * hw)
        want = (acc.astype(np.float64) * sc.astype(np.float64)[None, :, None] + bi.astype(np.float64)[None, :, None]).astype(np.float32)
        got = y.reshape(n, cout, hw * hw)
        bad = np.argwhere(~np.isclose(got, want, rtol=1e-5, atol=1e-6))
        print("last run vs oracle:", len(bad), "mismatches")
        for (b_, c_, p_) in bad[:12]:
            g_, w_ = got[b_, c_, p_], want[b_, c_, p_]
            a_ = acc[b_, c_, p_]
            # which (channel, accumulator) would explain the value?
            cand = [(c2, (g_ - bi[c2]) / sc[c2]) for c2 in range(cout) if abs(round((g_ - bi[c2]) / sc[c2]) - (g_ - bi[c2]) / sc[c2]) < 2e-2 and round((g_ - bi[c2]) / sc[c2]) == a_]
            print("  img", b_, "ch", c_, "hw", p_, "got", g_, "want", w_, "acc", a_, "(got-b)/s", (g_ - bi[c_]) / sc[c_], "same acc with bias of channel", [c2 for c2, _ in cand][:4])
        ch = np.bincount(bad[:, 1] % 32, minlength=32); print("  mismatches by channel % 32:", ch.tolist())
        print("  by hw % 4:", np.bincount(bad[:, 2] % 4, minlength=4).tolist(), " by hw % 128:", np.flatnonzero(np.bincount(bad[:, 2] % 128, minlength=128)).tolist()[:40])
    if kind == capi.OUT_I32:
        s = plref.shape(n, cin, hw, hw, cout, 1, 1, (0, 0, 0, 0), (1, 1), (1, 1), 1)
        ref = plref.conv2d_acc(s, x, w, via_gemm=True).reshape(-1)
        print("run 0 vs oracle accumulators:", int((first != ref).sum()), "mismatches")

"""GPU parity tests: the HIP path, called through the C ABI (libplhip.so), against the CPU oracle on
identical seeded inputs and against the committed golden vectors.

Bar: int32 accumulators bit-exact; int8 outputs bit-exact vs the oracle's restatement of the reference
epilogue; fp32 outputs within 1e-5 relative (north_star) — in practice they are bit-identical because
both sides use a single fused multiply-add.  Sweep grids mirror lite/tests/math/conv_int8_compute_test.cc:
482-731 (thinned to keep one process / a few minutes).
"""
import os

import numpy as np
import pytest

from conftest import golden_files, load_golden

pytestmark = pytest.mark.gpu

FP32_RTOL = 1e-5  # north_star: "within 1e-5 after float dequant"


def _check_all_kinds(ctx, capi, plref, n, cin, h, w, cout, kh, kw, pads, st, dl, g, act, alpha, has_bias, rng,
                     depthwise=False, per_channel=True):
    s = plref.shape(n, cin, h, w, cout, kh, kw, pads, (st, st), (dl, dl), g)
    oh, ow = plref.out_dims(s)
    if oh < 1 or ow < 1:
        return 0
    x = rng.integers(-127, 128, (n, cin, h, w)).astype(np.int8)
    wt = rng.integers(-127, 128, (cout, cin // g, kh, kw)).astype(np.int8)
    bias = rng.uniform(-1, 1, cout).astype(np.float32) if has_bias else None
    kk = (cin // g) * kh * kw
    in_scale = 1.0 / 127
    w_scale = ((1 + np.arange(cout) % 7) / 127.0 / 4.0).astype(np.float32) if per_channel else np.array([1 / 127.0], np.float32)
    out_scale = kk / 127.0 if act != 2 else alpha / 127.0
    acc_ref = plref.conv2d_acc(s, x, wt)
    d = capi.conv_desc(n, cin, h, w, cout, kh, kw, pads, (st, st), (dl, dl), g, act, alpha)
    acc = ctx.conv2d(d, x, wt, None, None, capi.OUT_I32, depthwise=depthwise)
    assert np.array_equal(acc, acc_ref), "int32 accumulators differ"
    for int8_out, kind in ((0, capi.OUT_F32), (1, capi.OUT_I8)):
        sc, bi, al = plref.fold_scales(int8_out, in_scale, w_scale, out_scale, bias, cout, act, alpha)
        d.act_alpha = al
        y_ref = plref.epilogue(acc_ref, sc, bi, act, al, bool(int8_out))
        y = ctx.conv2d(d, x, wt, sc, bi if has_bias else None, kind, depthwise=depthwise)
        if int8_out:
            assert np.array_equal(y, y_ref), "int8 output differs"
        else:
            np.testing.assert_allclose(y, y_ref, rtol=FP32_RTOL, atol=1e-6)
    return 1


def test_selftest_known_answer(gpu_ctx):
    gpu_ctx.selftest()


CONV = golden_files("conv_")


@pytest.mark.parametrize("path", CONV, ids=[os.path.basename(p)[5:-4] for p in CONV])
def test_conv_golden(gpu_ctx, pkg, plref, path):
    capi = pkg.capi
    g = load_golden(path)
    n, cin, h, w = g["x"].shape
    cout, _, kh, kw = g["w"].shape
    st, dl, grp = int(g["stride"]), int(g["dil"]), int(g["groups"])
    act, alpha = int(g["act"]), float(g["alpha"])
    pads = tuple(int(p) for p in g["pads"])
    bias = g.get("bias")
    dw = grp == cin == cout and grp > 1
    d = capi.conv_desc(n, cin, h, w, cout, kh, kw, pads, (st, st), (dl, dl), grp, act, alpha)
    for depthwise in ([True, False] if dw and grp <= 8 else [dw]):
        acc = gpu_ctx.conv2d(d, g["x"], g["w"], None, None, capi.OUT_I32, depthwise=depthwise)
        assert np.array_equal(acc, g["acc_ref"])  # acc_ref comes from the reference's conv_basic<int8,int>
        for int8_out, kind, key in ((0, capi.OUT_F32, "y_f32"), (1, capi.OUT_I8, "y_i8")):
            sc, bi, al = plref.fold_scales(int8_out, float(g["in_scale"]), g["w_scale"], float(g["out_scale"]), bias, cout, act, alpha)
            d.act_alpha = al
            y = gpu_ctx.conv2d(d, g["x"], g["w"], sc, bi if bias is not None else None, kind, depthwise=depthwise)
            if int8_out:
                assert np.array_equal(y, g[key])
            else:
                np.testing.assert_allclose(y, g[key], rtol=FP32_RTOL, atol=1e-6)


@pytest.mark.parametrize("path", golden_files("gemm_"), ids=lambda p: os.path.basename(p)[5:-4])
def test_gemm_golden_as_1x1_conv(gpu_ctx, pkg, path):
    """basic_gemm<int8,int> golden: C = A(MxK) * B(KxN) == 1x1 conv with cout=M, cin=K, HW=N."""
    capi = pkg.capi
    g = load_golden(path)
    m, k = g["a"].shape
    n = g["b"].shape[1]
    d = capi.conv_desc(1, k, 1, n, m, 1, 1)
    acc = gpu_ctx.conv2d(d, g["b"].reshape(1, k, 1, n), g["a"].reshape(m, k, 1, 1), None, None, capi.OUT_I32)
    assert np.array_equal(acc.reshape(m, n), g["acc_ref"])


def test_dw3x3_sweep(gpu_ctx, pkg, plref):
    """conv_int8_compute_test.cc:482-513 (thinned)."""
    rng = np.random.default_rng(100)
    cnt = 0
    for st in (1, 2):
        for pad in (0, 1):
            for c in (1, 3, 5, 8, 16, 32):
                for h in (1, 3, 15, 33):
                    act = (0, 1, 2, 4)[cnt % 4]
                    cnt += _check_all_kinds(gpu_ctx, pkg.capi, plref, 1 + cnt % 2, c, h, h, c, 3, 3, (pad,) * 4, st, 1, c,
                                            act, 6.0 if act == 2 else 0.3, cnt % 2 == 0, rng, depthwise=True) or 1
    assert cnt > 40


def test_dw5x5_sweep(gpu_ctx, pkg, plref):
    """conv_int8_compute_test.cc:517-548 (thinned)."""
    rng = np.random.default_rng(101)
    cnt = 0
    for st in (1, 2):
        for pad in (0, 1, 2, 3, 4):
            for c in (1, 5, 15, 33):
                for h in (3, 15, 33, 112):
                    if h == 112 and c > 5:
                        continue
                    act = (0, 1, 2, 4)[cnt % 4]
                    cnt += _check_all_kinds(gpu_ctx, pkg.capi, plref, 1 + cnt % 2, c, h, h, c, 5, 5, (pad,) * 4, st, 1, c,
                                            act, 6.0 if act == 2 else 0.3, cnt % 2 == 1, rng, depthwise=True) or 1
    assert cnt > 40


def test_dw_generic_path_dilated_and_rect(gpu_ctx, pkg, plref):
    rng = np.random.default_rng(102)
    for (c, h, w, kh, kw, st, dl, pads) in [(4, 13, 11, 3, 3, 1, 2, (2, 2, 2, 2)), (3, 9, 17, 2, 3, 2, 1, (0, 1, 1, 0)),
                                           (6, 12, 12, 7, 7, 1, 1, (3, 3, 3, 3)), (2, 8, 8, 1, 1, 1, 1, (0, 0, 0, 0))]:
        s_ok = _check_all_kinds(gpu_ctx, pkg.capi, plref, 2, c, h, w, c, kh, kw, pads, st, dl, c, 1, 0.0, True, rng, depthwise=True)
        assert s_ok == 1


def test_conv1x1s1_sweep(gpu_ctx, pkg, plref):
    """conv_int8_compute_test.cc:552-586."""
    rng = np.random.default_rng(103)
    cnt = 0
    for cin in (1, 3, 8, 33):
        for cout in (1, 5, 17):
            for g in (1, 2):
                if cin % g or cout % g:
                    continue
                for h in (1, 9, 16, 33):
                    act = (0, 1, 2, 4)[cnt % 4]
                    cnt += _check_all_kinds(gpu_ctx, pkg.capi, plref, 1 + cnt % 2, cin, h, h, cout, 1, 1, (0, 0, 0, 0), 1, 1, g,
                                            act, 6.0 if act == 2 else 0.3, cnt % 2 == 0, rng)
    assert cnt > 30


def test_conv3x3_s1_s2_asymmetric_pads(gpu_ctx, pkg, plref):
    """conv_int8_compute_test.cc:590-673 (thinned)."""
    rng = np.random.default_rng(104)
    cnt = 0
    for st in (1, 2):
        for cin in (1, 3, 8, 33):
            for cout in (1, 5, 33):
                for pads in ((1, 1, 1, 1), (1, 2, 2, 1), (2, 1, 1, 2), (2, 2, 2, 2)):
                    h = (1, 7, 17, 33)[cnt % 4]
                    act = (0, 1, 2, 4)[(cnt // 2) % 4]
                    cnt += _check_all_kinds(gpu_ctx, pkg.capi, plref, 1 + cnt % 2, cin, h, h, cout, 3, 3, pads, st, 1, 1,
                                            act, 6.0 if act == 2 else 0.3, cnt % 2 == 0, rng) or 1
    assert cnt > 60


def test_conv_random_params(gpu_ctx, pkg, plref):
    """conv_int8_compute_test.cc:677-730 (seeded random subset of the grid)."""
    rng = np.random.default_rng(105)
    done = 0
    for _ in range(120):
        cin = int(rng.choice([1, 17, 8]))
        cout = int(rng.choice([1, 8, 17]))
        g = int(rng.choice([1, 2]))
        if cin % g or cout % g:
            continue
        kh, kw = int(rng.integers(1, 4)), int(rng.integers(1, 4))
        st, dl = int(rng.integers(1, 3)), int(rng.integers(1, 3))
        pads = tuple(int(v) for v in rng.integers(0, 3, 4))
        h = int(rng.choice([1, 3, 5, 19]))
        act = int(rng.choice([0, 1, 2, 4]))
        done += _check_all_kinds(gpu_ctx, pkg.capi, plref, int(rng.integers(1, 3)), cin, h, h, cout, kh, kw, pads, st, dl, g,
                                 act, 6.0 if act == 2 else 0.3, bool(rng.integers(0, 2)), rng, per_channel=bool(rng.integers(0, 2)))
    assert done > 40


def test_gemm_grid_as_1x1(gpu_ctx, pkg, plref):
    """gemm_int8_compute_test.cc:350-357 grid (M x N x K tails), via the 1x1 path incl. N % 4 != 0 (im2col route)."""
    rng = np.random.default_rng(106)
    capi = pkg.capi
    for m in (1, 3, 8, 32, 33, 35, 41, 397):
        for n in (1, 3, 13, 141, 512):
            k = int(rng.choice([1, 3, 8, 59, 60, 62, 67, 71]))
            a = rng.integers(-127, 128, (m, k)).astype(np.int8)
            b = rng.integers(-127, 128, (k, n)).astype(np.int8)
            d = capi.conv_desc(1, k, 1, n, m, 1, 1)
            acc = gpu_ctx.conv2d(d, b.reshape(1, k, 1, n), a.reshape(m, k, 1, 1), None, None, capi.OUT_I32)
            assert np.array_equal(acc.reshape(m, n), plref.gemm_acc(a, b)), (m, n, k)


def test_mobilenet_layer_shapes_small_batch(gpu_ctx, pkg, plref):
    """Appendix B shapes (pointwise + depthwise + first conv), batch 2, int8 out with relu."""
    rng = np.random.default_rng(107)
    capi = pkg.capi
    for (cin, cout, hw) in [(32, 64, 112), (64, 128, 56), (128, 128, 56), (256, 256, 28), (512, 512, 14), (512, 1024, 7), (1024, 1024, 7)]:
        n = 2 if hw <= 56 else 1
        assert _check_all_kinds(gpu_ctx, capi, plref, n, cin, hw, hw, cout, 1, 1, (0, 0, 0, 0), 1, 1, 1, 1, 0.0, True, rng) == 1
    for (c, hw, st) in [(32, 112, 1), (64, 112, 2), (128, 56, 1), (256, 28, 2), (512, 14, 1), (1024, 7, 1)]:
        assert _check_all_kinds(gpu_ctx, capi, plref, 2, c, hw, hw, c, 3, 3, (1, 1, 1, 1), st, 1, c, 1, 0.0, True, rng, depthwise=True) == 1
    assert _check_all_kinds(gpu_ctx, capi, plref, 1, 3, 224, 224, 32, 3, 3, (1, 1, 1, 1), 2, 1, 1, 1, 0.0, True, rng) == 1


def test_ring_gemm_column_space(gpu_ctx, pkg, plref):
    """The LDS-DMA ring kernel (M > 192 or 96 < M <= 128, K >= 97): images are padded to 16 columns, the last 16-byte piece
    of an image is end-aligned and its duplicate columns are skipped.  HW % 16 in {0, 1, 2, 4, 14}, HW % 4 != 0
    (byte-unaligned pieces, partial-lane stores), M / K tails, batch not filling the last 128-column tile."""
    rng = np.random.default_rng(110)
    capi = pkg.capi
    cnt = 0
    for (h, w) in [(4, 4), (1, 17), (4, 5), (5, 6), (7, 7), (5, 10), (14, 14), (3, 11)]:
        for (cin, cout) in [(128, 200), (130, 256), (97, 120), (160, 512)]:
            act = (0, 1, 2, 4)[cnt % 4]
            cnt += _check_all_kinds(gpu_ctx, capi, plref, 1 + cnt % 3, cin, h, w, cout, 1, 1, (0, 0, 0, 0), 1, 1, 1,
                                    act, 6.0 if act == 2 else 0.3, cnt % 2 == 0, rng)
    assert cnt == 32


def test_wide_gemm_tiles(gpu_ctx, pkg, plref):
    """The wide-tile kernel (gemm_wide_i8.hip: M >= 256, K in {128, 256, 512, 1024}; one 256 x 32*NTT tile per block): every
    tile width it can pick (plhip_debug_wide_ntt 4 / 7 / 8 and the automatic choice) over HW % 16 in {0, 1, 4, 9, 12, 14}
    (end-aligned last chunk, skipped duplicates, byte-unaligned pieces), M tails (300, 257: m tiles past M), batches that
    do not fill the last tile, all three output kinds and all activations."""
    rng = np.random.default_rng(301)
    capi = pkg.capi
    lib = capi.load()
    cnt = 0
    try:
        for ntt in (4, 7, 8, 0):
            lib.plhip_debug_wide_ntt(ntt)
            for (h, w) in [(4, 4), (1, 17), (14, 14), (7, 7), (5, 5), (28, 28), (3, 20)]:
                for (cin, cout) in [(128, 256), (256, 300), (512, 512), (1024, 257)]:
                    if ntt in (7, 8) and cin == 1024:
                        continue  # K = 1024 only fits the LDS with 4 n tiles
                    if (h * w > 200 and cin > 256) or (ntt == 0 and (cnt % 3)):
                        cnt += 1
                        continue
                    act = (0, 1, 2, 4)[cnt % 4]
                    n = 1 + (cnt % 5)
                    assert _check_all_kinds(gpu_ctx, capi, plref, n, cin, h, w, cout, 1, 1, (0, 0, 0, 0), 1, 1, 1,
                                            act, 6.0 if act == 2 else 0.3, cnt % 2 == 0, rng) == 1
                    cnt += 1
    finally:
        lib.plhip_debug_wide_ntt(-1)
    assert cnt > 60


def test_stem_mfma_variants(gpu_ctx, pkg, plref):
    """conv3x3s2 stem on the MFMA path (Cin <= 3, OW % 4 == 0): Cin 1/2/3, Cout tails and two M tiles, pads 0/1, more
    than 32 quads per row (two column tiles), OH % 4 != 0 (partly empty row groups), several images; plus shapes that
    must fall back to the dot4 kernel (OW % 4 != 0, Cin = 4)."""
    rng = np.random.default_rng(111)
    capi = pkg.capi
    cnt = 0
    for (n, cin, h, w, cout, pad) in [(1, 3, 16, 16, 32, 1), (2, 1, 18, 32, 8, 1), (3, 2, 9, 17, 40, 0), (1, 3, 21, 264, 64, 1),
                                      (2, 3, 10, 8, 33, (0, 1, 1, 0)), (1, 3, 15, 15, 16, 1), (2, 4, 16, 16, 24, 1)]:
        pads = (pad,) * 4 if isinstance(pad, int) else pad
        act = (1, 0, 2, 4)[cnt % 4]
        cnt += _check_all_kinds(gpu_ctx, capi, plref, n, cin, h, w, cout, 3, 3, pads, 2, 1, 1, act, 6.0 if act == 2 else 0.2,
                                cnt % 2 == 0, rng)
    assert cnt == 7


def test_stem_with_the_calib_in_front(gpu_ctx, pkg, plref):
    """plhip_conv2d_calib_int8 = calib[fp32_to_int8] + conv 3x3 s2 (Cin <= 3) in one launch (conv_stem_f32in.hip): bit-identical to
    the oracle's calib followed by its conv: accumulators, int8 and fp32 outputs; values on the rounding ties and beyond the
    saturation bound in the image; two column tiles, OH % 4 != 0, top padding 0, Cout tails; shapes without the form refused."""
    import ctypes
    rng = np.random.default_rng(141)
    capi = pkg.capi
    cnt = 0
    for (n, cin, h, w, cout, pads) in [(2, 3, 224, 224, 32, (1, 1, 1, 1)), (1, 3, 16, 16, 32, (1, 1, 1, 1)), (3, 1, 18, 32, 8, (1, 1, 1, 1)),
                                       (1, 3, 21, 264, 64, (1, 1, 1, 1)), (2, 2, 10, 8, 33, (0, 1, 1, 0)), (1, 3, 66, 520, 40, (1, 0, 1, 1))]:
        act = (1, 0, 2, 4)[cnt % 4]
        alpha = 6.0 if act == 2 else 0.2
        calib_scale = np.float32(1.0 / 127 * (1 + cnt % 3))
        xf = rng.uniform(-1.2, 1.2, (n, cin, h, w)).astype(np.float32) * np.float32(1 + cnt % 3)
        # exact ties (k + 0.5) * scale, values past +-127 * scale and zeros
        k = rng.integers(-140, 141, xf.size // 7).astype(np.float32)
        flat = xf.reshape(-1)
        flat[:k.size] = (k + np.float32(0.5)) * calib_scale
        flat[k.size:k.size + 16] = 0.0
        x = plref.calib_f32_to_i8(xf, float(calib_scale))
        wt = rng.integers(-127, 128, (cout, cin, 3, 3)).astype(np.int8)
        bias = rng.uniform(-1, 1, cout).astype(np.float32) if cnt % 2 == 0 else None
        w_scale = ((1 + np.arange(cout) % 7) / 127.0 / 4.0).astype(np.float32)
        out_scale = cin * 9 / 127.0 if act != 2 else alpha / 127.0
        s = plref.shape(n, cin, h, w, cout, 3, 3, pads, (2, 2), (1, 1), 1)
        acc_ref = plref.conv2d_acc(s, x, wt)
        d = capi.conv_desc(n, cin, h, w, cout, 3, 3, pads, (2, 2), (1, 1), 1, act, alpha)
        assert gpu_ctx.L.plhip_conv2d_calib_supported(ctypes.byref(d)) == 1
        acc = gpu_ctx.conv2d_calib(d, xf, float(calib_scale), wt, None, None, capi.OUT_I32)
        assert np.array_equal(acc, acc_ref), "int32 accumulators differ (%d of %d)" % ((acc != acc_ref).sum(), acc.size)
        for int8_out, kind in ((0, capi.OUT_F32), (1, capi.OUT_I8)):
            sc, bi, al = plref.fold_scales(int8_out, float(calib_scale), w_scale, out_scale, bias, cout, act, alpha)
            d.act_alpha = al
            y_ref = plref.epilogue(acc_ref, sc, bi, act, al, bool(int8_out))
            y = gpu_ctx.conv2d_calib(d, xf, float(calib_scale), wt, sc, bi if bias is not None else None, kind)
            if int8_out:
                assert np.array_equal(y, y_ref), "int8 output differs"
            else:
                np.testing.assert_allclose(y, y_ref, rtol=FP32_RTOL, atol=1e-6)
        cnt += 1
    # no one-launch form: left padding 0 / 2, W % 4 != 0, OW % 4 != 0, Cin = 4, stride 1, 7x7
    for (cin, h, w, k, st, pads) in [(3, 16, 16, 3, 2, (1, 1, 0, 1)), (3, 16, 16, 3, 2, (1, 1, 2, 2)), (3, 16, 18, 3, 2, (1, 1, 1, 1)),
                                     (3, 16, 20, 3, 2, (1, 1, 1, 1)), (4, 16, 16, 3, 2, (1, 1, 1, 1)), (3, 16, 16, 3, 1, (1, 1, 1, 1)),
                                     (3, 32, 32, 7, 2, (3, 3, 3, 3))]:
        d = capi.conv_desc(1, cin, h, w, 16, k, k, pads, (st, st), (1, 1), 1, 1, 0.0)
        assert gpu_ctx.L.plhip_conv2d_calib_supported(ctypes.byref(d)) == 0, (cin, h, w, k, st, pads)


def test_dw_fast_fetch_and_staging(gpu_ctx, pkg, plref):
    """depthwise 3x3 direct kernel: fast row fetch (pad <= 1, RS | OH) against the general fetch (pad 2, ragged OH), LDS
    output staging for narrow planes with even / odd OW, partial last waves (plane count not a multiple of the wave's
    strip count), first / last workgroup guards (single-plane tensors)."""
    rng = np.random.default_rng(112)
    capi = pkg.capi
    cnt = 0
    for st in (1, 2):
        for (c, h, w, pad) in [(5, 14, 14, 1), (37, 7, 7, 1), (3, 28, 28, 1), (1, 56, 56, 1), (9, 16, 12, 0), (4, 14, 14, 2),
                               (130, 14, 14, 1), (2, 112, 112, 1), (6, 8, 30, 1)]:
            act = (1, 2, 0, 4)[cnt % 4]
            cnt += _check_all_kinds(gpu_ctx, capi, plref, 1 + cnt % 2, c, h, w, c, 3, 3, (pad,) * 4, st, 1, c, act,
                                    6.0 if act == 2 else 0.3, cnt % 2 == 0, rng, depthwise=True)
    assert cnt == 18


def test_dw5x5_direct_fetch_and_staging(gpu_ctx, pkg, plref):
    """depthwise 5x5 on the direct strip kernel (pad <= 3; conv5x5s{1,2}_depthwise_int8.cc shapes): fast row fetch (pad <= 2,
    RS | OH: two rows at either end of a strip may leave the image) against the general fetch (pad 3, ragged OH), LDS output
    staging for narrow planes, partial last waves, single-plane tensors, the fifth tap of the last channel's last filter row."""
    rng = np.random.default_rng(139)
    capi = pkg.capi
    cnt = 0
    for st in (1, 2):
        for (c, h, w, pad) in [(5, 14, 14, 2), (37, 7, 7, 2), (3, 28, 28, 2), (1, 56, 56, 2), (9, 16, 12, 0), (4, 14, 14, 1),
                               (130, 14, 14, 2), (2, 112, 112, 2), (6, 8, 30, 2), (3, 9, 9, 3), (1, 5, 5, 0), (64, 28, 28, 2)]:
            act = (1, 2, 0, 4)[cnt % 4]
            cnt += _check_all_kinds(gpu_ctx, capi, plref, 1 + cnt % 2, c, h, w, c, 5, 5, (pad,) * 4, st, 1, c, act,
                                    6.0 if act == 2 else 0.3, cnt % 2 == 0, rng, depthwise=True)
    assert cnt == 24


def test_resnet50_and_mobilenetv2_layer_shapes(gpu_ctx, pkg, plref):
    """BASELINE configs C4 / C5 as parity cases (lite/tests/benchmark/src/convolution_configs.h:381-466, 839-891), batch 1-2:
    ResNet50: 7x7 s2 p3 stem, 3x3 s1 / s2 dense (im2col route), 1x1 stride-2 downsample, 1x1 with K = 2048;
    MobileNetV2: relu6 expansions (wide M), linear bottleneck projections (no activation), relu6 depthwise s1 / s2."""
    rng = np.random.default_rng(113)
    capi = pkg.capi
    R50 = [  # n, cin, h, cout, k, pad, stride, act
        (1, 3, 224, 64, 7, 3, 2, 1), (2, 64, 56, 64, 3, 1, 1, 1), (1, 128, 56, 128, 3, 1, 2, 1),
        (1, 256, 56, 512, 1, 0, 2, 0), (2, 2048, 7, 512, 1, 0, 1, 1), (1, 512, 7, 512, 3, 1, 1, 1)]
    for (n, cin, h, cout, k, pad, st, act) in R50:
        assert _check_all_kinds(gpu_ctx, capi, plref, n, cin, h, h, cout, k, k, (pad,) * 4, st, 1, 1, act, 0.0, True, rng) == 1
    V2 = [(2, 16, 112, 96, 1, 0, 1, 2), (2, 96, 56, 24, 1, 0, 1, 0), (2, 144, 56, 24, 1, 0, 1, 0), (2, 320, 7, 1280, 1, 0, 1, 2),
          (2, 160, 7, 960, 1, 0, 1, 2)]
    for (n, cin, h, cout, k, pad, st, act) in V2:
        assert _check_all_kinds(gpu_ctx, capi, plref, n, cin, h, h, cout, k, k, (pad,) * 4, st, 1, 1, act, 6.0, True, rng) == 1
    for (c, h, st) in [(96, 112, 2), (144, 56, 1), (192, 28, 2), (576, 14, 1), (960, 7, 1)]:
        assert _check_all_kinds(gpu_ctx, capi, plref, 2, c, h, h, c, 3, 3, (1, 1, 1, 1), st, 1, c, 2, 6.0, True, rng, depthwise=True) == 1


def test_implicit_gemm_route(gpu_ctx, pkg, plref):
    """Dense k x k stride-1 convs on the implicit-GEMM route (zero-padded input copy + ring kernel): kernel shapes 3x3,
    5x5, 1x3, 3x1, 7x7, 2x2; symmetric / asymmetric / zero pads; OW = 16, 17, 20, 30, 56 (end-aligned last piece, skipped
    duplicate columns); K tails (Cin*kh*kw % 32 != 0); 32- and 64-row wave tiles; batch 1-3; all output kinds."""
    rng = np.random.default_rng(114)
    capi = pkg.capi
    cases = [  # n, cin, h, w, cout, kh, kw, pads(t,b,l,r), act
        (2, 64, 20, 20, 128, 3, 3, (1, 1, 1, 1), 1), (1, 32, 18, 18, 256, 3, 3, (0, 0, 0, 0), 0), (3, 24, 16, 17, 200, 5, 5, (2, 2, 2, 2), 2),
        (2, 100, 9, 30, 256, 1, 3, (0, 0, 1, 1), 4), (1, 96, 30, 16, 128, 3, 1, (1, 1, 0, 0), 1), (1, 8, 24, 24, 256, 7, 7, (3, 3, 3, 3), 1),
        (2, 70, 17, 21, 320, 2, 2, (1, 0, 0, 1), 0), (1, 64, 56, 56, 128, 3, 3, (1, 1, 1, 1), 1), (2, 33, 12, 40, 512, 3, 3, (2, 1, 0, 2), 2)]
    for (n, cin, h, w, cout, kh, kw, pads, act) in cases:
        d = capi.conv_desc(n, cin, h, w, cout, kh, kw, pads, (1, 1), (1, 1), 1, act, 0.0)
        name = gpu_ctx.L.plhip_conv_impl_name(__import__("ctypes").byref(d))
        patch = kh == 3 and kw == 3 and cin % 32 == 0 and cin >= 64  # 3x3 with whole 32-channel chunks: the patch kernel
        assert name == (b"conv_patch_gemm_int8_mfma32x32x32" if patch else b"conv_implicit_gemm_int8_mfma32x32x32"), (cin, cout, kh, kw)
        assert _check_all_kinds(gpu_ctx, capi, plref, n, cin, h, w, cout, kh, kw, pads, 1, 1, 1, act, 6.0 if act == 2 else 0.25,
                                act != 4, rng) == 1


def test_conv1x1_stride2_subsample(gpu_ctx, pkg, plref):
    """1x1 stride-2 convs (ResNet50's shortcuts) gather their GEMM operand with subsample2_1x1_i8_kernel (16 bytes per thread)
    : even and odd input extents (the row's last quad), output planes that are not a multiple of 16 columns, quads across two
    output rows (OW = 14, 7, 4, 12), groups, bottom / right padding (falls back to the generic im2col kernel)."""
    rng = np.random.default_rng(133)
    capi = pkg.capi
    cases = [  # n, cin, h, w, cout, pads, groups, act
        (2, 64, 56, 56, 128, (0, 0, 0, 0), 1, 0), (3, 48, 15, 23, 40, (0, 0, 0, 0), 1, 1), (2, 32, 9, 8, 64, (0, 0, 0, 0), 2, 2),
        (1, 256, 28, 28, 96, (0, 0, 0, 0), 1, 1), (2, 24, 8, 8, 33, (0, 1, 0, 1), 1, 0), (2, 40, 13, 13, 50, (0, 0, 0, 0), 1, 4),
        (2, 96, 14, 14, 64, (0, 0, 0, 0), 1, 1), (3, 64, 28, 27, 48, (0, 0, 0, 0), 1, 0)]
    for (n, cin, h, w, cout, pads, g, act) in cases:
        assert _check_all_kinds(gpu_ctx, capi, plref, n, cin, h, w, cout, 1, 1, pads, 2, 1, g, act, 6.0 if act == 2 else 0.25, True, rng) == 1


def test_patch_conv_route(gpu_ctx, pkg, plref):
    """Dense 3x3 stride-1 convs with Cin % 32 == 0 on the patch kernel (conv_patch_i8.hip): register-resident weights
    (Cin = 64) in both wave layouts (M > 64: 4 m tiles x 1 pixel group per half, M <= 64: 2 x 2), weights through the ring
    (Cin > 64), M blocks (M > 128) and M tails, row pitches 16 .. 64 incl. the shared pad column (W + 1 a multiple of 8),
    asymmetric / zero pads, tiles that end inside an image / past it, several tiles per stream, every activation and
    output kind."""
    rng = np.random.default_rng(131)
    capi = pkg.capi
    cases = [  # n, cin, h, w, cout, pads(t,b,l,r), act
        (2, 64, 20, 20, 128, (1, 1, 1, 1), 1), (1, 64, 56, 56, 128, (1, 1, 1, 1), 2), (2, 64, 56, 56, 64, (1, 1, 1, 1), 1),
        (3, 64, 9, 15, 40, (1, 1, 1, 1), 0), (2, 64, 13, 27, 100, (0, 2, 2, 0), 4), (2, 128, 28, 28, 128, (1, 1, 1, 1), 1),
        (2, 256, 14, 14, 256, (1, 1, 1, 1), 1), (1, 96, 17, 33, 200, (1, 0, 0, 1), 2), (5, 160, 14, 14, 72, (0, 0, 0, 0), 0),
        (37, 64, 14, 14, 96, (1, 1, 1, 1), 1), (1, 64, 40, 58, 130, (1, 1, 1, 1), 4),
        # more tiles than tile streams (512): every stream works through two tiles (the ring, the DMA cursor and the
        # register-resident weights across a tile boundary), in each of the three kernel variants
        (530, 64, 6, 14, 96, (1, 1, 1, 1), 1), (530, 64, 6, 14, 32, (1, 1, 1, 1), 2), (530, 128, 6, 14, 72, (1, 1, 1, 1), 0),
        # 7-wide planes (row pitch 8): global mode — channel-major padded copy, tiles across images, output pieces that end
        # in the next image; batch 1 (less than a tile), odd batches, all three kernel variants, asymmetric pads
        (1, 64, 7, 7, 96, (1, 1, 1, 1), 1), (9, 64, 7, 7, 65, (1, 1, 1, 1), 2), (5, 512, 7, 7, 520, (1, 1, 1, 1), 1),
        (13, 96, 9, 6, 72, (1, 0, 1, 1), 0), (7, 64, 5, 5, 130, (2, 1, 2, 1), 4), (70, 128, 7, 7, 96, (1, 1, 1, 1), 1)]
    for (n, cin, h, w, cout, pads, act) in cases:
        d = capi.conv_desc(n, cin, h, w, cout, 3, 3, pads, (1, 1), (1, 1), 1, act, 0.0)
        assert gpu_ctx.L.plhip_conv_impl_name(__import__("ctypes").byref(d)) == b"conv_patch_gemm_int8_mfma32x32x32", (cin, cout, w, pads)
        assert _check_all_kinds(gpu_ctx, capi, plref, n, cin, h, w, cout, 3, 3, pads, 1, 1, 1, act, 6.0 if act == 2 else 0.25,
                                act != 4, rng) == 1, (n, cin, h, w, cout, pads, act)


def test_patch_conv_stride2_route(gpu_ctx, pkg, plref):
    """Dense 3x3 stride-2 convs with Cin % 32 == 0 and M > 64 on the patch kernel over phase planes (ResNet50's downsampling
    convs): even / odd extents, asymmetric and zero pads, M blocks and tails, more tiles than tile streams, planes smaller
    than a tile (global mode: channel-major phase copy), every activation and output kind."""
    rng = np.random.default_rng(137)
    capi = pkg.capi
    cases = [  # n, cin, h, w, cout, pads(t,b,l,r), act
        (2, 128, 56, 56, 128, (1, 1, 1, 1), 1), (2, 256, 28, 28, 256, (1, 1, 1, 1), 1), (3, 512, 14, 14, 512, (1, 1, 1, 1), 1),
        (2, 64, 20, 21, 96, (1, 0, 1, 0), 2), (1, 32, 33, 40, 130, (0, 1, 0, 1), 4), (530, 64, 12, 28, 96, (1, 1, 1, 1), 1),
        (9, 64, 14, 14, 65, (1, 1, 1, 1), 2), (1, 64, 13, 13, 96, (1, 1, 1, 1), 0), (5, 96, 9, 11, 72, (0, 0, 0, 0), 0),
        (2, 32, 57, 120, 70, (1, 1, 1, 1), 1)]
    for (n, cin, h, w, cout, pads, act) in cases:
        d = capi.conv_desc(n, cin, h, w, cout, 3, 3, pads, (2, 2), (1, 1), 1, act, 0.0)
        assert gpu_ctx.L.plhip_conv_impl_name(__import__("ctypes").byref(d)) == b"conv_patch_s2_gemm_int8_mfma32x32x32", (cin, cout, w, pads)
        assert _check_all_kinds(gpu_ctx, capi, plref, n, cin, h, w, cout, 3, 3, pads, 2, 1, 1, act, 6.0 if act == 2 else 0.25,
                                act != 4, rng) == 1, (n, cin, h, w, cout, pads, act)


_FC_CHILD = r"""
import sys
import numpy as np
sys.path.insert(0, %r)
import __graft_entry__ as ge
from oracle import plref
plref.build()
capi = ge.import_package().capi
rng = np.random.default_rng(145)
with capi.Context(0) as ctx:
    for (m, k, n) in [(5, 1024, 1000), (70, 64, 37), (33, 96, 1000), (128, 1024, 1000), (3, 2048, 130)]:
        x = rng.integers(-127, 128, (m, k)).astype(np.int8)
        w = rng.integers(-127, 128, (k, n)).astype(np.int8)
        sc = ((1 + np.arange(n) %% 5) / 127.0 / 127.0).astype(np.float32)
        bi = rng.uniform(-1, 1, n).astype(np.float32)
        y_ref, acc_ref = plref.fc(x, w, bi, sc, False, False)
        assert np.array_equal(ctx.fc(x, w, None, None, 0, capi.OUT_I32), acc_ref), (m, k, n)
        np.testing.assert_allclose(ctx.fc(x, w, sc, bi, 0, capi.OUT_F32), y_ref, rtol=1e-5, atol=1e-7)
        sc8 = (sc * 40.0).astype(np.float32)
        y8_ref, _ = plref.fc(x, w, bi, sc8, True, True)
        assert np.array_equal(ctx.fc(x, w, sc8, bi, 1, capi.OUT_I8), y8_ref), (m, k, n)
print("fc mfma form ok")
"""


def test_fc_mfma_form_in_a_subprocess(gpu_ctx):
    """The MFMA form of fc (k % 32 == 0) is no longer what the library picks by default (the LDS-staged dot4 kernel runs the
    network tails in half its time); it stays selectable (PLHIP_FC_MFMA=1, read once per process): its parity against the
    oracle runs in a child process with the variable set."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PLHIP_FC_MFMA="1")
    r = subprocess.run([sys.executable, "-c", _FC_CHILD % root], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "fc mfma form ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_stem_7x7_stride2_direct(gpu_ctx, pkg, plref):
    """7x7 stride-2 convs with Cin <= 3 and OW % 4 == 0 on the direct stem kernel (conv_stem7_i8.hip; ResNet50's conv1): the
    network shape, Cin 1 / 2 / 3, M tails and several m tiles, odd heights, asymmetric / zero pads, rows narrower than a quad
    tile, single-row-group tensors (every block fetches bytewise), every activation and output kind."""
    rng = np.random.default_rng(141)
    capi = pkg.capi
    cases = [  # n, cin, h, w, cout, pads(t,b,l,r), act
        (2, 3, 224, 224, 64, (3, 3, 3, 3), 1), (3, 3, 64, 64, 64, (3, 3, 3, 3), 2), (2, 1, 41, 64, 40, (3, 3, 3, 3), 0),
        (2, 2, 65, 64, 96, (3, 2, 3, 2), 4), (1, 3, 30, 69, 33, (0, 0, 0, 0), 1), (5, 3, 18, 32, 64, (3, 3, 3, 3), 1),
        (1, 3, 9, 16, 8, (3, 3, 3, 3), 0), (2, 3, 112, 522, 32, (2, 3, 1, 2), 2)]
    for (n, cin, h, w, cout, pads, act) in cases:
        d = capi.conv_desc(n, cin, h, w, cout, 7, 7, pads, (2, 2), (1, 1), 1, act, 0.0)
        assert gpu_ctx.L.plhip_conv_impl_name(__import__("ctypes").byref(d)) == b"conv_7x7s2_direct_int8_mfma32x32x32", (cin, cout, w, pads)
        assert _check_all_kinds(gpu_ctx, capi, plref, n, cin, h, w, cout, 7, 7, pads, 2, 1, 1, act, 6.0 if act == 2 else 0.25,
                                act != 4, rng) == 1, (n, cin, h, w, cout, pads, act)


def test_fc_calib_pool_softmax(gpu_ctx, pkg, plref):
    capi = pkg.capi
    g = load_golden(golden_files("fc_")[0])
    assert np.array_equal(gpu_ctx.fc(g["x"], g["w"], None, None, 0, capi.OUT_I32), g["acc_ref"])
    np.testing.assert_allclose(gpu_ctx.fc(g["x"], g["w"], g["scale"], g["bias"], 1, capi.OUT_F32), g["y_f32"], rtol=FP32_RTOL, atol=1e-7)
    assert np.array_equal(gpu_ctx.fc(g["x"], g["w"], g["scale8"], g["bias8"], 1, capi.OUT_I8), g["y_i8"])
    rng = np.random.default_rng(108)
    # fc_compute.cc shapes: MobileNet tail (k=1024, n=1000) at a ragged batch; k % 4 != 0 tail
    # (k % 32 == 0 -> MFMA kernel, incl. ragged m / n tiles and a split-K remainder; otherwise the dot4 kernels)
    for (m, k, n) in [(5, 1024, 1000), (1, 7, 3), (17, 66, 257), (70, 64, 37), (33, 96, 1000), (128, 1024, 1000)]:
        x = rng.integers(-127, 128, (m, k)).astype(np.int8)
        w = rng.integers(-127, 128, (k, n)).astype(np.int8)
        sc = ((1 + np.arange(n) % 5) / 127.0 / 127.0).astype(np.float32)
        bi = rng.uniform(-1, 1, n).astype(np.float32)
        y_ref, acc_ref = plref.fc(x, w, bi, sc, False, False)
        assert np.array_equal(gpu_ctx.fc(x, w, None, None, 0, capi.OUT_I32), acc_ref)
        np.testing.assert_allclose(gpu_ctx.fc(x, w, sc, bi, 0, capi.OUT_F32), y_ref, rtol=FP32_RTOL, atol=1e-7)
        sc8 = (sc * 40.0).astype(np.float32)  # int8-out: scale folded with 1 / out_scale
        y8_ref, _ = plref.fc(x, w, bi, sc8, True, True)
        assert np.array_equal(gpu_ctx.fc(x, w, sc8, bi, 1, capi.OUT_I8), y8_ref)
    c = load_golden(golden_files("calib")[0])
    assert np.array_equal(gpu_ctx.calib_f32_to_i8(c["x"], float(c["scale"])), c["q"])
    assert np.array_equal(gpu_ctx.calib_i8_to_f32(c["q"], float(c["scale"])), c["deq"])
    xf = rng.uniform(-4, 4, (3, 5, 7, 9)).astype(np.float32)
    assert np.array_equal(gpu_ctx.calib_f32_to_i8(xf, 1 / 31.0), plref.calib_f32_to_i8(xf, 1 / 31.0))
    xp = rng.uniform(-1, 1, (3, 20, 7, 7)).astype(np.float32)
    np.testing.assert_allclose(gpu_ctx.global_avg_pool(xp), plref.global_avg_pool(xp), rtol=1e-5, atol=1e-6)
    xs = rng.uniform(-5, 5, (4, 1000)).astype(np.float32)
    np.testing.assert_allclose(gpu_ctx.softmax(xs), plref.softmax(xs), rtol=1e-5, atol=1e-7)


def test_patch_conv_random_shapes(gpu_ctx, pkg, plref):
    """Random sweep over the patch kernel's shape space (the reference's random grid, conv_int8_compute_test.cc:676-731, narrowed
    to 3x3 stride 1 with whole 32-channel chunks): batch 1-9, Cin 64-192, 6-60 rows / columns (row pitches 16-64 incl. the
    shared pad column), Cout 32-300 (M tails, M blocks), pads 0-2 per side, every activation, bias on / off."""
    rng = np.random.default_rng(135)
    capi = pkg.capi
    done = 0
    for _ in range(60):
        cin = int(rng.choice([64, 96, 128, 160, 192]))
        cout = int(rng.integers(32, 301))
        if cout <= 64 and (cin != 64 or True):  # (M <= 64 runs the 2 x 2 layout only with Cin = 64 and row pitches >= 16: keep it simple)
            cout += 64
        n, h, w = int(rng.integers(1, 10)), int(rng.integers(4, 61)), int(rng.integers(4, 61))
        pads = tuple(int(v) for v in rng.integers(0, 3, 4))
        act = int(rng.choice([0, 1, 2, 4]))
        d = capi.conv_desc(n, cin, h, w, cout, 3, 3, pads, (1, 1), (1, 1), 1, act, 0.0)
        if gpu_ctx.L.plhip_conv_impl_name(__import__("ctypes").byref(d)) != b"conv_patch_gemm_int8_mfma32x32x32":
            continue  # (row pitch outside 8..64)
        if n * cout * h * w * cin > 6e9 / 9:
            continue  # keep the scalar oracle in seconds
        done += _check_all_kinds(gpu_ctx, capi, plref, n, cin, h, w, cout, 3, 3, pads, 1, 1, 1, act, 6.0 if act == 2 else 0.25,
                                 bool(rng.integers(0, 2)), rng)
    assert done >= 20, done


def test_new_route_random_shapes(gpu_ctx, pkg, plref):
    """Random sweeps (the reference's random grid, conv_int8_compute_test.cc:676-731, narrowed to each kernel's shape space) over
    the stride-2 patch kernel (Cin 32-160 in 32s, Cout 65-260, 5-120 rows / columns, pads 0-2), the 7x7 stride-2 stem (Cin 1-3,
    Cout 8-100, OW % 4 == 0, pads 0-3) and the direct depthwise 5x5 kernel (1-70 channels, stride 1 / 2, pads 0-3, 5-60 rows /
    columns): every activation, bias on / off."""
    rng = np.random.default_rng(143)
    capi = pkg.capi
    byref = __import__("ctypes").byref
    done = [0, 0, 0]
    for _ in range(40):
        cin, cout = int(rng.choice([32, 64, 96, 128, 160])), int(rng.integers(65, 261))
        n, h, w = int(rng.integers(1, 8)), int(rng.integers(5, 121)), int(rng.integers(5, 121))
        pads = tuple(int(v) for v in rng.integers(0, 3, 4))
        act = int(rng.choice([0, 1, 2, 4]))
        d = capi.conv_desc(n, cin, h, w, cout, 3, 3, pads, (2, 2), (1, 1), 1, act, 0.0)
        if gpu_ctx.L.plhip_conv_impl_name(byref(d)) != b"conv_patch_s2_gemm_int8_mfma32x32x32" or n * cout * h * w * cin > 1.0e9:
            continue
        done[0] += _check_all_kinds(gpu_ctx, capi, plref, n, cin, h, w, cout, 3, 3, pads, 2, 1, 1, act, 6.0 if act == 2 else 0.25,
                                    bool(rng.integers(0, 2)), rng)
    for _ in range(40):
        cin, cout = int(rng.integers(1, 4)), int(rng.integers(8, 101))
        n, h, w = int(rng.integers(1, 5)), int(rng.integers(7, 150)), int(rng.integers(16, 200))
        pads = tuple(int(v) for v in rng.integers(0, 4, 4))
        act = int(rng.choice([0, 1, 2, 4]))
        d = capi.conv_desc(n, cin, h, w, cout, 7, 7, pads, (2, 2), (1, 1), 1, act, 0.0)
        if gpu_ctx.L.plhip_conv_impl_name(byref(d)) != b"conv_7x7s2_direct_int8_mfma32x32x32":
            continue  # (OW % 4 != 0: the implicit GEMM)
        done[1] += _check_all_kinds(gpu_ctx, capi, plref, n, cin, h, w, cout, 7, 7, pads, 2, 1, 1, act, 6.0 if act == 2 else 0.25,
                                    bool(rng.integers(0, 2)), rng)
    for _ in range(40):
        c, st = int(rng.integers(1, 71)), int(rng.integers(1, 3))
        n, h, w = int(rng.integers(1, 4)), int(rng.integers(5, 61)), int(rng.integers(5, 61))
        pads = tuple(int(v) for v in rng.integers(0, 4, 4))
        act = int(rng.choice([0, 1, 2, 4]))
        done[2] += _check_all_kinds(gpu_ctx, capi, plref, n, c, h, w, c, 5, 5, pads, st, 1, c, act, 6.0 if act == 2 else 0.3,
                                    bool(rng.integers(0, 2)), rng, depthwise=True)
    assert done[0] >= 12 and done[1] >= 6 and done[2] >= 30, done


def test_full_size_properties_c2(gpu_ctx, pkg, plref):
    """BASELINE config #2 at full size (N=32, 64->128, 56x56, k3 s1 p1): too big for the scalar oracle in
    seconds, so check size-independent properties: (i) linearity in the weights acc(w1+w2) = acc(w1)+acc(w2);
    (ii) batch independence (image b alone gives the same slice); (iii) a sampled set of outputs vs the oracle."""
    capi = pkg.capi
    rng = np.random.default_rng(109)
    n, cin, cout, hw = 32, 64, 128, 56
    x = rng.integers(-127, 128, (n, cin, hw, hw)).astype(np.int8)
    w1 = rng.integers(-63, 64, (cout, cin, 3, 3)).astype(np.int8)
    w2 = rng.integers(-63, 64, (cout, cin, 3, 3)).astype(np.int8)
    d = capi.conv_desc(n, cin, hw, hw, cout, 3, 3, (1, 1, 1, 1), (1, 1), (1, 1), 1)
    a1 = gpu_ctx.conv2d(d, x, w1, None, None, capi.OUT_I32)
    a2 = gpu_ctx.conv2d(d, x, w2, None, None, capi.OUT_I32)
    a12 = gpu_ctx.conv2d(d, x, (w1 + w2).astype(np.int8), None, None, capi.OUT_I32)
    assert np.array_equal(a12, a1 + a2)
    d1 = capi.conv_desc(1, cin, hw, hw, cout, 3, 3, (1, 1, 1, 1), (1, 1), (1, 1), 1)
    b = 17
    assert np.array_equal(gpu_ctx.conv2d(d1, x[b:b + 1], w1, None, None, capi.OUT_I32)[0], a1[b])
    s1 = plref.shape(1, cin, hw, hw, cout, 3, 3, (1, 1, 1, 1), (1, 1), (1, 1), 1)
    assert np.array_equal(plref.conv2d_acc(s1, x[3:4], w1)[0], a1[3])

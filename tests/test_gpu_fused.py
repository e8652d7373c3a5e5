"""GPU: fused depthwise3x3 -> pointwise1x1 (plhip_dwpw_fused_int8) must be bit-identical to the two-kernel path and to
the oracle's two-stage computation (depthwise int8_out, then 1x1 conv)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _case(ctx, capi, plref, rng, n, c, h, w, stride, pad, m, dw_act, pw_act, int8_out, pw_alpha=0.0, dw_alpha=0.0):
    x = rng.integers(-127, 128, (n, c, h, w)).astype(np.int8)
    w_dw = rng.integers(-127, 128, (c, 1, 3, 3)).astype(np.int8)
    w_pw = rng.integers(-127, 128, (m, c, 1, 1)).astype(np.int8)
    b_dw = rng.uniform(-1, 1, c).astype(np.float32)
    b_pw = rng.uniform(-1, 1, m).astype(np.float32)
    ws_dw = ((1 + np.arange(c) % 7) / 127.0 / 4.0).astype(np.float32)
    ws_pw = ((1 + np.arange(m) % 5) / 127.0 / 4.0).astype(np.float32)
    in_s, mid_s = 1 / 127.0, (9 / 127.0 if dw_act != 2 else dw_alpha / 127.0)
    out_s = c / 127.0 / 8 if pw_act != 2 else pw_alpha / 127.0
    sd = plref.shape(n, c, h, w, c, 3, 3, pad, (stride, stride), (1, 1), c)
    oh, ow = plref.out_dims(sd)
    # oracle, stage 1: depthwise int8_out
    s1, b1, a1 = plref.fold_scales(1, in_s, ws_dw, mid_s, b_dw, c, dw_act, dw_alpha)
    d_ref, _ = plref.conv2d(sd, x, w_dw, b_dw, in_s, ws_dw, mid_s, dw_act, dw_alpha, True)
    # stage 2: pointwise
    sp = plref.shape(n, c, oh, ow, m, 1, 1, (0, 0, 0, 0), (1, 1), (1, 1), 1)
    y_ref, acc_ref = plref.conv2d(sp, d_ref, w_pw, b_pw, mid_s, ws_pw, out_s, pw_act, pw_alpha, int8_out)
    s2, b2, a2 = plref.fold_scales(int(int8_out), mid_s, ws_pw, out_s, b_pw, m, pw_act, pw_alpha)
    d_dw = capi.conv_desc(n, c, h, w, c, 3, 3, pad, (stride, stride), (1, 1), c, dw_act, a1)
    import ctypes
    if not ctx.L.plhip_dwpw_fused_supported(ctypes.byref(d_dw), m, capi.OUT_I8 if int8_out else capi.OUT_F32):
        return False  # outside the fused path (the predictor runs the two kernels): nothing to compare
    acc = ctx.dwpw_fused(d_dw, x, w_dw, s1, b1, w_pw, None, None, pw_act, a2, capi.OUT_I32)
    assert np.array_equal(acc, acc_ref), "fused int32 accumulators differ"
    y = ctx.dwpw_fused(d_dw, x, w_dw, s1, b1, w_pw, s2, b2, pw_act, a2, capi.OUT_I8 if int8_out else capi.OUT_F32)
    if int8_out:
        assert np.array_equal(y, y_ref), "fused int8 output differs"
    else:
        np.testing.assert_allclose(y, y_ref, rtol=1e-5, atol=1e-6)
    return True


def test_fused_matches_two_stage_oracle(gpu_ctx, pkg, plref):
    rng = np.random.default_rng(300)
    capi = pkg.capi
    cases = [
        # n, c, h, w, stride, pad(t,b,l,r), m, dw_act, pw_act, int8_out
        (2, 512, 14, 14, 1, (1, 1, 1, 1), 512, 1, 1, True),     # MobileNet dw8 / pw8: 4 rounds, two m tiles per wave
        (3, 128, 14, 14, 1, (1, 1, 1, 1), 256, 1, 1, True),     # one round (produce only, then consume only), one m tile per wave
        (1, 256, 14, 14, 1, (1, 1, 1, 1), 512, 2, 2, True),     # relu6 both; a single image: both tiles are first / last planes
        (5, 384, 14, 14, 1, (1, 1, 1, 1), 256, 0, 4, True),     # no depthwise activation, leaky pointwise, 3 rounds, 10 tiles on 16 blocks
        (2, 256, 14, 14, 1, (1, 1, 1, 1), 256, 1, 0, False),    # fp32 output
        (1, 128, 14, 14, 1, (1, 1, 1, 1), 512, 4, 1, True),     # leaky depthwise
        (9, 512, 14, 14, 1, (1, 1, 1, 1), 512, 1, 2, True),     # 18 tiles: ragged XCD shares
        # the streaming kernel of the large planes (fused_dwpw_stream.hip)
        (2, 32, 112, 112, 1, (1, 1, 1, 1), 64, 1, 1, True),     # dw2 / pw2: 2-row tiles, 2 m x 2 n wave splits
        (3, 128, 56, 56, 1, (1, 1, 1, 1), 128, 2, 2, True),     # dw4 / pw4, relu6 both: 4-row tiles, a wave = one m tile x 7 n tiles
        (2, 256, 28, 28, 1, (1, 1, 1, 1), 256, 1, 1, True),     # dw6 / pw6: 8-row tiles, the last tile of an image half empty
        (1, 128, 56, 56, 1, (1, 1, 1, 1), 128, 0, 4, False),    # no depthwise activation, leaky pointwise, fp32 output
        (1, 32, 112, 112, 1, (1, 1, 1, 1), 64, 4, 0, True),     # leaky depthwise, no pointwise activation
        # ... and its stride-2 form (12 aligned bytes per input row; the tensor's first quad fetches from column 0)
        (2, 64, 112, 112, 2, (1, 1, 1, 1), 128, 1, 1, True),    # dw3 / pw3
        (3, 128, 56, 56, 2, (1, 1, 1, 1), 256, 2, 2, True),     # dw5 / pw5, relu6 both: the last tile of an image half empty
        (1, 64, 112, 112, 2, (1, 1, 1, 1), 128, 0, 4, False),   # no depthwise activation, leaky pointwise, fp32 output
        (1, 128, 56, 56, 2, (1, 1, 1, 1), 256, 4, 0, True),     # leaky depthwise
        # ... on the 14-wide plane: 16-slot rows (2 junk), half-image tiles, the output channels in two passes, 14-byte row stores
        (3, 256, 28, 28, 2, (1, 1, 1, 1), 512, 1, 1, True),     # dw7 / pw7
        (1, 256, 28, 28, 2, (1, 1, 1, 1), 512, 2, 0, False),    # one image (its first and last lanes fetch shifted), fp32 output
        (2, 256, 28, 28, 2, (1, 1, 1, 1), 512, 4, 4, True),     # leaky both
        # the 7 x 7 planes (fused_dwpw_small.hip): one image per block pair, lane = (channel, output row)
        (3, 512, 14, 14, 2, (1, 1, 1, 1), 1024, 1, 1, True),    # dw13 / pw13
        (2, 1024, 7, 7, 1, (1, 1, 1, 1), 1024, 1, 1, False),    # dw14 / pw14: fp32 output (what the pool reads)
        (1, 1024, 7, 7, 1, (1, 1, 1, 1), 1024, 2, 2, True),     # one image (its first lane fetches shifted), relu6 both, int8 output
        (1, 512, 14, 14, 2, (1, 1, 1, 1), 1024, 4, 0, False),   # leaky depthwise, fp32 output
        (5, 512, 14, 14, 2, (1, 1, 1, 1), 1024, 0, 4, True),    # 10 blocks on 16: ragged XCD shares; leaky pointwise
        # outside the fused path (the predictor runs the two kernels): reported as unsupported
        (2, 32, 16, 16, 1, (1, 1, 1, 1), 64, 1, 1, True),
        (2, 64, 16, 16, 2, (1, 1, 1, 1), 128, 1, 1, True),
        (2, 64, 112, 112, 2, (0, 1, 0, 1), 128, 1, 1, True),    # stride 2 with the padding on the other side
        (2, 64, 56, 56, 1, (1, 1, 1, 1), 128, 1, 1, True),      # a large plane with another channel count
        (2, 96, 7, 7, 1, (1, 1, 1, 1), 160, 1, 1, False),
        (1, 40, 9, 13, 2, (0, 1, 1, 0), 33, 0, 4, True),
        (5, 256, 28, 28, 2, (1, 1, 1, 1), 256, 2, 2, True),
        (1, 512, 7, 7, 1, (1, 1, 1, 1), 1024, 1, 1, False),
        (1, 16, 112, 112, 1, (1, 1, 1, 1), 24, 1, 0, True),
        (2, 512, 14, 14, 1, (0, 1, 1, 1), 512, 1, 1, True),     # top padding 0: outside
        (2, 192, 14, 14, 1, (1, 1, 1, 1), 256, 1, 1, True),     # C % 128 != 0: outside
    ]
    ran = []
    for (n, c, h, w, st, pad, m, da, pa, i8) in cases:
        ran.append(_case(gpu_ctx, capi, plref, rng, n, c, h, w, st, pad, m, da, pa, i8, pw_alpha=(6.0 if pa == 2 else 0.3),
                         dw_alpha=(6.0 if da == 2 else (0.2 if da == 4 else 0.0))))
    print("fused cases run:", ran)
    assert ran[:24] == [True] * 24 and not any(ran[24:]), ran


def test_fused_pair_with_the_global_average_as_output(gpu_ctx, pkg, plref):
    """PLHIP_OUT_F32_GAP: the pair's fp32 output averaged over each 7 x 7 plane in the launch that produces it = the instructions
    conv2d[fp32_out] -> pool2d(avg, global) of the reference program (pooling.cc:1006-): equal to the oracle's global average of
    the oracle's fp32 output within 1e-5 (the sum's order differs; the values summed are the bit-identical fp32 outputs)."""
    import ctypes
    capi = pkg.capi
    rng = np.random.default_rng(301)
    for (n, c, hw, st, m, dw_act, pw_act) in [(2, 1024, 7, 1, 1024, 1, 1), (3, 512, 14, 2, 1024, 1, 0), (1, 1024, 7, 1, 1024, 2, 4)]:
        x = rng.integers(-127, 128, (n, c, hw, hw)).astype(np.int8)
        w_dw = rng.integers(-127, 128, (c, 1, 3, 3)).astype(np.int8)
        w_pw = rng.integers(-127, 128, (m, c, 1, 1)).astype(np.int8)
        b_dw = rng.uniform(-1, 1, c).astype(np.float32)
        b_pw = rng.uniform(-1, 1, m).astype(np.float32)
        ws_dw = ((1 + np.arange(c) % 7) / 127.0 / 4.0).astype(np.float32)
        ws_pw = ((1 + np.arange(m) % 5) / 127.0 / 4.0).astype(np.float32)
        dw_alpha = 6.0 if dw_act == 2 else 0.0
        in_s, mid_s = 1 / 127.0, (9 / 127.0 if dw_act != 2 else dw_alpha / 127.0)
        sd = plref.shape(n, c, hw, hw, c, 3, 3, (1, 1, 1, 1), (st, st), (1, 1), c)
        oh, ow = plref.out_dims(sd)
        s1, b1, a1 = plref.fold_scales(1, in_s, ws_dw, mid_s, b_dw, c, dw_act, dw_alpha)
        d_ref, _ = plref.conv2d(sd, x, w_dw, b_dw, in_s, ws_dw, mid_s, dw_act, dw_alpha, True)
        sp = plref.shape(n, c, oh, ow, m, 1, 1, (0, 0, 0, 0), (1, 1), (1, 1), 1)
        y_ref, _ = plref.conv2d(sp, d_ref, w_pw, b_pw, mid_s, ws_pw, 1.0, pw_act, 0.3, False)
        s2, b2, a2 = plref.fold_scales(0, mid_s, ws_pw, 1.0, b_pw, m, pw_act, 0.3)
        want = y_ref.astype(np.float64).mean(axis=(2, 3)).astype(np.float32).reshape(n, m, 1, 1)
        d_dw = capi.conv_desc(n, c, hw, hw, c, 3, 3, (1, 1, 1, 1), (st, st), (1, 1), c, dw_act, a1)
        assert gpu_ctx.L.plhip_dwpw_fused_supported(ctypes.byref(d_dw), m, capi.OUT_F32_GAP) == 1
        got = gpu_ctx.dwpw_fused(d_dw, x, w_dw, s1, b1, w_pw, s2, b2, pw_act, a2, capi.OUT_F32_GAP)
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6)
    # the planes of the other fused kernels have no averaged form
    d14 = capi.conv_desc(2, 512, 14, 14, 512, 3, 3, (1, 1, 1, 1), (1, 1), (1, 1), 512, 1, 0.0)
    assert gpu_ctx.L.plhip_dwpw_fused_supported(ctypes.byref(d14), 512, capi.OUT_F32_GAP) == 0
    assert gpu_ctx.L.plhip_dwpw_fused_supported(ctypes.byref(d14), 512, capi.OUT_F32) == 1
    for (c, hw, st, m) in [(128, 56, 1, 128), (256, 28, 1, 256), (256, 28, 2, 512), (64, 112, 2, 128)]:  # the streaming kernel's shapes
        dd = capi.conv_desc(2, c, hw, hw, c, 3, 3, (1, 1, 1, 1), (st, st), (1, 1), c, 1, 0.0)
        assert gpu_ctx.L.plhip_dwpw_fused_supported(ctypes.byref(dd), m, capi.OUT_F32_GAP) == 0, (c, hw, st, m)
        assert gpu_ctx.L.plhip_dwpw_fused_supported(ctypes.byref(dd), m, capi.OUT_F32) == 1, (c, hw, st, m)


def test_fused_unsupported_shapes_are_reported(gpu_ctx, pkg):
    import ctypes as C
    capi = pkg.capi
    d = capi.conv_desc(1, 8, 16, 16, 8, 5, 5, (2, 2, 2, 2), (1, 1), (1, 1), 8)  # 5x5 depthwise: not fused
    z = gpu_ctx.malloc(1 << 16)
    st = gpu_ctx.L.plhip_dwpw_fused_int8(gpu_ctx.h, C.byref(d), z, z, z, None, 8, z, z, None, 0, 0.0, z, capi.OUT_I8)
    assert st == -3
    gpu_ctx.free(z)

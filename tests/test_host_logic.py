"""CPU: the C-ABI library loads and exports every declared symbol; host-side bit tricks used by the
kernels are equivalent to the oracle's libm formulation."""
import ctypes
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(pkg):
    hdr = open(os.path.join(ROOT, "include", "plhip.h")).read()
    declared = sorted(set(re.findall(r"\b(plhip_[a-z0-9_]+)\s*\(", hdr)))
    assert set(declared) == set(pkg.capi.EXPORTS), set(declared) ^ set(pkg.capi.EXPORTS)
    lib = pkg.capi.load()
    for name in declared:
        assert hasattr(lib, name), name


def test_no_device_is_an_error_not_a_fallback(pkg):
    lib = pkg.capi.load()
    if lib.plhip_device_count() > 0:
        return
    h = ctypes.c_void_p()
    assert lib.plhip_ctx_create(0, ctypes.byref(h)) != 0
    try:
        pkg.capi.Context(0)
    except pkg.capi.PlhipError:
        pass
    else:
        raise AssertionError("Context() must raise without a GPU")


def test_descriptor_helpers(pkg):
    capi = pkg.capi
    lib = capi.load()
    d = capi.conv_desc(32, 64, 56, 56, 128, 3, 3, (1, 1, 1, 1), (1, 1), (1, 1), 1)
    assert capi.out_hw(d) == (56, 56)
    # C2: K = 576 -> 18 K-steps, M = 128 -> 4 fragment rows: 4*18 KiB packed
    assert lib.plhip_conv_packed_weight_bytes(ctypes.byref(d)) == 4 * 18 * 1024
    # C2 runs on the patch kernel (3x3 s1, Cin % 32 == 0): the workspace is the zero-padded input copy with rows of 64 bytes
    # (58 rounded up to a multiple of 8) + slack, not the im2col buffer; the packed weights keep their size (tap-major order)
    assert lib.plhip_conv_workspace_bytes(ctypes.byref(d)) == 32 * 64 * 58 * 64 + 4096  # (a multiple of 16)
    assert lib.plhip_conv_impl_name(ctypes.byref(d)) == b"conv_patch_gemm_int8_mfma32x32x32"
    # 3x3 with a channel tail (Cin % 32 != 0) stays on the ring kernel's implicit GEMM: 58 x 58 planes + slack
    d48 = capi.conv_desc(32, 48, 56, 56, 128, 3, 3, (1, 1, 1, 1), (1, 1), (1, 1), 1)
    assert lib.plhip_conv_workspace_bytes(ctypes.byref(d48)) == 32 * 48 * 58 * 58 + 64
    assert lib.plhip_conv_impl_name(ctypes.byref(d48)) == b"conv_implicit_gemm_int8_mfma32x32x32"
    # 3x3 stride 2, Cin % 32 == 0, M > 64: the patch kernel over the PHASE-SPLIT padded copy: 4 planes of ceil(58/2) rows of
    # roundup(ceil(58/2), 8) bytes per channel + slack; the packed weights: 9 fragments per (m tile, 32 channels)
    s2 = capi.conv_desc(32, 64, 56, 56, 128, 3, 3, (1, 1, 1, 1), (2, 2), (1, 1), 1)
    assert lib.plhip_conv_workspace_bytes(ctypes.byref(s2)) == 32 * 64 * 4 * 29 * 32 + 4096
    assert lib.plhip_conv_packed_weight_bytes(ctypes.byref(s2)) == 4 * 2 * 9 * 1024
    assert lib.plhip_conv_impl_name(ctypes.byref(s2)) == b"conv_patch_s2_gemm_int8_mfma32x32x32"
    # ... with a channel tail or M <= 64: the ring kernel's implicit GEMM on its own phase-split copy (rows rounded up to 4)
    s2b = capi.conv_desc(32, 48, 56, 56, 128, 3, 3, (1, 1, 1, 1), (2, 2), (1, 1), 1)
    assert lib.plhip_conv_workspace_bytes(ctypes.byref(s2b)) == 32 * 48 * 4 * 29 * 32 + 64
    assert lib.plhip_conv_impl_name(ctypes.byref(s2b)) == b"conv_implicit_gemm_int8_mfma32x32x32"
    s2c = capi.conv_desc(32, 64, 56, 56, 64, 3, 3, (1, 1, 1, 1), (2, 2), (1, 1), 1)
    assert lib.plhip_conv_impl_name(ctypes.byref(s2c)) == b"conv_implicit_gemm_int8_mfma32x32x32"
    # ResNet50's 7x7 stride-2 stem (Cin <= 3, OW % 4 == 0): its own direct kernel, no workspace, 6 K-steps of A fragments per m tile
    st7 = capi.conv_desc(256, 3, 224, 224, 64, 7, 7, (3, 3, 3, 3), (2, 2), (1, 1), 1)
    assert lib.plhip_conv_impl_name(ctypes.byref(st7)) == b"conv_7x7s2_direct_int8_mfma32x32x32"
    assert lib.plhip_conv_workspace_bytes(ctypes.byref(st7)) == 0
    assert lib.plhip_conv_packed_weight_bytes(ctypes.byref(st7)) == 2 * 6 * 1024
    st7b = capi.conv_desc(2, 4, 224, 224, 64, 7, 7, (3, 3, 3, 3), (2, 2), (1, 1), 1)  # 28 filter rows: the implicit GEMM
    assert lib.plhip_conv_impl_name(ctypes.byref(st7b)) == b"conv_implicit_gemm_int8_mfma32x32x32"
    st7c = capi.conv_desc(2, 3, 224, 226, 64, 7, 7, (3, 3, 3, 3), (2, 2), (1, 1), 1)  # OW = 113: not whole quads
    assert lib.plhip_conv_impl_name(ctypes.byref(st7c)) == b"conv_implicit_gemm_int8_mfma32x32x32"
    dil = capi.conv_desc(32, 64, 56, 56, 128, 3, 3, (2, 2, 2, 2), (1, 1), (2, 2), 1)  # dilation stays on im2col + GEMM
    assert lib.plhip_conv_workspace_bytes(ctypes.byref(dil)) == 32 * 576 * 3136
    assert lib.plhip_conv_impl_name(ctypes.byref(dil)) == b"conv_im2col_gemm_int8_mfma32x32x32"
    p = capi.conv_desc(128, 512, 14, 14, 512, 1, 1)
    assert lib.plhip_conv_workspace_bytes(ctypes.byref(p)) == 0
    assert lib.plhip_conv_impl_name(ctypes.byref(p)) == b"conv1x1s1_gemm_int8_mfma32x32x32"
    bad = capi.conv_desc(1, 6, 8, 8, 8, 3, 3, groups=4)  # cin % groups != 0
    assert lib.plhip_conv_packed_weight_bytes(ctypes.byref(bad)) == 0


def test_round_half_away_integer_trick(plref):
    """Device epilogue: q = (t + 1 + (t>>31)) >> 1, t = trunc(clamp(2y, -254, 254)) == clamp(roundf(y))."""
    rng = np.random.default_rng(5)
    y = np.concatenate([
        rng.uniform(-140, 140, 200000).astype(np.float32),
        (rng.integers(-300, 300, 20000) / 2.0).astype(np.float32),  # exact ties
        np.nextafter(np.float32(0.5), np.float32(0)).reshape(1), np.nextafter(np.float32(-0.5), np.float32(0)).reshape(1),
        np.nextafter((rng.integers(-260, 260, 20000) / 2.0).astype(np.float32), np.float32(1e9)),
        np.nextafter((rng.integers(-260, 260, 20000) / 2.0).astype(np.float32), np.float32(-1e9)),
        np.array([0.0, -0.0, 127.49999, 127.5, -127.5, 1e30, -1e30, np.inf, -np.inf], np.float32),
    ]).astype(np.float32)
    y2 = np.clip(y + y, np.float32(-254), np.float32(254))
    t = np.trunc(y2).astype(np.int32)
    q = (t + 1 + (t >> 31)) >> 1
    lib = plref.lib()
    ref = np.array([lib.plref_round_sat_i8(ctypes.c_float(v)) for v in y[:5000]], np.int32)
    assert np.array_equal(q[:5000], ref)
    # vectorised check of the rest through the calib oracle (scale 1 => q = round_sat(y))
    assert np.array_equal(q, plref.calib_f32_to_i8(y, 1.0).astype(np.int32))


def test_xcd_contiguous_maps_cover_every_tile_once():
    """Host mirror of the device-side work maps (csrc/gemm_i8.hip xcd_tile_map, the vb maps of the depthwise / stem
    kernels): block b runs on XCD b % 8 and takes the (b % 8)-th eighth of the work; every tile / virtual block must be
    visited exactly once, the padding blocks must fall outside, and the blocks sharing a B tile must sit on one XCD."""
    for mtb_n in (1, 2, 3, 4):
        for NT in (1, 7, 8, 9, 64, 196, 197, 203):
            ntx = (NT + 7) >> 3
            grid = 8 * mtb_n * ntx                     # launcher: mtb_n * roundup(NT, 8)
            assert grid == mtb_n * ((NT + 7) // 8 * 8)
            seen = {}
            for b in range(grid):
                x, q = b & 7, b >> 3
                j = q // mtb_n
                mtb, nt = q - j * mtb_n, x * ntx + j
                if nt >= NT:
                    continue
                assert (mtb, nt) not in seen
                seen[(mtb, nt)] = x
            assert len(seen) == mtb_n * NT
            for nt in range(NT):                       # one XCD per B tile
                assert len({seen[(m, nt)] for m in range(mtb_n)}) == 1
            xcd_of_tile = [seen[(0, nt)] for nt in range(NT)]
            assert xcd_of_tile == sorted(xcd_of_tile)  # contiguous ranges, in order
    for nb in (1, 5, 8, 9, 1000, 6272):               # depthwise / stem: virtual block id
        per = (nb + 7) >> 3
        vbs = [((b & 7) * per + (b >> 3)) for b in range(8 * per)]
        live = [v for v in vbs if v < nb]
        assert sorted(live) == list(range(nb))


def test_magic_division_is_exact_below_2_31():
    """Host mirror of fastdiv_u31 (csrc/dw_common.h) and of the (magic, shift) pairs launch_dw_direct_s prepares:
    q = mulhi(n, floor(2^(31+s)/d) + 1) >> (s-1), s = ceil(log2 d), must equal n // d for every n < 2^31."""
    rng = np.random.default_rng(5)
    for d in list(range(1, 130)) + [196, 255, 257, 1000, 1023, 1025, 4097, 65535, 65537, 1 << 20, (1 << 20) + 1, (1 << 30) - 1]:
        if d & (d - 1) == 0:
            magic, sh = 0, d.bit_length() - 1
        else:
            sc = (d - 1).bit_length()
            magic, sh = ((1 << (31 + sc)) // d) + 1, sc - 1
            assert magic < (1 << 32)
        ns = np.concatenate([np.arange(0, min(4 * d + 2, 5000)), (np.arange(1, 200) * d) - 1, np.arange(1, 200) * d,
                             rng.integers(0, 1 << 31, 2000), np.array([(1 << 31) - 1, (1 << 31) - d, ((1 << 31) - 1) // d * d])])
        ns = ns[(ns >= 0) & (ns < (1 << 31))].astype(np.uint64)
        q = (ns >> np.uint64(sh)) if magic == 0 else (((ns * np.uint64(magic)) >> np.uint64(32)) >> np.uint64(sh))
        assert np.array_equal(q, ns // np.uint64(d)), d



def test_fused_dwpw_support_query_is_host_only(pkg):
    """plhip_dwpw_fused_supported is a pure function of the descriptors (no device): the fused kernel takes the 14 x 14
    stride-1 pairs with 128 | C <= 512 and M = 256 / 512 (MobileNetV1's five 512 -> 512 pairs) and, on its streaming kernel,
    the 112 / 56 / 28-wide stride-1 pairs 32 -> 64, 128 -> 128, 256 -> 256 and the stride-2 pairs 112 -> 56 (64 -> 128),
    56 -> 28 (128 -> 256) and 28 -> 14 (256 -> 512) and, on the small-plane kernel, the two pairs that end on 7 x 7 (512 -> 1024
    stride 2, 1024 -> 1024), every output kind; other planes, strides, filters, channel counts are refused (the predictor then
    runs the two kernels)."""
    capi = pkg.capi
    lib = capi.load()

    def q(n, c, hw, s, m, out=capi.OUT_I8, k=3, groups=None, dil=1):
        d = capi.conv_desc(n, c, hw, hw, c, k, k, (k // 2,) * 4, (s, s), (dil, dil), c if groups is None else groups, capi.ACT_RELU, 0.0)
        return lib.plhip_dwpw_fused_supported(ctypes.byref(d), m, out)

    for (c, m) in [(512, 512), (128, 256), (256, 512), (384, 256)]:
        for out in (capi.OUT_I8, capi.OUT_F32, capi.OUT_I32):
            assert q(128, c, 14, 1, m, out=out) == 1 and q(1, c, 14, 1, m, out=out) == 1, (c, m, out)
    # the streaming kernel of the large planes: MobileNetV1's other stride-1 pairs
    for (c, hw, m) in [(32, 112, 64), (128, 56, 128), (256, 28, 256)]:
        for out in (capi.OUT_I8, capi.OUT_F32, capi.OUT_I32):
            assert q(128, c, hw, 1, m, out=out) == 1 and q(1, c, hw, 1, m, out=out) == 1, (c, hw, m, out)
    for out in (capi.OUT_I8, capi.OUT_F32, capi.OUT_I32, capi.OUT_F32_GAP):  # the 7 x 7 planes (also with the plane average as output)
        assert q(128, 1024, 7, 1, 1024, out=out) == 1 and q(1, 512, 14, 2, 1024, out=out) == 1, out
    # ... which no other kernel has
    assert q(128, 512, 14, 1, 512, out=capi.OUT_F32_GAP) == 0 and q(128, 128, 56, 1, 128, out=capi.OUT_F32_GAP) == 0
    assert q(128, 64, 112, 2, 128, out=capi.OUT_F32_GAP) == 0 and q(128, 256, 28, 2, 512, out=capi.OUT_F32_GAP) == 0
    for (c, hw, m) in [(512, 7, 1024), (1024, 7, 512), (64, 112, 128), (32, 112, 32), (128, 28, 128), (256, 56, 256)]:
        assert q(128, c, hw, 1, m) == 0, (c, hw, m)
    for out in (capi.OUT_I8, capi.OUT_F32, capi.OUT_I32):  # stride 2: dw3 / pw3, dw5 / pw5
        assert q(128, 64, 112, 2, 128, out=out) == 1 and q(1, 128, 56, 2, 256, out=out) == 1 and q(7, 256, 28, 2, 512, out=out) == 1, out
    assert q(128, 64, 112, 2, 64) == 0 and q(128, 128, 56, 2, 128) == 0 and q(128, 256, 28, 2, 256) == 0 and q(8, 64, 56, 2, 128) == 0
    assert q(128, 512, 14, 2, 512) == 0 and q(128, 512, 14, 1, 1024) == 0 and q(128, 192, 14, 1, 256) == 0 and q(128, 640, 14, 1, 512) == 0
    assert q(128, 512, 14, 1, 512, k=5) == 0 and q(128, 512, 14, 1, 512, dil=2) == 0 and q(128, 512, 14, 1, 512, groups=1) == 0


def test_calib_conv_support_query_is_host_only(pkg):
    """plhip_conv2d_calib_supported (calib[fp32_to_int8] + conv in one launch) is a pure function of the descriptor: the 3x3
    stride-2 stem with Cin <= 3, left padding 1, top padding <= 1, W % 4 == 0 and OW % 4 == 0; everything else is refused (the
    builder then keeps the two instructions)."""
    capi = pkg.capi
    lib = capi.load()

    def q(cin, h, w, k=3, s=2, pads=(1, 1, 1, 1), cout=32, groups=1):
        d = capi.conv_desc(4, cin, h, w, cout, k, k, pads, (s, s), (1, 1), groups, capi.ACT_RELU, 0.0)
        return lib.plhip_conv2d_calib_supported(ctypes.byref(d))

    assert q(3, 224, 224) == 1 and q(1, 64, 64) == 1 and q(3, 64, 64, pads=(0, 1, 1, 0)) == 1 and q(3, 224, 224, cout=40) == 1
    assert q(4, 224, 224) == 0 and q(3, 224, 224, s=1) == 0 and q(3, 224, 224, k=7, pads=(3, 3, 3, 3)) == 0
    assert q(3, 224, 224, pads=(1, 1, 0, 1)) == 0 and q(3, 224, 226) == 0 and q(3, 224, 228) == 0 and q(3, 224, 224, pads=(2, 2, 1, 1)) == 0

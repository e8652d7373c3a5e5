"""CPU: pin the oracle (oracle/plref.c) against the golden vectors minted from the reference's own
scalar oracle, and (in the authoring container) against the reference header compiled in place."""
import os

import numpy as np
import pytest

from conftest import golden_files, load_golden

CONV = golden_files("conv_")
GEMM = golden_files("gemm_")


def _shape(plref, g):
    n, cin, h, w = g["x"].shape
    cout, _, kh, kw = g["w"].shape
    st, dl = int(g["stride"]), int(g["dil"])
    return plref.shape(n, cin, h, w, cout, kh, kw, tuple(int(p) for p in g["pads"]), (st, st), (dl, dl), int(g["groups"]))


def test_golden_present():
    assert len(CONV) >= 12 and len(GEMM) >= 4


@pytest.mark.parametrize("path", CONV, ids=[os.path.basename(p)[5:-4] for p in CONV])
def test_conv_acc_matches_reference_golden(plref, path):
    g = load_golden(path)
    s = _shape(plref, g)
    assert np.array_equal(plref.conv2d_acc(s, g["x"], g["w"]), g["acc_ref"])
    # the im2col+GEMM structuring (the timed CPU baseline) gives the same accumulators
    assert np.array_equal(plref.conv2d_acc(s, g["x"], g["w"], via_gemm=True), g["acc_ref"])


@pytest.mark.parametrize("path", CONV, ids=[os.path.basename(p)[5:-4] for p in CONV])
def test_conv_epilogue_matches_golden_and_reference_tolerance(plref, path):
    g = load_golden(path)
    bias = g.get("bias")
    cout = g["w"].shape[0]
    act, alpha = int(g["act"]), float(g["alpha"])
    for int8_out, key in ((0, "y_f32"), (1, "y_i8")):
        sc, bi, al = plref.fold_scales(int8_out, float(g["in_scale"]), g["w_scale"], float(g["out_scale"]), bias, cout, act, alpha)
        y = plref.epilogue(g["acc_ref"], sc, bi, act, al, bool(int8_out))
        assert np.array_equal(y, g[key])
    # reference methodology (conv_int8_compute_test.cc:364-431): fp32-out vs the float baseline
    base = g["f32_baseline"]
    diff = np.abs(g["y_f32"] - base)
    ratio = diff / (np.abs(base) + 1e-6)
    # the reference's rule, nothing weaker (:371-372): fail iff max ratio > 1e-5 AND max diff > 5e-5
    assert not (ratio.max() > 1e-5 and diff.max() > 5e-5), "fp32 epilogue strays from the reference float baseline"
    # int8-out: |delta| <= 1 LSB everywhere, and fewer than max(10, 1%) mismatches
    q_base = plref.calib_f32_to_i8(base, float(g["out_scale"]))
    d8 = np.abs(g["y_i8"].astype(np.int32) - q_base.astype(np.int32))
    assert d8.max() <= 1
    assert (d8 != 0).sum() < max(10, int(0.01 * d8.size))


@pytest.mark.parametrize("path", GEMM, ids=[os.path.basename(p)[5:-4] for p in GEMM])
def test_gemm_acc_matches_reference_golden(plref, path):
    g = load_golden(path)
    assert np.array_equal(plref.gemm_acc(g["a"], g["b"]), g["acc_ref"])


def test_fc_and_calib_golden(plref):
    g = load_golden(golden_files("fc_")[0])
    y, acc = plref.fc(g["x"], g["w"], g["bias"], g["scale"], True, False)
    assert np.array_equal(acc, g["acc_ref"]) and np.array_equal(y, g["y_f32"])
    y8, _ = plref.fc(g["x"], g["w"], g["bias8"], g["scale8"], True, True)
    assert np.array_equal(y8, g["y_i8"])
    c = load_golden(golden_files("calib")[0])
    assert np.array_equal(plref.calib_f32_to_i8(c["x"], float(c["scale"])), c["q"])
    assert np.array_equal(plref.calib_i8_to_f32(c["q"], float(c["scale"])), c["deq"])
    # ties away from zero, floor at -127 (type_trans.cc:183-184)
    q = plref.calib_f32_to_i8(np.array([0.5, -0.5, 1.5, 2.5, -2.5, 0.49999997, 300, -300], np.float32), 1.0)
    assert q.tolist() == [1, -1, 2, 3, -3, 0, 127, -127]


def test_oracle_vs_reference_header_in_place(plref):
    """Authoring container only: the reference's conv_basic / basic_gemm compiled from /root/reference."""
    if plref.ref_lib() is None:
        pytest.skip("oracle/_ref not built (reference tree absent)")
    rng = np.random.default_rng(11)
    # grid borrowed from conv_int8_compute_test.cc:676-731 (random k/stride/pad/dilation/group)
    for cin, cout, g in ((1, 1, 1), (17, 8, 1), (8, 8, 2), (6, 6, 6)):
        for kh, kw in ((1, 1), (2, 3), (3, 3), (1, 2)):
            for st in (1, 2):
                for pads in ((0, 0, 0, 0), (1, 2, 0, 1), (2, 2, 2, 2)):
                    for dl in (1, 2):
                        for h in (1, 3, 5, 19):
                            s = plref.shape(2, cin, h, h, cout, kh, kw, pads, (st, st), (dl, dl), g)
                            oh, ow = plref.out_dims(s)
                            if oh < 1 or ow < 1:
                                continue
                            x = rng.integers(-127, 128, (2, cin, h, h)).astype(np.int8)
                            w = rng.integers(-127, 128, (cout, cin // g, kh, kw)).astype(np.int8)
                            assert np.array_equal(plref.conv2d_acc(s, x, w), plref.ref_conv_acc(s, x, w))
    # GEMM grid gemm_int8_compute_test.cc:350-357 (subset)
    for m in (1, 3, 33, 38):
        for n in (1, 13, 141):
            for k in (1, 3, 59, 67):
                a = rng.integers(-127, 128, (m, k)).astype(np.int8)
                b = rng.integers(-127, 128, (k, n)).astype(np.int8)
                assert np.array_equal(plref.gemm_acc(a, b), plref.ref_gemm_acc(a, b))


def test_im2col_layout(plref):
    """K x N row-major, k = c*kh*kw + r*kw + q (conv_impl.cc:103-153): im2col + GEMM == direct conv."""
    rng = np.random.default_rng(3)
    x = rng.integers(-127, 128, (5, 9, 7)).astype(np.int8)
    w = rng.integers(-127, 128, (4, 5, 3, 2)).astype(np.int8)
    col = plref.im2col(x, 3, 2, (1, 0, 1, 1), (2, 1), (1, 2))
    s = plref.shape(1, 5, 9, 7, 4, 3, 2, (1, 0, 1, 1), (2, 1), (1, 2), 1)
    acc = plref.conv2d_acc(s, x[None], w)
    assert np.array_equal(plref.gemm_acc(w.reshape(4, -1), col).reshape(acc.shape), acc)

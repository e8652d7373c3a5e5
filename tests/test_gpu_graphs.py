"""-m gpu: the glue ops and the whole ResNet50-INT8 / MobileNetV2-INT8 programs (BASELINE configs C4 / C5) against the
oracle graph, every variable of the lowered program; and the MobileNetV1 program (C3) at its own batch, 128."""
import importlib

import numpy as np
import pytest

from oracle import graph_oracle
import mbv1_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lite(pkg):
    return importlib.import_module("paddle_lite_amd.liteapi")


@pytest.fixture(scope="module")
def wl(pkg):
    return importlib.import_module("paddle_lite_amd.workloads")


def test_pool2d_sweep_vs_oracle(gpu_ctx, plref):
    """pooling_basic semantics (pooling.cc:38-215): max and avg (exclusive and not), pads 0-2 incl. asymmetric,
    ceil_mode, windows hanging over the edge.  Same operation order on both sides: bit-exact."""
    rng = np.random.default_rng(300)
    cases = [  # n, c, h, w, k, s, pads, ceil
        (2, 64, 112, 112, 3, 2, (1, 1, 1, 1), False),   # ResNet50 pool1
        (1, 3, 7, 9, 2, 2, (0, 0, 0, 0), False), (2, 5, 15, 15, 3, 1, (1, 1, 1, 1), False),
        (1, 4, 8, 8, 3, 2, (0, 0, 0, 0), True), (1, 2, 14, 14, 2, 2, (1, 1, 1, 1), False),
        (1, 3, 13, 17, 3, 2, (0, 1, 1, 2), False), (3, 7, 5, 5, 5, 1, (2, 2, 2, 2), False),
        (1, 1, 6, 6, 3, 3, (1, 1, 1, 1), True), (1, 2, 33, 3, 3, 2, (1, 1, 1, 1), False),
    ]
    for (n, c, h, w, k, s, pads, ceil) in cases:
        x = rng.standard_normal((n, c, h, w)).astype(np.float32)
        for typ, excl in (("max", True), ("avg", True), ("avg", False)):
            got = gpu_ctx.pool2d(x, typ, (k, k), (s, s), pads, exclusive=excl, ceil_mode=ceil)
            ref = plref.pool2d(x, typ, (k, k), (s, s), pads, exclusive=excl, ceil_mode=ceil)
            assert got.shape == ref.shape
            assert np.array_equal(got, ref), (h, w, k, s, pads, typ, excl)


def test_elementwise_add_vs_oracle(gpu_ctx, plref):
    rng = np.random.default_rng(301)
    for shape in [(2, 256, 56, 56), (1, 3, 5, 7), (1, 1, 1, 1), (3, 24, 17, 17)]:
        x = rng.standard_normal(shape).astype(np.float32)
        y = rng.standard_normal(shape).astype(np.float32)
        for relu in (False, True):
            assert np.array_equal(gpu_ctx.elementwise_add(x, y, relu), plref.elementwise_add(x, y, relu))


def test_fc_both_reference_routes(gpu_ctx, plref, pkg):
    """fp32-out fc: bit 1 of the flag word selects gemm_s8 + fill_bias_fc's two roundings (fc_compute.cc:250-266);
    both routes bit-exact against their oracle restatements, for the MFMA (k % 32 == 0) and the dot4 kernels."""
    capi = pkg.capi
    rng = np.random.default_rng(302)
    for (m, k, n) in [(9, 256, 513), (4, 100, 37), (33, 1024, 1000)]:
        x = rng.integers(-127, 128, (m, k)).astype(np.int8)
        w = rng.integers(-127, 128, (k, n)).astype(np.int8)
        bias = rng.uniform(-1, 1, n).astype(np.float32)
        sc = np.full(n, 1.7 / 127 / 127, np.float32)
        for relu in (0, 1):
            y0, _ = plref.fc(x, w, bias, sc, relu, False, route=0)
            y1, _ = plref.fc(x, w, bias, sc, relu, False, route=1)
            assert np.array_equal(gpu_ctx.fc(x, w, sc, bias, relu, capi.OUT_F32), y0)
            assert np.array_equal(gpu_ctx.fc(x, w, sc, bias, relu | 2, capi.OUT_F32), y1)


def test_fc_kernel_classes_pick_the_reference_route(lite, plref):
    """FcCompute<kInt8,kFloat> with a single weight scale and m > 1 takes the gemm_s8 route, with per-column scales
    the gemv route (check_fc_use_gemm, fc_compute.cc:66-81); FcCompute<kInt8,kInt8> (alias int8out) driven through
    the kernel class."""
    rng = np.random.default_rng(303)
    m, k, n = 6, 128, 70
    x = rng.integers(-127, 128, (m, k)).astype(np.int8)
    w = rng.integers(-127, 128, (k, n)).astype(np.int8)
    bias = rng.uniform(-1, 1, n).astype(np.float32)
    for ws, int8_out in [(np.array([0.9 / 127], np.float32), False), (((1 + np.arange(n) % 5) / 127.0 / 4).astype(np.float32), False),
                         (np.array([0.9 / 127], np.float32), True), (((1 + np.arange(n) % 5) / 127.0 / 4).astype(np.float32), True)]:
        in_scale, out_scale = 1 / 127.0, 0.05
        p = lite.Predictor(0)
        try:
            p.add_feed("x", x.shape, lite.PREC_INT8)
            p.add_io_copy("x", "xd", True)
            p.add_fc("xd", "yd", w, bias, in_scale, ws, out_scale, int8_out, True)
            p.add_io_copy("yd", "y", False)
            p.set_input("x", x)
            p.run()
            y = p.get_var("y", np.int8 if int8_out else np.float32)
            names = p.kernel_names()
        finally:
            p.close()
        wsn = np.resize(ws, n).astype(np.float32)
        if int8_out:
            sc = (wsn * np.float32(in_scale) / np.float32(out_scale)).astype(np.float32)
            ref, _ = plref.fc(x, w, (bias / np.float32(out_scale)).astype(np.float32), sc, True, True)
            assert any("fc:hip/int8_t" in s_ or "int8out" in s_ for s_ in names)
        else:
            sc = (wsn * np.float32(in_scale)).astype(np.float32)
            ref, _ = plref.fc(x, w, bias, sc, True, False, route=plref.fc_route(m, ws.size))
        assert np.array_equal(y, ref), (ws.size, int8_out)


def _run_graph(lite, wl, net, img, fuse=False, fuse_dwpw=None):
    p = lite.Predictor(0)
    try:
        out = wl.emit_graph(p, net, img.shape[0], fuse=fuse, fuse_dwpw=fuse_dwpw)
        fetched = p.graph_lower()
        assert fetched == [out]
        p.set_input(net["input"], img)
        p.run()
        p.run(skip_io_copy=False)  # second launch: ReInitWhenNeeded no-op paths
        return p, out
    except Exception:
        p.close()
        raise


def _compare_all(p, ref, net):
    n_i8 = n_f32 = 0
    for name, want in ref.items():
        dev_name = name
        got = p.get_var(dev_name, want.dtype)
        assert got.shape == want.shape, name
        if want.dtype == np.int8:
            assert np.array_equal(got, want), "%s: %d of %d int8 values differ" % (name, (got != want).sum(), want.size)
            n_i8 += 1
        else:
            np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5, err_msg=name)
            n_f32 += 1
    return n_i8, n_f32


def test_resnet50_int8_program_vs_oracle_graph(lite, wl, plref):
    """BASELINE config C4's graph (53 convs, max pool, 16 residual adds with fused relu, avg pool, fc, softmax) at
    batch 2, 224x224: every int8 tensor of the lowered program bit-exact, fp32 tensors within 1e-5."""
    net = wl.resnet50_net()
    img = np.random.default_rng(310).uniform(-1, 1, (2, 3, 224, 224)).astype(np.float32)
    ref = graph_oracle.forward(plref, net, img)
    p, out = _run_graph(lite, wl, net, img)
    try:
        names = p.kernel_names()
        assert sum(n.startswith("conv2d") for n in names) == 53
        assert sum("fusion_elementwise_add_activation" in n for n in names) == 16
        assert sum(n.startswith("calib") for n in names) == 18 and sum(n.startswith("pool2d") for n in names) == 2
        n_i8, n_f32 = _compare_all(p, ref, net)
        assert n_i8 == 18 + 32 and n_f32 >= 21 + 16 + 2
        prob = p.get_var(out, np.float32)
        np.testing.assert_allclose(prob, ref["prob"], rtol=1e-4, atol=1e-7)
        for name in ["res2a/precision_trans", "res3d_branch2b", "res5c_branch2a"]:
            got = p.get_var(name, np.int8)
            assert 0.02 < (got != 0).mean() and (np.abs(got.astype(np.int32)) == 127).mean() < 0.2, name
    finally:
        p.close()


def test_mobilenet_v2_int8_program_vs_oracle_graph(lite, wl, plref):
    """BASELINE config C5's graph (relu6 epilogues, linear bottlenecks, 10 residual adds) at batch 3, 224x224."""
    net = wl.mobilenet_v2_net()
    img = np.random.default_rng(311).uniform(-1, 1, (3, 3, 224, 224)).astype(np.float32)
    ref = graph_oracle.forward(plref, net, img)
    p, out = _run_graph(lite, wl, net, img)
    try:
        names = p.kernel_names()
        assert sum(n.startswith("depthwise_conv2d") for n in names) == 17
        assert sum(n.startswith("conv2d") for n in names) == 35
        assert sum(n.startswith("elementwise_add") for n in names) == 10
        _compare_all(p, ref, net)
        # the relu6 clip is visible in int8: 6 / (8/127) = 95
        e = p.get_var("b2_expand", np.int8)
        assert e.max() == 95 and (e == 95).mean() > 1e-4
        np.testing.assert_allclose(p.get_var(out, np.float32), ref["prob"], rtol=1e-4, atol=1e-7)
    finally:
        p.close()


def test_mobilenet_v1_int8_program_at_batch_128(lite, wl, plref):
    """C3 at its own batch: grid sizes, XCD tile maps and the 2-blocks-per-CU regime of the benchmark run, compared with
    the oracle graph (im2col + GEMM form of the oracle for speed: same accumulators bit for bit, tests/test_oracle.py)."""
    B = 128
    W = wl.make_mobilenet_v1_weights(seed=1234)
    img = np.random.default_rng(312).uniform(-1, 1, (B, 3, 224, 224)).astype(np.float32)
    ref = mbv1_oracle.forward(plref, wl, W, img, via_gemm=True)
    p = lite.Predictor(0)
    try:
        out = wl.build_mobilenet_v1(p, W, B)
        p.set_input("image", img)
        p.run()
        for name, want in ref.items():
            if want.dtype != np.int8:
                continue
            got = p.get_var(name, np.int8)
            assert np.array_equal(got, want), "%s: %d of %d differ" % (name, (got != want).sum(), want.size)
        np.testing.assert_allclose(p.get_var("pw14", np.float32), ref["pw14"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(p.get_var("logits", np.float32), ref["logits"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(p.get_var(out, np.float32), ref["prob"], rtol=1e-4, atol=1e-7)
    finally:
        p.close()


def _big_batch_check(lite, wl, plref, net, B, seed, fuse, mid):
    """The program at its BENCHMARK batch (grid sizes, XCD tile maps, multi-round wide tiles, phase-split copies of that
    size): the first two images must reproduce the batch-2 run bit for bit (that run is compared with the oracle variable
    by variable in the tests above), and one image from the middle of the batch is compared with the oracle directly."""
    rng = np.random.default_rng(seed)
    c, h, w = net["input_shape"]
    img = rng.uniform(-1, 1, (B, c, h, w)).astype(np.float32)
    ref_mid = graph_oracle.forward(plref, net, img[mid:mid + 1], via_gemm=True)
    small, out = _run_graph(lite, wl, net, img[:2], fuse=fuse)
    try:
        big, out_b = _run_graph(lite, wl, net, img, fuse=fuse)
        try:
            assert out == out_b
            n_i8 = n_f32 = 0
            for name, want in ref_mid.items():
                try:
                    s_ = small.get_var(name, want.dtype)
                except Exception:  # noqa: BLE001  (a variable the fused program does not materialise)
                    continue
                g_ = big.get_var(name, want.dtype, max_bytes=int(want.nbytes) * B + 64)
                assert g_.shape[0] == B and s_.shape[0] == 2, name
                if want.dtype == np.int8:
                    assert np.array_equal(g_[:2], s_), "%s: batch-%d prefix differs from the batch-2 run" % (name, B)
                    assert np.array_equal(g_[mid:mid + 1], want), "%s: image %d differs from the oracle" % (name, mid)
                    n_i8 += 1
                else:
                    np.testing.assert_array_equal(g_[:2], s_, err_msg=name)  # same kernels, same order: identical
                    np.testing.assert_allclose(g_[mid:mid + 1], want, rtol=1e-4 if name == out else 1e-5, atol=1e-5, err_msg=name)
                    n_f32 += 1
            return n_i8, n_f32
        finally:
            big.close()
    finally:
        small.close()


def test_resnet50_at_the_benchmark_batch_256(lite, wl, plref):
    """BASELINE config C4 at batch 256 (what bench.py --config c4 times), default (fused) program."""
    n_i8, n_f32 = _big_batch_check(lite, wl, plref, wl.resnet50_net(), 256, 320, fuse=True, mid=137)
    assert n_i8 >= 30 and n_f32 >= 2


def test_mobilenet_v2_at_the_benchmark_batch_1024(lite, wl, plref):
    """BASELINE config C5's global batch on one GPU (bench.py --config c5 on one rank times exactly this)."""
    n_i8, n_f32 = _big_batch_check(lite, wl, plref, wl.mobilenet_v2_net(), 1024, 321, fuse=True, mid=611)
    assert n_i8 >= 30 and n_f32 >= 2


def test_mobilenet_v2_at_the_per_rank_shard_of_an_8_gpu_run(lite, wl, plref):
    """BASELINE config C5 on 8 GPUs: 1024 images split by shard_range = 128 per rank (`bench.py --config c5 --gpus 8` runs
    exactly this program on every rank): the grid sizes, tile maps and XCD shares of THAT batch, prefix == the batch-2 run,
    one mid-batch image == the oracle."""
    n_i8, n_f32 = _big_batch_check(lite, wl, plref, wl.mobilenet_v2_net(), 128, 322, fuse=True, mid=77)
    assert n_i8 >= 30 and n_f32 >= 2


def test_pointwise_7x7_rows_at_the_end_of_an_allocation(gpu_ctx, plref, pkg):
    """HW = 49 rows are not a multiple of 4 bytes: the ring kernel's END-aligned 16-byte pieces must use the true row
    length (round 1 passed the length rounded up to 4 and read 3 bytes past the last row).  pw13's shape at batch 128:
    the input is exactly 784 pages and Context.to_device allocates exactly its size, so it ends where the allocation ends."""
    capi = pkg.capi
    rng = np.random.default_rng(320)
    n, cin, cout = 128, 512, 1024
    x = rng.integers(-127, 128, (n, cin, 7, 7)).astype(np.int8)
    assert x.nbytes % 4096 == 0
    w = rng.integers(-127, 128, (cout, cin, 1, 1)).astype(np.int8)
    d = capi.conv_desc(n, cin, 7, 7, cout, 1, 1)
    acc = gpu_ctx.conv2d(d, x, w, None, None, capi.OUT_I32)
    s = plref.shape(n, cin, 7, 7, cout, 1, 1, (0, 0, 0, 0), (1, 1), (1, 1), 1)
    assert np.array_equal(acc, plref.conv2d_acc(s, x, w, via_gemm=True))


def test_implicit_gemm_short_rows_and_small_m(gpu_ctx, plref, pkg):
    """Dense 3x3 stride-1 convs take the implicit-GEMM route of the transposed-read ring kernel for any M > 32 and for
    output rows down to 7 columns (one start-aligned 16-byte chunk per row whose trailing columns are garbage and must
    never be stored): ResNet50's 56x56 M = 64, 14x14 and 7x7 layers.  int32 accumulators, int8 and fp32 outputs."""
    capi = pkg.capi
    rng = np.random.default_rng(330)
    for (n, cin, cout, hw, k, pad) in [(3, 32, 48, 14, 3, 1), (2, 16, 64, 7, 3, 1), (2, 24, 40, 9, 3, 1), (1, 16, 96, 15, 3, 1),
                                       (2, 12, 64, 11, 5, 2), (2, 64, 64, 56, 3, 1)]:
        x = rng.integers(-127, 128, (n, cin, hw, hw)).astype(np.int8)
        w = rng.integers(-127, 128, (cout, cin, k, k)).astype(np.int8)
        bias = rng.uniform(-1, 1, cout).astype(np.float32)
        wsc = ((1 + np.arange(cout) % 7) / 127.0 / 4.0).astype(np.float32)
        d = capi.conv_desc(n, cin, hw, hw, cout, k, k, (pad,) * 4, (1, 1), (1, 1), 1, capi.ACT_RELU, 0.0)
        # (the 64 -> 64 56x56 layer has whole 32-channel chunks: it runs on the patch kernel, conv_patch_i8.hip)
        assert capi.load().plhip_conv_impl_name(d).decode().startswith(("conv_implicit_gemm", "conv_patch_gemm")), (cin, cout, hw)
        s = plref.shape(n, cin, hw, hw, cout, k, k, (pad,) * 4, (1, 1), (1, 1), 1)
        acc_ref = plref.conv2d_acc(s, x, w)
        assert np.array_equal(gpu_ctx.conv2d(d, x, w, None, None, capi.OUT_I32), acc_ref), (cin, cout, hw)
        for int8_out, kind in ((1, capi.OUT_I8), (0, capi.OUT_F32)):
            sc, bi, al = plref.fold_scales(int8_out, 1 / 127.0, wsc, cin * k * k / 127.0, bias, cout, 1, 0.0)
            y = gpu_ctx.conv2d(d, x, w, sc, bi, kind)
            ref = plref.epilogue(acc_ref, sc, bi, 1, al, bool(int8_out))
            if int8_out:
                assert np.array_equal(y, ref), (cin, cout, hw)
            else:
                np.testing.assert_allclose(y, ref, rtol=1e-5, atol=1e-6)


def test_implicit_gemm_stride2_phase_split(gpu_ctx, plref, pkg):
    """Dense stride-2 convs (ResNet50's 7x7 stem and 3x3 downsampling convs) run as implicit GEMM on a PHASE-SPLIT padded
    copy (4 planes per channel: rows / columns 2y+p, 2x+q), so each tap row is contiguous in ow again.  Odd and even
    extents, asymmetric pads, 3x3 / 5x5 / 7x7, short output rows; int32 accumulators, int8 and fp32 outputs."""
    capi = pkg.capi
    rng = np.random.default_rng(331)
    cases = [  # n, cin, cout, h, w, k, pads(t, b, l, r)
        (2, 3, 64, 64, 64, 7, (3, 3, 3, 3)),       # the stem's shape class (K = 147): direct kernel
        (2, 4, 64, 64, 64, 7, (3, 3, 3, 3)),       # Cin = 4 (28 filter rows): stays on the implicit GEMM
        (2, 32, 64, 28, 28, 3, (1, 1, 1, 1)),      # res3a/4a/5a branch2b class
        (3, 16, 48, 15, 17, 3, (1, 1, 1, 1)),      # odd extents
        (2, 16, 40, 14, 14, 3, (0, 1, 0, 1)),      # asymmetric pads, 7-column output rows
        (1, 8, 96, 33, 31, 5, (2, 2, 2, 2)),
        (2, 128, 128, 56, 56, 3, (1, 1, 1, 1)),    # res3a_branch2b itself: Cin % 32 == 0 and M > 64, the patch kernel's stride-2 form
    ]
    for (n, cin, cout, h, wd, k, pads) in cases:
        x = rng.integers(-127, 128, (n, cin, h, wd)).astype(np.int8)
        w = rng.integers(-127, 128, (cout, cin, k, k)).astype(np.int8)
        bias = rng.uniform(-1, 1, cout).astype(np.float32)
        wsc = ((1 + np.arange(cout) % 5) / 127.0 / 4.0).astype(np.float32)
        d = capi.conv_desc(n, cin, h, wd, cout, k, k, pads, (2, 2), (1, 1), 1, capi.ACT_RELU, 0.0)
        want = "conv_patch_s2_gemm" if (k == 3 and cin % 32 == 0 and cout > 64) else "conv_implicit_gemm"
        if k == 7 and cin <= 3:
            want = "conv_7x7s2_direct"  # the stem's own kernel (conv_stem7_i8.hip)
        assert capi.load().plhip_conv_impl_name(d).decode().startswith(want), (cin, cout, h, k)
        s = plref.shape(n, cin, h, wd, cout, k, k, pads, (2, 2), (1, 1), 1)
        acc_ref = plref.conv2d_acc(s, x, w)
        assert np.array_equal(gpu_ctx.conv2d(d, x, w, None, None, capi.OUT_I32), acc_ref), (cin, cout, h, k)
        for int8_out, kind in ((1, capi.OUT_I8), (0, capi.OUT_F32)):
            sc, bi, al = plref.fold_scales(int8_out, 1 / 127.0, wsc, cin * k * k / 127.0, bias, cout, 1, 0.0)
            y = gpu_ctx.conv2d(d, x, w, sc, bi, kind)
            ref = plref.epilogue(acc_ref, sc, bi, 1, al, bool(int8_out))
            if int8_out:
                assert np.array_equal(y, ref), (cin, cout, h, k)
            else:
                np.testing.assert_allclose(y, ref, rtol=1e-5, atol=1e-6)
    # a 1x1 stride-2 conv would read one phase plane of four: it keeps the strided-copy GEMM route
    d = capi.conv_desc(2, 256, 56, 56, 512, 1, 1, (0, 0, 0, 0), (2, 2), (1, 1), 1, capi.ACT_NONE, 0.0)
    assert not capi.load().plhip_conv_impl_name(d).decode().startswith("conv_implicit_gemm")


@pytest.mark.parametrize("which", ["resnet50", "mobilenet_v2"])
def test_fused_programs_equal_the_oracle_graph(lite, wl, plref, which):
    """Default (fused) lowering: conv + residual add + relu + calib in one launch, int8 max pool behind the stem.  Every
    variable the fused program still produces must equal the oracle's variable of the same name: int8 bit for bit, fp32
    within 1e-5 — i.e. the fusion changes the instruction count, not one byte of the data."""
    net = wl.resnet50_net() if which == "resnet50" else wl.mobilenet_v2_net()
    img = np.random.default_rng(340).uniform(-1, 1, (2, 3, 224, 224)).astype(np.float32)
    ref = graph_oracle.forward(plref, net, img)
    p, out = _run_graph(lite, wl, net, img, fuse=True)
    try:
        plan = p.graph_plan()
        # (MobileNetV2: 59 with the conv tails fused, one less with the calib in front of its 3x3 stride-2 stem taken over: fusion F)
        assert len(plan) == (61 if which == "resnet50" else 58)
        assert ("+calib_in=" in plan[1]) == (which == "mobilenet_v2")
        live = set()
        for l in plan:
            o = l.split(" out=")[1].split(" ")[0]
            if not l.endswith("-f32") and " -f32" not in l:
                live.add(o)
            if "+calib=" in l:
                live.add(l.split("+calib=")[1].split(" ")[0])
        n_i8 = 0
        for name, want in ref.items():
            if name not in live:
                continue
            got = p.get_var(name, want.dtype)
            assert got.shape == want.shape, name
            if want.dtype == np.int8:
                assert np.array_equal(got, want), "%s: %d of %d int8 values differ" % (name, (got != want).sum(), want.size)
                n_i8 += 1
            else:
                np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5, err_msg=name)
        assert n_i8 >= (49 if which == "resnet50" else 50)
        np.testing.assert_allclose(p.get_var(out, np.float32), ref["prob"], rtol=1e-4, atol=1e-7)
    finally:
        p.close()


def test_fused_conv_tail_through_the_c_abi(gpu_ctx, plref, pkg):
    """plhip_conv2d_int8_fused == conv[fp32_out] ; elementwise_add (+relu) ; calib, run one by one through the oracle:
    1x1 (first- and second-generation GEMM kernels), implicit 3x3, ragged HW, with and without each stage."""
    capi = pkg.capi
    rng = np.random.default_rng(341)
    for (n, cin, cout, hw, k, pad) in [(2, 64, 256, 14, 1, 0), (3, 16, 24, 7, 1, 0), (2, 160, 96, 9, 1, 0), (2, 32, 64, 14, 3, 1),
                                       (1, 512, 128, 28, 1, 0)]:
        x = rng.integers(-127, 128, (n, cin, hw, hw)).astype(np.int8)
        w = rng.integers(-127, 128, (cout, cin, k, k)).astype(np.int8)
        bias = rng.uniform(-1, 1, cout).astype(np.float32)
        wsc = ((1 + np.arange(cout) % 7) / 127.0 / 4.0).astype(np.float32)
        d = capi.conv_desc(n, cin, hw, hw, cout, k, k, (pad,) * 4, (1, 1), (1, 1), 1, capi.ACT_NONE, 0.0)
        s = plref.shape(n, cin, hw, hw, cout, k, k, (pad,) * 4, (1, 1), (1, 1), 1)
        sc, bi, _ = plref.fold_scales(0, 1 / 127.0, wsc, 1.0, bias, cout, 0, 0.0)
        y_ref, _ = plref.conv2d(s, x, w, bias, 1 / 127.0, wsc, 1.0, 0, 0.0, False)
        res = rng.standard_normal(y_ref.shape).astype(np.float32) * np.float32(y_ref.std())
        cs = float(np.abs(y_ref).max() / 100.0)
        for (use_res, relu, use_calib, want_f32) in [(True, True, True, True), (True, False, True, False), (False, False, True, True),
                                                     (True, True, False, True), (False, False, True, False)]:
            z = plref.elementwise_add(y_ref, res, relu) if use_res else y_ref
            q = plref.calib_f32_to_i8(z, cs) if use_calib else None
            yf, yq = gpu_ctx.conv2d_fused(d, x, w, sc, bi, res if use_res else None, relu, cs if use_calib else None, want_f32)
            if want_f32:
                np.testing.assert_allclose(yf, z, rtol=1e-5, atol=1e-6)
            if use_calib:
                # the int8 copy must be the quantisation of the fp32 values THIS launch produced (bit for bit when those
                # equal the oracle's, which they do except for fma-contraction-free ties: checked through yf when present)
                assert np.array_equal(yq, q), (cin, cout, hw, k, use_res, relu, int((yq != q).sum()))


def test_int8_max_pool_commutes_with_calib(gpu_ctx, plref):
    rng = np.random.default_rng(342)
    x = (rng.standard_normal((2, 64, 112, 112)) * 1.5).astype(np.float32)
    scale = 4.0 / 127
    a = plref.calib_f32_to_i8(plref.pool2d(x, "max", (3, 3), (2, 2), (1, 1, 1, 1)), scale)
    b = gpu_ctx.pool2d(plref.calib_f32_to_i8(x, scale), "max", (3, 3), (2, 2), (1, 1, 1, 1))
    assert b.dtype == np.int8 and np.array_equal(a, b)
    # 3x3 stride 2 has its own kernel (12-byte row windows, -128 outside the image): odd extents, every pad mix, tiny planes
    for (h, w, k, s, pads) in [(7, 9, 2, 2, (0, 0, 0, 0)), (13, 17, 3, 2, (0, 1, 1, 2)), (6, 6, 3, 3, (1, 1, 1, 1)),
                               (15, 15, 3, 2, (1, 1, 1, 1)), (8, 4, 3, 2, (1, 0, 1, 0)), (3, 4, 3, 2, (1, 1, 1, 1)),
                               (9, 33, 3, 2, (0, 0, 0, 0)), (28, 30, 3, 2, (1, 1, 0, 1))]:
        xi = rng.integers(-128, -100, (2, 3, h, w)).astype(np.int8) if h == 15 else rng.integers(-127, 128, (2, 3, h, w)).astype(np.int8)
        want = plref.pool2d(xi.astype(np.float32), "max", (k, k), (s, s), pads).astype(np.int8)
        assert np.array_equal(gpu_ctx.pool2d(xi, "max", (k, k), (s, s), pads), want)


@pytest.mark.parametrize("which,batch,mode", [("mobilenet_v1", 2, None), ("mobilenet_v1", 9, None), ("mobilenet_v1", 3, True), ("mobilenet_v2", 3, True)])
def test_dwpw_fusion_equals_the_oracle_graph(lite, wl, plref, which, batch, mode):
    """Depthwise -> pointwise fusion through the predictor and the kernel class (GraphBuilder fusion D; HipConvFusion::pw_*,
    lite/kernels/hip/conv_fusion.h).  mode None = the DEFAULT lowering: the pairs the fused kernels take (all thirteen of MobileNetV1 at
    224 x 224) are ONE launch of plhip_dwpw_fused_int8 each; mode True = every eligible pair is one
    instruction, the shapes outside the kernel as two launches inside it.  Every variable the program still produces equals
    the oracle's: int8 bit for bit, fp32 within 1e-5."""
    net = wl.mobilenet_v1_net() if which == "mobilenet_v1" else wl.mobilenet_v2_net()
    img = np.random.default_rng(350 + batch).uniform(-1, 1, (batch, 3, 224, 224)).astype(np.float32)
    ref = graph_oracle.forward(plref, net, img)
    p, out = _run_graph(lite, wl, net, img, fuse=True, fuse_dwpw=mode)
    try:
        plan = p.graph_plan()
        fused_lines = [l for l in plan if "+pw=" in l]
        assert len(fused_lines) == (13 if which == "mobilenet_v1" else 2)
        names = p.kernel_names()
        n_one_launch = sum("conv_depthwise_3x3_pointwise_1x1_fused" in n for n in names)
        n_two = sum("conv_depthwise_int8_hip+conv1x1s1" in n for n in names)
        assert n_one_launch + n_two == len(fused_lines), names
        assert n_one_launch == (13 if which == "mobilenet_v1" else 0), names
        gone = {l.split(" via=")[1].split(" ")[0] for l in fused_lines}
        n_i8 = 0
        for l in plan:
            if not (l.startswith("conv2d/") or l.startswith("depthwise_conv2d/")):
                continue
            name = l.split(" out=")[1].split(" ")[0]
            assert name not in gone
            if " -f32" in l:
                continue
            want = ref[name]
            got = p.get_var(name, want.dtype)
            if want.dtype == np.int8:
                assert np.array_equal(got, want), "%s: %d of %d int8 values differ" % (name, (got != want).sum(), want.size)
                n_i8 += 1
            else:
                np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5, err_msg=name)
        assert n_i8 >= 10
        np.testing.assert_allclose(p.get_var(out, np.float32), ref["prob"], rtol=1e-4, atol=1e-7)
    finally:
        p.close()


def test_fused_program_with_its_feed_resized_off_the_fused_kernels(lite, wl, plref):
    """The default program of MobileNetV1 is lowered for 224 x 224 (every pair one launch, the pool and the stem's calib taken over);
    then the feed is resized to 192 x 192, where NO fused kernel has the planes: every fused instruction must fall back inside its
    kernel object (ConvCompute::ReInitWhenNeeded: depthwise + 1x1 as two launches through a private tensor, + the global average
    pool kernel behind the last pair; the stem keeps its one-launch form, 192 % 4 == 0) and still produce the oracle's numbers;
    back at 224 the first result is reproduced."""
    net224, net192 = wl.mobilenet_v1_net(), wl.mobilenet_v1_net(res=192)
    rng = np.random.default_rng(360)
    img224 = rng.uniform(-1, 1, (2, 3, 224, 224)).astype(np.float32)
    img192 = rng.uniform(-1, 1, (2, 3, 192, 192)).astype(np.float32)
    ref224 = graph_oracle.forward(plref, net224, img224)
    ref192 = graph_oracle.forward(plref, net192, img192)
    p, out = _run_graph(lite, wl, net224, img224, fuse=True)
    try:
        assert sum("+pw=" in l for l in p.graph_plan()) == 13 and any("+pool=" in l for l in p.graph_plan())
        dev_out = out[:-len("/host")]
        first = p.get_var(dev_out, np.float32)
        np.testing.assert_allclose(first, ref224[dev_out], rtol=1e-5, atol=1e-6)
        p.add_feed(net224["input"], img192.shape, lite.PREC_FLOAT)
        p.set_input(net224["input"], img192)
        p.run()
        names = p.kernel_names()
        assert sum("conv_depthwise_int8_hip+conv1x1s1" in n for n in names) == 13, names   # every pair fell back
        assert any("+pooling_global_avg" in n and "conv_depthwise_int8_hip+" in n for n in names), names
        for v in ("conv1", "pw2", "pw7", "pw13", "pool", dev_out):
            want = ref192[v]
            got = p.get_var(v, want.dtype)
            assert got.shape == want.shape, v
            if want.dtype == np.int8:
                assert np.array_equal(got, want), v
            else:
                np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5, err_msg=v)
        p.add_feed(net224["input"], img224.shape, lite.PREC_FLOAT)
        p.set_input(net224["input"], img224)
        p.run()
        assert np.array_equal(p.get_var(dev_out, np.float32), first)
        assert sum("conv_depthwise_3x3_pointwise_1x1_fused" in n for n in p.kernel_names()) == 13
    finally:
        p.close()

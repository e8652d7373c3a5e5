"""Host logic of graph mode (lite/api/graph_builder.*): kernel pick and cast placement, checked on the CPU (no device)
against the hand-written Appendix-D program and against oracle/graph_oracle.py's independent restatement of the same
reference passes; plus the oracle's new glue ops against torch's CPU ops (independent cross-check)."""
import importlib

import numpy as np
import pytest

from oracle import graph_oracle


@pytest.fixture(scope="module")
def lite(pkg):
    return importlib.import_module("paddle_lite_amd.liteapi")


@pytest.fixture(scope="module")
def wl(pkg):
    return importlib.import_module("paddle_lite_amd.workloads")


def _plan(lite, wl, net, batch=2, fuse=False, fuse_dwpw=False):
    p = lite.Predictor(planner=True)
    try:
        wl.emit_graph(p, net, batch, fuse=fuse, fuse_dwpw=fuse_dwpw)
        return p.graph_plan()
    finally:
        p.close()


def test_default_dwpw_fusion_takes_the_pairs_the_fused_kernel_takes(lite, wl):
    """Default lowering (GraphBuilder fusion D, mode 2): the shapes are propagated from the feed and a depthwise conv takes
    its 1x1 consumer over only where plhip_dwpw_fused_supported says the fused kernel runs the pair as ONE launch:
    all thirteen pairs of MobileNetV1 at 224 x 224 (32 -> 64 @112, 128 -> 128 @56, 256 -> 256 @28 and the stride-2 pairs 64 -> 128
    @112 -> 56, 128 -> 256 @56 -> 28, 256 -> 512 @28 -> 14 on the streaming kernel, the five 512 -> 512 @14 on the resident-image
    kernel, the two that end on 7 x 7 on the small-plane kernel), at any batch; at 192 x 192 no plane has a kernel and the program
    keeps its 26 conv instructions; MobileNetV2 has no such pair."""
    net = wl.mobilenet_v1_net()
    for batch in (1, 128):
        off, dflt = _plan(lite, wl, net, batch=batch, fuse=True, fuse_dwpw=False), _plan(lite, wl, net, batch=batch, fuse=True, fuse_dwpw=None)
        fl = [l for l in dflt if "+pw=" in l]
        assert len(dflt) == len(off) - 15 and len(fl) == 13
        assert [l.split(" out=")[1].split(" ")[0] for l in fl] == ["pw%d" % i for i in range(2, 14)] + ["pool"]
        assert [l.split(" via=")[1].split(" ")[0] for l in fl] == ["dw%d" % i for i in range(2, 15)]
        # (E) the last pair also takes the global average pool behind it (PLHIP_OUT_F32_GAP): pw14's plane is never written
        assert fl[-1].endswith("+pw=conv2d/fp32_out via=dw14 +pool=avg/global pw_out=pw14") and not any(l.startswith("pool2d/") for l in dflt)
        assert sum("+pool=" in l for l in dflt) == 1
        # (F) the stem conv takes the calib[fp32_to_int8] in front of it (plhip_conv2d_calib_int8): the int8 image is never written
        assert dflt[1].startswith("conv2d/int8_out in=image/target_trans out=conv1 ") and "+calib_in=image/precision_trans in_scale=" in dflt[1]
        assert off[1].startswith("calib/fp32_to_int8 in=image/target_trans out=image/precision_trans") and len(dflt) == 19
    n192 = wl.mobilenet_v1_net(res=192)
    assert not any("+pw=" in l for l in _plan(lite, wl, n192, fuse=True, fuse_dwpw=None))
    forced = _plan(lite, wl, n192, fuse=True, fuse_dwpw=True)
    assert len([l for l in forced if "+pw=" in l]) == 13 and not any("+pool=" in l for l in forced)  # (the pool only where the kernel takes it)
    v2 = wl.mobilenet_v2_net(res=64)
    assert not any("+pw=" in l for l in _plan(lite, wl, v2, fuse=True, fuse_dwpw=None))
    # the reference program (no kHIP fusion) is untouched by the default
    assert not any("+pw=" in l for l in _plan(lite, wl, net, fuse=False, fuse_dwpw=None))


def test_opt_in_dwpw_fusion_plan(lite, wl):
    """GraphBuilder::set_fuse_dwpw(true): every depthwise_conv2d[int8_out] of MobileNetV1 has exactly one consumer, a plain 1x1
    conv: 13 instructions disappear, the fused line carries the depthwise output scale (= the pointwise input scale) and the
    pointwise kernel choice; MobileNetV2's project convs that carry a fused residual tail keep their own instruction."""
    net = wl.mobilenet_v1_net()
    base, fused = _plan(lite, wl, net, fuse=True), _plan(lite, wl, net, fuse=True, fuse_dwpw=True)
    assert len(fused) == len(base) - 13
    fl = [l for l in fused if "+pw=" in l]
    assert len(fl) == 13 and all(l.startswith("depthwise_conv2d/int8_out ") for l in fl)
    assert sum("+pw=conv2d/int8_out" in l for l in fl) == 12 and "+pw=conv2d/fp32_out via=dw14" in fl[-1]
    W = wl.make_mobilenet_v1_weights(1234)
    assert " out=pw2 " in fl[0] and "via=dw2" in fl[0] and "oscale=%.9g" % float(W["pw2"]["in_scale"]) in fl[0]
    assert "pw_oscale=%.9g" % float(W["dw3"]["in_scale"]) in fl[0]
    assert not any(l.startswith("conv2d/") and " in=dw" in l for l in fused)
    assert not any("+pw=" in l for l in base)  # base = fusion D switched off
    v2 = wl.mobilenet_v2_net(res=64)
    b2, f2 = _plan(lite, wl, v2, fuse=True), _plan(lite, wl, v2, fuse=True, fuse_dwpw=True)
    # a project conv that carries a fused tail (residual add and / or the calib copy for the next block) is not taken over
    plain = sum(1 for l in b2 if l.startswith("conv2d/") and l.split(" out=")[0].endswith("_dw") and "+add=" not in l and "+calib=" not in l)
    assert 1 <= plain < 17 and sum("+pw=" in l for l in f2) == plain and len(f2) == len(b2) - plain


def test_mobilenet_v1_graph_mode_arrives_at_appendix_d(lite, wl):
    plan = _plan(lite, wl, wl.mobilenet_v1_net())
    kinds = [l.split(" ")[0] for l in plan]
    expect = ["io_copy/host_to_device", "calib/fp32_to_int8", "conv2d/int8_out"]
    for i in range(13):
        expect += ["depthwise_conv2d/int8_out", "conv2d/int8_out" if i < 12 else "conv2d/fp32_out"]
    expect += ["pool2d/def", "calib/fp32_to_int8", "fc/fp32out", "softmax/def", "io_copy/device_to_host"]
    assert kinds == expect
    W = wl.make_mobilenet_v1_weights(1234)
    # output scale of an int8_out conv = the input scale of its consumer (static_kernel_pick_pass.cc:118-121)
    assert "oscale=%.9g" % float(W["dw2"]["in_scale"]) in plan[2]
    assert "scale=%.9g" % float(W["input_scale"]) in plan[1]
    assert "scale=%.9g" % float(W["pool_scale"]) in plan[-4]


@pytest.mark.parametrize("which", ["resnet50", "mobilenet_v2"])
def test_plan_matches_independent_restatement(lite, wl, which):
    net = wl.resnet50_net(res=64) if which == "resnet50" else wl.mobilenet_v2_net(res=64)
    plan = _plan(lite, wl, net)
    ref = graph_oracle.plan(net)
    body = plan[1:-1]  # io_copy at both ends
    assert plan[0].startswith("io_copy/host_to_device") and plan[-1].startswith("io_copy/device_to_host")
    assert len(body) == len(ref)
    n_fp32_convs = n_calib = 0
    for line, (kind, s) in zip(body, ref):
        if kind == "calib":
            n_calib += 1
            assert line.startswith("calib/fp32_to_int8"), line
            assert " out=%s " % s["dst"] in line and "scale=%.9g" % s["scale"] in line, (line, s)
        else:
            o = s["o"]
            assert " out=%s" % o["name"] in line, (line, o["name"])
            if o["op"] in graph_oracle.INT8_OPS:
                alias = ("int8out" if s["int8_out"] else "fp32out") if o["op"] == "fc" else ("int8_out" if s["int8_out"] else "fp32_out")
                assert line.startswith(o["op"] + "/" + alias), (line, alias)
                n_fp32_convs += not s["int8_out"]
                if s["int8_out"]:
                    assert "oscale=%.9g" % s["oscale"] in line
                ins = line.split(" in=")[1].split(" ")[0].split(",")
                assert [i.replace("/target_trans", "") for i in ins] == s["ins"]
    if which == "resnet50":
        # conv1 (-> max pool), 16 x branch2c and 4 x branch1 (-> add), fc; one calib per residual-stream tensor:
        # image, pool1, 16 block outputs (the last feeds pool5 only), pool5
        assert n_fp32_convs == 1 + 16 + 4 + 1 and n_calib == 1 + 1 + 15 + 1
    else:
        # projects that feed an add (10), projects whose output is the next block's expand input AND its add operand
        # (b2, b4, b7, b11, b14), conv_last (-> pool), fc
        assert n_fp32_convs == 10 + 5 + 1 + 1


def test_oracle_pool2d_against_torch(plref):
    torch = pytest.importorskip("torch")
    F = torch.nn.functional
    rng = np.random.default_rng(5)
    for (h, w, k, s, p) in [(112, 112, 3, 2, 1), (7, 9, 2, 2, 0), (15, 15, 3, 1, 1), (8, 8, 3, 2, 0), (14, 14, 2, 2, 1)]:
        x = rng.standard_normal((2, 3, h, w)).astype(np.float32)
        got = plref.pool2d(x, "max", (k, k), (s, s), (p, p, p, p))
        ref = F.max_pool2d(torch.from_numpy(x), k, s, p).numpy()
        assert np.array_equal(got, ref), (h, k, s, p)
        got = plref.pool2d(x, "avg", (k, k), (s, s), (p, p, p, p), exclusive=True)
        ref = F.avg_pool2d(torch.from_numpy(x), k, s, p, count_include_pad=False).numpy()
        np.testing.assert_allclose(got, ref, rtol=1e-6, atol=1e-6)
    # windows of the reference's graphs: ResNet50's 3x3 s2 p1 max pool halves 112 -> 56
    assert plref.pool2d(np.zeros((1, 1, 112, 112), np.float32), "max", (3, 3), (2, 2), (1, 1, 1, 1)).shape[2:] == (56, 56)


def test_oracle_elementwise_add(plref):
    rng = np.random.default_rng(6)
    x = rng.standard_normal((2, 5, 7, 3)).astype(np.float32)
    y = rng.standard_normal((2, 5, 7, 3)).astype(np.float32)
    assert np.array_equal(plref.elementwise_add(x, y), x + y)
    assert np.array_equal(plref.elementwise_add(x, y, True), np.maximum(x + y, 0))


def test_fc_routes_differ_by_at_most_one_ulp(plref):
    """gemm_s8 + fill_bias_fc rounds twice, gemv_int8's vmlaq once.  The extra rounding is the
    product's: the two results are at most half an ulp of the product plus one ulp of the result apart — and really
    different somewhere (otherwise the route switch would be untested)."""
    rng = np.random.default_rng(7)
    m, k, n = 9, 256, 513
    x = rng.integers(-127, 128, (m, k)).astype(np.int8)
    w = rng.integers(-127, 128, (k, n)).astype(np.int8)
    bias = rng.uniform(-1, 1, n).astype(np.float32)
    sc = np.full(n, 1.7 / 127 / 127, np.float32)
    y0, acc = plref.fc(x, w, bias, sc, False, False, route=0)
    y1, _ = plref.fc(x, w, bias, sc, False, False, route=1)
    prod = acc.astype(np.float32) * sc
    bound = 0.5 * np.spacing(np.abs(prod)) + np.spacing(np.maximum(np.abs(y0), np.abs(y1)))
    assert np.all(np.abs(y0.astype(np.float64) - y1) <= bound)
    assert np.any(y0 != y1)
    # route 1 is exactly "product rounded, then sum rounded"
    assert np.array_equal(y1, (acc.astype(np.float32) * sc) + bias)
    assert plref.fc_route(1, 1) == 0 and plref.fc_route(4, 1) == 1 and plref.fc_route(4, n) == 0


def test_khip_fusions_rewrite_the_residual_tails(lite, wl):
    """graph_builder.cc FuseSteps: conv[fp32_out] -> add (+relu) -> calib becomes one conv instruction (the LATER conv
    operand of the add takes it over), conv -> pool2d(max) -> calib becomes conv+calib -> int8 max pool; nothing else
    changes.  Same variables, fewer instructions."""
    net = wl.resnet50_net(res=64)
    ref, fused = _plan(lite, wl, net, fuse=False), _plan(lite, wl, net, fuse=True)
    assert len(ref) == 93 and len(fused) == 61
    assert not any(l.startswith(("elementwise_add", "fusion_elementwise_add_activation")) for l in fused)
    assert sum(l.startswith("calib") for l in fused) == 2  # the network input and pool5 only
    assert fused[2].startswith("conv2d/fp32_out in=image/precision_trans out=conv1 +calib=conv1/precision_trans") and fused[2].endswith("-f32")
    assert fused[3] == "pool2d/def in=conv1/precision_trans out=pool1/precision_trans int8"
    # res2a: branch1 is emitted after branch2c, so IT carries the add; identity blocks: branch2c does
    assert any(l.startswith("conv2d/fp32_out in=pool1/precision_trans out=res2a +add=res2a_branch2c +relu +calib=res2a/precision_trans") for l in fused)
    assert any(l.startswith("conv2d/fp32_out in=res2b_branch2b out=res2b +add=res2a +relu +calib=res2b/precision_trans") for l in fused)
    # the last block's sum feeds only the average pool: no calib, fp32 kept
    assert any(l == "conv2d/fp32_out in=res5c_branch2b out=res5c +add=res5b +relu" for l in fused)
    # every int8 tensor an int8 conv consumes in the reference program is still produced under the same name
    need = {i for l in ref if "/int8_out" in l or "/fp32_out" in l for i in l.split(" in=")[1].split(" ")[0].split(",")}
    made = set()
    for l in fused:
        made.add(l.split(" out=")[1].split(" ")[0])
        if "+calib=" in l:
            made.add(l.split("+calib=")[1].split(" ")[0])
    assert need <= made
    net = wl.mobilenet_v2_net(res=64)
    ref, fused = _plan(lite, wl, net, fuse=False), _plan(lite, wl, net, fuse=True)
    assert len(ref) == 84 and len(fused) == 59 and not any(l.startswith("elementwise_add") for l in fused)

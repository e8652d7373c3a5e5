"""BASELINE config C1 (MobileNetV1 fp32, 1x3x224x224, the reference's x86 CPU plumbing): oracle/x86_path.py restates
lite/kernels/x86/conv_compute.h:48-150 (im2col + SGEMM per image and group, no bias / activation in the kernel) and the
fp32 program around it.  CPU only.  Parity is UNPINNED beyond 1e-5 relative (the reference's GEMM is MKLML's cblas_sgemm);
what pins the restatement here: the reference's own known-answer test at that boundary (all-ones 3x3 -> 27,
lite/kernels/x86/conv_compute_test.cc:40-95) and torch's fp32 conv2d as an independent cross-check."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def x86():
    from oracle import plref, x86_path
    plref.build()
    return x86_path


def test_reference_known_answer_all_ones_3x3(x86):
    # conv_compute_test.cc:40-95: x [1,3,3,3] and filter [1,3,3,3] all ones, stride 1, pad 0 -> one output, 27, within 1e-5
    y = x86.conv2d_f32(np.ones((1, 3, 3, 3), np.float32), np.ones((1, 3, 3, 3), np.float32), 1, 0, 1)
    assert y.shape == (1, 1, 1, 1) and abs(float(y[0, 0, 0, 0]) - 27.0) <= 1e-5


def test_conv_f32_matches_torch_within_1e5(x86):
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(5)
    for (n, cin, h, w, cout, k, st, pad, g) in [(2, 3, 17, 19, 8, 3, 2, 1, 1), (1, 16, 14, 14, 24, 1, 1, 0, 1), (2, 8, 9, 9, 8, 3, 1, 1, 8),
                                               (1, 12, 11, 7, 6, 3, 2, 1, 3), (1, 32, 7, 7, 64, 1, 1, 0, 1), (1, 4, 10, 10, 4, 5, 1, 2, 2)]:
        x = rng.uniform(-1, 1, (n, cin, h, w)).astype(np.float32)
        wt = rng.uniform(-1, 1, (cout, cin // g, k, k)).astype(np.float32)
        got = x86.conv2d_f32(x, wt, st, pad, g)
        want = torch.nn.functional.conv2d(torch.from_numpy(x).double(), torch.from_numpy(wt).double(), None, st, pad, 1, g).numpy()
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5)


def test_c1_whole_program_plumbing(x86, pkg):
    """The C1 program end to end at a reduced resolution: a distribution comes out, twice the same, and it equals the same
    fp32 graph evaluated with torch ops within 1e-4 (softmax of fp32 logits)."""
    torch = pytest.importorskip("torch")
    import importlib
    wl = importlib.import_module("paddle_lite_amd.workloads")
    net = wl.mobilenet_v1_net(seed=1234, res=64)
    model = x86.fp32_model(net)
    img = np.ones((1, 3, 64, 64), np.float32)  # test_mobilenetv1_lite_x86.cc:36-45: all ones
    p1, p2 = x86.forward(model, img), x86.forward(model, img)
    assert p1.shape == (1, 1000) and np.array_equal(p1, p2) and abs(float(p1.sum()) - 1.0) < 1e-5
    t = torch.from_numpy(img).double()
    for o in model:
        if o["op"] == "conv":
            t = torch.nn.functional.conv2d(t, torch.from_numpy(o["w"]).double(), None, o["stride"], o["pad"], 1, o["groups"])
            t = t / np.sqrt(1.0 + x86.EPS)
            if o["bias"] is not None:
                t = t + torch.from_numpy(o["bias"]).double().reshape(1, -1, 1, 1)
            t = torch.relu(t) if o["relu"] else t
        elif o["op"] == "gap":
            t = t.mean(dim=(2, 3))
        elif o["op"] == "fc":
            t = t @ torch.from_numpy(o["w"]).double() + torch.from_numpy(o["bias"]).double()
        else:
            t = torch.softmax(t, dim=1)
    np.testing.assert_allclose(p1, t.numpy(), rtol=1e-4, atol=1e-7)


def test_c1_timing_record_shape(x86, pkg):
    import importlib
    wl = importlib.import_module("paddle_lite_amd.workloads")
    r = x86.time_c1(wl.mobilenet_v1_net(seed=1234, res=32), seconds=0.2, warmup=1)
    assert r["repeats"] >= 3 and r["avg_ms"] > 0 and r["min_ms"] <= r["avg_ms"] and abs(r["prob_sum"] - 1.0) < 1e-5

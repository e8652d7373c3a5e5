"""patches/*.patch (kHIP touch-points inside a real Paddle-Lite tree) must apply to the reference:
tools/check_patches.sh runs `git apply --check` + `git apply` on scratch copies of the touched files.  The reference
tree exists in the build container only; elsewhere the test is skipped."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_patches_apply_to_the_reference_tree():
    if not os.path.isdir("/root/reference/lite"):
        pytest.skip("reference tree not present on this machine")
    p = subprocess.run([os.path.join(ROOT, "tools", "check_patches.sh")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    out = p.stdout.decode()
    assert p.returncode == 0, out
    assert "all 6 patches apply" in out
    for name in os.listdir(os.path.join(ROOT, "patches")):
        assert name.endswith(".patch") and ("applies: " + name) in out


def test_patch_0006_carries_the_matcher_this_repository_runs():
    """patches/0006 adds the kHIP conv-tail fusions as a mir pass (lite/core/mir/fusion/hip_conv_tail_fuse_pass.{h,cc}, registered
    in optimizer.h's pass list in front of runtime_context_assign_pass).  Its pattern matcher is the header GraphBuilder::FuseSteps
    includes (paddle-lite_amd/lite/core/mir/fusion/hip_conv_tail_matcher.h), byte for byte: what tests/test_graph_lowering.py
    and the whole-graph GPU tests prove about the 61-instruction ResNet50 plan is proved about the pass's decisions."""
    patch = open(os.path.join(ROOT, "patches", "0006-mir-hip-conv-tail-fuse-pass.patch")).read()
    hdr = open(os.path.join(ROOT, "paddle-lite_amd", "lite", "core", "mir", "fusion", "hip_conv_tail_matcher.h")).read()
    start = patch.index("+++ b/lite/core/mir/fusion/hip_conv_tail_matcher.h")
    body = patch[start:].split("\n", 2)[2]            # skip the +++ line and the @@ hunk header
    end = body.find("\ndiff --git ")
    body = body if end < 0 else body[:end + 1]
    carried = "".join(l[1:] + "\n" for l in body.split("\n")[:-1] if l.startswith("+"))
    assert carried == hdr
    assert '"hip_conv_tail_fuse_pass",' in patch and "USE_MIR_PASS(hip_conv_tail_fuse_pass);" in patch
    assert "REGISTER_MIR_PASS(hip_conv_tail_fuse_pass" in patch and "fusion::MatchConvTails(&prog);" in patch
    src = open(os.path.join(ROOT, "paddle-lite_amd", "lite", "api", "graph_builder.cc")).read()
    assert "mir::fusion::MatchConvTails(&prog);" in src

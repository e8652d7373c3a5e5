"""Build-time guard for the A-in-registers GEMM kernels (csrc/gemm_i8.hip, load_a_regs): their fragment loads are issued
through inline asm that the compiler's wait-count pass cannot see, which is sound only if no instruction other than the
consuming MFMAs touches the destination registers while a load may be in flight.  tools/check_areg_isa.py compiles the
file to ISA (hipcc, ~1.5 min, no GPU needed) and checks every NG > 0 instantiation."""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc") and shutil.which("hipcc") is None, reason="needs hipcc")
def test_areg_kernels_have_no_copies_of_in_flight_fragments():
    # on the ISA the build kept (csrc/Makefile, -save-temps=obj) when it is there, else a fresh compile (~1.5 min)
    import subprocess
    import sys
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_areg_isa.py"), "--asm-dir", os.path.join(ROOT, "paddle-lite_amd", "csrc"),
                        "--quiet"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    out = p.stdout.decode()
    assert p.returncode == 0, out[-3000:]
    assert "A-in-register kernels checked, 0 problems" in out

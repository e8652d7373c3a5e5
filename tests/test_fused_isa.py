"""Build-time guard for the kernels that fetch MFMA operands with inline-asm loads and wait for them with hand-counted
s_waitcnt (csrc/fused_dwpw_i8.hip: weight fragments; csrc/gemm_tr_i8.hip: the LDS reads of the DMA ring): no instruction
other than the consuming MFMAs may touch a destination register between such a load and its MFMAs — a register copy or a
spill inserted by the compiler would move stale data.  tools/check_fused_isa.py compiles both files to ISA (hipcc, ~2 min,
no GPU needed) and checks every instantiation."""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc") and shutil.which("hipcc") is None, reason="needs hipcc")
def test_fused_and_tr_kernels_have_no_touch_of_in_flight_operands():
    # on the ISA the build kept (csrc/Makefile, -save-temps=obj) when it is there, else a fresh compile (~1.5 min)
    import subprocess
    import sys
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_fused_isa.py"), "--asm-dir", os.path.join(ROOT, "paddle-lite_amd", "csrc"),
                        "--quiet"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    out = p.stdout.decode()
    assert p.returncode == 0, out[-3000:]
    assert "transposed-read kernels checked, 0 problems" in out

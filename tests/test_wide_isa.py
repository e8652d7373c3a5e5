"""Build-time guard of the wide-tile GEMM (csrc/gemm_wide_kernel.h): its weight fragments (inline-asm global_load_dwordx4)
and activation fragments (inline-asm ds_read_b64_tr_b8) are waited for with hand-counted s_waitcnt; no instruction may
touch a destination register while its load can be in flight.  tools/check_wide_isa.py replays every instantiation's ISA
against the hardware's in-order completion model.  csrc/Makefile runs the same check on every build; this test runs it on
the ISA the build kept (or compiles it: ~1.5 min of hipcc, no GPU)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not available")
def test_no_instruction_touches_an_in_flight_operand_register():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_wide_isa.py"), "--asm-dir",
                        os.path.join(ROOT, "paddle-lite_amd", "csrc")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    out = p.stdout.decode()
    assert p.returncode == 0, out[-3000:]
    assert "kernels checked, 0 problems" in out

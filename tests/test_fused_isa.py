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
def test_fused_and_tr_kernels_never_touch_in_flight_operand_registers():
    spec = importlib.util.spec_from_file_location("check_fused_isa", os.path.join(ROOT, "tools", "check_fused_isa.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main() == 0

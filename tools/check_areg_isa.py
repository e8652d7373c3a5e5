#!/usr/bin/env python3
"""Static check of the A-in-registers GEMM kernels (gemm_i8_dma_kernel<..., NG > 0>).

Their A fragments are fetched with an inline-asm `global_load_dwordx4` that the compiler's wait-count pass cannot see
(see the comment at load_a_regs in csrc/gemm_i8.hip).  That is only sound if, between such a load and the MFMAs that
consume the fragment, NO other instruction reads or overwrites the destination registers (a register copy would move
stale data while the load is still in flight).  This script compiles csrc/gemm_i8.hip to ISA and verifies exactly that
for every NG > 0 instantiation.  The GPU parity suite is the dynamic guard; this is the build-time one.

Usage: python tools/check_areg_isa.py            (exit code 0 = all kernels clean)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "paddle-lite_amd", "csrc", "gemm_i8.hip")
REG = re.compile(r"v\[(\d+):(\d+)\]|v(\d+)")


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def check_kernel(name, lines):
    held = {}  # register -> [fragment id, uses]
    frag_regs = {}
    nfrag = 0
    errs = []
    nload = nmfma = 0
    in_asm = False  # only the inline-asm loads are hidden from the compiler; its own loads (e.g. the fused residual
                    # operand in the epilogue) are tracked by its wait-count pass and need no check
    for ln, raw in lines:
        if "#ASMSTART" in raw:
            in_asm = True
        elif "#ASMEND" in raw:
            in_asm = False
        ins = raw.split(";")[0].strip()
        if not ins or ins.startswith(".") or ins.endswith(":"):
            continue
        parts = ins.split(None, 1)
        op = parts[0]
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        if op == "global_load_dwordx4" and in_asm and len(ops) >= 2 and "off" in ins and "lds" not in op:
            dst = regs(ops[0])
            addr = regs(ops[1])
            for r in addr:
                if r in held and held[r][1] < 4:
                    errs.append("%s:%d address %s reads an unconsumed fragment register v%d" % (name, ln, ops[1], r))
            for r in dst:
                if r in held and held[r][1] < 4:
                    errs.append("%s:%d load overwrites unconsumed fragment register v%d" % (name, ln, r))
            nfrag += 1
            frag_regs[nfrag] = dst
            for r in dst:
                held[r] = [nfrag, 0]
            nload += 1
            continue
        if op.startswith("v_mfma"):
            nmfma += 1
            a = regs(ops[1])
            ids = {held[r][0] for r in a if r in held}
            if len(ids) == 1 and all(r in held for r in a):
                for r in a:
                    held[r][1] += 1
            elif ids:
                errs.append("%s:%d MFMA A operand %s mixes fragment registers" % (name, ln, ops[1]))
            # B operand / accumulators must not touch live fragments
            for o in [ops[0]] + ops[2:]:
                for r in regs(o):
                    if r in held and held[r][1] < 4:
                        errs.append("%s:%d MFMA operand %s touches unconsumed fragment register v%d" % (name, ln, o, r))
                    held.pop(r, None) if r in regs(ops[0]) else None
            continue
        # any other instruction: must not read or write a fragment that still has MFMAs to feed
        used = set()
        for o in ops:
            used |= regs(o)
        for r in used:
            if r in held:
                if held[r][1] < 4:
                    errs.append("%s:%d `%s` touches unconsumed fragment register v%d (fragment %d, %d of 4 uses)" %
                                (name, ln, ins, r, held[r][0], held[r][1]))
                else:
                    held.pop(r)
    return nload, nmfma, errs


def main():
    kept = None
    if "--asm-dir" in sys.argv:  # the ISA csrc/Makefile keeps from the real build (-save-temps=obj): no second compile
        kept = os.path.join(sys.argv[sys.argv.index("--asm-dir") + 1], "gemm_i8-hip-amdgcn-amd-amdhsa-gfx950.s")
    if kept and os.path.exists(kept):
        text = open(kept).read().splitlines()
    else:
        with tempfile.TemporaryDirectory() as td:
            asm = os.path.join(td, "gemm.s")
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-mllvm",
                                   "-amdgpu-mfma-vgpr-form=1", "-S", "--cuda-device-only", "-o", asm, SRC],
                                  stderr=subprocess.DEVNULL)
            text = open(asm).read().splitlines()
    kern = re.compile(r"^(_ZN5plhip18gemm_i8_dma_kernelILi(\d)ELi(\d)ELb([01])ELb([01])ELi4ELi(\d+)EEEvNS_8GemmArgsE):")
    cur, body, total_err, nk = None, [], [], 0
    for i, line in enumerate(text, 1):
        m = kern.match(line)
        if m:
            cur, body = (m.group(1), int(m.group(6))), []
            continue
        if cur is not None:
            body.append((i, line))
            if "s_endpgm" in line:
                if cur[1] > 0:
                    nload, nmfma, errs = check_kernel(cur[0], body)
                    nk += 1
                    if errs or "--quiet" not in sys.argv:
                        print("%-70s NG=%d  %3d fragment loads, %4d MFMAs: %s" % (cur[0][:70], cur[1], nload, nmfma,
                                                                                "clean" if not errs else "%d PROBLEMS" % len(errs)))
                    total_err += errs
                cur = None
    for e in total_err[:40]:
        print("  ", e)
    print("%d A-in-register kernels checked, %d problems" % (nk, len(total_err)))
    return 1 if total_err or nk == 0 else 0


if __name__ == "__main__":
    sys.exit(main())

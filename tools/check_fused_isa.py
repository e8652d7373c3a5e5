#!/usr/bin/env python3
"""Static check of the fused depthwise -> pointwise kernel (csrc/fused_dwpw_i8.hip) and the transposed-read GEMM
(csrc/gemm_tr_i8.hip): both fetch operands with inline-asm loads the compiler's wait-count pass cannot see and wait for
them with hand-counted s_waitcnt.  That is sound only if nothing but the consuming MFMAs (and the hand-placed waits)
touches a destination register of such a load while it may be in flight — a register copy or a spill inserted by the
compiler would move stale data.

fused_dwpw_i8.hip: the weight fragments (`global_load_dwordx4` inside #ASMSTART/#ASMEND): from such a load until the
register has fed its 4 MFMAs, no other instruction may read or write it (once dead, the compiler may reuse it).
gemm_tr_i8.hip: the LDS reads of the DMA ring (`ds_read_b64_tr_b8`: 2 MFMAs per register, `ds_read_b128`: 4).

Usage: python tools/check_fused_isa.py      (exit code 0 = all kernels clean; ~2 min of hipcc, no GPU needed)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "paddle-lite_amd", "csrc")
REG = re.compile(r"v\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def parse(lines):
    """-> [(line no, op, operands, in_asm)] of the instructions of one kernel"""
    out, in_asm = [], False
    for ln, raw in lines:
        if "#ASMSTART" in raw:
            in_asm = True
        elif "#ASMEND" in raw:
            in_asm = False
        ins = raw.split(";")[0].strip()
        if not ins or ins.startswith(".") or ins.endswith(":") or ins.startswith("#"):
            continue
        parts = ins.split(None, 1)
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        out.append((ln, parts[0], ops, in_asm))
    return out


def check_kernel(name, lines, load_ops):
    """load_ops: {asm load mnemonic: MFMA uses of one loaded register before it is dead}.  Linear scan in program text
    order (the K loops are unrolled so that a fragment's consumers follow its load in the text, or — across the loop's
    back edge — the text restarts with the same fragments in flight, which the prologue's loads stand in for): a register
    is HELD from its asm load until it has fed its MFMAs; while held, only MFMAs may touch it."""
    ins = parse(lines)
    mfmas = [i for i, (ln, op, ops, a) in enumerate(ins) if op.startswith("v_mfma")]
    nload = sum(1 for (ln, op, ops, a) in ins if a and op in load_ops)
    if not nload or not mfmas:
        return 0, 0, ["%s: no asm operand loads / no MFMAs found" % name]
    held = {}  # register -> [uses so far, uses expected]
    errs = []
    for i in range(0, mfmas[-1] + 1):
        ln, op, ops, a = ins[i]
        if a and op in load_ops:
            for o in ops[1:]:
                for r in regs(o):
                    if r in held:
                        errs.append("%s:%d address of `%s` reads in-flight operand register v%d" % (name, ln, op, r))
            for r in regs(ops[0]):
                # a held register loaded again: the source's own doing (the two arms of an if / else follow each other in
                # the text), not a compiler-inserted hazard: the hold simply restarts
                held[r] = [0, load_ops[op]]
            continue
        if op.startswith("v_mfma"):
            for o in (ops[1], ops[2]):
                for r in regs(o):
                    if r in held:
                        held[r][0] += 1
                        if held[r][0] >= held[r][1]:
                            del held[r]
            for o in [ops[0]] + ops[3:]:
                for r in regs(o):
                    if r in held:
                        errs.append("%s:%d MFMA accumulator %s touches in-flight operand register v%d" % (name, ln, o, r))
            continue
        touched = set()
        for o in ops:
            touched |= regs(o)
        for r in sorted(touched & set(held)):
            errs.append("%s:%d `%s %s` touches in-flight operand register v%d (%d of %d MFMA uses)" %
                        (name, ln, op, ", ".join(ops), r, held[r][0], held[r][1]))
    return nload, len(mfmas), errs


def kept_asm(src):
    """--asm-dir DIR: the ISA csrc/Makefile keeps from the real build (-save-temps=obj), so that the check costs no compile"""
    if "--asm-dir" in sys.argv:
        p = os.path.join(sys.argv[sys.argv.index("--asm-dir") + 1], src[:-4] + "-hip-amdgcn-amd-amdhsa-gfx950.s")
        if os.path.exists(p):
            return p
    return None


def run(src, kernel_re, load_ops):
    kept = kept_asm(src)
    if kept:
        text = open(kept).read().splitlines()
    else:
        with tempfile.TemporaryDirectory() as td:
            asm = os.path.join(td, "k.s")
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-mllvm",
                                   "-amdgpu-mfma-vgpr-form=1", "-S", "--cuda-device-only", "-o", asm, os.path.join(CSRC, src)],
                                  stderr=subprocess.DEVNULL)
            text = open(asm).read().splitlines()
    kern = re.compile(kernel_re)
    cur, body, total, nk = None, [], [], 0
    for i, line in enumerate(text, 1):
        m = kern.match(line)
        if m:
            cur, body = m.group(1), []
            continue
        if cur is not None:
            body.append((i, line))
            if "s_endpgm" in line:
                nl, nm, errs = check_kernel(cur, body, load_ops)
                nk += 1
                if errs or "--quiet" not in sys.argv:
                    print("%-72s %3d asm operand loads, %4d MFMAs: %s" % (cur[:72], nl, nm, "clean" if not errs else "%d PROBLEMS" % len(errs)))
                total += errs
                cur = None
    return nk, total


def main():
    nk1, e1 = run("fused_dwpw_i8.hip", r"^(_ZN5plhip17fused_dwpw_kernelI\w+EvNS_9FusedArgsE):", {"global_load_dwordx4": 4})
    nk2, e2 = run("gemm_tr_i8.hip", r"^(_ZN5plhip17gemm_i8_tr_kernelI\w+EvNS_8GemmArgsE):", {"ds_read_b64_tr_b8": 2, "ds_read_b128": 4})
    for e in (e1 + e2)[:40]:
        print("  ", e)
    print("%d fused + %d transposed-read kernels checked, %d problems" % (nk1, nk2, len(e1) + len(e2)))
    return 1 if e1 or e2 or nk2 == 0 else 0  # (no fused kernels in the default build: make EXPERIMENTS=1)


if __name__ == "__main__":
    sys.exit(main())

#!/usr/bin/env python3
"""Static check of the wide-tile GEMM kernels (csrc/gemm_wide_kernel.h, instantiated by gemm_wide_n4/n7/n8.hip) and of the
patch convolution kernels (csrc/conv_patch_kernel.h, instantiated by conv_patch_i8 / _stat_b / _stream.hip).

They fetch the weight fragments with inline-asm `global_load_dwordx4` and the activation fragments with inline-asm
`ds_read_b64_tr_b8`, neither of which the compiler's wait-count pass sees, and wait for them with hand-counted
`s_waitcnt`.  That is sound only if no instruction reads or writes a destination register of such a load while the load
may still be in flight (a register copy, a spill or a re-use as a temporary would move or lose data).  This script walks
the ISA of every instantiation with the hardware's own model: vector-memory operations complete in issue order (vmcnt),
LDS operations complete in issue order (lgkmcnt; scalar loads are not in these kernels' loops); an `s_waitcnt vmcnt(N)` /
`lgkmcnt(N)` retires all but the N youngest.  Every instruction in between is checked against the destination
registers of the inline-asm loads still outstanding.

Usage: python tools/check_wide_isa.py [file.s ...]   (no argument: compiles the three translation units itself, ~1 min;
       exit code 0 = all kernels clean).  csrc/Makefile runs it on the ISA it keeps from the real build."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "paddle-lite_amd", "csrc")
REG = re.compile(r"v\[(\d+):(\d+)\]|\bv(\d+)\b")
VM_OPS = ("global_load", "global_store", "global_atomic", "buffer_load", "buffer_store", "flat_load", "flat_store", "scratch_")
KERN = re.compile(r"^(_ZN5plhip19gemm_i8_wide_kernelI\w+EvNS_8GemmArgsE|_ZN5plhip20conv_patch_i8_kernelI\w+EvNS_9PatchArgsE):")
UNITS = ("gemm_wide_n4", "gemm_wide_n7", "gemm_wide_n8", "conv_patch_i8", "conv_patch_stat_b", "conv_patch_stream", "conv_patch_s2")


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def check_kernel(name, lines):
    vm, lgkm = [], []  # outstanding operations in issue order: (line, set of destination registers of an ASM load, or empty)
    errs, n_w, n_f, in_asm = [], 0, 0, False
    # `if (first round) s_waitcnt vmcnt(A) else s_waitcnt vmcnt(B)` (conv_patch_kernel.h) reads, in this linear walk, as two
    # waits in a row with only branches / labels between them: they are alternatives, the walk applies the WEAKER one
    pending_vm = None  # (count, [ops issued since]) of an asm vmcnt wait not applied yet
    for ln, raw in lines:
        if "#ASMSTART" in raw:
            in_asm = True
            continue
        if "#ASMEND" in raw:
            in_asm = False
            continue
        ins = raw.split(";")[0].strip()
        if not ins or ins.startswith(".") or ins.endswith(":"):
            continue
        parts = ins.split(None, 1)
        op = parts[0]
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        if pending_vm is not None and not (op == "s_waitcnt" and in_asm) and not op.startswith(("s_cbranch", "s_branch")):
            del vm[:max(0, len(vm) - pending_vm)]
            pending_vm = None
        if op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", ins)
            if m and in_asm and "lgkmcnt" not in ins:
                pending_vm = int(m.group(1)) if pending_vm is None else max(pending_vm, int(m.group(1)))
            elif m:
                del vm[:max(0, len(vm) - int(m.group(1)))]
            m = re.search(r"lgkmcnt\((\d+)\)", ins)
            if m:
                del lgkm[:max(0, len(lgkm) - int(m.group(1)))]
            continue
        if op in ("s_endpgm",):
            break
        touched = set()
        for o in ops:
            touched |= regs(o)
        live_vm = set().union(*[d for _, d in vm]) if vm else set()
        live_ds = set().union(*[d for _, d in lgkm]) if lgkm else set()
        hit = touched & (live_vm | live_ds)
        # an in-flight load's destination may be NAMED by nothing at all - except that a later asm load may not reuse it either
        if hit:
            errs.append("%s:%d `%s` touches v%s while an inline-asm load into it may be in flight" % (name[:60], ln, ins[:70], sorted(hit)[:4]))
        if op.startswith(VM_OPS):
            dst = regs(ops[0]) if (in_asm and op.startswith(("global_load_dwordx4", "global_load_dword")) and "lds" not in op) else set()
            n_w += bool(dst)
            vm.append((ln, dst))
        elif op.startswith("ds_") or op.startswith("s_load") or op.startswith("s_buffer_load"):
            dst = regs(ops[0]) if (in_asm and op in ("ds_read_b64_tr_b8", "ds_read_b128")) else set()
            n_f += bool(dst)
            lgkm.append((ln, dst))
    return n_w, n_f, errs


def main():
    files = [a for a in sys.argv[1:] if a.endswith(".s")]
    if "--asm-dir" in sys.argv:
        d = sys.argv[sys.argv.index("--asm-dir") + 1]
        files = [os.path.join(d, n + "-hip-amdgcn-amd-amdhsa-gfx950.s") for n in UNITS]
        files = [f for f in files if os.path.exists(f)]
    tmp = None
    if not files:
        tmp = tempfile.TemporaryDirectory()
        for n in UNITS:
            out = os.path.join(tmp.name, n + ".s")
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-mllvm",
                                   "-amdgpu-mfma-vgpr-form=1", "-S", "--cuda-device-only", "-I", CSRC, "-o", out,
                                   os.path.join(CSRC, n + ".hip")], stderr=subprocess.DEVNULL)
            files.append(out)
    nk, total = 0, []
    for f in files:
        cur, body = None, []
        for i, line in enumerate(open(f).read().splitlines(), 1):
            m = KERN.match(line)
            if m:
                cur, body = m.group(1), []
                continue
            if cur is not None:
                body.append((i, line))
                if "s_endpgm" in line:
                    n_w, n_f, errs = check_kernel(cur, body)
                    nk += 1
                    if errs or "-v" in os.environ.get("CHECK_WIDE_VERBOSE", ""):
                        print("%-64s %3d weight loads, %4d fragment reads: %s" % (cur[:64], n_w, n_f, "clean" if not errs else "%d PROBLEMS" % len(errs)))
                    total += errs
                    cur = None
    for e in total[:40]:
        print("  ", e)
    print("%d wide-tile / patch kernels checked, %d problems" % (nk, len(total)))
    return 1 if total or nk == 0 else 0


if __name__ == "__main__":
    sys.exit(main())

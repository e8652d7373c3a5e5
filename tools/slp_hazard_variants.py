#!/usr/bin/env python3
"""Designed experiment for the nondeterministic fp32-epilogue result of round 3: build gemm_i8.hip WITH the SLP
vectoriser (the failing build), then re-assemble its device ISA with ONE edit per variant and link one library per variant.
Running tools/dbg_flaky.py once against each library says which edit removes the failure, i.e. which mechanism it is:

  v0  the SLP build as hipcc emits it (the failing reference)
  v1  16 wait states behind every global_store_dwordx4      -> a late store-DATA read racing the next row's v_cvt writes
  v2  8 wait states in front of every v_pk_fma_f32 with op_sel:[0,1,1] -> a v_cvt_f32_i32 -> v_pk_fma_f32 forwarding distance
  v3  every v_pk_fma_f32 ... op_sel:[0,1,1] replaced by two v_fma_f32 on the same registers -> the packed op_sel form itself

CPU only (hipcc cross-compiles); output: build/slp/libplhip_slp_v{0,1,2,3}.so; run: PLHIP_LIB_PATH=build/slp/libplhip_slp_vN.so python tools/dbg_flaky.py 64 64 56 256 f32 8.  Usage: python tools/slp_hazard_variants.py
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "paddle-lite_amd", "csrc")
OUT = os.path.join(ROOT, "build", "slp")  # build/ is git-ignored but travels to the GPU box (gpurun_out/ does not)
LLVM = "/opt/rocm/lib/llvm/bin"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-mllvm", "-amdgpu-mfma-vgpr-form=1"]
DEV_S = "gemm_i8-hip-amdgcn-amd-amdhsa-gfx950.s"
HOST_S = "gemm_i8-host-x86_64-unknown-linux-gnu.s"
PK = re.compile(r"^\s*v_pk_fma_f32 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\] op_sel:\[0,1,1\]\s*$")


def run(cmd, **kw):
    subprocess.check_call(cmd, **kw)


ONLY = "gemm_i8_lds_kernelILi1ELi1E"  # the failing route's kernel (32-row wave tiles, fp32 output); elsewhere the extra bytes push branches past simm16


def edit(lines, variant):
    out, n, inside = [], 0, False
    for ln in lines:
        if ln.startswith("_ZN") and ln.rstrip().split(":")[0].startswith("_ZN5plhip"):
            inside = ONLY in ln
        m = PK.match(ln) if inside else None
        if not inside:
            out.append(ln)
            continue
        if variant == 1 and "global_store_dwordx4" in ln:
            out += [ln, "\ts_nop 7\n", "\ts_nop 7\n"]
            n += 1
        elif variant == 2 and m:
            out += ["\ts_nop 7\n", ln]
            n += 1
        elif variant == 3 and m:
            a, b, c, d, e, f, g, h = (int(v) for v in m.groups())
            # lo = c * f + h, hi = d * f + h (op_sel picks the HIGH register of src1 / src2 for both halves)
            if a in (d, f, h):  # the low result would clobber a source of the high one: high first
                assert b not in (c, f, h), ln
                out += ["\tv_fma_f32 v%d, v%d, v%d, v%d\n" % (b, d, f, h), "\tv_fma_f32 v%d, v%d, v%d, v%d\n" % (a, c, f, h)]
            else:
                out += ["\tv_fma_f32 v%d, v%d, v%d, v%d\n" % (a, c, f, h), "\tv_fma_f32 v%d, v%d, v%d, v%d\n" % (b, d, f, h)]
            n += 1
        else:
            out.append(ln)
    return out, n


def main():
    os.makedirs(OUT, exist_ok=True)
    work = os.path.join(ROOT, "gpurun_out", "slp", "work")  # ~0.5 GB of assembly: stays here
    os.makedirs(work, exist_ok=True)
    if not os.path.exists(os.path.join(work, DEV_S)):
        run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-save-temps=obj", "-c", os.path.join(CSRC, "gemm_i8.hip"), "-o", os.path.join(work, "gemm_i8_slp.o")], cwd=work)
    dev = open(os.path.join(work, DEV_S)).readlines()
    host = open(os.path.join(work, HOST_S)).readlines()
    # the host assembly embeds the offload bundle as one .asciz in section .hip_fatbin: swap it for an .incbin of the variant's
    fat = [i for i, ln in enumerate(host) if ln.startswith("\t.asciz\t\"__CLANG_OFFLOAD_BUNDLE__")]
    assert len(fat) == 1 and host[fat[0] + 1].startswith("\t.size"), "unexpected host assembly layout"
    label = host[fat[0] + 1].split()[1].rstrip(",")
    others = [os.path.join(CSRC, o) for o in sorted(os.listdir(CSRC)) if o.endswith(".o") and o != "gemm_i8.o"]
    assert others, "build the library first (make -C paddle-lite_amd/csrc)"
    for v in range(4):
        d, n = edit(dev, v)
        vs = os.path.join(work, "dev_v%d.s" % v)
        open(vs, "w").writelines(d)
        vo, vout, vfb = vs[:-2] + ".o", vs[:-2] + ".out", vs[:-2] + ".hipfb"
        run([LLVM + "/clang", "-cc1as", "-triple", "amdgcn-amd-amdhsa", "-filetype", "obj", "-target-cpu", "gfx950", "-mrelocation-model", "pic", "-o", vo, vs])
        run([LLVM + "/lld", "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", vout, vo])
        run([LLVM + "/clang-offload-bundler", "-type=o", "-bundle-align=4096", "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950",
             "-input=/dev/null", "-input=" + vout, "-output=" + vfb])
        h = list(host)
        h[fat[0]] = "\t.incbin\t\"%s\"\n" % vfb
        h[fat[0] + 1] = "\t.size\t%s, %d\n" % (label, os.path.getsize(vfb))
        hs = os.path.join(work, "host_v%d.s" % v)
        open(hs, "w").writelines(h)
        ho = os.path.join(work, "gemm_i8_v%d.o" % v)
        run([LLVM + "/clang", "-c", hs, "-o", ho])
        lib = os.path.join(OUT, "libplhip_slp_v%d.so" % v)
        run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-o", lib] + others + [ho])
        print("v%d: %d edits -> %s" % (v, n, lib))


if __name__ == "__main__":
    sys.exit(main())

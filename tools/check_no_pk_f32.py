#!/usr/bin/env python3
"""Build-time guard: no packed-fp32 VALU instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) with an `op_sel:` modifier
in any device object (a LOW result taking a source from the HIGH register of its pair), and none at all in a kernel that
also issues MFMAs.

Why (round 4; tools/slp_hazard_variants.py is the experiment): on MI355X `v_pk_fma_f32 ... op_sel:[0,1,1]` --
the form hipcc's SLP vectoriser emits for the second row of an fp32 epilogue, whose LOW result takes src1 / src2 from the HIGH
registers of their pairs -- intermittently returned src2 alone (product read as zero) in lanes 32-63: conv2d[fp32_out] 64 -> 256
at 56x56 differed on every launch.  Re-assembling the same kernel with that one instruction replaced by two v_fma_f32 on the
same registers (no other change, same waits, same schedule) is exact; 16 wait states behind every 128-bit store or 8 in front
of the packed instruction change nothing.  The library is therefore built with -fno-slp-vectorize, and this check makes the
build fail if that form comes back (a compiler bump, a dropped flag, a hand-written v_pk_*).  The plain form (no op_sel: each
half from its own half; what the backend selects for explicit float2 / float4 arithmetic in the streaming glue kernels) never
failed -- the e = 0 rows of the same epilogue used op_sel_hi only and were always exact -- and stays allowed outside MFMA
kernels; beside MFMAs any packed form costs more issue time than two scalar ones and is refused too.

Usage: python tools/check_no_pk_f32.py --asm-dir paddle-lite_amd/csrc     (reads the *-gfx950.s the build keeps)
"""
import argparse
import glob
import os
import re
import sys

BAD = re.compile(r"^\s*(v_pk_fma_f32|v_pk_mul_f32|v_pk_add_f32)\b")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--asm-dir", required=True)
    ap.add_argument("--expect", type=int, default=0, help="minimum number of ISA files that must be present")
    ap.add_argument("--quiet", action="store_true")
    args = ap.parse_args()
    files = sorted(glob.glob(os.path.join(args.asm_dir, "*-hip-amdgcn-amd-amdhsa-gfx950.s")))
    if len(files) < max(1, args.expect):
        print("check_no_pk_f32: %d ISA files under %s, expected >= %d (build with -save-temps=obj)" % (len(files), args.asm_dir, max(1, args.expect)))
        return 2
    bad = plain = 0
    for f in files:
        kernel, pk, has_mfma = "?", [], False

        def close():
            nonlocal bad, plain
            for (n, txt) in pk:
                if "op_sel:" in txt or has_mfma:
                    bad += 1
                    if bad <= 10:
                        print("%s:%d: %s in %s%s" % (os.path.basename(f), n, txt, kernel[:90], " (MFMA kernel)" if has_mfma else ""))
                else:
                    plain += 1

        for n, ln in enumerate(open(f, errors="replace"), 1):
            if ln.startswith("_Z") and ":" in ln:
                close()
                kernel, pk, has_mfma = ln.split(":")[0], [], False
            elif BAD.match(ln):
                pk.append((n, ln.strip()))
            elif "v_mfma_" in ln:
                has_mfma = True
        close()
    if not args.quiet or bad:
        print("check_no_pk_f32: %d files checked, %d refused packed-fp32 instructions (op_sel form, or beside MFMAs), %d plain ones in glue kernels" % (len(files), bad, plain))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

#!/usr/bin/env python3
"""Writes a scratch `lite/operators/op_params.h` made of the REFERENCE's own struct text (read from --reference, written
under --out, a directory outside this repository) for the parameter structs the kHIP kernel classes receive.  Compiling
paddle-lite_amd/lite/kernels/hip/*.cc with that directory first on the include path proves that the kernel sources use
only fields the reference's structs have: the plugin boundary is a drop-in, not a fork of the schema
(tests/test_boundary_reference_params.py).  Nothing of the reference is copied into the repository."""
import argparse
import os
import re

STRUCTS = ["ParamBase", "IoCopyParam", "CalibParam", "FcParam", "SoftmaxParam", "ActivationParam", "ConvParam", "PoolParam",
           "ElementwiseParam", "FusionElementwiseActivationParam"]


def struct_text(src, name):
    m = re.search(r"^struct %s\b[^{;]*\{" % re.escape(name), src, re.M)
    assert m, "struct %s not found in the reference header" % name
    i, depth = m.end(), 1
    while depth:
        c = src[i]
        depth += (c == "{") - (c == "}")
        i += 1
    j = src.index(";", i)
    return src[m.start():j + 1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--out", required=True, help="include root to create lite/operators/op_params.h under")
    a = ap.parse_args()
    src = open(os.path.join(a.reference, "lite", "operators", "op_params.h")).read()
    m = re.search(r"^#define WITH_INT8_CONFIG(?:.*\\\n)*.*\n", src, re.M)
    assert m, "WITH_INT8_CONFIG not found"
    parts = ["// GENERATED from the reference's lite/operators/op_params.h (struct text verbatim); scratch file, never committed\n"
             "#pragma once\n#include <memory>\n#include <string>\n#include <vector>\n\n#include \"lite/api/paddle_place.h\"\n"
             "#include \"lite/core/tensor.h\"\n\nnamespace paddle {\nnamespace lite {\nnamespace operators {\n\n"]
    parts.append(struct_text(src, "ParamBase") + "\n\n" + m.group(0) + "\n")
    for s in STRUCTS[1:]:
        parts.append(struct_text(src, s) + "\n\n")
    parts.append("}  // namespace operators\n}  // namespace lite\n}  // namespace paddle\n")
    d = os.path.join(a.out, "lite", "operators")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "op_params.h"), "w") as f:
        f.write("".join(parts))
    print(os.path.join(d, "op_params.h"))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py` into per-kernel-family HBM traffic per launch.

Usage: python tools/pmc_traffic.py <fetch_dir> <write_dir> [out.json]
Counter units (MI355X_MICROARCH.md §HBM): FETCH_SIZE / WRITE_SIZE are in KiB of L2<->fabric traffic; on gfx950 FETCH_SIZE
reports exactly 1/2 of the bytes of wide (16 B/lane) coalesced streaming reads; other access widths are uncalibrated, so the
raw value, the x2-corrected value and the calibration ratio measured on `calib_f32_to_i8_kernel` (a pure 16 B/lane stream
whose byte count is known: 4 B read per element) are all reported.
"""
import collections
import csv
import glob
import json
import sys


def load(d, counter):
    f = glob.glob(d + "/*/*_counter_collection.csv") + glob.glob(d + "/*_counter_collection.csv")
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == counter:
            per[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return per


def family(name):
    if "fused_dwpw" in name:
        return "dwpw_fused"
    if "depthwise" in name:
        return "depthwise3x3"
    if "conv3x3s2" in name or "conv7x7s2_stem" in name:
        return "stem_conv"
    if "conv_patch_i8_kernel" in name:
        return "conv3x3_patch"  # dense 3x3 stride-1 convs with whole 32-channel chunks (conv_patch_i8.hip); their padded copy below
    if "pad_rows8" in name or "pad_phase8" in name:
        return "conv3x3_patch_pad"
    if "gemm_i8_tr_kernel" in name:
        return "conv_implicit_gemm"  # dense k x k convs (the implicit-GEMM route); 1x1 convs never use this kernel by default
    if "gemm_i8" in name:
        return "pointwise1x1"
    if "calib_f32_to_i8" in name:
        return "calib_f32_to_i8"
    return None


def main():
    fd, wd = sys.argv[1], sys.argv[2]
    out = sys.argv[3] if len(sys.argv) > 3 else "profiles/pmc_traffic.json"
    fetch, write = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
    fam = collections.defaultdict(lambda: {"fetch_kib": 0.0, "write_kib": 0.0, "launches": 0})
    for k, v in fetch.items():
        f = family(k)
        if f:
            fam[f]["fetch_kib"] += sum(v)
            fam[f]["launches"] += len(v)
    for k, v in write.items():
        f = family(k)
        if f:
            fam[f]["write_kib"] += sum(v)
    res = {}
    for f, d in fam.items():
        n = max(1, d["launches"])
        res[f] = {"launches_profiled": d["launches"],
                  "fetch_bytes_per_launch_raw": d["fetch_kib"] * 1024 / n,
                  "fetch_bytes_per_launch_x2": 2 * d["fetch_kib"] * 1024 / n,
                  "write_bytes_per_launch": d["write_kib"] * 1024 / n}
    try:
        import os
        import subprocess
        res["commit"] = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], cwd=os.path.dirname(os.path.abspath(__file__)),
                                                stderr=subprocess.DEVNULL).decode().strip()
    except Exception:  # noqa: BLE001 - no git on the GPU box
        res["commit"] = "unknown"
    try:  # the kernel sources this pass measured: bench.py refuses the file when they have changed since
        import importlib.util
        import os
        spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
        b_ = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(b_)
        res["csrc_sha256"] = b_.csrc_sha256()
    except Exception as e:  # noqa: BLE001
        res["csrc_sha256"] = "unknown: %s" % e
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()

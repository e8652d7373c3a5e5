#!/usr/bin/env python3
"""Copy the evidence bundle of tools/collect_evidence.sh (gpurun_out/evidence/) into profiles/ under the round's names,
aggregate the PMC passes per kernel, regenerate profiles/pmc_traffic.json and print the numbers DESIGN.md quotes.
Usage: python tools/update_profiles.py [round_tag=r01]"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
E = os.path.join(ROOT, "gpurun_out", "evidence")
P = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"


def cp(src, dst):
    shutil.copyfile(os.path.join(E, src), os.path.join(P, "%s_final_%s" % (tag, dst)))


cp("stats_serial/p_kernel_stats.csv", "kernel_stats_serial.csv")
cp("stats_default/p_kernel_stats.csv", "kernel_stats_default.csv")
cp("opbench.txt", "opbench.txt")
cp("layer_table.txt", "layer_table.txt")
cp("bench.json", "bench.json")
cp("bench_inflight1.json", "bench_inflight1.json")
for f in ("bench_c4.json", "bench_c5.json", "bench_c2.json", "bench_force_dist.json", "layer_table_c4.txt", "layer_table_c5.txt",
          "opbench_b256.txt", "opbench_pw_wide_off.txt", "wide_timeline_pw8.txt", "wide_timeline_pw6.txt", "wide_timeline_pw13.txt",
          "wide_timeline_pw14.txt", "gemm_timeline_pw8_ring.txt", "c2bench.txt", "pmc_sq_c3.csv", "pmc_sq_c4.csv",
          "bench_driver_form.json", "opbench_resnet50_3x3.txt", "opbench_resnet50_3x3_patch_off.txt", "opbench_resnet50_3x3_no_epilogue.txt",
          "opbench_dw5x5.txt", "opbench_dw5x5_lds_band.txt", "opbench_resnet50_3x3_s2_patch_off.txt", "patch_timeline_res3a_s2.txt",
          "patch_timeline_res5a_s2.txt", "patch_timeline_c2.txt", "patch_timeline_res2.txt", "patch_timeline_res4.txt",
          "opbench_fused.txt", "fused_timeline_b128.txt", "fused_timeline_b256.txt", "fused_timeline_exp1.txt", "fused_timeline_exp2.txt",
          "fused_timeline_exp3.txt", "fused_timeline_exp19.txt", "fused_timeline_exp4.txt", "probe_coexec.txt", "probe_cvt_rtz.txt",
          "bench_dwpw_off.json", "bench_dwpw_off_inflight1.json", "bench_stream_off.json", "bench_stream_off_inflight1.json",
          "stream_timeline_56.txt", "stream_timeline_112.txt", "stream_timeline_28.txt", "stream_timeline_112_s2.txt",
          "stream_timeline_56_s2.txt", "stream_timeline_28_s2.txt", "small_timeline_14_s2.txt", "small_timeline_7_f32.txt",
          "small_timeline_7_f32_two_blocks.txt", "fused_vs_two_kernels.txt", "opbench_fused_small_two_blocks.txt", "bench_stride1_only.json",
          "bench_small_off.json", "bench_small_two_blocks.json", "bench_no_pool_tail.json"):
    if os.path.exists(os.path.join(E, f)):
        cp(f, f)
for kind, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    f = glob.glob(os.path.join(E, kind, "*_counter_collection.csv"))[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            a = agg[r["Kernel_Name"][:90]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    with open(os.path.join(P, "%s_final_pmc_%s_size.csv" % (tag, kind)), "w") as o:
        o.write("kernel,dispatches,%s_sum_KiB,%s_per_dispatch_KiB\n" % (counter, counter))
        for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            o.write('"%s",%d,%.1f,%.1f\n' % (k, n, v, v / n))
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"), os.path.join(E, "fetch"),
                       os.path.join(E, "write"), os.path.join(P, "pmc_traffic.json")], stdout=subprocess.DEVNULL)
if os.path.isdir(os.path.join(E, "stats_s2")):
    cp("stats_s2/p_kernel_stats.csv", "kernel_stats_s2_layers.csv")
if os.path.isdir(os.path.join(E, "stats_c2")):
    cp("stats_c2/p_kernel_stats.csv", "kernel_stats_c2bench.csv")
if os.path.isdir(os.path.join(E, "fetch_c2")):
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"), os.path.join(E, "fetch_c2"),
                           os.path.join(E, "write_c2"), os.path.join(P, "pmc_traffic_c2.json")], stdout=subprocess.DEVNULL)
if os.path.isdir(os.path.join(E, "fetch_c4")):
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"), os.path.join(E, "fetch_c4"),
                           os.path.join(E, "write_c4"), os.path.join(P, "pmc_traffic_c4.json")], stdout=subprocess.DEVNULL)


def g(f):
    return json.loads(open(os.path.join(E, f)).read().strip().splitlines()[-1])


for f in ("bench.json", "bench_driver_form.json", "bench_inflight1.json", "bench_c4.json", "bench_c5.json", "bench_c2.json", "bench_force_dist.json"):
    d = g(f)
    print("%-24s %9.1f img/s  %.4f ms/step  roofline.frac %.4f" % (f, d["value"], d["ms_per_step"], d["roofline"]["frac"]))
print(open(os.path.join(P, "%s_final_opbench.txt" % tag)).read().splitlines()[-1])
_t = json.load(open(os.path.join(P, "pmc_traffic.json")))
print(json.dumps({k: _t[k] for k in ("dwpw_fused", "stem_conv", "pointwise1x1") if k in _t}))

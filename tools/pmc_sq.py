#!/usr/bin/env python3
"""Aggregate one `rocprofv3 --kernel-trace --pmc <SQ counters>` pass of bench.py per kernel: where the waves' cycles go.
Usage: python tools/pmc_sq.py <dir with *_counter_collection.csv> [out.csv]
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles
the matrix pipe is busy summed over SIMDs (MI355X_MICROARCH.md, rocprofv3 PMC slots)."""
import collections
import csv
import glob
import sys


def main():
    d = sys.argv[1]
    out = sys.argv[2] if len(sys.argv) > 2 else None
    f = (glob.glob(d + "/*/*_counter_collection.csv") + glob.glob(d + "/*_counter_collection.csv"))[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-70:]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r["Dispatch_Id"], k)
        if key not in seen:
            seen.add(key)
            n[k] += 1
    cols = sorted({c for v in agg.values() for c in v})
    rows = []
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
        wc = v.get("SQ_WAVE_CYCLES", 0) or 1
        rows.append([k, n[k]] + ["%.0f" % (v.get(c, 0) / n[k]) for c in cols] +
                    ["%.3f" % (v.get("SQ_WAIT_ANY", 0) / wc), "%.3f" % (v.get("SQ_WAIT_INST_ANY", 0) / wc),
                     "%.3f" % (v.get("SQ_ACTIVE_INST_VALU", 0) / wc), "%.3f" % (v.get("SQ_ACTIVE_INST_ANY", 0) / wc)])
    hdr = ["kernel", "dispatches"] + [c + "_per_dispatch" for c in cols] + ["wait_any/wave_cycles", "wait_inst/wave_cycles",
                                                                             "active_valu/wave_cycles", "active_any/wave_cycles"]
    lines = [",".join(hdr)] + [",".join('"%s"' % x if i == 0 else str(x) for i, x in enumerate(r)) for r in rows]
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main()

#!/bin/bash
# round-3 helper (run through gpurun): whole GPU suite, then the conv table (batch ${B:-128}) and the default bench line
set -o pipefail
O=gpurun_out/r3; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/gpu_suite.txt 2>&1 || { tail -40 $O/gpu_suite.txt; exit 1; }
tail -2 $O/gpu_suite.txt
timeout -k 10 300 python tools/opbench.py all --batch ${B:-128} 2>&1 | cut -c1-80 | grep -v "fused\|2-krn" > $O/opbench_b${B:-128}.txt || exit 1
cat $O/opbench_b${B:-128}.txt
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_q.json 2> $O/bench_q.err || { tail -20 $O/bench_q.err; exit 1; }
python - <<'PY'
import json
l=json.load(open("gpurun_out/r3/bench_q.json"))
print(l["value"], l["windows"]["ms_per_step_min_median_max"], l["selfcheck"]["ok"], l["single_stream"]["value"], l["roofline"]["frac"], {k:(v["ms"]) for k,v in l["kernels"].items()})
PY

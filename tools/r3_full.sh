#!/bin/bash
# round-3 helper (run through gpurun): the whole GPU suite, smoke, and the default bench line
set -o pipefail
O=gpurun_out/r3; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/gpu_suite.txt 2>&1 || { tail -40 $O/gpu_suite.txt; exit 1; }
tail -3 $O/gpu_suite.txt
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1 || { tail -20 $O/smoke.txt; exit 1; }
tail -1 $O/smoke.txt
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err || { tail -20 $O/bench_default.err; exit 1; }
python - <<'PY'
import json
l=json.load(open("gpurun_out/r3/bench_default.json"))
print(l["value"], l["ms_per_step"], l["windows"], l["selfcheck"], l["single_stream"]["value"])
print(l["roofline"]["kernel"], l["roofline"]["frac"], {k:(v["ms"]) for k,v in l["kernels"].items()})
print(l["cpu_baseline"])
PY

mkdir -p gpurun_out/r3
for n in 1 2 3 4 5 6; do timeout -k 10 200 python bench.py --steps 100 --warmup 20 --windows 3 --no-cpu-baseline --no-selfcheck --inflight $n 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('inflight', $n, l['value'], l['windows']['ms_per_step_min_median_max'])"; done > gpurun_out/r3/inflight.txt 2>&1
cat gpurun_out/r3/inflight.txt

set -o pipefail
O=gpurun_out/p7; mkdir -p $O
ROUNDS=4 PLHIP_GEMM_DEBUG=32 timeout -k 10 120 python tools/patch_timeline.py --n 256 --cin 64 --cout 64 --hw 56 > $O/tl_res2.txt 2>&1; cat $O/tl_res2.txt
ROUNDS=4 PLHIP_PATCH_DELAY=60 PLHIP_GEMM_DEBUG=32 timeout -k 10 120 python tools/patch_timeline.py --n 256 --cin 64 --cout 64 --hw 56 > $O/tl_res2_d60.txt 2>&1; tail -12 $O/tl_res2_d60.txt
ROUNDS=2 PLHIP_GEMM_DEBUG=32 timeout -k 10 120 python tools/patch_timeline.py --n 256 --cin 256 --cout 256 --hw 14 > $O/tl_res4.txt 2>&1; cat $O/tl_res4.txt
timeout -k 10 300 python bench.py --config c4 --steps 10 --warmup 3 --no-cpu-baseline > $O/c4.json 2> $O/c4.err; tail -3 $O/c4.err; python - <<'PY'
import json
l=json.load(open("gpurun_out/p7/c4.json"))
print(l["value"], l["single_stream"]["value"], {k:round(v["ms"],3) for k,v in l["kernels"].items()})
PY

set -o pipefail
O=gpurun_out/p4; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_graphs.py -x -q -m gpu -k "patch_conv_route or implicit_gemm or resnet50_int8_program or full_size_properties_c2" > $O/t.txt 2>&1; echo rc=$? >> $O/t.txt; tail -4 $O/t.txt
for d in 0 20 40 80; do echo "== delay $d"; PLHIP_PATCH_DELAY=$d timeout -k 10 120 python tools/c2bench.py 2>&1 | grep "int8 out"; done > $O/c2_delay.txt 2>&1; cat $O/c2_delay.txt
PLHIP_GEMM_DEBUG=32 timeout -k 10 120 python tools/patch_timeline.py > $O/tl_c2.txt 2>&1; cat $O/tl_c2.txt
for sh in "256 64 56 64" "256 128 28 128" "256 256 14 256"; do set -- $sh; for d in 0 40; do echo "== delay $d"; PLHIP_PATCH_DELAY=$d timeout -k 10 120 python tools/c2bench.py --n $1 --cin $2 --hw $3 --cout $4 2>&1 | grep "int8 out"; done; PLHIP_CONV_PATCH=0 timeout -k 10 120 python tools/c2bench.py --n $1 --cin $2 --hw $3 --cout $4 2>&1 | grep "int8 out"; done > $O/resnet.txt 2>&1; cat $O/resnet.txt

mkdir -p gpurun_out/r2b
timeout -k 5 200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_graphs.py -q -m gpu -x 2>&1 | tail -3
for tr in 1 0; do
echo "== PLHIP_GEMM_TR=$tr"
PLHIP_GEMM_TR=$tr timeout -k 5 60 python tools/c2bench.py | head -1
PLHIP_GEMM_TR=$tr timeout -k 5 60 python tools/c2bench.py --n 256 --cin 64 --cout 64 --hw 56 | head -1
PLHIP_GEMM_TR=$tr timeout -k 5 60 python tools/c2bench.py --n 256 --cin 128 --cout 128 --hw 28 | head -1
PLHIP_GEMM_TR=$tr timeout -k 5 60 python tools/c2bench.py --n 256 --cin 256 --cout 256 --hw 14 | head -1
PLHIP_GEMM_TR=$tr timeout -k 5 60 python tools/c2bench.py --n 256 --cin 512 --cout 512 --hw 7 | head -1
done

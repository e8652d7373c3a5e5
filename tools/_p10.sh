timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_graphs.py -x -q -m gpu -k "patch_conv_route or implicit_gemm or resnet50 or full_size_properties_c2" 2>&1 | tail -2
timeout -k 10 200 python tools/opbench.py all --net resnet50_3x3 --batch 256 2>&1 | cut -c1-110 | head -4
cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/p10 -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/c2bench.py > /dev/null 2>&1; head -4 $GRAFT_REPO_ROOT/gpurun_out/p10/p_kernel_stats.csv | cut -c1-140

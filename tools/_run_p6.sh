set -o pipefail
O=gpurun_out/p6; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_graphs.py -x -q -m gpu -k "patch_conv_route or implicit_gemm or resnet50_int8_program or full_size_properties_c2" > $O/t.txt 2>&1; echo rc=$? >> $O/t.txt; tail -4 $O/t.txt
cd /tmp && export TMPDIR=/tmp
for sh in "32 64 56 128" "256 64 56 64" "256 128 28 128" "256 256 14 256"; do set -- $sh
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_$1_$2 -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/c2bench.py --n $1 --cin $2 --hw $3 --cout $4 > $GRAFT_REPO_ROOT/$O/prof_$1_$2.log 2>&1
  grep "int8 out" $GRAFT_REPO_ROOT/$O/prof_$1_$2.log
  head -4 $GRAFT_REPO_ROOT/$O/prof_$1_$2/p_kernel_stats.csv | cut -c1-150
done

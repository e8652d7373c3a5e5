set -o pipefail
O=gpurun_out/p8; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_c4 -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --config c4 --inflight 1 --steps 6 --warmup 2 --windows 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/c4.json 2> $GRAFT_REPO_ROOT/$O/c4.err
cd $GRAFT_REPO_ROOT
head -30 $O/prof_c4/p_kernel_stats.csv | cut -c1-170

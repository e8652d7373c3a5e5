#!/bin/bash
# Collects the round's measurement evidence on the GPU box (run through gpurun from the repo root):
#   kernel-trace stats of bench.py (default command = 3 steps in flight, and the serial form whose per-kernel times are the
#   ones behind `roofline`), FETCH_SIZE / WRITE_SIZE PMC passes (separate runs, kernel-trace only, serial form), the
#   per-layer table, the GEMM timeline, and plain bench lines.  Everything lands in gpurun_out/evidence/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/evidence
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats_default -o p --output-format csv -- python $R/bench.py --no-cpu-baseline --steps 30 > $O/bench_under_rocprof_default.json 2> $O/stats_default.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats_serial -o p --output-format csv -- python $R/bench.py --no-cpu-baseline --inflight 1 --steps 30 > $O/bench_under_rocprof_serial.json 2> $O/stats_serial.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o f --output-format csv -- python $R/bench.py --no-cpu-baseline --inflight 1 --steps 3 --warmup 1 > /dev/null 2> $O/fetch.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o w --output-format csv -- python $R/bench.py --no-cpu-baseline --inflight 1 --steps 3 --warmup 1 > /dev/null 2> $O/write.err || exit 1
# where the waves' cycles go: one SQ counter pass (8 SQ slots), serial form, aggregated per kernel by tools/pmc_sq.py
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU -d $O/sq_c3 -o s --output-format csv -- python $R/bench.py --no-cpu-baseline --inflight 1 --steps 3 --warmup 1 > /dev/null 2> $O/sq_c3.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU -d $O/sq_c4 -o s --output-format csv -- python $R/bench.py --config c4 --no-cpu-baseline --inflight 1 --steps 2 --warmup 1 > /dev/null 2> $O/sq_c4.err || exit 1
cd $R
python tools/pmc_sq.py $O/sq_c3 $O/pmc_sq_c3.csv > /dev/null || exit 1
python tools/pmc_sq.py $O/sq_c4 $O/pmc_sq_c4.csv > /dev/null || exit 1
timeout -k 10 200 python tools/opbench.py all 2>&1 | cut -c1-110 | grep -v "fused\|2-krn" > $O/opbench.txt || exit 1
PLHIP_GEMM_DEBUG=32 timeout -k 10 100 python tools/gemm_timeline.py pw8 > $O/gemm_timeline_pw8.txt 2>&1 || exit 1
timeout -k 10 400 python bench.py --layer-table > $O/bench.json 2> $O/layer_table.txt || exit 1
timeout -k 10 200 python bench.py --no-cpu-baseline --inflight 1 > $O/bench_inflight1.json 2>/dev/null || exit 1
timeout -k 10 200 python bench.py --no-cpu-baseline --inflight 2 > $O/bench_inflight2.json 2>/dev/null || exit 1
timeout -k 10 200 python bench.py --no-cpu-baseline --inflight 4 > $O/bench_inflight4.json 2>/dev/null || exit 1
# the other BASELINE configs (whole graphs), the per-layer table of ResNet50, C2 and its per-kernel split
timeout -k 10 300 python bench.py --config c4 --no-cpu-baseline --layer-table > $O/bench_c4.json 2> $O/layer_table_c4.txt || exit 1
timeout -k 10 300 python bench.py --config c5 --no-cpu-baseline --layer-table > $O/bench_c5.json 2> $O/layer_table_c5.txt || exit 1
timeout -k 10 300 python bench.py --config c2 --no-cpu-baseline > $O/bench_c2.json 2>/dev/null || exit 1
# the N > 1 data path on one GPU: one rank, RCCL broadcast / scatter / per-step all_gather
PLHIP_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_force_dist.json 2>/dev/null || exit 1
# per-layer tables at batch 256, the fused depthwise -> pointwise pairs, and their timelines
timeout -k 10 200 python tools/opbench.py all --batch 256 2>&1 | cut -c1-110 | grep -v "fused\|2-krn" > $O/opbench_b256.txt || exit 1
timeout -k 10 200 python tools/opbench.py fused --batch 128 2>&1 | cut -c1-110 > $O/opbench_fused.txt || exit 1
PLHIP_FUSED_DEBUG=96 timeout -k 10 100 python tools/fused_timeline.py dw8 > $O/fused_timeline_dw8.txt 2>&1 || exit 1
timeout -k 10 120 python tools/fused_concurrency.py --threads 3 > $O/fused_concurrency.txt 2>&1 || exit 1
PLHIP_GEMM_TR=2 PLHIP_GEMM_DEBUG=32 timeout -k 10 100 python tools/gemm_timeline.py pw8 --tr > $O/gemm_tr_timeline_pw8.txt 2>&1 || exit 1
timeout -k 10 100 python tools/c2bench.py > $O/c2bench.txt 2>&1 || exit 1
tail -c 300 $O/bench.json

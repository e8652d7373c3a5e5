#!/bin/bash
# Collects the round's measurement evidence on the GPU box (run through gpurun from the repo root):
#   kernel-trace stats of bench.py (default command = 3 steps in flight, and the serial form whose per-kernel times are the
#   ones behind `roofline`), FETCH_SIZE / WRITE_SIZE PMC passes (separate runs, kernel-trace only, serial form; c3 and c4),
#   SQ counter passes, the per-layer tables (batch 128 and 256), the wide-tile GEMM timelines, and the bench lines of every
#   BASELINE config WITH their cpu_baseline legs.  Everything lands in gpurun_out/evidence/ (tools/update_profiles.py copies
#   it under profiles/).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/evidence
# PART=1: the rocprofv3 passes; PART=2: tables, timelines and bench lines (each half stays under gpurun's 20-minute limit);
# default: both
PART=${PART:-all}
if [ "$PART" != "2" ]; then rm -rf $O; fi
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
if [ "$PART" != "2" ]; then
SQ="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU"
B="--no-cpu-baseline --no-selfcheck --windows 1"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats_default -o p --output-format csv -- python $R/bench.py $B --steps 30 > $O/bench_under_rocprof_default.json 2> $O/stats_default.err || exit 1
echo "stats default done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats_serial -o p --output-format csv -- python $R/bench.py $B --inflight 1 --steps 30 > $O/bench_under_rocprof_serial.json 2> $O/stats_serial.err || exit 1
echo "stats serial done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o f --output-format csv -- python $R/bench.py $B --inflight 1 --steps 3 --warmup 1 > /dev/null 2> $O/fetch.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o w --output-format csv -- python $R/bench.py $B --inflight 1 --steps 3 --warmup 1 > /dev/null 2> $O/write.err || exit 1
echo "pmc c3 done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch_c4 -o f --output-format csv -- python $R/bench.py --config c4 $B --inflight 1 --steps 2 --warmup 1 > /dev/null 2> $O/fetch_c4.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write_c4 -o w --output-format csv -- python $R/bench.py --config c4 $B --inflight 1 --steps 2 --warmup 1 > /dev/null 2> $O/write_c4.err || exit 1
echo "pmc c4 done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch_c2 -o f --output-format csv -- python $R/bench.py --config c2 $B --inflight 1 --steps 4 --warmup 1 > /dev/null 2> $O/fetch_c2.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write_c2 -o w --output-format csv -- python $R/bench.py --config c2 $B --inflight 1 --steps 4 --warmup 1 > /dev/null 2> $O/write_c2.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats_c2 -o p --output-format csv -- python $R/tools/c2bench.py > $O/c2bench_under_rocprof.txt 2> $O/stats_c2.err || exit 1
echo "pmc / stats c2 done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SQ -d $O/sq_c3 -o s --output-format csv -- python $R/bench.py $B --inflight 1 --steps 3 --warmup 1 > /dev/null 2> $O/sq_c3.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SQ -d $O/sq_c4 -o s --output-format csv -- python $R/bench.py --config c4 $B --inflight 1 --steps 2 --warmup 1 > /dev/null 2> $O/sq_c4.err || exit 1
echo "sq done"
cd $R
python tools/pmc_sq.py $O/sq_c3 $O/pmc_sq_c3.csv > /dev/null || exit 1
python tools/pmc_sq.py $O/sq_c4 $O/pmc_sq_c4.csv > /dev/null || exit 1
# the stride-2 patch route: phase copy + kernel per downsampling layer
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/stats_s2 -o p --output-format csv -- python $R/tools/opbench.py s2 --net resnet50_3x3 --batch 256 > $O/opbench_s2_under_rocprof.txt 2> $O/stats_s2.err || exit 1
fi
if [ "$PART" = "1" ]; then exit 0; fi
cd $R
timeout -k 10 200 python tools/opbench.py all 2>&1 | cut -c1-110 | grep -v "fused\|2-krn" > $O/opbench.txt || exit 1
# round 4: the fused depthwise -> pointwise kernel: the 13 pairs (fused where the kernel takes them, else the two kernels), its
# in-kernel timeline at batch 128 / 256 and in its timing experiments, the probes behind its design
timeout -k 10 200 python tools/opbench.py fused 2>&1 | cut -c1-110 | grep "fused\|2-krn" > $O/opbench_fused.txt || exit 1
timeout -k 10 100 python tools/fused_timeline.py > $O/fused_timeline_b128.txt 2>&1 || exit 1
timeout -k 10 100 python tools/fused_timeline.py --batch 256 > $O/fused_timeline_b256.txt 2>&1 || exit 1
for e in 1 2 3 19 4; do timeout -k 10 100 python tools/fused_timeline.py --exp $e > $O/fused_timeline_exp$e.txt 2>&1 || exit 1; done
for cfg in "128 128 56" "32 64 112" "256 256 28"; do set -- $cfg; timeout -k 10 100 python tools/stream_timeline.py --c $1 --m $2 --hw $3 > $O/stream_timeline_$3.txt 2>&1 || exit 1; done
# ... its stride-2 forms and the small-plane kernel (the 7 x 7 pairs; one / two blocks per image; the pooled output), and every pair
# fused against the two kernels (wall clock over 50 launches)
for cfg in "64 128 112" "128 256 56" "256 512 28"; do set -- $cfg; timeout -k 10 100 python tools/stream_timeline.py --c $1 --m $2 --hw $3 --stride 2 > $O/stream_timeline_$3_s2.txt 2>&1 || exit 1; done
timeout -k 10 100 python tools/stream_timeline.py --c 512 --m 1024 --hw 14 --stride 2 > $O/small_timeline_14_s2.txt 2>&1 || exit 1
timeout -k 10 100 python tools/stream_timeline.py --c 1024 --m 1024 --hw 7 --f32 > $O/small_timeline_7_f32.txt 2>&1 || exit 1
PLHIP_FUSED_SMALL=2 timeout -k 10 100 python tools/stream_timeline.py --c 1024 --m 1024 --hw 7 --f32 > $O/small_timeline_7_f32_two_blocks.txt 2>&1 || exit 1
: > $O/fused_vs_two_kernels.txt
for cfg in "32 64 112 1" "64 128 112 2" "128 128 56 1" "128 256 56 2" "256 256 28 1" "256 512 28 2" "512 512 14 1" "512 1024 14 2" "1024 1024 7 1"; do set -- $cfg
  echo "dw3x3 s$4 + pw1x1 $1 -> $2 @$3:" >> $O/fused_vs_two_kernels.txt
  timeout -k 10 120 python tools/fused_run.py --c $1 --m $2 --hw $3 --stride $4 --two --reps 50 2>&1 | grep "us per" >> $O/fused_vs_two_kernels.txt || exit 1
done
PLHIP_FUSED_SMALL=2 timeout -k 10 200 python tools/opbench.py fused 2>&1 | cut -c1-110 | grep "fused\|2-krn" > $O/opbench_fused_small_two_blocks.txt || exit 1
[ -x tools/_probe_coexec ] && ./tools/_probe_coexec > $O/probe_coexec.txt 2>&1
[ -x tools/_probe_rtz ] && ./tools/_probe_rtz > $O/probe_cvt_rtz.txt 2>&1
PLHIP_FUSED_STREAM=0 PLHIP_FUSED_SMALL=0 timeout -k 10 200 python bench.py --no-cpu-baseline > $O/bench_stream_off.json 2>/dev/null || exit 1
PLHIP_FUSED_STREAM=0 PLHIP_FUSED_SMALL=0 timeout -k 10 200 python bench.py --no-cpu-baseline --inflight 1 > $O/bench_stream_off_inflight1.json 2>/dev/null || exit 1
PLHIP_FUSED_STREAM=2 PLHIP_FUSED_SMALL=0 timeout -k 10 200 python bench.py --no-cpu-baseline > $O/bench_stride1_only.json 2>/dev/null || exit 1
PLHIP_FUSED_SMALL=0 timeout -k 10 200 python bench.py --no-cpu-baseline > $O/bench_small_off.json 2>/dev/null || exit 1
PLHIP_FUSED_SMALL=2 timeout -k 10 200 python bench.py --no-cpu-baseline > $O/bench_small_two_blocks.json 2>/dev/null || exit 1
PLHIP_BENCH_FUSE_DWPW=1 timeout -k 10 200 python bench.py --no-cpu-baseline > $O/bench_no_pool_tail.json 2>/dev/null || exit 1
PLHIP_BENCH_FUSE_DWPW=0 timeout -k 10 200 python bench.py --no-cpu-baseline > $O/bench_dwpw_off.json 2>/dev/null || exit 1
PLHIP_BENCH_FUSE_DWPW=0 timeout -k 10 200 python bench.py --no-cpu-baseline --inflight 1 > $O/bench_dwpw_off_inflight1.json 2>/dev/null || exit 1
timeout -k 10 200 python tools/opbench.py all --batch 256 2>&1 | cut -c1-110 | grep -v "fused\|2-krn" > $O/opbench_b256.txt || exit 1
PLHIP_GEMM_WIDE=0 timeout -k 10 200 python tools/opbench.py pw 2>&1 | cut -c1-110 > $O/opbench_pw_wide_off.txt || exit 1
for l in pw8 pw6 pw13; do PLHIP_GEMM_DEBUG=32 timeout -k 10 100 python tools/wide_timeline.py $l > $O/wide_timeline_$l.txt 2>&1 || exit 1; done
# (pw14 has an fp32 output: it runs on the ring kernels by default; its wide-kernel timeline needs the tile forced)
PLHIP_WIDE_NTT=4 PLHIP_GEMM_DEBUG=32 timeout -k 10 100 python tools/wide_timeline.py pw14 > $O/wide_timeline_pw14.txt 2>&1 || exit 1
PLHIP_GEMM_WIDE=0 PLHIP_GEMM_DEBUG=32 timeout -k 10 100 python tools/gemm_timeline.py pw8 > $O/gemm_timeline_pw8_ring.txt 2>&1 || exit 1
# the patch kernel (conv_patch_i8.hip): config #2 and ResNet50's 3x3 layers, A/B against the ring kernel's implicit GEMM, the
# in-kernel timelines, the time without the epilogue, and the depthwise 5x5 rows
timeout -k 10 200 python tools/opbench.py all --net resnet50_3x3 --batch 256 2>&1 | cut -c1-130 > $O/opbench_resnet50_3x3.txt || exit 1
PLHIP_CONV_PATCH=0 timeout -k 10 200 python tools/opbench.py all --net resnet50_3x3 --batch 256 2>&1 | cut -c1-130 > $O/opbench_resnet50_3x3_patch_off.txt || exit 1
PLHIP_PATCH_DEBUG=1 timeout -k 10 200 python tools/opbench.py all --net resnet50_3x3 --batch 256 2>&1 | cut -c1-130 > $O/opbench_resnet50_3x3_no_epilogue.txt || exit 1
PLHIP_CONV_PATCH_S2=0 PLHIP_STEM7=0 timeout -k 10 200 python tools/opbench.py s2 --net resnet50_3x3 --batch 256 2>&1 | cut -c1-130 > $O/opbench_resnet50_3x3_s2_patch_off.txt || exit 1
timeout -k 10 200 python tools/opbench.py all --net dw5x5 2>&1 | cut -c1-110 > $O/opbench_dw5x5.txt || exit 1
PLHIP_DW5_DIRECT=0 timeout -k 10 200 python tools/opbench.py all --net dw5x5 2>&1 | cut -c1-110 > $O/opbench_dw5x5_lds_band.txt || exit 1
PLHIP_PATCH_DEBUG=32 timeout -k 10 100 python tools/patch_timeline.py > $O/patch_timeline_c2.txt 2>&1 || exit 1
ROUNDS=4 PLHIP_PATCH_DEBUG=32 timeout -k 10 100 python tools/patch_timeline.py --n 256 --cin 64 --cout 64 --hw 56 > $O/patch_timeline_res2.txt 2>&1 || exit 1
PLHIP_PATCH_DEBUG=32 timeout -k 10 100 python tools/patch_timeline.py --n 256 --cin 256 --cout 256 --hw 14 > $O/patch_timeline_res4.txt 2>&1 || exit 1
PLHIP_PATCH_DEBUG=32 timeout -k 10 100 python tools/patch_timeline.py --n 256 --cin 128 --cout 128 --hw 56 --stride 2 > $O/patch_timeline_res3a_s2.txt 2>&1 || exit 1
PLHIP_PATCH_DEBUG=32 timeout -k 10 100 python tools/patch_timeline.py --n 256 --cin 512 --cout 512 --hw 14 --stride 2 > $O/patch_timeline_res5a_s2.txt 2>&1 || exit 1
echo "tables done"
timeout -k 10 400 python bench.py --layer-table > $O/bench.json 2> $O/layer_table.txt || exit 1
timeout -k 10 200 python bench.py --no-cpu-baseline --inflight 1 > $O/bench_inflight1.json 2>/dev/null || exit 1
timeout -k 10 200 python bench.py --steps 20 --warmup 5 > $O/bench_driver_form.json 2>/dev/null || exit 1
echo "c3 done"
# the other BASELINE configs (whole graphs, their own batch), with their cpu_baseline legs and per-instruction tables
timeout -k 10 400 python bench.py --config c4 --layer-table > $O/bench_c4.json 2> $O/layer_table_c4.txt || exit 1
timeout -k 10 400 python bench.py --config c5 --layer-table > $O/bench_c5.json 2> $O/layer_table_c5.txt || exit 1
timeout -k 10 300 python bench.py --config c2 > $O/bench_c2.json 2>/dev/null || exit 1
# the N > 1 data path on one GPU: one rank, RCCL broadcast / scatter / per-step all_gather
PLHIP_BENCH_FORCE_DIST=1 PLHIP_BENCH_WATCHDOG=200 timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_force_dist.json 2> $O/force_dist.err || { echo "force_dist rc=$?"; tail -30 $O/force_dist.err; exit 1; }
timeout -k 10 100 python tools/c2bench.py > $O/c2bench.txt 2>&1 || exit 1
tail -c 400 $O/bench.json

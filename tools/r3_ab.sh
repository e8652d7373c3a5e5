#!/bin/bash
# round-3 A/B helper (run through gpurun): parity of the wide-tile GEMM, its timeline, and the pointwise table with it off / on
set -o pipefail
O=gpurun_out/r3; mkdir -p $O; rm -f $O/ab_pw.txt
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "wide or mobilenet_layer or ring_gemm or conv1x1 or gemm_g" > $O/t_wide.txt 2>&1 || { tail -30 $O/t_wide.txt; exit 1; }
tail -3 $O/t_wide.txt
for l in ${TL:-pw8}; do PLHIP_GEMM_DEBUG=32 timeout -k 10 120 python tools/wide_timeline.py $l --batch ${B:-128} > $O/wide_tl_$l.txt 2>&1 || exit 1; cat $O/wide_tl_$l.txt; done
for v in ${VARS:-PLHIP_GEMM_WIDE=0 PLHIP_GEMM_WIDE=1}; do
  echo "== $v" >> $O/ab_pw.txt
  env $v timeout -k 10 200 python tools/opbench.py pw --batch ${B:-128} 2>&1 | cut -c1-80 >> $O/ab_pw.txt || exit 1
done
cat $O/ab_pw.txt

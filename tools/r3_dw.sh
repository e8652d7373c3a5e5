#!/bin/bash
# round-3 helper (run through gpurun): parity of the depthwise paths, then the depthwise table with the MFMA kernel off / on
set -o pipefail
O=gpurun_out/r3; mkdir -p $O; rm -f $O/ab_dw.txt
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py -x -q -k "dw or golden or mobilenet or depthwise or fused" > $O/t_dw.txt 2>&1 || { tail -40 $O/t_dw.txt; exit 1; }
tail -3 $O/t_dw.txt
for v in ${VARS:-PLHIP_DW_MFMA=0 PLHIP_DW_MFMA=1}; do
  echo "== $v" >> $O/ab_dw.txt
  env $v timeout -k 10 200 python tools/opbench.py dw --batch ${B:-128} 2>&1 | cut -c1-80 >> $O/ab_dw.txt || exit 1
done
cat $O/ab_dw.txt

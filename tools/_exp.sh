#!/bin/bash
# scratch: timing experiments of the fused kernel
mkdir -p gpurun_out/fz
for d in 0 1 2 4 5 6 7; do
  echo "== PLHIP_FUSED_DEBUG=$d"
  PLHIP_FUSED_DEBUG=$d timeout -k 10 200 python tools/opbench.py fused --batch 128 2>&1 | grep -E "dw2 |dw4 |dw6 |dw8 |dw13|dw14|fused total"
done

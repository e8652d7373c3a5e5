mkdir -p gpurun_out/r2b
for cfg in 0 1; do for d in 0 50 100 150; do
  echo "cfg $cfg delay $d: $(PLHIP_TR_CFG=$cfg PLHIP_TR_DELAY=$d timeout -k 5 60 python tools/opbench.py pw8 --batch 128 | head -1)"
done; done
PLHIP_TR_CFG=1 PLHIP_TR_DELAY=100 timeout -k 5 60 python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -2

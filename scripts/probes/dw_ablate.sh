#!/bin/bash
# in-kernel weight gradients: correctness + per-kernel times as built and with the probe switches of LayerArgsDW::ablate (key 17)
python -m pytest tests/test_gpu_parity.py -x -q -k "in_kernel_weight_gradients" 2>&1 | tail -3
for ab in ${@:-0 7}; do
  echo "== ablate $ab"
  bash scripts/stats_step.sh bf16 16=1 17=$ab 2>&1 | grep -E "wsdw|dw_gather|wreduce_wnorm_batch|ms_per_step"
done

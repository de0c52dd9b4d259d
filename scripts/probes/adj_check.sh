#!/bin/bash
# adjoint chains: equality tests + per-kernel times (bf16 and fp16x3)
python -m pytest tests/test_gpu_parity.py -x -q -k "fused_adjoint or g2 or golden_g2 or sdf_outputs" 2>&1 | tail -3
bash scripts/stats_step.sh bf16 2>&1 | grep -E "adjoint|ms_per_step"
bash scripts/stats_step.sh fp16x3 2>&1 | grep -E "adjoint|sdf_chain|ms_per_step"

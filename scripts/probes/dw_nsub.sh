for n in 4 8 16; do echo "== nsub $n"; bash scripts/stats_step.sh bf16 16=1 18=$n 2>&1 | grep -E "dw_gather|wreduce_wnorm_batch"; done

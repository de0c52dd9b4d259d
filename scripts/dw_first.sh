set -x
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "in_kernel_weight_gradients" 2>&1 | tail -15
for k in 0 1; do timeout 300 python scripts/bench_tune.py 16=$k -- --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-prof 2>&1 | grep '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bf16 key16=$k', d['ms_per_step'])"; done
for k in 0 1; do timeout 300 python scripts/bench_tune.py 16=$k -- --precision fp16x3 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-prof 2>&1 | grep '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp16x3 key16=$k', d['ms_per_step'])"; done

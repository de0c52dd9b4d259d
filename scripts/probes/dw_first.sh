# A/B of a tuning key on the bench step (no profiler): bash scripts/probes/dw_first.sh KEY "V0 V1" [precisions]
key=${1:-16}; vals=${2:-"0 1"}; precs=${3:-"bf16 fp16x3"}
for prec in $precs; do for rep in 1 2; do for v in $vals; do
  timeout 300 python scripts/bench_tune.py $key=$v -- --precision $prec --steps 30 --warmup 5 --no-secondary --no-cpu-baseline --no-prof 2>&1 | grep '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$prec key$key=$v', round(d['ms_per_step'],4))"
done; done; done

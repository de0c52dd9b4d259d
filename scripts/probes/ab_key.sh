# A/B of one tuning key on the bench step:  bash scripts/probes/ab_key.sh KEY V0 V1 [precision]
K=$1; A=$2; B=$3; P=${4:-bf16}
for r in 1 2; do for v in $A $B; do
python scripts/bench_tune.py $K=$v -- --precision $P --no-secondary --no-cpu-baseline --no-prof --steps 40 2>/dev/null | grep '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('key $K=$v', d['ms_per_step'])"
done; done

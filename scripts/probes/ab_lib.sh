# A/B of library builds through NEAT_HIP_LIB:  bash scripts/probes/ab_lib.sh [precision] -- the default build against every abl_libs/libneat_*.so
P=${1:-bf16}
for r in 1 2; do
for lib in "" $PWD/abl_libs/libneat_*.so; do
NEAT_HIP_LIB=$lib python bench.py --precision $P --no-secondary --no-cpu-baseline --steps 40 2>/dev/null | grep '^{"metric"' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['all_kernels']
print('%-14s' % ('$lib'.split('libneat_')[-1] or 'default'), 'ms', round(d['ms_per_step'],4), ' '.join('%s %.1f' % (n.split('_kernel')[0], v['avg_us']) for n, v in k.items()))"
done; done

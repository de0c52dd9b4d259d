for r in 1 2; do
for lib in "" "$PWD/abl_libs/libneat_narrow.so"; do
NEAT_HIP_LIB=$lib python bench.py --no-secondary --no-cpu-baseline --steps 40 2>/dev/null | grep '^{"metric"' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['all_kernels']
print('lib', '$lib'[-14:] or 'wide(default)', 'ms', round(d['ms_per_step'],4), 'fused', round(k['sdf_fused_kernel']['avg_us'],1), 'adjoint', round(k['sdf_adjoint_kernel']['avg_us'],1))"
done; done

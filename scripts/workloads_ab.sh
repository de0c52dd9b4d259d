for t in 16=0 16=2; do echo "== $t"; NEAT_TUNING=$t python scripts/bench_workloads.py 2>&1 | grep workload | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['workload'][:60], d.get('sampler','')[:30], round(d.get('ms_per_step', d.get('ms_per_chunk',0)),3))"; done

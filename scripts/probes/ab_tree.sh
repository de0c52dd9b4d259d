# same-box A/B of two source trees (this one against a copy of an older commit under abl_old/, see NOTEBOOK):
#   bash scripts/probes/ab_tree.sh    -> C2 bench line and the secondary workloads of both trees, alternating
R=$PWD
for r in 1 2; do
for t in $R/abl_old $R; do
  cd $t
  PYTHONPATH=$t python bench.py --no-cpu-baseline --steps 40 2>/dev/null | grep '^{' | python -c "
import sys,json; d=json.loads(sys.stdin.read()); s=d.get('secondary',{})
print('$t'.split('/')[-1].ljust(8), 'c2 %.4f' % d['ms_per_step'], ' '.join('%s %.3f' % (k.replace('_512x128','').replace('_2048x128','').replace('_64+64',''), v['ms_per_step']) for k,v in s.items() if 'ms_per_step' in v))"
done; done

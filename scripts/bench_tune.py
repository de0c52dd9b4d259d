"""bench.py with tuning keys: python scripts/bench_tune.py key=value ... [-- bench args]"""
import os, sys, runpy
import torch  # first: libneat_hip.so must bind to the HIP runtime torch loads
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neat_amd import _lib
args = sys.argv[1:]
rest = []
if '--' in args:
    i = args.index('--'); rest = args[i + 1:]; args = args[:i]
for kv in args:
    k, v = kv.split('=')
    _lib.check(_lib.lib().neat_set_tuning(int(k), int(v)), f"tuning {kv}")
sys.argv = ['bench.py'] + rest
runpy.run_path(os.path.join(ROOT, 'bench.py'), run_name='__main__')

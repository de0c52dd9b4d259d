"""Time the fused primal kernel (values mode) of an ablated library build: NEAT_LIB=path python scripts/probes/probe_fused_abl.py GEN"""
import os, sys, torch
sys.path.insert(0, '.')
from neat_amd import _lib
if os.environ.get("NEAT_LIB"):
    _lib.LIB_PATH = os.environ["NEAT_LIB"]
from neat_amd import networks, synth
gen = int(sys.argv[1]); P = 133120
dev = torch.device('cuda:0')
lib = _lib.lib()
m = networks.VolSDFNetwork(synth.ABC_NEAT_A_MODEL_CONF)
m.load_state_dict({k: torch.tensor(v) for k, v in synth.synth_state_dict(42, "rough").items()})
m.to(dev).eval().set_precision("bf16")
x = (torch.rand(P, 3, generator=torch.Generator().manual_seed(0)) * 4 - 2).to(dev)
_lib.check(lib.neat_set_tuning(4, gen), "neat_set_tuning")
with torch.no_grad():
    for _ in range(3): m.implicit_network.get_sdf_vals(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): m.implicit_network.get_sdf_vals(x)
    e1.record(); torch.cuda.synchronize()
print(f"{os.environ.get('NEAT_LIB', 'default')} gen {gen}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")

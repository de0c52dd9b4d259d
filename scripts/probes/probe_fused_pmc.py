"""One fused-primal kernel generation in a loop (values mode or save mode), for rocprofv3 --pmc runs.
    python scripts/probes/probe_fused_pmc.py GEN MODE [P]      MODE: values | save"""
import os, sys
import torch
sys.path.insert(0, '.')
from neat_amd import _lib
if os.environ.get('NEAT_LIB'):
    _lib.LIB_PATH = os.environ['NEAT_LIB']      # a probe build from scripts/probes/abl_build.sh
from neat_amd import networks, synth
gen, mode = int(sys.argv[1]), sys.argv[2]
P = int(sys.argv[3]) if len(sys.argv) > 3 else 133120
dev = torch.device('cuda:0')
lib = _lib.lib()
m = networks.VolSDFNetwork(synth.ABC_NEAT_A_MODEL_CONF)
m.load_state_dict({k: torch.tensor(v) for k, v in synth.synth_state_dict(42, "rough").items()})
m.to(dev).eval().set_precision("bf16")
x = (torch.rand(P, 3, generator=torch.Generator().manual_seed(0)) * 4 - 2).to(dev)
_lib.check(lib.neat_set_tuning(4, gen), "neat_set_tuning")
with torch.no_grad():
    for _ in range(4):
        if mode == "values":
            m.implicit_network.get_sdf_vals(x)
        else:
            m.implicit_network.get_outputs(x)
torch.cuda.synchronize()

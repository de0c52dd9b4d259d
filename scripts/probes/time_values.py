"""time the fused primal kernel in values mode (no activation stores) vs save mode at the C2 point count"""
import sys, time, torch
sys.path.insert(0, '.')
from neat_amd import synth, networks, ops, _lib
dev = torch.device('cuda:0')
m = networks.VolSDFNetwork(synth.ABC_NEAT_A_MODEL_CONF)
m.load_state_dict({k: torch.tensor(v) for k, v in synth.synth_state_dict(42, 'rough').items()})
m.to(dev).eval().set_precision('bf16')
x = (torch.rand(133120, 3, device=dev) - 0.5) * 4
for nt in (2, 3, 4):
    _lib.lib().neat_set_tuning(5, nt)
    with torch.no_grad():
        for _ in range(3): m.implicit_network.get_sdf_vals(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): m.implicit_network.get_sdf_vals(x)
        torch.cuda.synchronize(); tv = (time.perf_counter() - t0) / 20
        for _ in range(3): m.implicit_network.get_outputs(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): m.implicit_network.get_outputs(x)
        torch.cuda.synchronize(); to = (time.perf_counter() - t0) / 20
    print(f"NT={nt}: get_sdf_vals (values mode, incl. small glue) {tv*1e6:.0f} us ; get_outputs (save mode + adjoint chain + finalize) {to*1e6:.0f} us")

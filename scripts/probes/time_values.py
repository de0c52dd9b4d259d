"""Values-mode fused SDF forward (one sampler round's query, 1024 x 128 points) timed three ways: library HIP events around eager launches,
a captured graph of 20 launches between torch events, and (under rocprofv3 --kernel-trace) the tracer's durations.
  python scripts/probes/time_values.py [precision] [z: synth|linspace]"""
import ctypes, sys, time
import torch
sys.path.insert(0, ".")
from neat_amd import _lib, synth
from neat_amd.train import Trainer, synthetic_batch

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
zkind = sys.argv[2] if len(sys.argv) > 2 else "synth"
dev = torch.device("cuda:0")
lib = _lib.lib()
sd = {k: torch.tensor(v) for k, v in synth.synth_state_dict(42, "rough").items()}
_, inp, gt = synthetic_batch(42, 1024, dev)
tr = Trainer(device=dev, state_dict=sd)
tr.model.set_precision(prec).eval()
net = tr.model.implicit_network
with torch.no_grad():
    dirs, cam = tr.model._rays(inp)
    z = torch.tensor(synth.synth_z_vals(42, 1024, 128)).to(dev) if zkind == "synth" else torch.linspace(0, 6, 128, device=dev).repeat(1024, 1).contiguous()
    for _ in range(5):
        net.get_sdf_vals_rays(cam, dirs, z)
    torch.cuda.synchronize()
    lib.neat_prof_enable(1)
    for _ in range(20):
        net.get_sdf_vals_rays(cam, dirs, z)
    torch.cuda.synchronize()
    ms, fl, n, by = ctypes.c_double(), ctypes.c_double(), ctypes.c_int(), ctypes.c_double()
    lib.neat_prof_collect(2, ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(n), ctypes.byref(by))
    lib.neat_prof_enable(0)
    print(f"{prec} {zkind}: library events, eager: {1e3 * ms.value / n.value:.1f} us per launch ({n.value} launches)")
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            net.get_sdf_vals_rays(cam, dirs, z)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print(f"{prec} {zkind}: graph of 20 (points kernel + values kernel): {1e3 * e0.elapsed_time(e1) / 100:.1f} us per pair")

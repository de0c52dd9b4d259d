import sys, time, torch
sys.path.insert(0, '.')
import bench
from neat_amd import synth
from neat_amd.wireframe import WireframeGraph
from oracle import neat_oracle as O
R, S = 128, 128
sd = synth.synth_state_dict(42, "rough"); sc = synth.synth_scene(seed=42, n_rays=R)
z = torch.tensor(synth.synth_z_vals(42, R, S))
wf = WireframeGraph(torch.tensor(sc["wf_vertices"]), torch.tensor(sc["wf_vconf"]), torch.tensor(sc["wf_edges"]), torch.tensor(sc["wf_weights"]), 512, 512)
inp = {k: torch.tensor(sc[k]) for k in ("intrinsics", "pose", "uv", "uv_proj")}
for th in (8, 16, 32, 64):
    torch.set_num_threads(th)
    p = O.params_from_numpy(sd, requires_grad=True)
    ts = []
    for it in range(3):
        t0 = time.perf_counter()
        rand = {"eik_idx": torch.randint(S, (R,)), "eik_uniform": torch.empty(R, 3).uniform_(-3, 3)}
        out = O.full_forward(p, inp, wf.line_segments(), wf.vertices, training=True, rand=rand, z_vals=z)
        lo = O.neat_loss(out, torch.tensor(sc["gt_rgb"]), torch.tensor(sc["gt_lines2d"]))
        lo["loss"].backward()
        ts.append(time.perf_counter() - t0)
    print(th, 'threads', ['%.2f' % t for t in ts], 'samples/s', R * S / min(ts), flush=True)

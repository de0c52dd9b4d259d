"""Is a replayed step host-bound?  For C2 / the sampler step / C4's per-rank shape: wall time per step of N replays WITH the final
synchronize (what bench.py reports), host time per step of the enqueue loop alone (no synchronize inside the timed region: the loop
returns as soon as the last replay is queued), and the GPU time between two events recorded around the N replays.
host << total: the host runs ahead, the step is GPU time.  host ~= total: the host is the bottleneck."""
import copy, json, sys, time
import torch
sys.path.insert(0, '.')
from neat_amd import synth
from neat_amd.train import Trainer, synthetic_batch
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
N = 40
dev = torch.device("cuda:0")

def measure(label, tr, inp, gt):
    for _ in range(3):
        tr.step(inp, gt)
    tr.capture(inp, gt)
    for _ in range(20):
        tr.step(inp, gt)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(N):
        tr.step(inp, gt)
    e1.record()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_total = time.perf_counter() - t0
    # the host's own cost per step: a SHORT burst after a synchronize (the queue never fills, nothing throttles the host)
    burst = []
    for _ in range(6):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(3):
            tr.step(inp, gt)
        burst.append((time.perf_counter() - t1) / 3)
    torch.cuda.synchronize()
    print(json.dumps({"workload": label, "host_unthrottled_ms_per_step": 1e3 * min(burst), "precision": prec, "total_ms_per_step": 1e3 * t_total / N, "host_enqueue_ms_per_step": 1e3 * t_host / N,
                      "gpu_event_ms_per_step": e0.elapsed_time(e1) / N}), flush=True)

sd = {k: torch.tensor(v) for k, v in synth.synth_state_dict(42, "rough").items()}
tr = Trainer(device=dev, state_dict=sd)
tr.model.set_precision(prec)
_, inp, gt = synthetic_batch(42, 1024, dev)
tr.model.z_vals_override = torch.tensor(synth.synth_z_vals(42, 1024, 128)).to(dev)
measure("C2 (1024 x 128 given samples)", tr, inp, gt)
tr._graphs.clear()
tr.model.z_vals_override = None
tr.model.ray_sampler.sync_free = True
measure("sampler step (ErrorBoundSampler, device-decided rounds)", tr, inp, gt)
conf = copy.deepcopy(synth.ABC_NEAT_A_MODEL_CONF)
conf.update(dbscan_enabled=True, use_median=False)
conf["global_junctions"] = dict(conf["global_junctions"], num_junctions=1024)
sd4 = {k: torch.tensor(v) for k, v in synth.synth_state_dict(42, "rough", num_junctions=1024).items()}
tr4 = Trainer(model_conf=conf, device=dev, state_dict=sd4)
tr4.model.set_precision(prec)
_, inp4, gt4 = synthetic_batch(42, 512, dev)
tr4.model.z_vals_override = torch.tensor(synth.synth_z_vals(42, 512, 128)).to(dev)
measure("C4 per-rank shape (512 x 128, DTU switches)", tr4, inp4, gt4)

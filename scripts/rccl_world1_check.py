"""RCCL on the ONE GPU a builder box has: a world-size-1 `nccl` process group, the train step's flat gradient
bucket through a real dist.all_reduce on the device buffer -- eager steps, then steps replayed from the HIP graph with the
all-reduce between replay and Adam -- and the trajectory must equal the same steps without any process group.  Prints whether the
RCCL library is mapped into the process.  Launched by tests/test_dp_gpu.py (needs NEAT_FORCE_DIST=1, MASTER_ADDR / MASTER_PORT)."""
import os, sys
import torch
import torch.distributed as dist
sys.path.insert(0, '.')
from neat_amd import dp, synth
from neat_amd.train import Trainer, synthetic_batch

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)


def run(with_group):
    torch.manual_seed(42)
    tr = Trainer(device=dev, state_dict={k: torch.tensor(v) for k, v in synth.synth_state_dict(42, "rough").items()})
    tr.bucket.force_collective = with_group
    _, inp, gt = synthetic_batch(42, 128, dev)
    tr.model.z_vals_override = torch.tensor(synth.synth_z_vals(42, 128, 64)).to(dev)
    tr.model.set_precision("bf16")
    losses = []
    for _ in range(2):
        losses.append(float(tr.step(inp, gt)[1]["loss"].detach()))
    ok = tr.capture(inp, gt)
    for _ in range(3):
        losses.append(float(tr.step(inp, gt)[1]["loss"].detach()))
    # the in-place path on a caller-owned flat gradient buffer
    flat = torch.arange(1000, device=dev, dtype=torch.float32)
    tr.bucket.all_reduce_mean(flat)
    torch.cuda.synchronize()
    return ok, tr.replays, losses, torch.cat([p.detach().reshape(-1) for p in tr.model.parameters()]).cpu(), flat.cpu()


assert os.environ.get("NEAT_FORCE_DIST") == "1"
base = run(False)
rank, world, local = dp.init_from_env(backend="nccl")
assert dist.is_initialized() and dist.get_backend() == "nccl" and world == 1
got = run(True)
dist.barrier()
maps = open("/proc/self/maps").read()
rccl = sorted({line.split()[-1] for line in maps.splitlines() if "rccl" in line.lower() or "nccl" in line.lower()})
print("graph captured:", got[0], "replays:", got[1], "losses:", ["%.6f" % v for v in got[2]])
print("RCCL mapped:", rccl)
assert got[0] and got[1] >= 3
assert base[2] == got[2] and torch.equal(base[3], got[3]), "the world-1 all-reduce changed the trajectory"
assert torch.equal(got[4], torch.arange(1000, dtype=torch.float32))
assert rccl, "no RCCL library in the process map"
print("RCCL world-1 check OK")
dist.destroy_process_group()

"""Two ranks on ONE GPU over gloo (RCCL refuses two ranks per device): HIP-graph train step + flat-bucket all-reduce.
Checks that ranks with different rays end every step with identical parameters.  Launch:
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 scripts/dp_graph_check.py"""
import os, sys
import torch
import torch.distributed as dist
sys.path.insert(0, '.')
from neat_amd import dp, synth
from neat_amd.train import Trainer, synthetic_batch

rank, world, _ = dp.init_from_env(backend="gloo")
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
torch.manual_seed(dp.rank_seed(42, rank))
tr = Trainer(device=dev, state_dict={k: torch.tensor(v) for k, v in synth.synth_state_dict(42, "rough").items()})
_, inp, gt = synthetic_batch(dp.rank_seed(42, rank), 128, dev, view=rank)
tr.model.z_vals_override = torch.tensor(synth.synth_z_vals(dp.rank_seed(42, rank), 128, 64)).to(dev)
tr.model.set_precision("bf16")
for _ in range(2):
    tr.step(inp, gt)
ok = tr.capture(inp, gt)
for _ in range(4):
    _, lo = tr.step(inp, gt)
flat = torch.cat([p.detach().reshape(-1) for p in tr.model.parameters()]).cpu()
gathered = [torch.empty_like(flat) for _ in range(world)]
dist.all_gather(gathered, flat)
diff = max(float((g - gathered[0]).abs().max()) for g in gathered)
if rank == 0:
    print("in-place all-reduce groups:", [len(g) for g in tr.bucket._plan], "of", len(tr.bucket.params), "tensors")
    print(f"graph captured: {ok} ({tr.capture_error!r}); loss {float(lo['loss']):.5f}; max parameter difference across ranks: {diff:.3e}")
    assert diff == 0.0
dist.destroy_process_group()

"""Two ranks: HIP-graph train step + ONE flat gradient all-reduce + Adam per step.  Ranks render different views with different rays
and must end every step with identical parameters.  On a box with fewer GPUs than ranks the ranks share a device and the
all-reduce goes through gloo (RCCL refuses two ranks per device); with a GPU per rank it is RCCL.  Launch:
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 scripts/dp_graph_check.py
(tests/test_dp_gpu.py does)"""
import os, sys
import torch
import torch.distributed as dist
sys.path.insert(0, '.')
from neat_amd import dp, synth
from neat_amd.train import Trainer, synthetic_batch

ndev = torch.cuda.device_count()
backend = "nccl" if ndev >= int(os.environ.get("WORLD_SIZE", "1")) else "gloo"
torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % ndev)
rank, world, local = dp.init_from_env(backend=backend)
dev = torch.device("cuda", local % ndev)
torch.manual_seed(dp.rank_seed(42, rank))
tr = Trainer(device=dev, state_dict={k: torch.tensor(v) for k, v in synth.synth_state_dict(42, "rough").items()})
_, inp, gt = synthetic_batch(dp.rank_seed(42, rank), 128, dev, view=rank)
tr.model.z_vals_override = torch.tensor(synth.synth_z_vals(dp.rank_seed(42, rank), 128, 64)).to(dev)
tr.model.set_precision("bf16")
for _ in range(2):
    tr.step(inp, gt)
fault_rank = int(os.environ.get("NEAT_TEST_CAPTURE_FAULT_RANK", "-1"))      # tests: the capture fails on this rank only
if rank == fault_rank:
    def _fault():
        raise RuntimeError("injected capture fault")
    tr._capture_fault = _fault
ok = tr.capture(inp, gt)
# the agreement bench.py makes: a rank whose capture failed sends everybody to eager steps.  Trainer.capture took the same
# number of optimizer steps (= all-reduces) on every rank whatever its outcome, so this small collective pairs with itself.
ok_all = torch.tensor([1.0 if ok else 0.0], device=dev)
dist.all_reduce(ok_all, op=dist.ReduceOp.MIN)
if ok and float(ok_all.item()) == 0.0:
    tr._graphs.clear()
for _ in range(4):
    _, lo = tr.step(inp, gt)
flat = torch.cat([p.detach().reshape(-1) for p in tr.model.parameters()]).cpu()
gathered = [torch.empty_like(flat) for _ in range(world)]
dist.all_gather(gathered, flat)
diff = max(float((g - gathered[0]).abs().max()) for g in gathered)
if rank == 0:
    print(f"backend {backend}; graph captured: {ok} ({tr.capture_error!r}); replays {tr.replays}; loss {float(lo['loss'].detach()):.5f}; "
          f"max parameter difference across ranks: {diff:.3e}", flush=True)
    if fault_rank < 0:
        assert ok and tr.replays >= 4
    assert diff == 0.0 and torch.isfinite(lo["loss"]).all()
dist.destroy_process_group()

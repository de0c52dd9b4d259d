import sys, copy, torch
sys.path.insert(0, '.')
from neat_amd import synth, ops
from neat_amd.train import Trainer, synthetic_batch
dev = torch.device("cuda:0")
conf = copy.deepcopy(synth.ABC_NEAT_A_MODEL_CONF)
conf.update(dbscan_enabled=True, use_median=False)
conf["global_junctions"] = dict(conf["global_junctions"], num_junctions=1024)
sd = synth.synth_state_dict(42, "rough", num_junctions=1024)
tr3 = Trainer(model_conf=conf, device=dev, state_dict={k: torch.tensor(v) for k, v in sd.items()})
tr3.model.set_precision("bf16")
_, inp3, gt3 = synthetic_batch(42, 2048, dev)
tr3.model.z_vals_override = torch.tensor(synth.synth_z_vals(42, 2048, 128)).to(dev)
orig = ops.dbscan_means
def dbg(pts, eps):
    c, v, n = orig(pts, eps)
    torch.cuda.synchronize()
    print("dbscan clusters", int(n.item()), "valid", int(v.sum()), flush=True)
    return c, v, n
ops.dbscan_means = dbg
orig_l = ops.linear_sum_assignment
def dbgl(cost, rm=None, cm=None):
    print("lsap", tuple(cost.shape), None if rm is None else int(rm.sum()), None if cm is None else int(cm.sum()), flush=True)
    torch.save({"cost": cost.detach().cpu(), "rm": None if rm is None else rm.cpu(), "cm": None if cm is None else cm.cpu()}, f"gpurun_out/lsap_case_{cost.shape[1]}.pt")
    r = orig_l(cost, rm, cm)
    torch.cuda.synchronize()
    print("  n_match", int(r[2].item()), flush=True)
    return r
ops.linear_sum_assignment = dbgl
for i in range(3):
    out, lo = tr3.step(inp3, gt3)
    torch.cuda.synchronize()
    print("step", i, float(lo["loss"].detach()), flush=True)

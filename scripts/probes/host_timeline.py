"""Where does a train step spend host time?  (no syncs inside; one sync at the end of each step)"""
import sys, time, torch
sys.path.insert(0, '.')
from neat_amd import synth, networks, ops
from neat_amd.train import Trainer, synthetic_batch
import neat_amd.networks as N
dev = torch.device('cuda:0')
prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
tr = Trainer(device=dev, state_dict={k: torch.tensor(v) for k, v in synth.synth_state_dict(42, 'rough').items()})
_, inp, gt = synthetic_batch(42, 1024, dev)
tr.model.z_vals_override = torch.tensor(synth.synth_z_vals(42, 1024, 128)).to(dev)
tr.model.set_precision(prec)
marks = []
def mark(name): marks.append((name, time.perf_counter()))
# wrap pieces
orig_render = tr.model._render
def render(*a, **k):
    mark('pre-render'); r = orig_render(*a, **k); mark('render-fwd-launched'); return r
tr.model._render = render
orig_eik = tr.model._eikonal
def eik(*a, **k):
    r = orig_eik(*a, **k); mark('eikonal-launched'); return r
tr.model._eikonal = eik
for it in range(6):
    marks.clear()
    torch.cuda.synchronize(); mark('start')
    out = tr.model(inp); mark('forward-done(host)')
    lo = tr.loss(out, gt); mark('loss-done(host)')
    tr.optimizer.zero_grad(set_to_none=True)
    lo['loss'].backward(); mark('backward-launched')
    tr.optimizer.step(); tr.scheduler.step(); mark('optimizer-launched')
    torch.cuda.synchronize(); mark('gpu-drained')
t0 = marks[0][1]
prev = t0
for n, t in marks[1:]:
    print(f"{n:28s} +{(t-prev)*1e3:7.2f} ms   (t={(t-t0)*1e3:7.2f})")
    prev = t

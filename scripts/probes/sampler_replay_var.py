"""How stable is the replayed sampler step?  Fresh Trainer + capture + 20 timed replays, several times in one process."""
import sys, time, torch
sys.path.insert(0, '.')
from neat_amd import synth
from neat_amd.train import Trainer, synthetic_batch
dev = torch.device('cuda:0')
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
sd = {k: torch.tensor(v) for k, v in synth.synth_state_dict(42, "rough").items()}
_, inp, gt = synthetic_batch(42, 1024, dev)
for trial in range(5):
    torch.manual_seed(42)
    tr = Trainer(device=dev, state_dict=sd)
    tr.model.set_precision(prec)
    tr.model.ray_sampler.sync_free = True
    for _ in range(3):
        tr.step(inp, gt)
    ok = tr.capture(inp, gt)
    ts = []
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            tr.step(inp, gt)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 20 * 1e3)
    # GPU-side time of one replay alone (events around graph.replay)
    e = next(iter(tr._graphs.values()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record(); e.graph.replay(); e1.record(); torch.cuda.synchronize()
    print(f"trial {trial}: captured {ok}, ms/step {['%.3f' % t for t in ts]}, one replay alone {e0.elapsed_time(e1):.3f} ms, rounds {tr.model.ray_sampler.rounds_taken()}", flush=True)
    del tr

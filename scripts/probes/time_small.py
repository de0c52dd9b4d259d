"""Device time of the small junction-block / optimiser launches (HIP events, 50 repetitions each)."""
import sys, torch
sys.path.insert(0, '.')
from neat_amd import ops
from neat_amd.optim import FlatAdam
dev = torch.device('cuda:0')
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for J in (64, 1024):
    lat = torch.randn(J, 256, device=dev, requires_grad=True)
    lin = [torch.nn.Linear(256, 256).to(dev), torch.nn.Linear(256, 256).to(dev), torch.nn.Linear(256, 3).to(dev)]
    def fwd(): return ops.ffn_junctions(lat, lin)
    def fb():
        y = ops.ffn_junctions(lat, lin); y.sum().backward()
    print(f"ffn J={J}: forward {timeit(fwd):.1f} us, forward+backward {timeit(fb):.1f} us (incl. torch glue)")
ps = [torch.nn.Parameter(torch.randn(n, device=dev)) for n in (256 * 256,) * 17 + (217, 257, 3, 6, 1, 256 * 39, 64 * 256)]
opt = FlatAdam(ps, lr=1e-3)
for p in opt.param_groups[0]['params']: p.grad = torch.randn_like(p)
print(f"flat adam {sum(p.numel() for p in ps)} params: {timeit(opt.step):.1f} us")
A = torch.randn(4, 4, device=dev)
print(f"inv_small 4x4: {timeit(lambda: ops.inv_small(A)):.1f} us")

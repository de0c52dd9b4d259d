"""Device time of neat_lsap at the sizes of a training step (HIP events over 200 back-to-back calls each; the first row is the
junction matching of the model forward, 8 wireframe vertices x 2048 line end points, the second the loss's matching)."""
import sys, torch
sys.path.insert(0, '.')
from neat_amd import ops
dev = torch.device('cuda:0')
def timeit(fn, n=50):
    """GPU time per call: n calls captured into ONE HIP graph (no host work between them), replayed 5 times."""
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * n) * 1e3
g = torch.Generator().manual_seed(0)
for nr, nc, masked in ((8, 2048, False), (8, 64, True), (13, 2048, False), (64, 1024, True), (300, 4096, False)):
    cost = (torch.rand(nr, nc, generator=g) * 100).to(dev)
    mask = (torch.rand(nr, generator=g) < 0.7).to(dev) if masked else None
    print(f"lsap {nr:4d} x {nc:4d}{' masked' if masked else '       '}: {timeit(lambda: ops.linear_sum_assignment(cost, mask)):7.1f} us per call (graph node)")

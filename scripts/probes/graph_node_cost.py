"""What does one dependent kernel node cost inside a replayed HIP graph?  N tiny in-place adds on a 256-element tensor, captured on one
stream, replayed; also the same N launches eagerly (queue kept full).   python scripts/probes/graph_node_cost.py"""
import time, torch
dev = torch.device("cuda:0")
x = torch.zeros(256, device=dev)
for N in (100, 1000):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            x.add_(1.0)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        for _ in range(N):
            x.add_(1.0)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(f"graph of {N} dependent tiny kernels: {1e6 * dt / N:.2f} us per node")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(200):
        x.add_(1.0)
    e0.record()
    for _ in range(N):
        x.add_(1.0)
    e1.record()
    torch.cuda.synchronize()
    print(f"eager, {N} launches back to back: {1e3 * e0.elapsed_time(e1) / N:.2f} us per launch (GPU time between events)")

# the library's own tiny kernels: the SAME kernel N times vs a cycle over several different ones (instruction fetch?)
import sys
sys.path.insert(0, '.')
from neat_amd import ops
K3 = torch.eye(3, device=dev); w2c = torch.eye(4, device=dev)[:3].contiguous(); X = torch.rand(1024, 3, device=dev) + 2.0
o = torch.zeros(1024, 3, device=dev); d = torch.nn.functional.normalize(torch.rand(1024, 3, device=dev), dim=-1); n = d.clone()
c2 = torch.rand(8, 2, device=dev); g2 = torch.rand(64, 2, device=dev)
same = [lambda: ops.project2d(K3, w2c, X)]
mixed = [lambda: ops.project2d(K3, w2c, X), lambda: ops.l3d_points(X, o, d, n), lambda: ops.junction_cost(c2, g2), lambda: X.abs(), lambda: X + 1.0, lambda: X.sign()]
for name, fns in (("one library kernel repeated", same), ("six different tiny kernels in turn", mixed)):
    with torch.no_grad():
        for f in fns:
            f()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(300):
                fns[i % len(fns)]()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            g.replay()
        torch.cuda.synchronize()
        print(f"{name}: {1e6 * (time.perf_counter() - t0) / 20 / 300:.2f} us per node")

# does a tiny node cost more after a kernel that leaves dirty lines in the L2s (end-of-kernel write-back)?
x = torch.zeros(1024, device=dev)
def timed(build, reps=20):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        build()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / reps
for mb in (1, 8, 32, 128):
    big = torch.empty(mb * 262144, device=dev); big.fill_(0.0); x.add_(1.0); torch.cuda.synchronize()
    def a():
        for _ in range(100):
            big.fill_(1.0)
    def b():
        for _ in range(100):
            big.fill_(1.0); x.add_(1.0)
    def c():
        for _ in range(100):
            big.fill_(1.0); x.add_(1.0); x.add_(1.0); x.add_(1.0)
    ta, tb, tc = timed(a), timed(b), timed(c)
    print(f"writer of {mb} MB: alone {ta / 100:.2f} us; + one tiny node {(tb - ta) / 100:.2f} us; + three tiny nodes {(tc - ta) / 100:.2f} us")

"""world_size-2 `gloo` test of the data-parallel exchange (neat_amd/dp.py) on CPU: after the bucket all-reduce every
rank holds the mean gradient, including parameters that received no gradient on one rank."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from neat_amd import dp
    r, w, _ = dp.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(7)), torch.nn.Parameter(torch.randn(()))]
    g = torch.Generator().manual_seed(100 + rank)
    params[0].grad = torch.randn(5, 3, generator=g)
    params[1].grad = torch.randn(7, generator=g) if rank == 0 else None        # missing on rank 1
    params[2].grad = torch.randn((), generator=g)
    bucket = dp.FlatGradBucket(params)
    assert bucket.numel == 15 + 7 + 1
    bucket.all_reduce_mean()
    torch.save([p.grad for p in params], os.path.join(out_dir, f"grads{rank}.pt"))
    # Several steps in which the ranks' gradient memory behaves DIFFERENTLY (rank 0 gets fresh allocations every step, rank 1 keeps
    # writing into the same tensors, and from step 2 on into the bucket's own views): every rank must still issue exactly one
    # collective of the same size per step -- the bucket's decisions may not depend on this rank's addresses (dp.py).
    q = [torch.nn.Parameter(torch.zeros(4, 2)), torch.nn.Parameter(torch.zeros(3)), torch.nn.Parameter(torch.zeros(2))]
    b2 = dp.FlatGradBucket(q)
    keep = [torch.zeros(4, 2), torch.zeros(3), torch.zeros(2)]
    calls = []
    orig = dist.all_reduce
    dist.all_reduce = lambda t, *a, **k: (calls.append(tuple(t.shape)), orig(t, *a, **k))[1]
    try:
        for step in range(4):
            vals = [torch.full(s.shape, float((rank + 1) * (step + 1) * (i + 1))) for i, s in enumerate(keep)]
            for i, prm in enumerate(q):
                if rank == 0:
                    prm.grad = vals[i].clone()                      # fresh allocation: new address every step
                elif step < 2:
                    keep[i].copy_(vals[i]); prm.grad = keep[i]       # same tensors every step
                else:
                    prm.grad.copy_(vals[i])                          # in place, into the views the bucket installed
            if step == 3 and rank == 1:
                q[1].grad = None                                     # a parameter without gradient on one rank only
            b2.all_reduce_mean()
            for i, prm in enumerate(q):
                expect = 1.5 * (step + 1) * (i + 1)
                if step == 3 and i == 1:
                    expect = 0.5 * 1 * (step + 1) * (i + 1)         # rank 1 contributed zeros
                assert torch.allclose(prm.grad, torch.full(keep[i].shape, expect)), (rank, step, i, prm.grad, expect)
    finally:
        dist.all_reduce = orig
    assert calls == [(13,)] * 4, calls                               # one flat all-reduce of all 8 + 3 + 2 elements per step, on every rank
    scal = dp.all_reduce_scalars({"loss": torch.tensor(float(rank + 1))})
    assert abs(float(scal["loss"]) - 1.5) < 1e-6
    assert dp.shard_rays(4096, world) == 2048 and dp.rank_seed(42, rank) == 42 + rank
    dist.destroy_process_group()


def test_flat_bucket_all_reduce_mean(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    g0 = torch.load(tmp_path / "grads0.pt")
    g1 = torch.load(tmp_path / "grads1.pt")
    gens = [torch.Generator().manual_seed(100 + r) for r in range(2)]
    a = [torch.randn(5, 3, generator=gens[0]), torch.randn(7, generator=gens[0]), torch.randn((), generator=gens[0])]
    b0 = torch.randn(5, 3, generator=gens[1])
    b2 = torch.randn((), generator=gens[1])
    expect = [(a[0] + b0) / 2, a[1] / 2, (a[2] + b2) / 2]
    for e, x, y in zip(expect, g0, g1):
        assert torch.allclose(x, e, atol=1e-7) and torch.allclose(y, e, atol=1e-7)


def test_single_process_is_a_no_op():
    from neat_amd import dp
    p = torch.nn.Parameter(torch.ones(3))
    p.grad = torch.full((3,), 2.0)
    dp.FlatGradBucket([p]).all_reduce_mean()
    assert torch.equal(p.grad, torch.full((3,), 2.0))

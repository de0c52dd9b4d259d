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
    # gradients that are views of one flat allocation (what the HIP backward returns) are reduced in place
    q = [torch.nn.Parameter(torch.zeros(4, 2)), torch.nn.Parameter(torch.zeros(3)), torch.nn.Parameter(torch.zeros(2))]
    flat = torch.arange(11, dtype=torch.float32) * (rank + 1)
    q[0].grad, q[1].grad = flat[:8].view(4, 2), flat[8:11]
    q[2].grad = torch.full((2,), float(rank))
    b2 = dp.FlatGradBucket(q)
    b2.all_reduce_mean()
    assert b2._plan == [(0, 1)]
    assert torch.allclose(flat, torch.arange(11, dtype=torch.float32) * 1.5) and q[0].grad.data_ptr() == flat.data_ptr()
    assert torch.allclose(q[2].grad, torch.full((2,), 0.5))
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

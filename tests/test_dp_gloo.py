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


def _grad_worker(rank, world, port, out_dir):
    """C4 arithmetic: each rank differentiates the mean-reduced losses of ITS rays (oracle on CPU = the reference's math), the bucket
    averages; must equal the gradient of the same losses over the concatenated batch."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from neat_amd import dp, synth
    from oracle import neat_oracle as O
    dp.init_from_env(backend="gloo")
    R, S = 16, 24
    p, full = _dp_problem(R, S)
    sl = slice(rank * R // world, (rank + 1) * R // world)
    loss = _dp_loss(O, p, full, sl)
    loss.backward()
    params = list(p.values())
    for q in params:
        q.requires_grad_(True)
    bucket = dp.FlatGradBucket(params)
    bucket.all_reduce_mean()
    torch.save({k: v.grad.clone() for k, v in p.items() if v.grad is not None}, os.path.join(out_dir, f"dpgrads{rank}.pt"))
    dist.destroy_process_group()


def _dp_problem(R, S):
    from neat_amd import synth
    from oracle import neat_oracle as O
    sd = synth.synth_state_dict(5, "rough")
    p = O.params_from_numpy(sd, requires_grad=True)
    sc = synth.synth_scene(seed=5, n_rays=R, view=0)
    z = torch.tensor(synth.synth_z_vals(5, R, S))
    gen = torch.Generator().manual_seed(5)
    eik = torch.empty(R, 3).uniform_(-3, 3, generator=gen)
    return p, (sc, z, eik)


def _dp_loss(O, p, full, sl):
    """rgb L1 (mean over rays) + 0.1 eikonal (mean over points): the ray-decomposable part of VolSDFLoss (loss_wfr.py:60-75)."""
    sc, z, eik = full
    T = torch.tensor
    dirs, origin = O.camera_rays(T(sc["uv"])[:, sl], T(sc["pose"]), T(sc["intrinsics"]))
    dirs = dirs.reshape(-1, 3)
    o = origin[:, None, :].expand(1, dirs.shape[0], 3).reshape(-1, 3)
    out = O.render_rays(p, o, dirs, z[sl])
    rgb_l = (out["rgb_values"] - T(sc["gt_rgb"])[0, sl]).abs().mean()
    gth = O.sdf_gradient(p, eik[sl])
    return rgb_l + 0.1 * ((gth.norm(2, dim=1) - 1) ** 2).mean()


def test_two_rank_average_equals_full_batch_gradient(tmp_path):
    from oracle import neat_oracle as O
    world, port = 2, _free_port()
    mp.spawn(_grad_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    g0 = torch.load(tmp_path / "dpgrads0.pt")
    g1 = torch.load(tmp_path / "dpgrads1.pt")
    p, full = _dp_problem(16, 24)
    _dp_loss(O, p, full, slice(0, 16)).backward()
    checked = 0
    for k, v in p.items():
        if v.grad is None:
            continue
        scale = float(v.grad.abs().max()) + 1e-12
        assert torch.allclose(g0[k], g1[k]), k                                  # both ranks hold the same averaged gradient
        assert float((g0[k] - v.grad).abs().max()) <= 2e-5 * scale + 1e-9, k      # = gradient of the full-batch mean losses
        checked += 1
    assert checked == 43          # 27 SDF + 15 rendering-head tensors + density.beta (the attraction head and the junction MLP see neither loss)

"""device-side linear_sum_assignment (neat_lsap through the C ABI) against scipy, which is what the
reference calls on the host (neat_wfr_rend_a.py:473, loss_wfr.py:108).  scipy is the oracle here."""
import numpy as np
import pytest
import torch
from scipy.optimize import linear_sum_assignment as scipy_lsa


def _run(cost, mask=None):
    from neat_amd import ops
    c = torch.tensor(cost, dtype=torch.float32).cuda()
    m = None if mask is None else torch.tensor(mask).cuda()
    r, c_, n = ops.linear_sum_assignment(c, m)
    n = int(n.item())
    r, c_ = r.cpu().numpy(), c_.cpu().numpy()
    assert (r[n:] == -1).all() and (c_[n:] == -1).all()
    return r[:n], c_[:n], n


@pytest.mark.gpu
@pytest.mark.parametrize("nr,nc,seed", [(8, 2048, 0), (1, 1, 1), (1, 7, 2), (7, 1, 3), (64, 64, 4), (100, 37, 5),
                                        (37, 100, 6), (300, 4096, 7), (1500, 90, 8)])
def test_matches_scipy(nr, nc, seed):
    rng = np.random.default_rng(seed)
    cost = rng.uniform(0, 100, (nr, nc)).astype(np.float32)
    r, c, n = _run(cost)
    sr, sc = scipy_lsa(cost)
    assert n == len(sr)
    assert (r == sr).all() and (c == sc).all()


@pytest.mark.gpu
@pytest.mark.parametrize("nr,nc,seed", [(12, 64, 0), (200, 64, 1), (64, 64, 2), (5, 3, 3)])
def test_row_mask_is_compaction(nr, nc, seed):
    """Masked rows behave like the reference's `[good]` compaction followed by scipy."""
    rng = np.random.default_rng(seed)
    cost = rng.uniform(0, 10, (nr, nc)).astype(np.float32)
    mask = rng.uniform(size=nr) < 0.6
    r, c, n = _run(cost, mask)
    keep = np.nonzero(mask)[0]
    sr, sc = scipy_lsa(cost[keep])
    assert n == len(sr)
    assert (r == keep[sr]).all() and (c == sc).all()
    r0, c0, n0 = _run(cost, np.zeros(nr, bool))
    assert n0 == 0


@pytest.mark.gpu
def test_ties_same_choice_and_cost():
    """Integer costs with many ties: same total cost always; and the same assignment, since the tie rule is scipy's."""
    rng = np.random.default_rng(11)
    for nr, nc in [(20, 20), (10, 50), (50, 10)]:
        cost = rng.integers(0, 4, (nr, nc)).astype(np.float32)
        r, c, n = _run(cost)
        sr, sc = scipy_lsa(cost)
        assert cost[r, c].sum() == cost[sr, sc].sum()
        assert len(set(c.tolist())) == n and len(set(r.tolist())) == n
        assert (r == sr).all() and (c == sc).all()


@pytest.mark.gpu
def test_non_finite_cost_is_reported():
    cost = np.full((3, 3), np.inf, np.float32)
    _, _, n = _run(cost)
    assert n == -1


@pytest.mark.gpu
def test_lazy_outputs_equal_compaction():
    from neat_amd.networks import JunctionOutputs
    good = torch.tensor([True, False, True]).cuda()
    pad = {"j3d_local": torch.arange(9.0).reshape(3, 3).cuda(), "j2d_local": torch.zeros(3, 2).cuda(),
           "j2d_local_calib": torch.ones(3, 2).cuda()}
    out = JunctionOutputs({"rgb_values": 1}, good, pad)
    assert "j3d_local" in out and "nope" not in out
    assert out["j3d_local"].shape == (2, 3) and out["j3d_local"][1, 0] == 6
    assert set(out.keys()) >= {"rgb_values", "j3d_local", "j2d_local", "j2d_local_calib"}


@pytest.mark.gpu
@pytest.mark.parametrize("nr,nc,seed", [(8, 200, 0), (60, 40, 1), (30, 30, 2)])
def test_col_mask_is_compaction(nr, nc, seed):
    """Masked columns behave like deleting them before scipy (padded candidate sets, e.g. the DBSCAN centres)."""
    from neat_amd import ops
    rng = np.random.default_rng(seed)
    cost = rng.uniform(0, 10, (nr, nc)).astype(np.float32)
    cmask = rng.uniform(size=nc) < 0.5
    rmask = rng.uniform(size=nr) < 0.8
    r, c, n = ops.linear_sum_assignment(torch.tensor(cost).cuda(), torch.tensor(rmask).cuda(), torch.tensor(cmask).cuda())
    n = int(n.item())
    r, c = r.cpu().numpy(), c.cpu().numpy()
    keep_r, keep_c = np.nonzero(rmask)[0], np.nonzero(cmask)[0]
    sr, sc = scipy_lsa(cost[np.ix_(keep_r, keep_c)])
    assert n == len(sr) and (r[n:] == -1).all()
    assert (r[:n] == keep_r[sr]).all() and (c[:n] == keep_c[sc]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("n,seed", [(64, 0), (2048, 1), (4096, 2), (8192, 3)])
def test_dbscan_means_vs_sklearn(n, seed):
    """neat_dbscan_means against sklearn.cluster.DBSCAN(eps=0.01, min_samples=2) + per-cluster means (what the reference
    computes on the host, rend_a :328-339): same clusters in the same order, same centres."""
    from sklearn.cluster import DBSCAN
    from neat_amd import ops
    rng = np.random.default_rng(seed)
    centres = rng.uniform(-1, 1, (max(n // 16, 2), 3))
    pts = centres[rng.integers(0, len(centres), n)] + rng.normal(0, 0.002, (n, 3))
    pts[: n // 8] = rng.uniform(-1, 1, (n // 8, 3))                     # isolated points = noise
    t = np.linspace(0, 1, n // 8)[:, None]
    pts[n // 8: n // 4] = np.array([0.5, -0.5, 0.2]) + t * np.array([0.6, 0.1, -0.3])      # a long chain: one cluster through many hops
    pts = pts.astype(np.float32)
    labels = DBSCAN(eps=0.01, min_samples=2).fit(pts).labels_
    ref = np.array([pts[labels == i].mean(axis=0) for i in range(labels.max() + 1)]).reshape(-1, 3)
    got, valid, count = ops.dbscan_means(torch.tensor(pts).cuda(), 0.01)
    k = int(count.item())
    assert k == len(ref) and int(valid.sum()) == k and bool(valid[:k].all())
    assert np.abs(got[:k].cpu().numpy() - ref).max() <= 1e-6
    assert float(got[k:].abs().max()) == 0.0 if k < got.shape[0] else True


@pytest.mark.gpu
def test_few_rows_many_columns_repeatedly():
    """Regression: the count returned by the in-kernel compaction used to be read after a fast thread could reset it
    (hang / fault with 1-2 participating rows, timing dependent).  Many launches of the shapes a training step produces."""
    from neat_amd import ops
    rng = np.random.default_rng(5)
    for it in range(60):
        cost = rng.uniform(0, 50, (8, 1024)).astype(np.float32)
        mask = np.zeros(8, bool)
        mask[rng.choice(8, 1 + it % 2, replace=False)] = True
        r, c, n = ops.linear_sum_assignment(torch.tensor(cost).cuda(), torch.tensor(mask).cuda())
        n = int(n.item())
        keep = np.nonzero(mask)[0]
        sr, sc = scipy_lsa(cost[keep])
        assert n == len(sr) and (r[:n].cpu().numpy() == keep[sr]).all() and (c[:n].cpu().numpy() == sc).all()


@pytest.mark.gpu
@pytest.mark.parametrize("nr,nc,seed", [(8, 2048, 21), (13, 2048, 22), (16, 1024, 23), (2, 64, 24), (9, 4096, 25), (8, 255, 26)])
def test_few_rows_many_columns_hard_cases(nr, nc, seed):
    """The shape of the training step's junction matching (a handful of wireframe vertices against thousands of line end points) with
    what a column-pruning search would get wrong: costs full of ties, every column present twice, non-finite entries, masks on both
    sides.  Same assignments as scipy.  (Round 5 tried such a search -- one wave over the union of the rows' n cheapest columns -- and
    dropped it: NOTEBOOK.)"""
    rng = np.random.default_rng(seed)
    cases = [rng.uniform(0, 100, (nr, nc)).astype(np.float32),
             rng.integers(0, 5, (nr, nc)).astype(np.float32),                       # ties everywhere
             np.round(rng.uniform(0, 100, (nr, nc)), 0).astype(np.float32)]        # some ties
    dup = rng.uniform(0, 100, (nr, nc)).astype(np.float32)
    dup[:, nc // 2:] = dup[:, :nc - nc // 2]                                        # every column twice: ties at every boundary
    cases.append(dup)
    bad = rng.uniform(0, 100, (nr, nc)).astype(np.float32)
    bad[rng.uniform(size=(nr, nc)) < 0.3] = np.inf
    cases.append(bad)
    for cost in cases:
        sr, sc = scipy_lsa(cost)
        r, c, n = _run(cost)
        assert n == len(sr) and (r == sr).all() and (c == sc).all(), (nr, nc)
    # masks on both sides
    from neat_amd import ops
    cost = rng.uniform(0, 100, (nr, nc)).astype(np.float32)
    rmask, cmask = rng.uniform(size=nr) < 0.8, rng.uniform(size=nc) < 0.7
    rmask[:2] = True
    ct = torch.tensor(cost).cuda()
    r, c, n = ops.linear_sum_assignment(ct, torch.tensor(rmask).cuda(), torch.tensor(cmask).cuda())
    n = int(n)
    kr, kc = np.nonzero(rmask)[0], np.nonzero(cmask)[0]
    sr, sc = scipy_lsa(cost[kr][:, kc])
    assert n == len(sr) and (r.cpu().numpy()[:n] == kr[sr]).all() and (c.cpu().numpy()[:n] == kc[sc]).all()

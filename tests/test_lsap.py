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

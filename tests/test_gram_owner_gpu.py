"""Row sets of a call that share partial Grams (csrc/ws_kernels.hpp: ws_block_owner_kernel).

The lanes of a CV grid bring their fold's row mask (reference: src/sparselm/model_selection.py:273, 304-323 hands every
(candidate, fold) fit to scikit-learn's splitters -- KFold's test folds are contiguous row ranges).  On a row block where a
mask is all ones the partial Gram of the working set's columns is the same matrix for every such mask, on its fold's test
rows it is nothing: the engine computes each block once and sums per set.  Held here against the route where every set
multiplies every block itself (SLM_NO_GRAM_OWNER=1: the sums have the same terms in the same order, so the coefficients
are bit-identical), and against the oracle on the training rows (model/_lasso.py:230-275).
"""

import numpy as np
import pytest

import oracle
from sparselm_amd import _engine

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    return _engine.get_engine(0)


def rel_inf(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def problem(seed, n, p, gs):
    rng = np.random.default_rng(seed)
    G = p // gs
    X = rng.standard_normal((n, p))
    groups = rng.permutation(np.repeat(np.arange(G), gs)).astype(np.int32)
    beta = np.zeros(p)
    for g in rng.choice(G, 6, replace=False):
        beta[groups == g] = rng.uniform(1.0, 3.0, gs) * rng.choice([-1.0, 1.0], gs)
    y = X @ beta + 2.0 * rng.standard_normal(n)
    return X, y, groups, G


def test_fold_masks_share_row_blocks_bit_identically(eng, monkeypatch):
    n, p, gs = 40_000, 800, 8
    X, y, groups, G = problem(5, n, p, gs)
    gidx, Gn = oracle.group_index(groups, p)
    g0 = X.T @ y / n
    bmax = float(np.max(np.sqrt(np.bincount(groups, weights=g0 * g0, minlength=G))))
    pts = [(0.2 * a, 0.8 * a, 0.0) for a in np.geomspace(0.9 * bmax, 0.05 * bmax, 6)]
    rng = np.random.default_rng(6)
    contiguous = np.arange(n) * 5 // n  # KFold(5): test folds are row ranges
    masks = [(contiguous != f).astype(float) for f in range(5)]
    masks.append((rng.permutation(n) % 5 != 0).astype(float))  # a shuffled fold: mixed on every block
    masks.append(np.where(contiguous == 2, 0.0, rng.uniform(0.5, 1.5, n)))  # general weights, zero on a row range
    masks.append(np.ones(n))  # no rows left out, as weights
    specs = [dict(points=pts, row_weight=m, n_eff=int(np.count_nonzero(m))) for m in masks for _ in range(2)]
    flags = _engine.FLAG_WORKING_SET | _engine.FLAG_FRESH_L
    with eng.dataset(X, y) as ds:
        ds.set_groups(gidx, Gn)
        shared = ds.solve_lanes(specs, tol=1e-10, flags=flags)
        monkeypatch.setenv("SLM_NO_GRAM_OWNER", "1")
        own = ds.solve_lanes(specs, tol=1e-10, flags=flags)
        monkeypatch.delenv("SLM_NO_GRAM_OWNER")
    assert all(r.converged for r in shared) and all(r.converged for r in own)
    assert shared[0].ws_builds >= 1  # (the working set served the call: its Grams are what is shared)
    for a, b in zip(shared, own):
        assert np.array_equal(a.betas, b.betas)
    # and the oracle on the rows a 0/1 mask keeps: two contiguous folds, the shuffled one
    for l in (0, 4, 10):
        m = masks[l // 2]
        tr = m > 0
        b = None
        for k, (sa, sb, _) in enumerate(pts):
            b, _ = oracle.fista(X[tr], y[tr], sa, sb, 0.0, gidx, Gn, beta0=b, tol=1e-13)
            assert rel_inf(shared[l].betas[k], b) < 1e-6, (l, k)


def test_masks_that_are_zero_everywhere_but_a_block_and_single_set(eng, monkeypatch):
    """Edge cases of the owner table: ONE row set with weights (nobody to share with; its test rows' blocks are skipped), and a
    mask that keeps a single row range (every other block all zeros)."""
    n, p, gs = 24_000, 400, 4
    X, y, groups, G = problem(8, n, p, gs)
    gidx, Gn = oracle.group_index(groups, p)
    g0 = X.T @ y / n
    bmax = float(np.max(np.sqrt(np.bincount(groups, weights=g0 * g0, minlength=G))))
    pts = [(0.0, a, 0.0) for a in np.geomspace(0.8 * bmax, 0.1 * bmax, 4)]
    keep = np.zeros(n)
    keep[5_000:17_000] = 1.0
    flags = _engine.FLAG_WORKING_SET | _engine.FLAG_FRESH_L
    for masks in ([keep], [keep, 1.0 - keep], [(np.arange(n) >= n // 3).astype(float)]):
        specs = [dict(points=pts, row_weight=m, n_eff=int(m.sum())) for m in masks]
        with eng.dataset(X, y) as ds:
            ds.set_groups(gidx, Gn)
            shared = ds.solve_lanes(specs, tol=1e-10, flags=flags)
            monkeypatch.setenv("SLM_NO_GRAM_OWNER", "1")
            own = ds.solve_lanes(specs, tol=1e-10, flags=flags)
            monkeypatch.delenv("SLM_NO_GRAM_OWNER")
        for m, a, b in zip(masks, shared, own):
            assert a.converged and b.converged
            assert np.array_equal(a.betas, b.betas)
            tr = m > 0
            ref, _ = oracle.fista(X[tr], y[tr], 0.0, pts[-1][1], 0.0, gidx, Gn, beta0=a.betas[-1], tol=1e-13)
            assert rel_inf(a.betas[-1], ref) < 1e-6

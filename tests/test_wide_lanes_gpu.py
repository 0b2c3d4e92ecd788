"""Thirty-two lanes per read of X (round 5): a call of up to SLM_MAX_LANES = 32 lanes is two halves of sixteen -- the width of
an MFMA operand -- whose residuals sit in two planes of R and whose X^T R is ONE read of X (`xtr32_mfma_kernel`); the residual
kernels, the working set's scoring and the model-Gram rounds run per half.  Parity: the same problems on thirty-two lanes, on
sixteen and on one plain lane; independent lanes with fold masks; a shared path; the rounds on the model Gram forced on
(SLM_MG=2).  Reference: the dispatch these lanes replace is joblib's, one fit per task
(/root/reference/src/sparselm/model_selection.py:273,304-323)."""

import numpy as np
import pytest

import oracle
from sparselm_amd import _engine

pytestmark = pytest.mark.gpu
WS, PLAIN = _engine.FLAG_WORKING_SET, _engine.FLAG_NO_WORKING_SET


@pytest.fixture(scope="module")
def eng():
    return _engine.get_engine(0)


def _problem(rng, n, p, k, noise):
    X = rng.standard_normal((n, p))
    bt = np.zeros(p)
    bt[rng.choice(p, k, replace=False)] = 5 * rng.uniform(0.2, 1.0, k) * rng.choice([-1, 1], k)
    return X, X @ bt + noise * rng.standard_normal(n)


def test_max_lanes_tells_where_thirty_two_are_served(eng):
    rng = np.random.default_rng(0)
    X, y = _problem(rng, 900, 300, 10, 1.0)
    with eng.dataset(X, y) as ds:
        assert ds.max_lanes(WS) == _engine.MAX_LANES_WIDE == 32  # working-set solves on the split pass
        assert ds.max_lanes(WS | _engine.FLAG_COVARIANCE) == 32  # covariance passes as well: the Gram product a launch per half
        assert ds.max_lanes(PLAIN) <= 16
        pts = [(0.1, 0.0, 0.0)]
        with pytest.raises(ValueError):
            ds.solve_lanes([dict(points=pts)] * 33, flags=WS)
        with pytest.raises(NotImplementedError):  # thirty-two lanes without the working set: no kernel serves them
            ds.solve_lanes([dict(points=pts)] * 20, flags=PLAIN)


@pytest.mark.parametrize("grouped", [False, True])
def test_thirty_two_independent_lanes_with_fold_masks(eng, grouped):
    """The cells of a grid: 4 folds x 8 paths of a dozen points, lanes 17..32 in the second half -- against sixteen at a time and
    one plain lane each, and (two lanes of the second half) against oracle.fista on the fold's rows."""
    rng = np.random.default_rng(3 + grouped)
    n, p = 2600, 480
    X, y = _problem(rng, n, p, 20, 2.0)
    fold = rng.integers(0, 4, n)
    masks = [(fold != f).astype(float) for f in range(4)]
    gid = rng.permutation(np.arange(p) % (p // 8)).astype(np.int32) if grouped else None
    G = p // 8 if grouped else p
    with eng.dataset(X, y) as ds:
        if grouped:
            ds.set_groups(gid, G)
        specs = []
        for f in range(4):
            m = masks[f]
            c = X.T @ (m * y) / m.sum()
            top = float(np.max(np.sqrt(np.bincount(gid, weights=c * c, minlength=G)))) if grouped else float(np.max(np.abs(c)))
            for r in range(8):
                al = np.geomspace(top, (0.02 + 0.01 * r) * top, 12)
                pts = np.c_[0.3 * al, 0.7 * al, 0 * al] if grouped else np.c_[al, 0 * al, 0 * al]
                specs.append(dict(points=pts, row_weight=m, n_eff=int(m.sum())))
        wide = ds.solve_lanes(specs, tol=1e-10, flags=WS)
        again = ds.solve_lanes(specs, tol=1e-10, flags=WS)
        half = ds.solve_lanes(specs[:16], tol=1e-10, flags=WS) + ds.solve_lanes(specs[16:], tol=1e-10, flags=WS)
        plain = [ds.solve_lanes([specs[l]], tol=1e-10, flags=PLAIN)[0] for l in (0, 15, 16, 23, 31)]
    assert len(wide) == 32 and all(r.converged for r in wide + half + plain)
    assert wide[0].grad_launches < half[0].grad_launches + half[16].grad_launches  # (one read of X for both halves)
    for a, b, c in zip(wide, half, again):
        assert np.max(np.abs(a.betas - b.betas)) < 1e-7 * np.max(np.abs(b.betas))
        assert np.array_equal(a.betas, c.betas)
    for l, q in zip((0, 15, 16, 23, 31), plain):
        assert np.max(np.abs(wide[l].betas - q.betas)) < 1e-7 * np.max(np.abs(q.betas))
    gidx, Gn = oracle.group_index(gid, p)
    for l in (17, 30):
        m = masks[l // 8]
        sa, sb, sd = specs[l]["points"][-1]
        bo, _ = oracle.fista(X[m > 0], y[m > 0], sa, sb, sd, gidx, Gn, beta0=wide[l].betas[-1], tol=1e-13)
        assert np.max(np.abs(wide[l].betas[-1] - bo)) < 1e-6 * np.max(np.abs(bo))


@pytest.mark.parametrize("grouped", [False, True])
def test_thirty_two_lanes_on_covariance_passes(eng, grouped):
    """The cells of a grid from the folds' Grams (SLM_FLAG_COVARIANCE), thirty-two to a call: every pass multiplies each fold's
    Gram by both halves' points (a launch per half).  Against sixteen at a time, against the same lanes over X, against
    oracle.fista on a fold's rows; lanes that finish early (short paths in the second half) leave their half idle."""
    rng = np.random.default_rng(21 + grouped)
    n, p = 2400, 420
    X, y = _problem(rng, n, p, 18, 2.0)
    fold = rng.integers(0, 4, n)
    masks = [(fold != f).astype(float) for f in range(4)]
    gid = rng.permutation(np.arange(p) % (p // 6)).astype(np.int32) if grouped else None
    G = p // 6 if grouped else p
    COV = _engine.FLAG_COVARIANCE
    with eng.dataset(X, y) as ds:
        if grouped:
            ds.set_groups(gid, G)
        ds.covariance_folds(masks, [int(m.sum()) for m in masks])
        specs = []
        for f in range(4):
            m = masks[f]
            c = X.T @ (m * y) / m.sum()
            top = float(np.max(np.sqrt(np.bincount(gid, weights=c * c, minlength=G)))) if grouped else float(np.max(np.abs(c)))
            for r in range(8):
                al = np.geomspace(top, (0.02 + 0.01 * r) * top, 12 if f < 2 else 3 + r)  # (the second half: shorter paths, done early)
                pts = np.c_[0.3 * al, 0.7 * al, 0 * al] if grouped else np.c_[al, 0 * al, 0 * al]
                specs.append(dict(points=pts, row_weight=m, n_eff=int(m.sum())))
        wide = ds.solve_lanes(specs, tol=1e-10, flags=WS | COV)
        again = ds.solve_lanes(specs, tol=1e-10, flags=WS | COV)
        half = ds.solve_lanes(specs[:16], tol=1e-10, flags=WS | COV) + ds.solve_lanes(specs[16:], tol=1e-10, flags=WS | COV)
        over_x = ds.solve_lanes(specs, tol=1e-10, flags=WS)
    assert len(wide) == 32 and all(r.converged for r in wide + half + over_x)
    for a, b, c, d in zip(wide, half, again, over_x):
        scale = np.max(np.abs(d.betas))
        assert np.max(np.abs(a.betas - b.betas)) < 1e-7 * scale
        assert np.max(np.abs(a.betas - d.betas)) < 1e-7 * scale
        assert np.array_equal(a.betas, c.betas)
    gidx, Gn = oracle.group_index(gid, p)
    for l in (5, 19, 31):
        m = masks[l // 8]
        sa, sb, sd = specs[l]["points"][-1]
        bo, _ = oracle.fista(X[m > 0], y[m > 0], sa, sb, sd, gidx, Gn, beta0=wide[l].betas[-1], tol=1e-13)
        assert np.max(np.abs(wide[l].betas[-1] - bo)) < 1e-6 * np.max(np.abs(bo))


def test_a_shared_path_on_thirty_two_lanes(eng):
    rng = np.random.default_rng(9)
    n, p = 3000, 520
    X, y = _problem(rng, n, p, 25, 3.0)
    with eng.dataset(X, y) as ds:
        g0, _ = ds.gradient(None)
        amax = float(np.max(np.abs(g0)))
        pts = [(a, 0.0, 0.0) for a in np.geomspace(amax, 5e-3 * amax, 70)]
        wide = ds.solve_path(pts, lanes=32, flags=WS, tol=1e-10)
        half = ds.solve_path(pts, lanes=16, flags=WS, tol=1e-10)
        one = ds.solve_path(pts, lanes=1, flags=PLAIN, tol=1e-10)
    assert wide.converged and half.converged and one.converged
    assert np.max(np.abs(wide.betas - one.betas)) < 1e-7 * np.max(np.abs(one.betas))
    assert np.max(np.abs(half.betas - one.betas)) < 1e-7 * np.max(np.abs(one.betas))


def test_model_gram_rounds_on_thirty_two_lanes(eng, monkeypatch):
    monkeypatch.setenv("SLM_MG", "2")  # (rounds at any size)
    rng = np.random.default_rng(13)
    n, p = 4200, 700
    X, y = _problem(rng, n, p, 30, 60.0)
    with eng.dataset(X, y) as ds:
        g0, _ = ds.gradient(None)
        amax = float(np.max(np.abs(g0)))
        pts = [(a, 0.0, 0.0) for a in np.geomspace(amax, 1e-3 * amax, 64)]
        wide = ds.solve_path(pts, lanes=32, flags=WS, tol=1e-9)
        no_rounds = ds.solve_path(pts, lanes=32, flags=WS | _engine.FLAG_NO_MODEL_GRAM, tol=1e-9)
        half = ds.solve_path(pts, lanes=16, flags=WS, tol=1e-9)
    assert wide.converged and no_rounds.converged and half.converged
    assert wide.mg_rounds > 0 and no_rounds.mg_rounds == 0 and np.count_nonzero(wide.betas[-1]) > 512
    assert wide.grad_launches < no_rounds.grad_launches
    assert np.max(np.abs(wide.betas - no_rounds.betas)) < 1e-6 * np.max(np.abs(no_rounds.betas))
    assert np.max(np.abs(wide.betas - half.betas)) < 1e-6 * np.max(np.abs(half.betas))

"""Working-set (Gram-assisted) refinement, sparse-lm_amd/csrc/ws_kernels.hpp.

The refinement only moves the point the next pass over X evaluates; every reported solution is
still verified by a true gradient under the unchanged stopping rule.  So with it (FLAG_WORKING_SET)
and without it (FLAG_NO_WORKING_SET) the engine must agree with the oracle and with itself to the
solver tolerance, on every penalty of the family, while spending fewer passes over X.
"""

import os

import numpy as np
import pytest

import oracle
from sparselm_amd import _engine

pytestmark = pytest.mark.gpu
WS, NO_WS = _engine.FLAG_WORKING_SET, _engine.FLAG_NO_WORKING_SET


@pytest.fixture(scope="module")
def eng():
    return _engine.get_engine(0)


def rel_inf(a, b):
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


def ofista(X, y, a, b=None, d=None, groups=None, G=None):
    """Oracle minimiser at tight tolerance (singleton groups when none are given)."""
    p = X.shape[1]
    if groups is None:
        groups, G = np.arange(p), p
    b = np.zeros(G) if b is None else b
    d = np.zeros(G) if d is None else d
    return oracle.fista(X, y, a, b, d, np.asarray(groups), G, tol=1e-13, max_iter=200000)[0]


def problem(n, p, n_inf, seed, groups=None, noise=1.0):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, p))
    beta = np.zeros(p)
    if groups is None:
        beta[rng.choice(p, n_inf, replace=False)] = rng.uniform(1, 5, n_inf) * rng.choice([-1, 1], n_inf)
    else:
        for g in rng.choice(groups.max() + 1, n_inf, replace=False):
            beta[groups == g] = rng.uniform(1, 3, int(np.sum(groups == g))) * rng.choice([-1, 1])
    y = X @ beta + noise * rng.standard_normal(n)
    return X, y


def alpha_path(X, y, k=12, lo=1e-2, groups=None, G=None):
    c = X.T @ y / X.shape[0]
    amax = np.max(np.abs(c)) if groups is None else np.max(np.sqrt(np.bincount(groups, weights=c * c, minlength=G)))
    return np.geomspace(amax, lo * amax, k)


@pytest.mark.parametrize("n,p,lanes", [(3000, 700, 1), (3000, 700, 4), (2500, 333, 3), (900, 61, 2), (400, 9, 1)])
def test_lasso_path_with_and_without_refinement(eng, n, p, lanes):
    X, y = problem(n, p, min(12, p // 2), seed=n + p)
    alphas = alpha_path(X, y)
    pts = [(a, 0, 0) for a in alphas]
    with eng.dataset(X, y) as ds:
        r1 = ds.solve_path(pts, tol=1e-11, lanes=lanes, flags=WS)
        r0 = ds.solve_path(pts, tol=1e-11, lanes=lanes, flags=NO_WS)
    assert r1.converged and r0.converged
    assert r1.ws_builds >= 1 and r1.ws_refined > 0 and r0.ws_builds == 0
    assert rel_inf(r1.betas, r0.betas) < 1e-8
    for k in (3, 7, 11):
        ref = ofista(X, y, alphas[k] * np.ones(p))
        assert rel_inf(r1.betas[k], ref) < 1e-8
    assert r1.grad_launches < r0.grad_launches  # that is the point


def test_refinement_takes_about_one_pass_per_point(eng):
    X, y = problem(6000, 900, 15, seed=1)
    alphas = alpha_path(X, y, k=24, lo=1e-2)
    pts = [(a, 0, 0) for a in alphas]
    with eng.dataset(X, y) as ds:
        r1 = ds.solve_path(pts, tol=1e-8, flags=WS)
        r0 = ds.solve_path(pts, tol=1e-8, flags=NO_WS)
    assert r1.converged
    assert r1.grad_launches <= 2 * len(alphas) + 4 * max(1, r1.ws_builds)
    assert r1.grad_launches < 0.7 * r0.grad_launches
    assert rel_inf(r1.betas, r0.betas) < 1e-6


@pytest.mark.parametrize("kind", ["group", "sparse_group", "ridged", "shuffled_sgl"])
def test_group_penalties_match_oracle_with_refinement(eng, kind):
    n, G, size = 2500, 40, 6
    p = G * size
    groups = np.repeat(np.arange(G), size)
    if kind == "shuffled_sgl":
        groups = np.random.default_rng(5).permutation(groups)
    X, y = problem(n, p, 5, seed=11, groups=groups)
    w = np.random.default_rng(2).uniform(0.5, 2.0, G)
    alphas = alpha_path(X, y, k=8, lo=3e-2, groups=groups, G=G)
    if kind == "group":
        sa, sb, sd = 0.0, 1.0, 0.0
    elif kind in ("sparse_group", "shuffled_sgl"):
        sa, sb, sd = 0.3, 0.7, 0.0
    else:
        sa, sb, sd = 0.0, 1.0, 1.0
    pts = [(sa * a, sb * a, sd * 0.2) for a in alphas]
    with eng.dataset(X, y) as ds:
        ds.set_groups(groups, G)
        r1 = ds.solve_path(pts, b=w, tol=1e-11, flags=WS, want_group_norms=True)
        r0 = ds.solve_path(pts, b=w, tol=1e-11, flags=NO_WS, want_group_norms=True)
    assert r1.converged and r1.ws_refined > 0
    assert rel_inf(r1.betas, r0.betas) < 1e-8
    np.testing.assert_allclose(r1.group_norms, r0.group_norms, rtol=0, atol=1e-8 * np.max(r0.group_norms))
    for k in (2, 7):
        ref = ofista(X, y, pts[k][0] * np.ones(p), pts[k][1] * w, pts[k][2] * np.ones(G), groups, G)
        assert rel_inf(r1.betas[k], ref) < 1e-8


def test_lanes_with_their_own_row_masks_get_their_own_gram(eng):
    # CV folds as lanes: each lane's Gram is X^T diag(mask) X / n_train
    n, p = 3000, 300
    X, y = problem(n, p, 10, seed=21)
    rng = np.random.default_rng(0)
    fold = rng.integers(0, 3, n)
    alphas = alpha_path(X, y, k=6, lo=5e-2)
    lanes = []
    for f in range(3):
        m = (fold != f).astype(float)
        lanes.append(dict(points=[(a, 0, 0) for a in alphas], row_weight=m, n_eff=int(m.sum())))
    with eng.dataset(X, y) as ds:
        res = ds.solve_lanes(lanes, tol=1e-11, flags=WS)
        res0 = ds.solve_lanes(lanes, tol=1e-11, flags=NO_WS)
    assert res[0].ws_builds >= 1 and res[0].ws_refined > 0
    for f in range(3):
        tr = fold != f
        assert res[f].converged
        assert rel_inf(res[f].betas, res0[f].betas) < 1e-8
        ref = ofista(X[tr], y[tr], alphas[-1] * np.ones(p))
        assert rel_inf(res[f].betas[-1], ref) < 1e-8


def test_weighted_l1_and_warm_start(eng):
    # the adaptive estimators' inner solves: per-coefficient weights, one point, warm start
    n, p = 2000, 400
    X, y = problem(n, p, 8, seed=31)
    a = np.random.default_rng(3).uniform(0.05, 2.0, p)
    with eng.dataset(X, y) as ds:
        cold = ds.solve_path([(0.1, 0, 0)], a=a, tol=1e-11, flags=WS)
        warm = ds.solve_path([(0.08, 0, 0)], a=a, beta0=cold.betas[0], tol=1e-11, flags=WS)
        plain = ds.solve_path([(0.08, 0, 0)], a=a, tol=1e-11, flags=NO_WS)
    assert warm.converged and rel_inf(warm.betas, plain.betas) < 1e-8
    assert warm.grad_launches <= plain.grad_launches
    ref = ofista(X, y, 0.08 * a)
    assert rel_inf(warm.betas[0], ref) < 1e-8


def test_dense_solutions_overflow_the_working_set_gracefully(eng):
    # more active coefficients than the 256 columns a working set holds: refinement switches itself
    # off and the plain iteration finishes the job
    n, p = 1500, 600
    rng = np.random.default_rng(4)
    X = rng.standard_normal((n, p))
    y = X @ rng.standard_normal(p) + 0.1 * rng.standard_normal(n)
    with eng.dataset(X, y) as ds:
        r1 = ds.solve_path([(1e-4, 0, 0)], tol=1e-10, flags=WS)
        r0 = ds.solve_path([(1e-4, 0, 0)], tol=1e-10, flags=NO_WS)
    assert r1.converged and np.count_nonzero(r1.betas[0]) > 256
    assert rel_inf(r1.betas, r0.betas) < 1e-7


def test_ols_limit_inside_the_working_set(eng):
    # alpha = 0 with p <= 256: every column is in W, the model IS the problem
    n, p = 800, 40
    X, y = problem(n, p, 40, seed=41)
    with eng.dataset(X, y) as ds:
        r1 = ds.solve_path([(0, 0, 0)], tol=1e-12, flags=WS)
    assert r1.converged
    ref = np.linalg.lstsq(X, y, rcond=None)[0]
    assert rel_inf(r1.betas[0], ref) < 1e-8


def test_ill_conditioned_problem_is_solved_to_kkt(eng):
    # p > n with strongly correlated columns: FISTA alone needs tens of thousands of passes here; the
    # model solve on the Gram does the same iterations without touching X.  Check optimality directly.
    rng = np.random.default_rng(6)
    n, p = 60, 200
    Z = rng.standard_normal((n, 8))
    X = Z @ rng.standard_normal((8, p)) + 0.05 * rng.standard_normal((n, p))
    y = X[:, :5] @ np.ones(5) + 0.01 * rng.standard_normal(n)
    alphas = alpha_path(X, y, k=6, lo=1e-2)
    pts = [(a, 0, 0) for a in alphas]
    with eng.dataset(X, y) as ds:
        r1 = ds.solve_path(pts, tol=1e-10, max_iter=100000, flags=WS)
        r0 = ds.solve_path(pts, tol=1e-10, max_iter=100000, flags=NO_WS)
    assert r1.converged
    assert r1.grad_launches < r0.grad_launches
    for a, b, b0 in zip(alphas, r1.betas, r0.betas):
        g = X.T @ (X @ b - y) / n
        assert np.max(np.abs(g)) <= a * (1 + 1e-6)  # dual feasibility
        on = b != 0
        np.testing.assert_allclose(g[on], -a * np.sign(b[on]), rtol=1e-6)  # stationarity on the support
        obj = lambda v: 0.5 * np.sum((X @ v - y) ** 2) / n + a * np.abs(v).sum()
        assert obj(b) <= obj(b0) * (1 + 1e-9)


def test_small_hard_problem_gets_the_working_set_late(eng):
    # default flags on a small problem: plain steps first; once a point has cost 24 passes the host
    # switches the refinement on and the path finishes in a handful of passes instead of > 100 000
    rng = np.random.default_rng(6)
    n, p = 60, 200
    Z = rng.standard_normal((n, 8))
    X = Z @ rng.standard_normal((8, p)) + 0.05 * rng.standard_normal((n, p))
    y = X[:, :5] @ np.ones(5) + 0.01 * rng.standard_normal(n)
    alphas = alpha_path(X, y, k=6, lo=1e-2)
    pts = [(a, 0, 0) for a in alphas]
    with eng.dataset(X, y) as ds:
        r = ds.solve_path(pts, tol=1e-10, max_iter=100000)
        easy = ds.solve_path([(alphas[1], 0, 0)], tol=1e-6, max_iter=100000, flags=0)
    assert r.converged and r.ws_builds >= 1 and r.grad_launches < 2000
    for a, b in zip(alphas, r.betas):
        g = X.T @ (X @ b - y) / n
        assert np.max(np.abs(g)) <= a * (1 + 1e-6)
    assert easy.converged


def test_easy_small_problem_stays_plain(eng):
    if os.environ.get("SLM_WS") == "1":
        pytest.skip("working set forced on by SLM_WS=1")
    X, y = problem(400, 100, 10, seed=51)
    with eng.dataset(X, y) as ds:
        r = ds.solve_path([(0.1, 0, 0)], tol=1e-8)
    assert r.converged and r.ws_builds == 0 and r.grad_launches < 24


@pytest.mark.parametrize("lanes", [5, 8, 10])
def test_many_lane_split_pass_path_matches_oracle(eng, lanes):
    # working set from the first pass => the split pass with eight lane slots (residuals from the
    # gathered columns, accumulate-only stream over X)
    n, p = 4000, 900
    X, y = problem(n, p, 14, seed=77)
    alphas = alpha_path(X, y, k=24, lo=1e-2)
    pts = [(a, 0, 0) for a in alphas]
    with eng.dataset(X, y) as ds:
        r8 = ds.solve_path(pts, tol=1e-11, lanes=lanes, flags=WS)
        r1 = ds.solve_path(pts, tol=1e-11, lanes=1, flags=NO_WS)
    assert r8.converged and r8.ws_refined > 0
    assert rel_inf(r8.betas, r1.betas) < 1e-8
    for k in (5, 23):
        assert rel_inf(r8.betas[k], ofista(X, y, alphas[k] * np.ones(p))) < 1e-8
    assert r8.grad_launches <= len(alphas) // lanes + 8


def test_split_pass_with_fold_masks_and_groups(eng):
    # six CV-fold lanes with their own row masks and Grams, sparse-group penalty, shuffled groups
    n, G, size = 3000, 30, 5
    p = G * size
    groups = np.random.default_rng(9).permutation(np.repeat(np.arange(G), size))
    X, y = problem(n, p, 4, seed=13, groups=groups)
    fold = np.random.default_rng(1).integers(0, 6, n)
    alphas = alpha_path(X, y, k=6, lo=5e-2, groups=groups, G=G)
    lanes = []
    for f in range(6):
        m = (fold != f).astype(float)
        lanes.append(dict(points=[(0.3 * a, 0.7 * a, 0) for a in alphas], row_weight=m, n_eff=int(m.sum())))
    with eng.dataset(X, y) as ds:
        ds.set_groups(groups, G)
        res = ds.solve_lanes(lanes, tol=1e-11, flags=WS)
        res0 = ds.solve_lanes(lanes[:4], tol=1e-11, flags=NO_WS)
    for f in range(6):
        assert res[f].converged
        tr = fold != f
        ref = ofista(X[tr], y[tr], 0.3 * alphas[-1] * np.ones(p), 0.7 * alphas[-1] * np.ones(G), None, groups, G)
        assert rel_inf(res[f].betas[-1], ref) < 1e-8
    for f in range(4):
        assert rel_inf(res[f].betas, res0[f].betas) < 1e-8


def test_plain_steps_inside_the_split_pass_use_both_rowdot_windows(eng):
    # seven lanes, a dense problem: the refinement gives up (> 512 non-zeros), every lane takes plain
    # steps whose residuals come from X (rowdot_ring_kernel, two launches of five lanes)
    n, p = 1500, 700
    rng = np.random.default_rng(4)
    X = rng.standard_normal((n, p))
    y = X @ rng.standard_normal(p) + 0.1 * rng.standard_normal(n)
    lanes = [dict(points=[(a, 0, 0)]) for a in np.geomspace(3e-4, 1e-4, 7)]
    with eng.dataset(X, y) as ds:
        res = ds.solve_lanes(lanes, tol=1e-10, flags=WS)
        ref = [ds.solve_path(l["points"], tol=1e-10, flags=NO_WS) for l in lanes]
    for r, r0 in zip(res, ref):
        assert r.converged and np.count_nonzero(r.betas[0]) > 512
        assert rel_inf(r.betas, r0.betas) < 1e-7


def test_estimators_on_large_X_take_the_working_set_path(monkeypatch):
    # n * ld >= 2^26: the engine's default policy switches the refinement and the split pass on; the
    # estimator surface (sample weights, intercept, groups, adaptive re-weighting) must give the same
    # coefficients as with the plain iteration (SLM_WS=0)
    import warnings

    from sparselm_amd.model import AdaptiveLasso, Lasso, SparseGroupLasso

    rng = np.random.default_rng(12)
    n, p = 70_000, 960  # 67.2M doubles
    X = rng.standard_normal((n, p))
    beta = np.zeros(p)
    beta[rng.choice(p, 20, replace=False)] = rng.uniform(1, 4, 20) * rng.choice([-1, 1], 20)
    y = X @ beta + 3.0 + rng.standard_normal(n)
    w = rng.uniform(0.5, 1.5, n)
    groups = np.repeat(np.arange(p // 8), 8)
    cases = [
        (lambda: Lasso(alpha=0.05, fit_intercept=True), dict(sample_weight=w)),
        (lambda: SparseGroupLasso(groups=groups, alpha=0.05, l1_ratio=0.4), {}),
        (lambda: AdaptiveLasso(alpha=0.05, max_iter=3), {}),
    ]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for make, kw in cases:
            monkeypatch.delenv("SLM_WS", raising=False)
            a = make().fit(X, y, **kw)
            monkeypatch.setenv("SLM_WS", "0")
            b = make().fit(X, y, **kw)
            assert rel_inf(a.coef_, b.coef_) < 1e-6, type(a).__name__
            assert abs(a.intercept_ - b.intercept_) <= 1e-6 * max(1.0, abs(b.intercept_))
            assert np.count_nonzero(a.coef_) >= 20


def test_working_set_in_row_sharded_mode_with_one_rank():
    # RCCL communicator with a single rank: the Gram parts go through the staging matrix, the
    # all-reduce and ws_publish_kernel; the gradient of the split pass is all-reduced too
    from sparselm_amd import distributed as D

    n, G, size = 6000, 40, 5
    p = G * size
    groups = np.random.default_rng(3).permutation(np.repeat(np.arange(G), size))
    X, y = problem(n, p, 5, seed=23, groups=groups)
    alphas = alpha_path(X, y, k=10, lo=3e-2, groups=groups, G=G)
    pts = [(0.3 * a, 0.7 * a, 0) for a in alphas]
    eng2 = _engine.Engine(0)
    try:
        D.init_row_sharding(eng2, rank=0, world_size=1)
        with eng2.dataset(X, y) as ds:
            ds.set_global_rows(n)
            ds.set_groups(groups, G)
            r1 = ds.solve_path(pts, tol=1e-11, lanes=4, flags=WS)
            r0 = ds.solve_path(pts, tol=1e-11, lanes=1, flags=NO_WS)
    finally:
        eng2.comm_destroy()
        eng2.close()
    assert r1.converged and r1.ws_builds >= 1 and r1.ws_refined > 0
    assert rel_inf(r1.betas, r0.betas) < 1e-8
    assert r1.grad_launches < r0.grad_launches
    ref = ofista(X, y, pts[-1][0] * np.ones(p), pts[-1][1] * np.ones(G), None, groups, G)
    assert rel_inf(r1.betas[-1], ref) < 1e-8


def test_interleaved_lanes_of_a_shared_path(eng, monkeypatch):
    # working set from the start + per-feature penalty: the lanes of a shared path take its points in
    # turn (lane l: l, l + B, ...); same coefficients as contiguous ranges, never more passes
    n, p = 5000, 800
    X, y = problem(n, p, 16, seed=91)
    alphas = alpha_path(X, y, k=23, lo=5e-3)  # 23 points over 5 lanes: ragged last round
    pts = [(a, 0, 0) for a in alphas]
    with eng.dataset(X, y) as ds:
        ri = ds.solve_path(pts, tol=1e-11, lanes=5, flags=WS)
        monkeypatch.setenv("SLM_NO_INTERLEAVE", "1")
        rc = ds.solve_path(pts, tol=1e-11, lanes=5, flags=WS)
        monkeypatch.delenv("SLM_NO_INTERLEAVE")
        r1 = ds.solve_path(pts, tol=1e-11, lanes=1, flags=NO_WS)
    assert ri.converged and rc.converged
    assert rel_inf(ri.betas, r1.betas) < 1e-8 and rel_inf(rc.betas, r1.betas) < 1e-8
    assert ri.grad_launches <= rc.grad_launches
    assert ri.grad_launches <= 1 + -(-len(alphas) // 5) + 2


def test_long_shared_paths_whose_last_band_goes_to_the_last_lanes(eng, monkeypatch):
    """Shared paths of 32-74 points on 8 / 12 / 16 interleaved lanes: the points beyond the last full band are
    walked by the last lanes (PathCtl::tail_pt).  Every point is solved exactly once, against the plain iteration;
    giving that band to the first lanes instead (SLM_NO_TAIL_BAND=1) changes the schedule, not the answers."""
    rng = np.random.default_rng(0)
    for case in range(10):
        n = int(rng.integers(800, 4000))
        p = int(rng.integers(100, 900))
        X = rng.standard_normal((n, p))
        beta = np.zeros(p)
        nz = rng.choice(p, int(rng.integers(3, 40)), replace=False)
        beta[nz] = rng.standard_normal(len(nz)) * 3
        y = X @ beta + rng.standard_normal(n) * rng.choice([0.1, 1.0, 5.0])
        amax = np.max(np.abs(X.T @ y)) / n
        K = int(rng.integers(32, 75))
        lanes = int(rng.choice([16, 16, 12, 8]))
        pts = [(a, 0, 0) for a in np.geomspace(amax, rng.choice([0.1, 0.01, 0.001]) * amax, K)]
        with eng.dataset(X, y) as ds:
            r = ds.solve_path(pts, lanes=lanes, flags=WS, tol=1e-10)
            q = ds.solve_path(pts, lanes=4, flags=NO_WS, tol=1e-11)
            monkeypatch.setenv("SLM_NO_TAIL_BAND", "1")
            o = ds.solve_path(pts, lanes=lanes, flags=WS, tol=1e-10)
            monkeypatch.delenv("SLM_NO_TAIL_BAND")
        assert r.converged and q.converged and o.converged, (case, K, lanes)
        assert np.all(r.n_iter >= 1) and np.all(o.n_iter >= 1)  # every point was visited
        assert rel_inf(r.betas, q.betas) < 1e-7, (case, K, lanes)
        assert rel_inf(o.betas, q.betas) < 1e-7, (case, K, lanes)


def test_randomised_cross_check_against_plain_iteration():
    """tools/ws_fuzz.py: random penalty kinds, group sizes, lane counts, fold masks and p > n; the working-set
    answer must match the plain iteration to 1e-6 or, where the minimiser is not unique, reach the same objective."""
    import subprocess
    import sys

    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "ws_fuzz.py"), "120", "7"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "FUZZ cases 120" in out.stdout and "flagged 0" in out.stdout


def test_vector_and_matrix_core_residual_kernels_agree(eng, monkeypatch):
    # residuals from the gathered columns: resid_mfma_kernel (default) against resid_ws_kernel (SLM_RESID_VEC=1)
    rng = np.random.default_rng(12)
    n, p = 2100, 640
    X = rng.standard_normal((n, p))
    beta = np.zeros(p)
    beta[rng.choice(p, 25, replace=False)] = rng.standard_normal(25) * 4
    y = X @ beta + rng.standard_normal(n)
    w = rng.uniform(0.2, 2.0, n)
    amax = float(np.max(np.abs(X.T @ (w * y)))) / n
    pts = [(a, 0.0, 0.0) for a in np.geomspace(amax, 0.01 * amax, 30)]
    out = {}
    for name, flag in (("mfma", "0"), ("vec", "1")):
        monkeypatch.setenv("SLM_RESID_VEC", flag)
        with eng.dataset(X, y, row_weight=w) as ds:
            out[name] = ds.solve_path(pts, tol=1e-11, lanes=12, flags=_engine.FLAG_WORKING_SET)
    assert out["mfma"].converged and out["vec"].converged and out["mfma"].ws_refined > 0
    scale = np.max(np.abs(out["vec"].betas))
    assert np.max(np.abs(out["mfma"].betas - out["vec"].betas)) < 1e-8 * scale


@pytest.mark.gpu
def test_lanes_that_share_a_gram_give_the_same_bits_every_time():
    """Grid rows of one CV fold share a row mask, hence one working-set Gram and one bound on its lambda_max, which
    every lane of the set may raise (curvature guard of the model solver) and publishes by an integer atomic max
    (ws_kernels.hpp).  The bound the next pass starts from must not depend on which lane ended last: five runs of
    the same eight-lane call, and one on a device-to-device copy, bit for bit."""
    rng = np.random.default_rng(21)
    n, p = 3000, 400
    X = rng.standard_normal((n, p)) + 0.6 * rng.standard_normal((n, 1))
    coef = np.zeros(p)
    coef[rng.choice(p, 20, replace=False)] = 3 * rng.standard_normal(20)
    y = X @ coef + rng.standard_normal(n)
    masks = [(np.arange(n) % 2 != f).astype(float) for f in range(2)]
    eng = _engine.get_engine(0)
    with eng.dataset(X, y) as ds:
        g0, _ = ds.gradient(None)
        amax = float(np.max(np.abs(g0)))
        specs = []
        for f in range(2):
            for scale in (1.0, 0.8, 0.6, 0.45):
                al = np.geomspace(scale * amax, 5e-3 * scale * amax, 14)
                specs.append(dict(points=np.c_[al, 0 * al, 0 * al], row_weight=masks[f], n_eff=int(masks[f].sum())))
        runs = [ds.solve_lanes(specs, tol=1e-11, flags=_engine.FLAG_WORKING_SET) for _ in range(5)]
        with ds.clone() as copy:
            runs.append(copy.solve_lanes(specs, tol=1e-11, flags=_engine.FLAG_WORKING_SET))
            copy_engine = copy.engine
        copy_engine.close()
    assert all(r.converged for run in runs for r in run) and runs[0][0].ws_refined > 0
    for run in runs[1:]:
        for a, b in zip(runs[0], run):
            np.testing.assert_array_equal(a.betas, b.betas)


def test_cold_shared_paths_open_on_a_row_sample(eng, monkeypatch):
    """Sample start (solve_core): a cold shared path chooses its first working set from an eighth of the rows, builds the
    model with the exact X_W^T y, and lets the first pass over X verify the first band -- one pass fewer, the same
    coefficients.  A head of rows that says nothing about the rest (here: rows ordered by |y|, the sample sees the
    smallest targets only) may cost that pass again, never a digit."""
    rng = np.random.default_rng(17)
    n, p = 70000, 320
    X = rng.standard_normal((n, p))
    beta = np.zeros(p)
    beta[rng.choice(p, 12, replace=False)] = rng.uniform(1, 6, 12) * rng.choice([-1, 1], 12)
    y = X @ beta + 2.0 * rng.standard_normal(n)
    amax = np.max(np.abs(X.T @ y)) / n
    pts = [(a, 0, 0) for a in np.geomspace(amax, 5e-3 * amax, 40)]
    with eng.dataset(X, y) as ds:
        fast = ds.solve_path(pts, tol=1e-10, lanes=16, flags=WS)
        monkeypatch.setenv("SLM_NO_SAMPLE_START", "1")
        full = ds.solve_path(pts, tol=1e-10, lanes=16, flags=WS)
        monkeypatch.delenv("SLM_NO_SAMPLE_START")
        plain = ds.solve_path(pts, tol=1e-11, lanes=4, flags=NO_WS)
    assert fast.converged and full.converged and plain.converged
    assert fast.grad_launches == full.grad_launches - 1, (fast.grad_launches, full.grad_launches)
    assert rel_inf(fast.betas, plain.betas) < 1e-8 and rel_inf(full.betas, plain.betas) < 1e-8

    # small problems (the sample forced on), heads of rows that mislead, group penalties on contiguous ranges
    monkeypatch.setenv("SLM_SAMPLE_START_MIN_ROWS", "64")
    for case in range(6):
        n, p = int(rng.integers(1500, 6000)), int(rng.integers(120, 700))
        X = rng.standard_normal((n, p))
        beta = np.zeros(p)
        nz = rng.choice(p, int(rng.integers(3, 30)), replace=False)
        beta[nz] = rng.standard_normal(len(nz)) * 3
        y = X @ beta + rng.standard_normal(n)
        if case % 2 == 0:  # the sample sees the rows with the smallest targets only
            order = np.argsort(np.abs(y))
            X, y = np.ascontiguousarray(X[order]), y[order]
        groups = None if case < 3 else rng.integers(0, 25, p)
        lanes = int(rng.choice([4, 8, 16]))
        with eng.dataset(X, y) as ds:
            if groups is not None:
                ds.set_groups(groups, 25)
                c = X.T @ y / n
                top = np.max(np.sqrt(np.bincount(groups, weights=c * c, minlength=25)))
                pts = [(0.0, a, 0.0) for a in np.geomspace(top, 0.02 * top, 24)]
            else:
                amax = np.max(np.abs(X.T @ y)) / n
                pts = [(a, 0, 0) for a in np.geomspace(amax, 0.01 * amax, 33)]
            r = ds.solve_path(pts, tol=1e-10, lanes=lanes, flags=WS)
            q = ds.solve_path(pts, tol=1e-11, lanes=min(lanes, 4), flags=NO_WS)
        assert r.converged and q.converged, case
        assert np.all(r.n_iter >= 1), case
        assert rel_inf(r.betas, q.betas) < 1e-7, (case, rel_inf(r.betas, q.betas))


def test_re_weighted_solves_take_over_the_working_set(eng, monkeypatch):
    """A solve that starts where the last one ended, on the same row sets, keeps that solve's working set -- columns,
    gathered copies, Grams: X and the rows have not changed, the penalty has -- and appends what the new penalty lets in:
    the same solutions as with a fresh selection (SLM_NO_WS_CARRY=1), never more passes, no build."""
    rng = np.random.default_rng(23)
    n, p, G = 4000, 900, 90
    X = rng.standard_normal((n, p))
    groups = np.repeat(np.arange(G), p // G)
    beta = np.zeros(p)
    for g in rng.choice(G, 8, replace=False):
        beta[groups == g] = rng.standard_normal(p // G) * 2
    y = X @ beta + rng.standard_normal(n)
    c = X.T @ y / n
    alpha = 0.08 * np.max(np.sqrt(np.bincount(groups, weights=c * c)))
    masks = [(np.arange(n) % 4 != f).astype(float) for f in range(3)]

    def rounds(ds):
        ds.set_groups(groups, G)
        specs = [dict(points=[(0.0, 1.0, 0.0)], b=alpha * np.ones(G), row_weight=m, n_eff=int(m.sum())) for m in masks]
        out, stats = [], []
        for _ in range(4):
            res = ds.solve_lanes(specs, tol=1e-10, flags=WS, want_group_norms=True)
            assert all(r.converged for r in res)
            out.append([r.betas[0].copy() for r in res])
            stats.append((res[0].grad_launches, res[0].ws_builds, res[0].ws_appends))
            for sp, r in zip(specs, res):
                sp["b"] = alpha * (alpha / (r.group_norms[0] + 1e-3))
                sp["beta0"] = r.betas[0].copy()
        return out, stats

    with eng.dataset(X, y) as ds:
        kept, st_kept = rounds(ds)
    monkeypatch.setenv("SLM_NO_WS_CARRY", "1")
    with eng.dataset(X, y) as ds:
        fresh, st_fresh = rounds(ds)
    monkeypatch.delenv("SLM_NO_WS_CARRY")
    assert st_kept[0] == st_fresh[0] and st_fresh[1][1] >= 1  # (the first solve is the same; later ones used to build anew)
    for k in range(1, 4):
        assert st_kept[k][1] == 0, st_kept  # no build in a solve that took the set over
        assert st_kept[k][0] <= st_fresh[k][0], (st_kept, st_fresh)
    for a_round, b_round in zip(kept, fresh):
        for u, v in zip(a_round, b_round):
            assert rel_inf(u, v) < 1e-7
    # against the oracle, the last round's first fold
    tr = masks[0] > 0
    gn_prev = np.sqrt(np.bincount(groups, weights=kept[2][0] ** 2, minlength=G))
    want = oracle.fista(X[tr], y[tr], 0.0, alpha * (alpha / (gn_prev + 1e-3)), 0.0, groups, G, tol=1e-13, max_iter=200000)[0]
    assert rel_inf(kept[3][0], want) < 1e-6


def test_fit_scans_large_designs_for_nan_and_infinity_on_the_device(monkeypatch):
    """`fit` leaves the NaN / infinity scan of a LARGE X to the device copy (model/_base.py: two thirds of a 4 GB fit were
    numpy's sum in check_array): the same ValueError as scikit-learn's, before anything is solved; clean data fit as before."""
    from sparselm_amd import model
    from sparselm_amd.model import _base

    rng = np.random.default_rng(41)
    n, p = 1500, 800  # 1.2 M entries: above the threshold
    X = rng.standard_normal((n, p))
    y = X[:, :5] @ rng.uniform(1, 3, 5) + rng.standard_normal(n)
    assert X.size >= _base._DEVICE_SCAN_FROM
    fitted = model.Lasso(alpha=0.05, fit_intercept=True).fit(X, y)
    monkeypatch.setattr(_base, "_DEVICE_SCAN_FROM", 1 << 62)  # the host's scan, as for small arrays
    on_host = model.Lasso(alpha=0.05, fit_intercept=True).fit(X, y)
    monkeypatch.undo()
    np.testing.assert_array_equal(fitted.coef_, on_host.coef_)
    assert fitted.intercept_ == on_host.intercept_
    for bad, word in ((np.nan, "NaN"), (np.inf, "infinity"), (-np.inf, "infinity")):
        Xb = X.copy()
        Xb[rng.integers(n), rng.integers(p)] = bad
        for est in (model.Lasso(alpha=0.05), model.GroupLasso(groups=np.arange(p) // 8, alpha=0.05, fit_intercept=True),
                    model.AdaptiveLasso(alpha=0.05)):
            with pytest.raises(ValueError, match=word):
                est.fit(Xb, y)
    yb = y.copy()
    yb[3] = np.nan
    with pytest.raises(ValueError, match="NaN"):
        model.Lasso(alpha=0.05).fit(X, yb)  # (targets are scanned on the host as always)

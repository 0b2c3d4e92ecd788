"""The on-chip solver for problems that fit a workgroup (csrc/small_kernels.hpp; SLM_FLAG_ON_CHIP): the sizes the
reference's own tests and README run (/root/reference/tests/conftest.py:17-19, README.md:42-55)."""

import os
import subprocess
import sys

import numpy as np
import pytest

import oracle
from sparselm_amd import _engine

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.fixture(scope="module")
def eng():
    return _engine.get_engine(0)


def test_reference_sized_fits_match_the_oracle(eng):
    """25 x 20 and 25 x 30 (p > n) with the reference fixture's law, 100 x 80 as in its README: Lasso, group, sparse
    group and ridged penalties through ONE launch each, against the oracle's proximal-gradient iteration."""
    rng = np.random.default_rng(0)
    for n, p, G in ((25, 20, 4), (25, 30, 6), (100, 80, 8)):
        X = rng.standard_normal((n, p))
        beta = np.zeros(p)
        beta[rng.choice(p, 10, replace=False)] = rng.uniform(1, 5, 10)
        y = X @ beta + 0.5 * rng.standard_normal(n) + 3.0
        groups = rng.permutation(np.arange(p) % G)
        with eng.dataset(X, y) as ds:
            for kind, pen in (("lasso", (0.3, 0.0, 0.0)), ("group", (0.0, 0.6, 0.0)), ("sgl", (0.2, 0.3, 0.0)),
                              ("ridged", (0.0, 0.5, 0.4))):
                gidx, GG = oracle.group_index(None if kind == "lasso" else groups, p)
                ds.set_groups(None if kind == "lasso" else groups, None if kind == "lasso" else G)
                res = ds.solve_path([pen], tol=1e-12, flags=_engine.FLAG_ON_CHIP, want_group_norms=True)
                assert res.converged and res.mode[0] == 2 and res.grad_launches == 1
                ref, _ = oracle.fista(X, y, pen[0], pen[1], pen[2], gidx, GG, tol=1e-14, max_iter=500000)
                # (p > n: compare objectives -- the minimiser need not be unique)
                f = lambda b: oracle.objective(X, y, b, pen[0], pen[1], pen[2], gidx, GG)  # noqa: E731
                assert f(res.betas[0]) <= f(ref) * (1 + 1e-12) + 1e-14
                if n > p:
                    np.testing.assert_allclose(res.betas[0], ref, rtol=0, atol=1e-9 * np.max(np.abs(ref)))
                np.testing.assert_allclose(res.group_norms[0], np.sqrt(np.bincount(gidx, weights=res.betas[0] ** 2, minlength=GG)),
                                           rtol=1e-12, atol=1e-15)
                # the reported residual is the minimal-norm subgradient at the reported point
                g, _ = ds.gradient(res.betas[0])
                kkt = oracle.kkt_residual(g, res.betas[0], pen[0] * np.ones(p), pen[1] * np.ones(GG), pen[2] * np.ones(GG), gidx, GG)
                assert abs(res.kkt[0] - kkt) <= 1e-8 * max(1.0, np.max(np.abs(g))) and kkt < 1e-8 * np.max(np.abs(g))


def test_paths_folds_and_warm_starts_in_one_launch(eng):
    """Sixteen lanes -- CV folds as row masks with their own 1/n -- each a warm-started 12-point path: one launch, the
    same coefficients as the general path lane by lane; a warm start and COLD_START change nothing but the sweeps."""
    rng = np.random.default_rng(1)
    n, p = 120, 60
    X = rng.standard_normal((n, p)) @ (np.eye(p) + 0.4 * rng.standard_normal((p, p)) / np.sqrt(p))
    y = X[:, :6] @ rng.uniform(1, 3, 6) + rng.standard_normal(n)
    amax = np.max(np.abs(X.T @ y)) / n
    pts = [(a, 0.0, 0.0) for a in np.geomspace(amax, 0.01 * amax, 12)]
    fold = rng.integers(0, 16, n)
    specs = [dict(points=pts, row_weight=(fold != f).astype(float), n_eff=int(np.sum(fold != f))) for f in range(16)]
    with eng.dataset(X, y) as ds:
        assert ds.max_lanes(_engine.FLAG_ON_CHIP) == _engine.MAX_CELLS == 64 and ds.max_lanes() < 16
        chip = ds.solve_lanes(specs, tol=1e-11, flags=_engine.FLAG_ON_CHIP)
        for spec, r in zip(specs, chip):
            ref = ds.solve_lanes([spec], tol=1e-11)[0]
            assert r.converged and np.all(r.mode == 2) and ref.converged and np.all(ref.mode != 2)
            np.testing.assert_allclose(r.betas, ref.betas, rtol=0, atol=1e-8 * np.max(np.abs(ref.betas)))
        cold = ds.solve_lanes(specs[:3], tol=1e-11, flags=_engine.FLAG_ON_CHIP | _engine.FLAG_COLD_START)
        warm = ds.solve_lanes([dict(s, beta0=chip[i].betas[3]) for i, s in enumerate(specs[:3])], tol=1e-11, flags=_engine.FLAG_ON_CHIP)
        for i in range(3):
            np.testing.assert_allclose(cold[i].betas, chip[i].betas, rtol=0, atol=1e-8 * np.max(np.abs(chip[i].betas)))
            np.testing.assert_allclose(warm[i].betas, chip[i].betas, rtol=0, atol=1e-8 * np.max(np.abs(chip[i].betas)))
            assert np.sum(cold[i].n_iter) > np.sum(chip[i].n_iter)
        # the shared-path entry point (ranges of one path as lanes) and the flags that name another iteration
        one = ds.solve_path(pts, tol=1e-11, flags=_engine.FLAG_ON_CHIP, lanes=4)
        ref = ds.solve_path(pts, tol=1e-11)
        assert np.all(one.mode == 2)
        np.testing.assert_allclose(one.betas, ref.betas, rtol=0, atol=1e-8 * np.max(np.abs(ref.betas)))
        assert np.all(ds.solve_path(pts, tol=1e-9, flags=_engine.FLAG_ON_CHIP | _engine.FLAG_FISTA_ONLY).mode == 0)


def test_what_does_not_fit_or_does_not_settle_takes_the_general_path(eng):
    rng = np.random.default_rng(2)
    X = rng.standard_normal((40, 200))  # p > 128: no Gram matrix in the LDS
    y = rng.standard_normal(40)
    with eng.dataset(X, y) as ds:
        r = ds.solve_path([(0.3, 0.0, 0.0)], flags=_engine.FLAG_ON_CHIP)
        assert r.converged and r.mode[0] != 2
    # nearly collinear columns: the coordinate iteration crawls; with two sweeps allowed it must hand over, and the
    # answer is the general path's
    n, p = 60, 12
    base = rng.standard_normal((n, 3))
    X = base @ rng.standard_normal((3, p)) + 1e-4 * rng.standard_normal((n, p))
    y = X @ rng.standard_normal(p) + 0.1 * rng.standard_normal(n)
    with eng.dataset(X, y) as ds:
        ref = ds.solve_path([(1e-3, 0.0, 0.0)], tol=1e-9, max_iter=200000)
        r = ds.solve_path([(1e-3, 0.0, 0.0)], tol=1e-9, max_iter=200000, flags=_engine.FLAG_ON_CHIP)
        assert r.converged and ref.converged
        f = lambda b: 0.5 * np.mean((X @ b - y) ** 2) + 1e-3 * np.sum(np.abs(b))  # noqa: E731
        assert f(r.betas[0]) <= f(ref.betas[0]) * (1 + 1e-9)
    with pytest.raises(_engine.NonFiniteError):
        bad = X.copy()
        bad[3, 2] = np.inf
        with eng.dataset(bad, y) as ds:
            ds.solve_path([(0.1, 0.0, 0.0)], flags=_engine.FLAG_ON_CHIP)


def test_fifty_cells_in_one_launch(eng):
    """More lanes than sixteen on the on-chip route (SLM_MAX_CELLS: the (candidate, fold) cells of a small grid search -- the
    reference's README example has fifty): ONE launch, a workgroup per cell; the same coefficients as sixteen at a time; a
    call the general path has to take (flags that ask for its iteration) still stops at sixteen lanes."""
    rng = np.random.default_rng(2)
    n, p = 100, 80
    X = rng.standard_normal((n, p))
    y = X[:, :8] @ rng.uniform(1, 3, 8) + 0.3 * rng.standard_normal(n)
    amax = np.max(np.abs(X.T @ y)) / n
    fold = rng.integers(0, 5, n)
    specs = []
    for c in range(10):
        for f in range(5):
            w = (fold != f).astype(float)
            specs.append(dict(points=[(amax * 0.6 ** c, 0.0, 0.0)], a=rng.uniform(0.5, 1.5, p), row_weight=w, n_eff=int(w.sum()),
                              beta0=(0.01 * rng.standard_normal(p) if c % 3 == 0 else None)))
    with eng.dataset(X, y) as ds:
        one = ds.solve_lanes(specs, tol=1e-11, flags=_engine.FLAG_ON_CHIP)
        assert len(one) == 50 and all(r.converged and r.mode[0] == 2 for r in one) and one[0].grad_launches == 1
        for k0 in range(0, 50, 16):
            part = ds.solve_lanes(specs[k0 : k0 + 16], tol=1e-11, flags=_engine.FLAG_ON_CHIP)
            for a, b in zip(one[k0 : k0 + 16], part):
                assert np.array_equal(a.betas, b.betas)
        ref = ds.solve_lanes(specs[:3], tol=1e-11)
        for a, b in zip(one[:3], ref):
            np.testing.assert_allclose(a.betas, b.betas, rtol=0, atol=1e-8 * np.max(np.abs(b.betas)))
        # (seventeen lanes fit the engine's count since round 5 -- two halves of the split pass -- but not this dataset's kernels)
        with pytest.raises((ValueError, NotImplementedError)):
            ds.solve_lanes(specs[:17], tol=1e-11, flags=_engine.FLAG_ON_CHIP | _engine.FLAG_FISTA_ONLY)
        with pytest.raises(ValueError):
            ds.solve_lanes(specs[:33], tol=1e-11, flags=_engine.FLAG_ON_CHIP | _engine.FLAG_FISTA_ONLY)
        with pytest.raises(ValueError):
            ds.solve_lanes(specs + specs[:15], tol=1e-11, flags=_engine.FLAG_ON_CHIP)
    # a dataset the on-chip solver does not take keeps the sixteen
    Xb = rng.standard_normal((300, 130))
    with eng.dataset(Xb, Xb[:, 0]) as big:
        assert big.max_lanes(_engine.FLAG_ON_CHIP) <= 16


def test_randomised_cross_check_against_the_general_path():
    """tools/on_chip_fuzz.py: 150 random problems of up to 128 features (penalty kinds, groups, weighted l1, ridge, fold
    masks, paths, warm starts, p > n, correlated columns), the on-chip call against the same cells on the general path."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "on_chip_fuzz.py"), "150", "3"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "flagged 0" in out.stdout.splitlines()[-1]


def test_standardized_sparse_group_splitting_in_one_launch(eng):
    """slm_solve_standardized_sgl (csrc/small_split_kernels.hpp): l1 + sum_g b_g ||X_g beta_g||_2 -- the reference's
    SparseGroupLasso(standardize=True), src/sparselm/model/_lasso.py:616-639 with the group norms of :249-252 -- with
    all sweeps of the splitting on chip, against the oracle's primal-dual iteration and the optimality conditions of the
    original problem; a rank-deficient group, groups in scattered order, an empty group, and the continuation of the
    splitting variables the adaptive loop uses."""
    rng = np.random.default_rng(5)
    for n, p, G in ((100, 80, 10), (25, 30, 6), (400, 100, 25), (60, 128, 16)):
        X = rng.standard_normal((n, p)) + 0.5 * rng.standard_normal((n, 1))
        groups = rng.permutation(np.arange(p) % G)
        cols = np.flatnonzero(groups == 0)
        X[:, cols[-1]] = 2.0 * X[:, cols[0]]  # a rank-deficient group
        beta = np.where(rng.random(p) < 0.25, rng.standard_normal(p), 0.0)
        y = X @ beta + 0.2 * rng.standard_normal(n)
        gidx, GG = oracle.group_index(groups, p)
        a, b = 0.05 * rng.uniform(0.5, 1.5, p), 0.1 * rng.uniform(0.5, 1.5, GG)
        with eng.dataset(X, y) as ds:
            ds.set_groups(groups, GG)
            coef, gn, rec = ds.solve_standardized_sgl(a, b, tol=1e-11, max_sweeps=5000, want_group_norms=True)
            assert rec["status"] == 0 and rec["mode"] == 2
            scale = np.max(np.abs(X.T @ y)) / n
            ztol = 1e-9 * np.max(np.abs(coef))
            assert oracle.kkt_standardized(X, y, a, b, gidx, GG, coef, zero_tol=ztol) < 1e-7 * scale
            fit_norms = np.array([np.linalg.norm(X[:, gidx == g] @ coef[gidx == g]) for g in range(GG)])
            np.testing.assert_allclose(gn, fit_norms, rtol=1e-9, atol=1e-12 * np.max(fit_norms))
            assert np.all(coef[np.isin(gidx, np.flatnonzero(fit_norms == 0.0))] == 0.0)  # groups that are out: exact zeros
            if n > p:
                ref, info = oracle.standardized_sparse_group(X, y, a, b, gidx, GG)
                assert info["converged"]
                # (the duplicated column makes the split inside group 0 non-unique in the group norm; the l1 term picks it)
                np.testing.assert_allclose(coef, ref, rtol=0, atol=1e-7 * np.max(np.abs(ref)))
            # continuation: the same call again from its own splitting variables stops after a sweep or two
            again, _, rec2 = ds.solve_standardized_sgl(a, b, beta0=coef, warm=True, tol=1e-11, max_sweeps=5000)
            assert rec2["status"] == 0 and rec2["n_iter"] <= 3 and rec2["n_iter"] < rec["n_iter"]
            np.testing.assert_allclose(again, coef, rtol=0, atol=1e-9 * np.max(np.abs(coef)))
    # an empty group, singleton groups (then the group norm is ||x_j|| |b_j|: a weighted Lasso)
    n, p = 50, 12
    X = rng.standard_normal((n, p))
    y = X[:, :3] @ np.array([2.0, -1.0, 0.5]) + 0.1 * rng.standard_normal(n)
    with eng.dataset(X, y) as ds:
        groups = np.array([0, 0, 0, 2, 2, 2, 3, 3, 3, 3, 3, 3])
        ds.set_groups(groups, 4)
        coef, gn, rec = ds.solve_standardized_sgl(0.02 * np.ones(p), 0.1 * np.ones(4), tol=1e-11, want_group_norms=True)
        assert rec["status"] == 0 and gn[1] == 0.0
        assert oracle.kkt_standardized(X, y, 0.02, 0.1, groups, 4, coef) < 1e-8
        ds.set_groups(None)
        coef, _, rec = ds.solve_standardized_sgl(0.02 * np.ones(p), 0.1 * np.ones(p), tol=1e-11)
        assert rec["status"] == 0
        wl = 0.02 + 0.1 * np.linalg.norm(X, axis=0)
        ref, _ = oracle.fista(X, y, wl, 0.0, 0.0, np.arange(p), p, tol=1e-14, max_iter=500000)
        np.testing.assert_allclose(coef, ref, rtol=0, atol=1e-8 * np.max(np.abs(ref)))
    # not for the kernel: too many features, row weights -> SLM_ERR_UNSUPPORTED (the estimator then runs the sweeps)
    X = rng.standard_normal((40, 130))
    with eng.dataset(X, X[:, 0]) as ds:
        with pytest.raises(NotImplementedError):
            ds.solve_standardized_sgl(np.ones(130), np.ones(130))
    X = rng.standard_normal((40, 10))
    with eng.dataset(X, X[:, 0], row_weight=np.linspace(0.5, 1.5, 40)) as ds:
        with pytest.raises(NotImplementedError):
            ds.solve_standardized_sgl(np.ones(10), np.ones(10))


def test_standardized_estimator_takes_the_on_chip_route_and_agrees_with_the_sweeps(golden):
    """SparseGroupLasso(standardize=True) / AdaptiveSparseGroupLasso through the estimator: on chip by default, the
    host-driven sweeps of model/_split.py with on_chip=False -- same coefficients, same number of adaptive rounds."""
    import warnings

    from sparselm_amd.model import AdaptiveSparseGroupLasso, SparseGroupLasso

    X, y, groups, gw = golden["grp_X"], golden["grp_y"], golden["grp_groups"], golden["grp_gw"]
    fits = {}
    for on_chip in (True, False):
        opts = {"tol": 1e-11, "max_iter": 200000, "on_chip": on_chip}
        m = SparseGroupLasso(groups=groups, alpha=0.4, l1_ratio=0.5, group_weights=gw, standardize=True, solver_options=opts).fit(X, y)
        assert m.solver_info_["converged"] and bool(m.solver_info_.get("on_chip", False)) == on_chip
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            am = AdaptiveSparseGroupLasso(groups=groups, alpha=0.8, l1_ratio=0.4, group_weights=gw, standardize=True,
                                          fit_intercept=True, solver_options=opts).fit(X, y)
        fits[on_chip] = (m, am)
    scale = np.max(np.abs(fits[False][0].coef_))
    assert np.max(np.abs(fits[True][0].coef_ - fits[False][0].coef_)) < 1e-8 * scale
    assert np.max(np.abs(fits[True][0].coef_ - golden["std_sgl_coef"])) < 1e-8 * scale
    assert fits[True][1].n_iter_ == fits[False][1].n_iter_ == int(golden["std_ada_sgl_niter"])
    assert np.max(np.abs(fits[True][1].coef_ - golden["std_ada_sgl_coef"])) < 1e-6 * np.max(np.abs(golden["std_ada_sgl_coef"]))


# ---- re-weighted rounds inside the launch (slm_solve_lanes_reweighted) ----------------------------------------------------
def _host_rounds(ds, spec, rule, rounds, tol):
    """The loop the kernel replaces, over solve_lanes: (betas per round, rounds run)."""
    coef_scale, group_scale, numer, eps, rtol, n_coef, n_group = rule
    a, b = np.array(spec["a"], dtype=float), np.array(spec["b"], dtype=float)
    beta, out = None, []
    for _ in range(rounds):
        res = ds.solve_lanes([dict(points=[(1.0, 1.0, 1.0)], a=a, b=b, d=spec["d"], beta0=beta, row_weight=spec.get("row_weight"),
                                   n_eff=spec.get("n_eff"))], tol=tol, want_group_norms=True, flags=_engine.FLAG_ON_CHIP)[0]
        assert res.converged
        if res.mode[0] != 2:  # the chip gave the round up (the general path solved it): the launch of all rounds refuses too
            return None, 0
        beta = res.betas[0].copy()
        out.append(beta)
        na, nb = a.copy(), b.copy()
        if coef_scale != 0.0:
            na[:n_coef] = coef_scale * (numer / (np.abs(beta[:n_coef]) + eps))
        if group_scale is not None:
            nb[:n_group] = np.asarray(group_scale) * (numer / (res.group_norms[0][:n_group] + eps))
        moved = np.sqrt(np.sum((na - a) ** 2) + np.sum((nb - b) ** 2))
        a, b = na, nb
        if moved <= rtol:
            break
    return out, len(out)


def test_re_weighted_rounds_in_one_launch_are_the_host_loop_bit_for_bit(eng):
    rng = np.random.default_rng(5)
    n, p, G = 100, 80, 8
    X = rng.standard_normal((n, p))
    coef = np.zeros(p)
    coef[rng.choice(p, 10, replace=False)] = rng.uniform(1, 5, 10)
    y = X @ coef + 0.5 * rng.standard_normal(n)
    groups = rng.permutation(np.arange(p) % G)
    gw = rng.uniform(0.5, 2.0, G)
    mask = (np.arange(n) % 5 != 2).astype(float)
    tol = 1e-10
    cases = []  # (groups, spec, rule, rounds)
    for alpha in (1e-3, 0.3, 3.0):
        cases.append((None, dict(a=alpha * np.ones(p), b=np.zeros(p), d=np.zeros(p)), (alpha, None, alpha, 1e-6, 1e-10, p, 0), 3))
        cases.append((groups, dict(a=np.zeros(p), b=alpha * np.ones(G), d=np.zeros(G)), (0.0, alpha * gw, alpha, 1e-6, 1e-10, 0, G), 4))
        cases.append((groups, dict(a=0.3 * alpha * np.ones(p), b=0.7 * alpha * np.ones(G), d=0.1 * np.ones(G), row_weight=mask, n_eff=int(mask.sum())),
                      (0.3 * alpha, 0.7 * alpha * gw, alpha, 1e-6, 1e-10, p, G), 3))
    # rounds that end early: the weights of an all-zero solution do not move after the second round
    cases.append((None, dict(a=50.0 * np.ones(p), b=np.zeros(p), d=np.zeros(p)), (50.0, None, 50.0, 1e-6, 1e-10, p, 0), 5))
    refused = 0
    with eng.dataset(X, y) as ds:
        for grp, spec, rule, rounds in cases:
            ds.set_groups(grp, None if grp is None else G)
            want, r_want = _host_rounds(ds, spec, rule, rounds, tol)
            if want is None:
                with pytest.raises(NotImplementedError):
                    ds.solve_lanes_reweighted([dict(points=np.ones((rounds, 3)), reweight=rule, **spec)], tol=tol)
                refused += 1
                continue
            (res,), (r,) = ds.solve_lanes_reweighted([dict(points=np.ones((rounds, 3)), reweight=rule, **spec)], tol=tol)
            assert r == r_want, (rule, r, r_want)
            for k in range(r):
                np.testing.assert_array_equal(res.betas[k], want[k])
                assert res.mode[k] == 2 and res.n_iter[k] > 0
            for k in range(r, rounds):
                assert res.mode[k] == 3 and res.n_iter[k] == 0
        assert r_want < 5  # (the last case did end early)
        assert refused <= 3, refused
        # many lanes, each its own rule and number of rounds, in one launch
        ds.set_groups(None, None)
        alphas = np.geomspace(0.02, 5.0, 40)
        specs = [dict(points=np.ones((2 + i % 3, 3)), a=al * np.ones(p), b=np.zeros(p), d=np.zeros(p), reweight=(al, None, al, 1e-6, 1e-10, p, 0))
                 for i, al in enumerate(alphas)]
        results, rounds = ds.solve_lanes_reweighted(specs, tol=tol)
        assert results[0].grad_launches == 1
        for i in (0, 7, 22, 39):
            want, r_want = _host_rounds(ds, specs[i], specs[i]["reweight"], 2 + i % 3, tol)
            assert want is not None and rounds[i] == r_want
            np.testing.assert_array_equal(results[i].betas[rounds[i] - 1], want[-1])


def test_re_weighted_rounds_are_refused_where_the_chip_does_not_apply(eng):
    rng = np.random.default_rng(6)
    X = rng.standard_normal((300, 200))
    y = rng.standard_normal(300)
    with eng.dataset(X, y) as ds:  # p > 128: the caller keeps its loop
        with pytest.raises(NotImplementedError):
            ds.solve_lanes_reweighted([dict(points=np.ones((3, 3)), a=np.ones(200), reweight=(1.0, None, 1.0, 1e-6, 1e-10, 200, 0))])
    X = rng.standard_normal((40, 30))
    y = rng.standard_normal(40)
    with eng.dataset(X, y) as ds:
        with pytest.raises(ValueError):
            ds.solve_lanes_reweighted([dict(points=np.ones((3, 3)), a=np.ones(30), reweight=(1.0, None, 1.0, -1.0, 1e-10, 30, 0))])
        with pytest.raises(ValueError):
            ds.solve_lanes_reweighted([dict(points=np.ones((3, 3)), a=np.ones(30), reweight=(1.0, None, 1.0, 1e-6, 1e-10, 31, 0))])


@pytest.mark.parametrize("kind", ["lasso", "group", "sparse_group", "ridged", "overlap"])
def test_adaptive_searches_with_rounds_on_chip_equal_the_host_loop(kind, monkeypatch):
    """GridSearchCV over the Adaptive* estimators at the README's size: rounds inside the launch against the loop of calls
    (SLM_HOST_ROUNDS=1) -- the same scores, the same refit."""
    from sklearn.datasets import make_regression

    from sparselm_amd import model
    from sparselm_amd.model_selection import GridSearchCV

    X, y = make_regression(n_samples=100, n_features=60, n_informative=10, random_state=3)
    groups = np.arange(60) % 12
    est = {
        "lasso": lambda: model.AdaptiveLasso(fit_intercept=True),
        "group": lambda: model.AdaptiveGroupLasso(groups=groups, group_weights=np.linspace(0.5, 2.0, 12)),
        "sparse_group": lambda: model.AdaptiveSparseGroupLasso(groups=groups, l1_ratio=0.4, max_iter=4),
        "ridged": lambda: model.AdaptiveRidgedGroupLasso(groups=groups, delta=(0.5,), fit_intercept=True),
        "overlap": lambda: model.AdaptiveOverlapGroupLasso(group_list=[[g, (g + 1) % 12] for g in groups], max_iter=2),
    }[kind]
    grid = {"alpha": np.logspace(-3, 1, 6)}
    import warnings

    calls = []
    inner = _engine.Dataset.solve_lanes_reweighted

    def counted(self, lanes, **kw):
        out = inner(self, lanes, **kw)
        calls.append(len(lanes))
        return out

    monkeypatch.setattr(_engine.Dataset, "solve_lanes_reweighted", counted)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        on_chip = GridSearchCV(est(), grid, cv=4).fit(X, y)
        # the 6 x 4 cells in one launch, the refit in another (the overlap estimator duplicates columns on the host: its
        # search goes cell by cell through fit, a launch per fit where the duplicated columns' rounds settle on chip)
        if kind == "overlap":
            assert calls and set(calls) == {1}, calls
        else:
            assert calls == [24, 1], calls
        n_calls = len(calls)
        monkeypatch.setenv("SLM_HOST_ROUNDS", "1")
        host = GridSearchCV(est(), grid, cv=4).fit(X, y)
        assert len(calls) == n_calls
    np.testing.assert_array_equal(on_chip.cv_results_["mean_test_score"], host.cv_results_["mean_test_score"])
    np.testing.assert_array_equal(on_chip.best_estimator_.coef_, host.best_estimator_.coef_)
    np.testing.assert_array_equal(on_chip.best_estimator_.adaptive_weights_, host.best_estimator_.adaptive_weights_)
    assert on_chip.best_estimator_.n_iter_ == host.best_estimator_.n_iter_


def test_a_rule_with_a_zero_scale_still_counts_its_round(monkeypatch):
    """AdaptiveLasso(alpha=0): the rule's coefficient scale is zero, the renewed weights are the old ones (zero) and the loop
    ends after ONE round -- as the reference's loop does (model/_adaptive_lasso.py:206-232: `n_iter_` = 1), and as the
    loop of calls does here (SLM_HOST_ROUNDS=1).  (Round-4 advice: the kernel used to count no round at all there, the
    estimator then indexed its results with -1 and reported `n_iter_` = 0.)"""
    from sparselm_amd.model import AdaptiveLasso

    rng = np.random.default_rng(2)
    X = rng.standard_normal((40, 12))
    y = X @ rng.standard_normal(12) + 0.1 * rng.standard_normal(40)
    chip = AdaptiveLasso(alpha=0.0, max_iter=4, fit_intercept=False).fit(X, y)
    monkeypatch.setenv("SLM_HOST_ROUNDS", "1")
    host = AdaptiveLasso(alpha=0.0, max_iter=4, fit_intercept=False).fit(X, y)
    assert chip.n_iter_ == host.n_iter_ == 1
    np.testing.assert_allclose(chip.coef_, host.coef_, rtol=0, atol=1e-9 * np.max(np.abs(host.coef_)))
    np.testing.assert_allclose(chip.coef_, np.linalg.lstsq(X, y, rcond=None)[0], rtol=0, atol=1e-6 * np.max(np.abs(host.coef_)))

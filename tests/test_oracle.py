"""The oracle against its pins: reference known answers, scikit-learn CD, closed forms, KKT.

CPU only.  These tests are what entitles the oracle to judge the HIP engine (DESIGN.md "Oracle").
"""

import numpy as np
import numpy.testing as npt
import pytest

import oracle
from oracle import cref


def test_reference_toy_known_answers(golden):
    # /root/reference/tests/test_lasso.py:29-61 (6 decimals there)
    X, y, T = golden["toy_X"], golden["toy_y"], golden["toy_T"]
    for alpha, coef, pred in zip(golden["toy_alpha"], golden["toy_coef"], golden["toy_pred"]):
        r = oracle.fit_lasso(X, y, alpha)
        npt.assert_array_almost_equal(r["coef"], [coef])
        npt.assert_array_almost_equal(T @ r["coef"] + r["intercept"], pred)


def test_lasso_vs_sklearn_cd(golden):
    X, y = golden["l1_X"], golden["l1_y"]
    for k, alpha in enumerate(golden["l1_alpha"]):
        r = oracle.fit_lasso(X, y, alpha)
        npt.assert_allclose(r["coef"], golden["l1_coef"][k], rtol=0, atol=1e-10 * np.max(abs(golden["l1_coef"][k])) + 1e-13)
        r = oracle.fit_lasso(X, y, alpha, fit_intercept=True)
        npt.assert_allclose(r["coef"], golden["l1_coef_icpt"][k], rtol=0, atol=1e-9)
        npt.assert_allclose(r["intercept"], golden["l1_icpt"][k], rtol=1e-10)
        r = oracle.fit_lasso(X, y, alpha, fit_intercept=True, sample_weight=golden["l1_sw"])
        npt.assert_allclose(r["coef"], golden["l1_coef_sw"][k], rtol=0, atol=1e-9)
        npt.assert_allclose(r["intercept"], golden["l1_icpt_sw"][k], rtol=1e-10)


def test_weighted_l1_vs_sklearn_cd(golden):
    X, y, w = golden["l1_X"], golden["l1_y"], golden["wl1_w"]
    p = X.shape[1]
    gidx, G = oracle.group_index(None, p)
    beta, info = oracle.fista(X, y, w, 0.0, 0.0, gidx, G)
    assert info["converged"]
    npt.assert_allclose(beta, golden["wl1_coef"], rtol=0, atol=1e-10)


def test_weighted_least_squares_closed_form(golden):
    # /root/reference/tests/test_ols.py:35-66, reached as the alpha = 0 member of the family
    X, y, sw = golden["ols_X"], golden["ols_y"], golden["ols_sw"]
    r = oracle.fit_lasso(X, y, 0.0, sample_weight=sw)
    npt.assert_allclose(r["coef"], golden["ols_coef"], rtol=1e-7)
    r = oracle.fit_lasso(X, y, 0.0, sample_weight=sw, fit_intercept=True)
    npt.assert_allclose(r["coef"], golden["ols_coef_icpt"], rtol=1e-7)
    npt.assert_allclose(r["intercept"], golden["ols_icpt"], rtol=1e-7)


def test_adaptive_lasso_sequence_vs_sklearn_cd(golden):
    # inner solves pinned by sklearn on X/w; weights by alpha * alpha/(|b|+eps)  (Appendix A-7)
    X, y = golden["l1_X"], golden["l1_y"]
    alpha = float(golden["ada_sk_alpha"])
    for k in (1, 2, 3):
        r = oracle.fit_adaptive_lasso(X, y, alpha=alpha, max_iter=k)
        npt.assert_allclose(r["coef"], golden["ada_sk_coefs"][k - 1], rtol=0, atol=2e-9)
        assert r["n_iter"] == k
        npt.assert_allclose(r["weights"], golden["ada_sk_weights"][k], rtol=1e-6)


@pytest.mark.parametrize("kind", ["gl", "sgl", "rgl"])
def test_group_fits_are_kkt_certified(golden, kind):
    X, y, groups, gw = golden["grp_X"], golden["grp_y"], golden["grp_groups"], golden["grp_gw"]
    alpha = float(golden["grp_alpha"])
    n, p = X.shape
    gidx, G = oracle.group_index(groups, p)
    if kind == "gl":
        r = oracle.fit_group_lasso(X, y, groups=groups, alpha=alpha, group_weights=gw)
        a, b, d = np.zeros(p), alpha * gw, np.zeros(G)
    elif kind == "sgl":
        r = oracle.fit_sparse_group_lasso(X, y, groups=groups, l1_ratio=0.3, alpha=alpha, group_weights=gw)
        a, b, d = 0.3 * alpha * np.ones(p), 0.7 * alpha * gw, np.zeros(G)
    else:
        r = oracle.fit_ridged_group_lasso(X, y, groups=groups, alpha=alpha, delta=golden["grp_delta"], group_weights=gw)
        a, b, d = np.zeros(p), alpha * gw, golden["grp_delta"]
    beta = r["coef"]
    npt.assert_allclose(beta, golden[f"grp_{kind}_coef"], rtol=0, atol=1e-10)
    grad = X.T @ (X @ beta - y) / n
    mu = np.linalg.eigvalsh(X.T @ X / n)[0]
    assert mu > 0.1
    assert oracle.kkt_residual(grad, beta, a, b, d, gidx, G) / mu < 1e-9 * np.linalg.norm(beta)
    assert oracle.prox_fixed_point_residual(grad, beta, a, b, d, gidx, G) < 1e-9
    # group all-or-nothing structure (reference tests/test_lasso.py:106-111) for the pure group penalties
    if kind != "sgl":
        for g in range(G):
            m = beta[gidx == g] != 0
            assert m.all() or (~m).all()


@pytest.mark.parametrize("kind", ["gl", "sgl", "rgl", "ada_gl", "ada_sgl", "ada_rgl"])
def test_group_family_agrees_with_the_second_solver(golden, kind):
    """The second pin of the group family: tests/golden/second_solver.py (active-set method, Newton's method
    on the optimality conditions of the face; no code shared with the oracle's proximal-gradient iteration) is
    run here AND its committed results are compared -- oracle, fresh second solve and fixture agree to 1e-8."""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import second_solver as S

    X, y, groups, gw = golden["grp_X"], golden["grp_y"], golden["grp_groups"], golden["grp_gw"]
    alpha = float(golden["grp_alpha"])
    if kind == "gl":
        mine = oracle.fit_group_lasso(X, y, groups=groups, alpha=alpha, group_weights=gw)["coef"]
        other = S.group_lasso(X, y, groups, alpha, gw)
        stored = golden["grp_gl_coef2"]
    elif kind == "sgl":
        mine = oracle.fit_sparse_group_lasso(X, y, groups=groups, l1_ratio=0.3, alpha=alpha, group_weights=gw)["coef"]
        other = S.group_lasso(X, y, groups, alpha, gw, l1_ratio=0.3)
        stored = golden["grp_sgl_coef2"]
    elif kind == "rgl":
        mine = oracle.fit_ridged_group_lasso(X, y, groups=groups, alpha=alpha, delta=golden["grp_delta"], group_weights=gw)["coef"]
        other = S.group_lasso(X, y, groups, alpha, gw, delta=golden["grp_delta"])
        stored = golden["grp_rgl_coef2"]
    else:
        kw = {"ada_gl": {}, "ada_sgl": {"l1_ratio": 0.4}, "ada_rgl": {"delta": (0.7,)}}[kind]
        fit = {"ada_gl": oracle.fit_adaptive_group_lasso, "ada_sgl": oracle.fit_adaptive_sparse_group_lasso,
               "ada_rgl": oracle.fit_adaptive_ridged_group_lasso}[kind]
        r = fit(X, y, groups=groups, alpha=1.5, group_weights=gw, fit_intercept=True, **kw)
        r2 = S.adaptive(X, y, groups, 1.5, gw, fit_intercept=True, **kw)
        assert r["n_iter"] == r2["n_iter"] == int(golden[f"{kind}_niter2"])
        npt.assert_allclose(r["intercept"], r2["intercept"], rtol=1e-9)
        mine, other, stored = r["coef"], r2["coef"], golden[f"{kind}_coef2"]
    scale = np.max(np.abs(other))
    assert np.max(np.abs(mine - other)) <= 1e-8 * scale
    assert np.max(np.abs(other - stored)) <= 1e-8 * scale
    assert np.array_equal(mine != 0, other != 0)  # same support, exact zeros on both sides


def test_standardized_sparse_group_solution_is_certified_and_meets_its_special_cases(golden):
    """oracle/primal_dual.py: the primal-dual solution of the standardised sparse-group problem (reference
    model/_lasso.py:627-639 with :249-252) satisfies the optimality conditions of the original problem
    (certificate computed from the coefficients alone), is the fixture, and at the ends of l1_ratio is the
    Lasso (scikit-learn's coordinate descent) and the standardised GroupLasso (whitened FISTA)."""
    from sklearn.linear_model import Lasso as SkLasso

    X, y, groups, gw = golden["grp_X"], golden["grp_y"], golden["grp_groups"], golden["grp_gw"]
    n, p = X.shape
    gidx, G = oracle.group_index(groups, p)
    scale = np.max(np.abs(X.T @ y)) / n
    r = oracle.fit_sparse_group_lasso(X, y, groups=groups, l1_ratio=0.5, alpha=0.4, group_weights=gw, standardize=True)
    assert r["info"]["converged"]
    npt.assert_allclose(r["coef"], golden["std_sgl_coef"], rtol=0, atol=1e-10 * np.max(np.abs(r["coef"])))
    assert oracle.kkt_standardized(X, y, 0.2 * np.ones(p), 0.2 * gw, gidx, G, r["coef"]) < 1e-11 * scale
    # a perturbed point is not certified
    bad = r["coef"].copy()
    bad[np.argmax(np.abs(bad))] *= 1.001
    assert oracle.kkt_standardized(X, y, 0.2 * np.ones(p), 0.2 * gw, gidx, G, bad) > 1e-6 * scale
    # l1_ratio = 1: the Lasso
    b1, _ = oracle.standardized_sparse_group(X, y, 0.4 * np.ones(p), np.zeros(G), gidx, G)
    sk = SkLasso(alpha=0.4, fit_intercept=False, tol=1e-15, max_iter=1_000_000).fit(X, y).coef_
    npt.assert_allclose(b1, sk, rtol=0, atol=1e-9 * np.max(np.abs(sk)))
    # l1_ratio = 0: the standardised group lasso, solved in whitened coordinates by the FISTA oracle
    b0, _ = oracle.standardized_sparse_group(X, y, np.zeros(p), 0.4 * gw, gidx, G)
    gl = oracle.fit_group_lasso(X, y, groups=groups, alpha=0.4, group_weights=gw, standardize=True)["coef"]
    npt.assert_allclose(b0, gl, rtol=0, atol=1e-8 * np.max(np.abs(gl)))


def test_prox_matches_bruteforce_minimiser(rng):
    # prox_s(v) = argmin_u 1/2||u - v||^2 + s*pen(u): check optimality by perturbation on a tiny case
    p = 7
    gidx = np.array([0, 0, 1, 1, 1, 2, 2])
    a = rng.uniform(0, 0.5, p)
    b = rng.uniform(0, 0.8, 3)
    d = rng.uniform(0, 1.0, 3)
    for _ in range(20):
        v = rng.normal(size=p)
        s = rng.uniform(0.1, 2.0)
        u = oracle.prox(v, s, a, b, d, gidx, 3)
        f = lambda x: 0.5 * np.sum((x - v) ** 2) + s * oracle.penalty_value(x, a, b, d, gidx, 3)
        f0 = f(u)
        for _ in range(200):
            assert f(u + 1e-4 * rng.normal(size=p)) >= f0 - 1e-12


def test_group_index_follows_sorted_unique_labels():
    gidx, G = oracle.group_index([7, 3, 7, 10, 3], 5)
    assert G == 3 and list(gidx) == [1, 0, 1, 2, 0]
    gidx, G = oracle.group_index(None, 4)
    assert G == 4 and list(gidx) == [0, 1, 2, 3]


def test_c_twin_matches_numpy_oracle(golden):
    X, y, groups, gw = golden["grp_X"], golden["grp_y"], golden["grp_groups"], golden["grp_gw"]
    n, p = X.shape
    gidx, G = oracle.group_index(groups, p)
    z = np.linspace(-1, 1, p)
    g, loss = cref.gradient(X, y, z)
    npt.assert_allclose(g, X.T @ (X @ z - y) / n, rtol=1e-12, atol=1e-12)
    npt.assert_allclose(loss, 0.5 * np.sum((X @ z - y) ** 2) / n, rtol=1e-13)
    w = np.linspace(0.5, 1.5, n)
    g, loss = cref.gradient(X, y, z, w)
    npt.assert_allclose(g, X.T @ (w * (X @ z - y)) / n, rtol=1e-12, atol=1e-12)
    alpha = float(golden["grp_alpha"])
    L = oracle.lipschitz(X)
    beta_c, it = cref.fista(X, y, 0.3 * alpha, 0.7 * alpha * gw, 0.0, gidx, G, L=L)
    assert it > 0
    npt.assert_allclose(beta_c, golden["grp_sgl_coef"], rtol=0, atol=1e-10)

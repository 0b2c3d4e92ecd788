"""The direct (Cholesky) step of the working-set model solver and the KKT-based stopping rule
(sparse-lm_amd/csrc/newton_kernels.hpp, ws_kernels.hpp `direct_step`, tail_kernels.hpp stopping rule).

The minimiser the reference returns (cvxpy's interior-point solve, src/sparselm/model/_base.py:512-519) does
not depend on the conditioning of X; a proximal-gradient stop on the step length does.  These tests pin the
two pieces that close that gap: the one-workgroup dense solve against numpy, and Lasso paths on strongly
correlated designs against scikit-learn's coordinate descent (same objective, `1/(2n)` scaling) at DEFAULT
solver options.
"""

import warnings

import numpy as np
import pytest

import oracle
from sparselm_amd import _engine

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    return _engine.get_engine(0)


def _spd(m, cond, seed):
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.standard_normal((m, m)))
    ev = np.geomspace(1.0, 1.0 / cond, m)
    return (Q * ev) @ Q.T, ev


@pytest.mark.parametrize("m", [1, 5, 16, 17, 37, 96, 200, 333, 512])
def test_dense_spd_solve_matches_numpy(eng, m):
    H, ev = _spd(m, 1e3, seed=m)
    rng = np.random.default_rng(m + 1)
    rhs = rng.standard_normal(m)
    x, mu = eng.dense_spd_solve(H, rhs)
    ref = np.linalg.solve(H, rhs)
    assert np.max(np.abs(x - ref)) <= 1e-10 * np.max(np.abs(ref))
    # two inverse-iteration steps from the solution: an estimate of lambda_min from above, within a small factor
    assert ev[-1] * (1 - 1e-9) <= mu <= 30 * ev[-1]


def test_dense_spd_solve_on_an_ill_conditioned_gram(eng):
    """The kind of matrix the model solver meets: the Gram of AR(1) columns with rho = 0.95."""
    rng = np.random.default_rng(0)
    n, m = 4000, 160
    E = rng.standard_normal((n, m))
    X = E.copy()
    for j in range(1, m):
        X[:, j] = 0.95 * X[:, j - 1] + np.sqrt(1 - 0.95**2) * E[:, j]
    H = X.T @ X / n
    rhs = rng.standard_normal(m)
    x, mu = eng.dense_spd_solve(H, rhs)
    ref = np.linalg.solve(H, rhs)
    assert np.max(np.abs(x - ref)) <= 1e-9 * np.max(np.abs(ref))
    lam = np.linalg.eigvalsh(H)[0]
    assert lam * (1 - 1e-9) <= mu <= 10 * lam


def test_dense_spd_solve_refuses_a_singular_matrix(eng):
    rng = np.random.default_rng(1)
    A = rng.standard_normal((40, 20))
    H = A @ A.T  # rank 20 of 40
    with pytest.raises(ValueError, match="positive definite"):
        eng.dense_spd_solve(H, np.ones(40))


def _ar1(rng, n, p, rho):
    E = rng.standard_normal((n, p))
    X = E.copy()
    for j in range(1, p):
        X[:, j] = rho * X[:, j - 1] + np.sqrt(1 - rho**2) * E[:, j]
    return X


def _low_rank(rng, n, p, r=8):
    return rng.standard_normal((n, r)) @ rng.standard_normal((r, p)) * 2.0 + 0.3 * rng.standard_normal((n, p))


def _sklearn_path(X, y, alphas):
    from sklearn.linear_model import lasso_path

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        _, ref, _ = lasso_path(X, y, alphas=alphas, precompute=X.T @ X, Xy=X.T @ y, tol=1e-14, max_iter=400000)
    return ref.T


@pytest.mark.parametrize("design", ["ar1_0.95", "lowrank+noise"])
def test_correlated_large_designs_reach_1e6_at_default_options(eng, design):
    """n = 70 000, p = 1 200 (large enough for the working set to run from the first pass), 30-alpha path to
    1e-3 alpha_max, DEFAULT tol: <= 1e-6 rel-inf against scikit-learn's coordinate descent at dual gap 1e-14,
    in at most 1.3 passes over X per path point (round 1: 1e-5 / 5e-4 and 271 / 1670 passes)."""
    rng = np.random.default_rng(0)
    n, p, K = 70_000, 1_200, 30
    X = _ar1(rng, n, p, 0.95) if design == "ar1_0.95" else _low_rank(rng, n, p)
    coef = np.zeros(p)
    coef[rng.choice(p, 25, replace=False)] = rng.standard_normal(25) * 3
    y = X @ coef + rng.standard_normal(n) * 2
    with eng.dataset(X, y) as ds:
        g0, _ = ds.gradient(None)
        amax = float(np.max(np.abs(g0)))
        alphas = np.geomspace(amax, 1e-3 * amax, K)
        res = ds.solve_path([(a, 0.0, 0.0) for a in alphas], lanes=16, flags=_engine.FLAG_FRESH_L)
    ref = _sklearn_path(X, y, alphas)
    err = np.max(np.abs(res.betas - ref)) / np.max(np.abs(ref))
    assert res.converged
    assert err <= 1e-6, err
    assert res.grad_launches <= 1.3 * K + 1, res.grad_launches
    assert res.ws_direct_steps > 0
    # the certificate that went with every accepted point: KKT residual over the strong-convexity estimate
    ok = res.mu > 0
    assert ok.all()
    assert np.all(res.kkt[1:] <= 1.01e-8 * res.mu[1:] * np.maximum(res.beta_norm[1:], 1e-300))


def test_direct_step_on_a_small_ill_conditioned_problem_matches_oracle(eng):
    """Working set forced on a small AR(1) design (K <= 112: the Gram sits in LDS): the direct step has to
    produce what the plain iteration converges to, in a handful of passes."""
    rng = np.random.default_rng(5)
    n, p = 3000, 300
    X = _ar1(rng, n, p, 0.97)
    coef = np.zeros(p)
    coef[rng.choice(p, 10, replace=False)] = rng.standard_normal(10) * 2
    y = X @ coef + 0.5 * rng.standard_normal(n)
    amax = float(np.max(np.abs(X.T @ y)) / n)
    alphas = np.geomspace(amax, 1e-2 * amax, 10)
    pts = [(a, 0.0, 0.0) for a in alphas]
    with eng.dataset(X, y) as ds:
        r = ds.solve_path(pts, lanes=4, flags=_engine.FLAG_WORKING_SET, tol=1e-9)
    assert r.converged
    ref = _sklearn_path(X, y, alphas)
    assert np.max(np.abs(r.betas - ref)) <= 1e-7 * np.max(np.abs(ref))
    assert r.grad_launches <= 3 * len(alphas)


def test_weighted_l1_with_ridge_takes_direct_steps(eng):
    """Singleton groups with b and d: the face Hessian is G_AA + diag(d), the face gradient carries (a + b) s."""
    rng = np.random.default_rng(7)
    n, p = 4000, 200
    X = _ar1(rng, n, p, 0.98)
    y = X[:, :5] @ np.array([2.0, -1.0, 1.5, 0.5, -2.0]) + 0.3 * rng.standard_normal(n)
    a = rng.uniform(0.5, 1.5, p) * 0.05
    b = rng.uniform(0.0, 1.0, p) * 0.02
    d = rng.uniform(0.0, 1.0, p) * 0.01
    with eng.dataset(X, y) as ds:
        r = ds.solve_path([(1.0, 1.0, 1.0)], a=a, b=b, d=d, flags=_engine.FLAG_WORKING_SET, tol=1e-10)
    assert r.converged
    gidx, G = oracle.group_index(None, p)
    ref, _ = oracle.fista(X, y, a, b, d, gidx, G, tol=1e-14, max_iter=400000)
    assert np.max(np.abs(r.betas[0] - ref)) <= 1e-7 * np.max(np.abs(ref))


@pytest.mark.parametrize("design", ["ar1_0.95", "lowrank+noise"])
@pytest.mark.parametrize("kind", ["group", "sparse_group", "ridged"])
def test_group_penalties_on_correlated_designs_take_newton_steps(eng, design, kind):
    """The direct step with real group norms (curvature (b_g / r_g)(I - u u^T) of the active groups in the face
    Hessian): GroupLasso / SparseGroupLasso / RidgedGroupLasso paths on strongly correlated columns, groups of
    eight, at DEFAULT options: within 1e-6 rel-inf of the plain iteration run to a much tighter tolerance, KKT
    residuals (oracle.kkt_residual on one more device gradient) small against the smallest eigenvalue of the
    Gram -- the strong-convexity modulus of the WHOLE objective, a pessimistic stand-in for the one on the face,
    which the group norms' own curvature lifts -- and a few passes per path (round 1: 45 ... 76 passes on the
    AR(1) design, 130 ms on the other)."""
    rng = np.random.default_rng(1)
    n, p, K = 40_000, 1_600, 12
    X = _ar1(rng, n, p, 0.95) if design == "ar1_0.95" else _low_rank(rng, n, p)
    groups = np.repeat(np.arange(p // 8), 8)
    G = p // 8
    coef = np.zeros(p)
    for g in rng.choice(G, 6, replace=False):
        coef[groups == g] = rng.standard_normal(8) * 2
    y = X @ coef + rng.standard_normal(n) * 2
    mu = float(np.linalg.eigvalsh(X.T @ X / n)[0])
    assert mu > 1e-3
    with eng.dataset(X, y) as ds:
        ds.set_groups(groups, G)
        g0, _ = ds.gradient(None)
        bmax = float(np.max(np.sqrt(np.bincount(groups, weights=g0 * g0))))
        al = np.geomspace(bmax, 1e-2 * bmax, K)
        if kind == "group":
            pts, d = [(0.0, a, 0.0) for a in al], None
        elif kind == "sparse_group":
            pts, d = [(0.3 * a, 0.7 * a, 0.0) for a in al], None
        else:
            pts, d = [(0.0, a, 1.0) for a in al], 0.05 * np.ones(G)
        res = ds.solve_path(pts, d=d, lanes=12, flags=_engine.FLAG_FRESH_L | _engine.FLAG_WORKING_SET, max_iter=20000)
        assert res.converged
        assert res.ws_direct_steps > 0
        assert res.grad_launches <= 3 * K, res.grad_launches
        ref = ds.solve_path(pts, d=d, lanes=4, flags=_engine.FLAG_NO_WORKING_SET, tol=1e-11, max_iter=400000)
        assert ref.converged
        assert np.max(np.abs(res.betas - ref.betas)) <= 1e-6 * np.max(np.abs(ref.betas))
        for k in (1, K // 2, K - 1):
            beta = res.betas[k]
            g, _ = ds.gradient(beta)
            sa, sb, sd = pts[k]
            kkt = oracle.kkt_residual(g, beta, sa * np.ones(p), sb * np.ones(G), sd * (d if d is not None else np.zeros(G)),
                                      groups, G)
            assert kkt / mu <= 1e-5 * np.max(np.abs(beta)), (k, kkt, mu)
        if kind != "sparse_group":  # group all-or-nothing (reference tests/test_lasso.py:106-111)
            active = np.bincount(groups, weights=(res.betas[-1] != 0), minlength=G)
            assert np.all((active == 0) | (active == 8))


def test_lanes_left_to_the_solver_with_direct_steps_give_what_one_kernel_gives(eng, monkeypatch):
    """The model solver is two launches: the iteration alone, then -- for the lanes that would take a direct step --
    the instance that carries them, which starts those solves over (ws_kernels.hpp, `ws_refine_lane<GROUPED, DIRECT>`).
    Nothing of an abandoned solve may leak: every lane on the second instance (SLM_WS_ONE_SOLVER=1) gives the same bits,
    and so does a second run."""
    rng = np.random.default_rng(11)
    n, p = 3000, 300
    X = _ar1(rng, n, p, 0.97)
    coef = np.zeros(p)
    coef[rng.choice(p, 10, replace=False)] = rng.standard_normal(10) * 2
    y = X @ coef + 0.5 * rng.standard_normal(n)
    amax = float(np.max(np.abs(X.T @ y)) / n)
    pts = [(a, 0.0, 0.0) for a in np.geomspace(amax, 1e-2 * amax, 12)]
    with eng.dataset(X, y) as ds:
        two = ds.solve_path(pts, lanes=4, flags=_engine.FLAG_WORKING_SET, tol=1e-9)
        again = ds.solve_path(pts, lanes=4, flags=_engine.FLAG_WORKING_SET, tol=1e-9)
        monkeypatch.setenv("SLM_WS_ONE_SOLVER", "1")
        one = ds.solve_path(pts, lanes=4, flags=_engine.FLAG_WORKING_SET, tol=1e-9)
    assert two.converged and one.converged
    assert two.ws_direct_steps > 0  # the design needs them: the light instance handed lanes over
    np.testing.assert_array_equal(two.betas, again.betas)
    np.testing.assert_array_equal(two.betas, one.betas)
    assert two.grad_launches == one.grad_launches and two.ws_direct_steps == one.ws_direct_steps


def test_spectral_steps_of_the_model_solver_save_iterations_not_accuracy(eng, monkeypatch):
    """Easy design: the model solver's spectral opening (default) against accelerated steps throughout (SLM_WS_BB=0):
    same passes, same solution to the solver's tolerance, fewer model iterations, no direct step either way."""
    rng = np.random.default_rng(12)
    n, p = 12000, 500  # (p / n as on the headline problem: faces with a condition number of 2-3)
    X = rng.standard_normal((n, p))
    coef = np.zeros(p)
    coef[rng.choice(p, 25, replace=False)] = 5 * rng.standard_normal(25)
    y = X @ coef + rng.standard_normal(n)
    amax = float(np.max(np.abs(X.T @ y)) / n)
    pts = [(a, 0.0, 0.0) for a in np.geomspace(amax, 1e-2 * amax, 24)]
    with eng.dataset(X, y) as ds:
        bb = ds.solve_path(pts, lanes=8, flags=_engine.FLAG_WORKING_SET, tol=1e-10)
        monkeypatch.setenv("SLM_WS_BB", "0")
        acc = ds.solve_path(pts, lanes=8, flags=_engine.FLAG_WORKING_SET, tol=1e-10)
    assert bb.converged and acc.converged
    assert np.max(np.abs(bb.betas - acc.betas)) <= 1e-8 * np.max(np.abs(acc.betas))
    assert bb.grad_launches <= acc.grad_launches + 1
    assert bb.ws_direct_steps == 0 and acc.ws_direct_steps == 0
    assert bb.ws_inner_iters < acc.ws_inner_iters

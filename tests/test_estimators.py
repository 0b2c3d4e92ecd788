"""Estimator surface: reference behaviours re-expressed on sparselm_amd.model.

Every test runs twice: with the CPU oracle injected behind the surface (``-m "not gpu"``: checks the
host logic -- validation, preprocessing, re-weighting loops) and through the real HIP engine
(``-m gpu``: the parity tests proper, calling through the C ABI).  Expected values come from the
committed golden fixtures and from the oracle's restatement of the reference semantics.

Mirrors /root/reference/tests/test_lasso.py and tests/test_common.py.
"""

import warnings
from inspect import signature

import numpy as np
import numpy.testing as npt
import pytest
from sklearn.model_selection import GridSearchCV

import oracle
from _oracle_backend import OracleBackend
from sparselm_amd import _backend
from sparselm_amd import model as spm
from sparselm_amd.model import (
    AdaptiveGroupLasso,
    AdaptiveLasso,
    AdaptiveOverlapGroupLasso,
    AdaptiveRidgedGroupLasso,
    AdaptiveSparseGroupLasso,
    GroupLasso,
    Lasso,
    OverlapGroupLasso,
    RidgedGroupLasso,
    SparseGroupLasso,
)

THRESHOLD = 1e-8
TIGHT = {"tol": 1e-12, "max_iter": 200000}
ESTIMATORS = [getattr(spm, n) for n in spm.__all__]
ADAPTIVE = [AdaptiveLasso, AdaptiveGroupLasso, AdaptiveSparseGroupLasso, AdaptiveRidgedGroupLasso, AdaptiveOverlapGroupLasso]


@pytest.fixture(params=["oracle", pytest.param("hip", marks=pytest.mark.gpu)])
def backend(request):
    if request.param == "oracle":
        with _backend.use_backend(OracleBackend()):
            yield "oracle"
    else:
        yield "hip"


def rel_inf(a, b):
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


# ---- reference known answers --------------------------------------------------------------------
def test_lasso_toy(backend):
    # /root/reference/tests/test_lasso.py:29-61
    X = [[-1], [0], [1]]
    Y = [-1, 0, 1]
    T = [[2], [3], [4]]
    for alpha, coef, pred in [(1e-8, [1], [2, 3, 4]), (0.1, [0.85], [1.7, 2.55, 3.4]),
                              (0.5, [0.25], [0.5, 0.75, 1.0]), (1.0, [0.0], [0, 0, 0])]:
        lasso = Lasso(alpha=alpha)
        lasso.fit(X, Y)
        npt.assert_array_almost_equal(lasso.coef_, coef)
        npt.assert_array_almost_equal(lasso.predict(T), pred)


def test_lasso_non_float_y(backend):
    # /root/reference/tests/test_lasso.py:64-74
    X = [[0, 0], [1, 1], [-1, -1]]
    lasso = Lasso(fit_intercept=False).fit(X, [0, 1, 2])
    lasso_float = Lasso(fit_intercept=False).fit(X, [0.0, 1.0, 2.0])
    npt.assert_array_equal(lasso.coef_, lasso_float.coef_)


def test_weighted_least_squares_closed_form(backend, golden):
    # /root/reference/tests/test_ols.py:35-66 through the alpha = 0 member of the family
    X, y, sw = golden["ols_X"], golden["ols_y"], golden["ols_sw"]
    reg = Lasso(alpha=0.0, solver_options=TIGHT).fit(X, y, sample_weight=sw)
    npt.assert_allclose(reg.coef_, golden["ols_coef"], rtol=1e-7)
    reg = Lasso(alpha=0.0, fit_intercept=True, solver_options=TIGHT).fit(X, y, sample_weight=sw)
    npt.assert_allclose(reg.coef_, golden["ols_coef_icpt"], rtol=1e-7)
    npt.assert_allclose(reg.intercept_, golden["ols_icpt"], rtol=1e-7)


# ---- golden parity ----------------------------------------------------------------------------------
def test_lasso_matches_sklearn_golden(backend, golden):
    X, y = golden["l1_X"], golden["l1_y"]
    for k, alpha in enumerate(golden["l1_alpha"]):
        m = Lasso(alpha=alpha, solver_options=TIGHT).fit(X, y)
        assert rel_inf(m.coef_, golden["l1_coef"][k]) < 1e-9
        m = Lasso(alpha=alpha, fit_intercept=True, solver_options=TIGHT).fit(X, y, sample_weight=golden["l1_sw"])
        assert rel_inf(m.coef_, golden["l1_coef_sw"][k]) < 1e-9
        npt.assert_allclose(m.intercept_, golden["l1_icpt_sw"][k], rtol=1e-9)


def test_default_tolerance_meets_1e6(backend, golden):
    # north_star: coefficients within 1e-6 rel-inf at default settings
    X, y = golden["l1_X"], golden["l1_y"]
    for k, alpha in enumerate(golden["l1_alpha"]):
        m = Lasso(alpha=alpha).fit(X, y)
        assert rel_inf(m.coef_, golden["l1_coef"][k]) < 1e-6


def test_group_family_matches_golden(backend, golden):
    X, y, groups, gw = golden["grp_X"], golden["grp_y"], golden["grp_groups"], golden["grp_gw"]
    alpha = float(golden["grp_alpha"])
    m = GroupLasso(groups=groups, alpha=alpha, group_weights=gw, solver_options=TIGHT).fit(X, y)
    assert rel_inf(m.coef_, golden["grp_gl_coef"]) < 1e-9
    m = SparseGroupLasso(groups=groups, l1_ratio=0.3, alpha=alpha, group_weights=gw, solver_options=TIGHT).fit(X, y)
    assert rel_inf(m.coef_, golden["grp_sgl_coef"]) < 1e-9
    m = RidgedGroupLasso(groups=groups, alpha=alpha, delta=golden["grp_delta"], group_weights=gw, solver_options=TIGHT).fit(X, y)
    assert rel_inf(m.coef_, golden["grp_rgl_coef"]) < 1e-9
    # list labels and float labels are accepted like ndarray ones (reference _lasso.py:185)
    m2 = GroupLasso(groups=[float(g) for g in groups], alpha=alpha, group_weights=list(gw), solver_options=TIGHT).fit(X, y)
    assert rel_inf(m2.coef_, golden["grp_gl_coef"]) < 1e-9


def test_adaptive_family_matches_golden(backend, golden):
    X, y, groups, gw = golden["grp_X"], golden["grp_y"], golden["grp_groups"], golden["grp_gw"]
    cases = [
        ("ada_l", AdaptiveLasso(alpha=1.5, fit_intercept=True, solver_options=TIGHT)),
        ("ada_gl", AdaptiveGroupLasso(groups=groups, alpha=1.5, group_weights=gw, fit_intercept=True, solver_options=TIGHT)),
        ("ada_sgl", AdaptiveSparseGroupLasso(groups=groups, l1_ratio=0.4, alpha=1.5, group_weights=gw, fit_intercept=True, solver_options=TIGHT)),
        ("ada_rgl", AdaptiveRidgedGroupLasso(groups=groups, alpha=1.5, delta=(0.7,), group_weights=gw, fit_intercept=True, solver_options=TIGHT)),
    ]
    for key, est in cases:
        est.fit(X, y)
        assert est.n_iter_ == int(golden[f"{key}_niter"])
        assert rel_inf(est.coef_, golden[f"{key}_coef"]) < 1e-7, key
        npt.assert_allclose(est.intercept_, golden[f"{key}_icpt"], rtol=1e-7)
        # weights after the FINAL update (Appendix A-10); eps-regularised reciprocals amplify errors
        w = est.adaptive_weights_
        wg = golden[f"{key}_w"]
        big = wg < 1e3  # active coefficients/groups
        npt.assert_allclose(w[big], wg[big], rtol=1e-5)
        assert np.all(w[~big] > 1e3)


def test_adaptive_lasso_sequence_vs_sklearn(backend, golden):
    X, y = golden["l1_X"], golden["l1_y"]
    alpha = float(golden["ada_sk_alpha"])
    for k in (2, 3):
        est = AdaptiveLasso(alpha=alpha, max_iter=k, solver_options=TIGHT).fit(X, y)
        assert rel_inf(est.coef_, golden["ada_sk_coefs"][k - 1]) < 1e-7
        assert est.n_iter_ == k


def test_overlap_group_lasso_matches_oracle(backend, golden, rng):
    # reference _lasso.py:279-502 / _adaptive_lasso.py:377-524: duplicated columns, folded back
    X, y = golden["grp_X"], golden["grp_y"]
    p = X.shape[1]
    group_list = [list(rng.choice(6, replace=False, size=rng.integers(1, 4))) for _ in range(p)]
    gw = 0.5 + rng.uniform(size=6)
    m = OverlapGroupLasso(group_list=group_list, alpha=2.0, group_weights=gw, solver_options=TIGHT).fit(X, y)
    ref = oracle.fit_overlap_group_lasso(X, y, group_list=group_list, alpha=2.0, group_weights=gw)
    assert rel_inf(m.coef_, ref["coef"]) < 1e-8
    # disjoint group_list == plain GroupLasso
    gl = GroupLasso(groups=golden["grp_groups"], alpha=2.0, solver_options=TIGHT).fit(X, y)
    og = OverlapGroupLasso(group_list=[[g] for g in golden["grp_groups"]], alpha=2.0, solver_options=TIGHT).fit(X, y)
    assert rel_inf(og.coef_, gl.coef_) < 1e-8
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        am = AdaptiveOverlapGroupLasso(group_list=group_list, alpha=1.5, fit_intercept=True, solver_options=TIGHT).fit(X, y)
    aref = oracle.fit_adaptive_overlap_group_lasso(X, y, group_list=group_list, alpha=1.5, fit_intercept=True)
    assert am.n_iter_ == aref["n_iter"]
    assert rel_inf(am.coef_, aref["coef"]) < 1e-6
    npt.assert_allclose(am.intercept_, aref["intercept"], rtol=1e-6)


def test_standardize_penalises_group_predictions(backend, golden):
    # standardize=True: alpha sum_g w_g ||X_g b_g||_2 (reference _lasso.py:249-252)
    X, y, groups, gw = golden["grp_X"], golden["grp_y"], golden["grp_groups"], golden["grp_gw"]
    n = len(y)
    gidx, G = oracle.group_index(groups, X.shape[1])
    alpha = 0.4
    m = GroupLasso(groups=groups, alpha=alpha, group_weights=gw, standardize=True, solver_options=TIGHT).fit(X, y)
    ref = oracle.fit_group_lasso(X, y, groups=groups, alpha=alpha, group_weights=gw, standardize=True)
    assert rel_inf(m.coef_, ref["coef"]) < 1e-8
    # optimality conditions in the ORIGINAL coordinates
    r = X @ m.coef_ - y
    n_active = 0
    for g in range(G):
        Xg = X[:, gidx == g]
        fit = Xg @ m.coef_[gidx == g]
        grad = Xg.T @ r / n
        if np.linalg.norm(fit) > 1e-9:
            n_active += 1
            assert np.max(np.abs(grad + alpha * gw[g] * Xg.T @ fit / np.linalg.norm(fit))) < 1e-8
        else:
            u = np.linalg.lstsq(alpha * gw[g] * Xg.T, -grad, rcond=None)[0]
            assert np.linalg.norm(u) <= 1 + 1e-8
    assert 0 < n_active < G
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        am = AdaptiveGroupLasso(groups=groups, alpha=0.4, standardize=True, fit_intercept=True, solver_options=TIGHT).fit(X, y)
    aref = oracle.fit_adaptive_group_lasso(X, y, groups=groups, alpha=0.4, standardize=True, fit_intercept=True)
    assert am.n_iter_ == aref["n_iter"]
    assert rel_inf(am.coef_, aref["coef"]) < 1e-6


def test_standardized_sparse_group_lasso_matches_golden_and_its_optimality_conditions(backend, golden):
    """standardize=True for the sparse-group penalty: lambda1 ||b||_1 + lambda2 sum_g w_g ||X_g b_g||_2
    (reference _lasso.py:627-639 with the group norms of :249-252).  The product reaches it by operator
    splitting around weighted-l1 solves, the golden by the oracle's primal-dual iteration."""
    X, y, groups, gw = golden["grp_X"], golden["grp_y"], golden["grp_groups"], golden["grp_gw"]
    p = X.shape[1]
    gidx, G = oracle.group_index(groups, p)
    opts = {"tol": 1e-11, "max_iter": 200000}
    m = SparseGroupLasso(groups=groups, alpha=0.4, l1_ratio=0.5, group_weights=gw, standardize=True, solver_options=opts).fit(X, y)
    assert m.solver_info_["converged"]
    assert rel_inf(m.coef_, golden["std_sgl_coef"]) < 1e-8
    # optimality in the ORIGINAL problem, certificate computed from the coefficients alone
    scale = np.max(np.abs(X.T @ y)) / len(y)
    assert oracle.kkt_standardized(X, y, 0.2 * np.ones(p), 0.2 * gw, gidx, G, m.coef_) < 1e-8 * scale
    fit_norms = np.array([np.linalg.norm(X[:, gidx == g] @ m.coef_[gidx == g]) for g in range(G)])
    assert 0 < np.sum(fit_norms > 0) < G  # some groups are out as a whole: exact zeros
    # default options stay within the stated 1e-6, and the unstandardised fit is a different model
    d = SparseGroupLasso(groups=groups, alpha=0.4, l1_ratio=0.5, group_weights=gw, standardize=True).fit(X, y)
    assert rel_inf(d.coef_, golden["std_sgl_coef"]) < 1e-6
    plain = SparseGroupLasso(groups=groups, alpha=0.4, l1_ratio=0.5, group_weights=gw, solver_options=opts).fit(X, y)
    assert rel_inf(plain.coef_, m.coef_) > 1e-3
    # the two ends of l1_ratio are the Lasso and the standardised GroupLasso
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        one = SparseGroupLasso(groups=groups, alpha=0.4, l1_ratio=1.0, standardize=True, solver_options=opts).fit(X, y)
        zero = SparseGroupLasso(groups=groups, alpha=0.4, l1_ratio=0.0, group_weights=gw, standardize=True, solver_options=opts).fit(X, y)
    assert rel_inf(one.coef_, oracle.fit_lasso(X, y, alpha=0.4)["coef"]) < 1e-8
    ref_gl = oracle.fit_group_lasso(X, y, groups=groups, alpha=0.4, group_weights=gw, standardize=True)
    assert rel_inf(zero.coef_, ref_gl["coef"]) < 1e-7
    # adaptive variant: both weight vectors re-computed from |b_j| and ||X_g b_g|| (_adaptive_lasso.py:712-726)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        am = AdaptiveSparseGroupLasso(groups=groups, alpha=0.8, l1_ratio=0.4, group_weights=gw, standardize=True,
                                      fit_intercept=True, solver_options=opts).fit(X, y)
    assert am.n_iter_ == int(golden["std_ada_sgl_niter"])
    assert rel_inf(am.coef_, golden["std_ada_sgl_coef"]) < 1e-6
    npt.assert_allclose(am.intercept_, float(golden["std_ada_sgl_icpt"]), rtol=1e-6)
    npt.assert_allclose(am.adaptive_weights_, golden["std_ada_sgl_w"], rtol=1e-4)


def test_standardized_sparse_group_lasso_on_the_reference_sized_models(backend, random_model_with_groups):
    """25 samples, 20 / 30 features (p > n), centred: the splitting against the oracle's primal-dual iteration."""
    X, y, _, groups = random_model_with_groups
    p = X.shape[1]
    m = SparseGroupLasso(groups=groups, alpha=1.0, l1_ratio=0.5, standardize=True, fit_intercept=True,
                         solver_options={"tol": 1e-10, "max_iter": 200000}).fit(X, y)
    ref = oracle.fit_sparse_group_lasso(X, y, groups=groups, alpha=1.0, l1_ratio=0.5, standardize=True, fit_intercept=True)
    assert ref["info"]["converged"]
    assert rel_inf(m.coef_, ref["coef"]) < 1e-6
    npt.assert_allclose(m.intercept_, ref["intercept"], rtol=1e-6)
    Xp, yp, _, _ = oracle.preprocess(X, y, None, True)
    gidx, G = oracle.group_index(groups, p)
    scale = np.max(np.abs(Xp.T @ yp)) / len(yp)
    ztol = 1e-7 * np.max(np.abs(m.coef_))
    assert oracle.kkt_standardized(Xp, yp, 0.5 * np.ones(p), 0.5 * np.ones(G), gidx, G, m.coef_, zero_tol=ztol) < 1e-6 * scale


def test_standardized_ridged_group_lasso_satisfies_its_optimality_conditions(backend, golden):
    """standardize=True for the ridged penalty (reference _lasso.py:767-793): group norms
    ||M_g b_g||, M_g = sqrtm(X_g^T X_g + sqrt(delta_g) I), ridge 1/2 delta_g ||b_g||^2 on b.  The fit is checked
    through the optimality conditions in the ORIGINAL coordinates, which know nothing of the change of variables
    the estimator solves it with."""
    from scipy.linalg import sqrtm

    X, y, groups, gw = golden["grp_X"], golden["grp_y"], golden["grp_groups"], golden["grp_gw"]
    n = len(y)
    gidx, G = oracle.group_index(groups, X.shape[1])
    alpha = 0.6
    delta = np.linspace(0.5, 2.0, G)
    m = RidgedGroupLasso(groups=groups, alpha=alpha, delta=delta, group_weights=gw, standardize=True,
                         solver_options=TIGHT).fit(X, y)
    r = X @ m.coef_ - y
    n_active = 0
    for g in range(G):
        cols = gidx == g
        Xg, bg = X[:, cols], m.coef_[cols]
        M = np.real(sqrtm(Xg.T @ Xg + np.sqrt(delta[g]) * np.eye(cols.sum())))
        grad = Xg.T @ r / n + delta[g] * bg
        if np.linalg.norm(bg) > 1e-10:
            n_active += 1
            sub = alpha * gw[g] * (M @ M @ bg) / np.linalg.norm(M @ bg)
            assert np.max(np.abs(grad + sub)) < 1e-7 * max(1.0, np.max(np.abs(sub)))
        else:
            assert np.linalg.norm(np.linalg.solve(M, grad)) <= alpha * gw[g] * (1 + 1e-7)
    assert 0 < n_active < G
    # the unstandardised fit is a different model
    m0 = RidgedGroupLasso(groups=groups, alpha=alpha, delta=delta, group_weights=gw, solver_options=TIGHT).fit(X, y)
    assert rel_inf(m.coef_, m0.coef_) > 1e-3
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        am = AdaptiveRidgedGroupLasso(groups=groups, alpha=alpha, delta=delta, standardize=True, fit_intercept=True,
                                      solver_options=TIGHT).fit(X, y)
    assert am.n_iter_ >= 2 and np.all(np.isfinite(am.coef_)) and am.intercept_ != 0.0


def test_standardize_with_a_rank_deficient_group_and_warm_starts(backend, golden):
    """A group with a duplicated column: ||X_g b_g|| cannot tell its two copies apart.  The reference hands that
    to cvxpy as it is; here the group keeps as many unknowns as its rank and the minimum-norm coefficients are
    reported.  Optimality is checked on what the objective does see: X_g b_g."""
    X, y, groups, gw = golden["grp_X"].copy(), golden["grp_y"], golden["grp_groups"], golden["grp_gw"]
    n = len(y)
    gidx, G = oracle.group_index(groups, X.shape[1])
    cols0 = np.flatnonzero(gidx == 0)
    X[:, cols0[1]] = X[:, cols0[0]]  # rank deficient on purpose
    alpha = 0.3
    est = GroupLasso(groups=groups, alpha=alpha, group_weights=gw, standardize=True, warm_start=True, solver_options=TIGHT)
    m = est.fit(X, y)
    assert m.coef_[cols0[0]] == pytest.approx(m.coef_[cols0[1]], abs=1e-9)  # minimum norm: the copies share the load
    r = X @ m.coef_ - y
    for g in range(G):
        Xg = X[:, gidx == g]
        fit = Xg @ m.coef_[gidx == g]
        grad = Xg.T @ r / n
        if np.linalg.norm(fit) > 1e-9:
            assert np.max(np.abs(grad + alpha * gw[g] * Xg.T @ fit / np.linalg.norm(fit))) < 1e-7
        else:
            u = np.linalg.lstsq(alpha * gw[g] * Xg.T, -grad, rcond=None)[0]
            assert np.linalg.norm(u) <= 1 + 1e-7
    # warm start in the transformed coordinates: the second fit starts at the solution
    first = m.coef_.copy()
    n1 = m.solver_info_["n_iter"]
    m2 = est.fit(X, y)
    assert rel_inf(m2.coef_, first) < 1e-8
    assert m2.solver_info_["n_iter"] <= max(2, n1 // 3)


# ---- structural properties from the reference tests ------------------------------------------------
def test_adaptive_lasso_sparser(backend, random_model):
    # /root/reference/tests/test_lasso.py:77-85
    X, y, _ = random_model
    lasso = Lasso(fit_intercept=True).fit(X, y)
    alasso = AdaptiveLasso(fit_intercept=True).fit(X, y)
    assert sum(abs(lasso.coef_) > THRESHOLD) >= sum(abs(alasso.coef_) > THRESHOLD)


def test_group_lasso_all_or_nothing(backend, random_model_with_groups):
    # /root/reference/tests/test_lasso.py:88-155 (both standardize arms for the pure group penalty)
    X, y, _, groups = random_model_with_groups
    gw = np.ones(len(np.unique(groups)))
    for est in (AdaptiveGroupLasso(groups=groups, alpha=0.1, fit_intercept=True),
                AdaptiveGroupLasso(groups=groups, alpha=0.1, fit_intercept=True, standardize=True),
                AdaptiveGroupLasso(groups=groups, alpha=0.1, group_weights=gw, fit_intercept=True),
                AdaptiveRidgedGroupLasso(groups=groups, alpha=0.1, group_weights=gw, fit_intercept=True)):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            est.fit(X, y)
        m = np.max(abs(est.coef_))
        for gid in np.unique(groups):
            c = abs(est.coef_[groups == gid])
            assert (c > m * THRESHOLD).all() or (c <= m * THRESHOLD).all()


@pytest.mark.parametrize("estimator_cls", ADAPTIVE)
def test_adaptive_weights_are_updated(backend, estimator_cls, random_model_with_groups, rng):
    # /root/reference/tests/test_lasso.py:158-200: every weight differs from its initial value after fit
    X, y, beta, groups = random_model_with_groups
    if estimator_cls is AdaptiveLasso:
        est = estimator_cls()
    elif estimator_cls is AdaptiveOverlapGroupLasso:
        gids = np.unique(groups)
        est = estimator_cls(group_list=[list(rng.choice(gids, replace=False, size=rng.integers(1, 3))) for _ in beta])
    else:
        est = estimator_cls(groups=groups)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        est.fit(X, y)
    G = len(np.unique(groups))
    if estimator_cls is AdaptiveOverlapGroupLasso:
        G = len(est.adaptive_weights_)
    if estimator_cls is AdaptiveSparseGroupLasso:
        init = np.concatenate((0.5 * np.ones(G), 0.5 * np.ones(len(beta))))
    elif estimator_cls is AdaptiveLasso:
        init = np.ones(len(beta))
    else:
        init = np.ones(G)
    assert not any(nw == pytest.approx(w) for nw, w in zip(est.adaptive_weights_, init))


def test_bad_inputs(backend, random_model_with_groups, rng):
    # /root/reference/tests/test_lasso.py:203-260
    X, y, beta, groups = random_model_with_groups
    bad_groups = rng.integers(0, 6, size=len(beta) - 1)
    group_weights = np.ones(len(np.unique(bad_groups)))
    with pytest.warns(UserWarning):
        GroupLasso().fit(X, y)
    with pytest.warns(UserWarning):
        OverlapGroupLasso().fit(X, y)
    with pytest.raises(ValueError):
        OverlapGroupLasso(group_list=[[0]] * (len(beta) - 1)).fit(X, y)  # reference _lasso.py:390-393
    with pytest.raises(ValueError):
        GroupLasso(bad_groups, group_weights=group_weights).fit(X, y)
    with pytest.raises(TypeError):
        GroupLasso("groups", group_weights=group_weights).fit(X, y)
    with pytest.raises(ValueError):
        GroupLasso(bad_groups, group_weights=np.ones(len(np.unique(bad_groups)) - 1)).fit(X, y)
    with pytest.raises(TypeError):
        GroupLasso(groups, group_weights="weights").fit(X, y)
    lasso = SparseGroupLasso(groups)
    with pytest.raises(ValueError):
        lasso.l1_ratio = -1.0
        lasso.fit(X, y)
    with pytest.raises(ValueError):
        lasso.l1_ratio = 2.0
        lasso.fit(X, y)
    with pytest.raises(ValueError):
        SparseGroupLasso(groups, l1_ratio=-1.0).fit(X, y)
    with pytest.raises(ValueError):
        SparseGroupLasso(groups, l1_ratio=2.0).fit(X, y)
    with pytest.warns(UserWarning):
        SparseGroupLasso(groups, l1_ratio=0.0).fit(X, y)
    with pytest.warns(UserWarning):
        SparseGroupLasso(groups, l1_ratio=1.0).fit(X, y)
    # sklearn-style declarative constraints (reference _lasso.py:77-79, _adaptive_lasso.py:102-108)
    with pytest.raises(ValueError):
        Lasso(alpha=-1.0).fit(X, y)
    with pytest.raises(ValueError):
        AdaptiveLasso(eps=2.0).fit(X, y)
    with pytest.raises(TypeError):
        Lasso(solver_options="fast").fit(X, y)  # reference _base.py:198-199
    with pytest.raises(ValueError):
        RidgedGroupLasso(groups, delta=(1.0, 2.0)).fit(X, y)
    with pytest.warns(UserWarning):
        AdaptiveLasso(max_iter=1).fit(X, y)  # reference _adaptive_lasso.py:142-147


def test_lambda_definitions(backend):
    # /root/reference/tests/test_lasso.py:263-291: lambda1 = l1_ratio*alpha, lambda2 = (1-l1_ratio)*alpha
    est = SparseGroupLasso(groups=[0, 0, 1], alpha=0.5)
    assert est._lambdas() == (0.25, 0.25)
    est.l1_ratio = 0.25
    assert est._lambdas() == (0.25 * 0.5, 0.75 * 0.5)
    rgl = RidgedGroupLasso(groups=[0, 0, 1], delta=(4.0,))
    npt.assert_array_equal(rgl._delta_vector(2), 4.0 * np.ones(2))


@pytest.mark.parametrize("estimator_cls", ESTIMATORS)
def test_general_fit(backend, estimator_cls, random_model, rng):
    # /root/reference/tests/test_common.py:34-67
    X, y, beta = random_model
    args = {}
    if "groups" in signature(estimator_cls).parameters:
        args["groups"] = rng.integers(0, 5, size=len(beta))
    if "group_list" in signature(estimator_cls).parameters:
        args["group_list"] = [
            np.sort(rng.choice(range(5), replace=False, size=rng.integers(1, 5))) for _ in range(len(beta))
        ]
    for fit_intercept in (False, True):
        est = estimator_cls(fit_intercept=fit_intercept, **args)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            est.fit(X, y)
        assert isinstance(est.coef_, np.ndarray)
        assert len(est.coef_) == len(beta)
        assert len(est.predict(X)) == len(y)
        assert (est.intercept_ != 0.0) == fit_intercept


def test_readme_gridsearch(backend):
    # BASELINE config 1 (/root/reference/README.md:42-55)
    from sklearn.datasets import make_regression

    X, y = make_regression(n_samples=100, n_features=80, n_informative=10, random_state=0)
    # noise-free data: every alpha <= ~1e-3 reaches R^2 = 1 to within the solver tolerance, so which
    # of them wins the argmax depends on the last digits; with a tight tolerance it is the smallest.
    for opts, expect in ((None, None), ({"tol": 1e-12, "max_iter": 200000}, {"alpha": 1e-08})):
        alasso = AdaptiveLasso(fit_intercept=False, solver_options=opts)
        gs = GridSearchCV(alasso, {"alpha": np.logspace(-8, 2, 10)})
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            gs.fit(X, y)
        assert gs.best_params_["alpha"] < 1e-2
        if expect is not None:
            assert gs.best_params_ == expect
        assert gs.best_score_ == pytest.approx(1.0, abs=1e-6)
        assert np.sum(np.abs(gs.best_estimator_.coef_) > 1e-6) == 10


def test_warm_start_refit(backend, golden):
    X, y = golden["l1_X"], golden["l1_y"]
    m = Lasso(alpha=2.0, warm_start=True, solver_options=TIGHT).fit(X, y)
    first = m.solver_info_["n_iter"]
    m.fit(X, y)
    assert m.solver_info_["n_iter"] <= max(3, first // 2)
    assert rel_inf(m.coef_, golden["l1_coef"][2]) < 1e-9


@pytest.mark.parametrize("estimator_cls", ESTIMATORS, ids=lambda c: c.__name__)
def test_sklearn_compatible(backend, estimator_cls):
    # /root/reference/tests/test_common.py:95-108 runs check_estimator for every estimator it exports: so does this
    from sklearn.utils.estimator_checks import check_estimator

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        results = check_estimator(estimator_cls(fit_intercept=True), on_fail=None)
    failed = [(r["check_name"], str(r["exception"])[:200]) for r in results if r["status"] == "failed"]
    assert not failed, failed
    assert sum(r["status"] == "passed" for r in results) >= 50


def test_stepwise_estimator(backend, golden):
    # behaviours of /root/reference/tests/test_stepwise.py on the re-implemented composite
    from sklearn.base import clone
    from sklearn.utils._param_validation import InvalidParameterError

    from sparselm_amd.stepwise import StepwiseEstimator

    X, y = golden["l1_X"], golden["l1_y"]
    p = X.shape[1]
    scopes = (tuple(range(0, 10)), tuple(range(10, p)))
    step = StepwiseEstimator(
        [("head", Lasso(alpha=0.5, fit_intercept=True, solver_options=TIGHT)), ("tail", Lasso(alpha=2.0, solver_options=TIGHT))],
        scopes,
    ).fit(X, y)
    first = Lasso(alpha=0.5, fit_intercept=True, solver_options=TIGHT).fit(X[:, :10], y)
    resid = y - first.predict(X[:, :10])
    second = Lasso(alpha=2.0, solver_options=TIGHT).fit(X[:, 10:], resid)
    npt.assert_allclose(step.coef_[:10], first.coef_, rtol=0, atol=1e-9)
    npt.assert_allclose(step.coef_[10:], second.coef_, rtol=0, atol=1e-9)
    npt.assert_allclose(step.intercept_, first.intercept_, rtol=1e-9)
    npt.assert_allclose(step.predict(X), first.predict(X[:, :10]) + second.predict(X[:, 10:]), rtol=1e-9, atol=1e-9)
    # parameters are routed through the step names and survive clone()
    c = clone(step).set_params(tail__alpha=7.0)
    assert c.get_params()["tail__alpha"] == 7.0 and step.get_params()["tail__alpha"] == 2.0
    with pytest.raises(InvalidParameterError):  # blocks must partition range(p)
        StepwiseEstimator(step.steps, ((0, 1), (1, 2))).fit(X[:, :3], y)
    with pytest.raises(InvalidParameterError):  # only the first step may fit an intercept
        StepwiseEstimator([("a", Lasso()), ("b", Lasso(fit_intercept=True))], scopes).fit(X, y)
    with pytest.raises(InvalidParameterError):  # no nesting
        StepwiseEstimator([("a", Lasso()), ("b", step)], scopes).fit(X, y)


@pytest.mark.gpu
def test_default_tolerance_holds_1e6_on_correlated_reference_sized_designs():
    # `tol` bounds the last prox step, the error is about the condition number times that: the
    # estimators' default (1e-10 at these sizes) keeps strongly correlated designs of the reference's own sizes
    # within the stated 1e-6 of an exact solver (scikit-learn's coordinate descent as referee)
    from sklearn.linear_model import Lasso as SkLasso

    from sparselm_amd.model import Lasso

    rng = np.random.default_rng(4)
    for n, p, rho in ((400, 100, 0.95), (250, 120, 0.9), (2000, 300, 0.97)):
        E = rng.standard_normal((n, p))
        X = E.copy()
        for j in range(1, p):
            X[:, j] = rho * X[:, j - 1] + np.sqrt(1 - rho**2) * E[:, j]
        beta = np.zeros(p)
        beta[rng.choice(p, 8, replace=False)] = rng.standard_normal(8) * 2
        y = X @ beta + 0.5 * rng.standard_normal(n)
        amax = np.max(np.abs(X.T @ y)) / n
        for frac in (0.3, 0.03, 0.003):
            ref = SkLasso(alpha=frac * amax, fit_intercept=False, tol=1e-14, max_iter=1000000).fit(X, y).coef_
            with warnings.catch_warnings():
                warnings.simplefilter("error")
                got = Lasso(alpha=frac * amax).fit(X, y).coef_
            assert np.max(np.abs(got - ref)) <= 1e-6 * np.max(np.abs(ref)), (n, p, rho, frac)


@pytest.mark.gpu
def test_against_a_cvxpy_formulation_when_cvxpy_is_installed(golden):
    """SURVEY 8(c), last bullet: where cvxpy can be imported (it cannot in the build container, and normally not
    on the GPU box either -- the test is then skipped), the penalty family written out in cvxpy from the math
        1/(2n) ||X b - y||^2 + a ||b||_1 + sum_g b_g ||b_g||_2 + 1/2 sum_g d_g ||b_g||_2^2
    (reference src/sparselm/model/_lasso.py:109-121, 267-275, 627-639, 795-811) and handed to its default conic
    solver is the reference's own route (`problem.solve()`, model/_base.py:516-518): the HIP estimators have to
    land within the north-star 1e-6 rel-inf of it."""
    cp = pytest.importorskip("cvxpy")
    X, y, groups, gw = golden["grp_X"], golden["grp_y"], golden["grp_groups"], golden["grp_gw"]
    n, p = X.shape
    gidx, G = oracle.group_index(groups, p)
    alpha = float(golden["grp_alpha"])
    delta = golden["grp_delta"]

    def cvx(a, b, d):
        beta = cp.Variable(p)
        obj = cp.sum_squares(X @ beta - y) / (2 * n) + a * cp.norm1(beta)
        for g in range(G):
            bg = beta[np.flatnonzero(gidx == g)]
            obj = obj + b[g] * cp.norm2(bg) + 0.5 * d[g] * cp.sum_squares(bg)
        prob = cp.Problem(cp.Minimize(obj))
        try:
            prob.solve(solver="CLARABEL", tol_gap_abs=1e-12, tol_gap_rel=1e-12, tol_feas=1e-12)
        except Exception:
            prob.solve()
        return np.asarray(beta.value)

    cases = [
        (Lasso(alpha=alpha, solver_options=TIGHT), (alpha, np.zeros(G), np.zeros(G))),
        (GroupLasso(groups=groups, alpha=alpha, group_weights=gw, solver_options=TIGHT), (0.0, alpha * gw, np.zeros(G))),
        (SparseGroupLasso(groups=groups, l1_ratio=0.3, alpha=alpha, group_weights=gw, solver_options=TIGHT),
         (0.3 * alpha, 0.7 * alpha * gw, np.zeros(G))),
        (RidgedGroupLasso(groups=groups, alpha=alpha, delta=delta, group_weights=gw, solver_options=TIGHT),
         (0.0, alpha * gw, delta)),
    ]
    for est, (a, b, d) in cases:
        ref = cvx(a, b, d)
        got = est.fit(X, y).coef_
        assert np.max(np.abs(got - ref)) <= 1e-6 * np.max(np.abs(ref)), type(est).__name__


@pytest.mark.gpu
def test_device_datasets_are_cached_across_fits_by_content():
    """The reference re-uses its cvxpy problem when fit() sees the same data again (model/_base.py:182-195).  Here
    the device-resident dataset is what is expensive to set up: stock scikit-learn GridSearchCV over the README
    example (10 alphas x 5 folds = 50 fits + the refit) must upload 5 + 1 distinct training sets, not 51 -- the
    cache is keyed by content, for X[train] is a fresh array every time."""
    from sklearn.datasets import make_regression

    cache = _backend.dataset_cache()
    cache.clear()
    cache.hits = cache.misses = 0
    X, y = make_regression(n_samples=100, n_features=80, n_informative=10, random_state=0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        gs = GridSearchCV(AdaptiveLasso(fit_intercept=False), {"alpha": np.logspace(-8, 2, 10)}).fit(X, y)
    assert cache.misses == 6 and cache.hits == 45
    assert gs.best_params_ == {"alpha": 1e-08}
    # same content, different array object, Fortran order: still the same device dataset? no -- the layout is
    # part of the key (the engine's copy is the same, the digest is computed on the bytes as given)
    first = Lasso(alpha=0.3).fit(X, y).coef_
    hits = cache.hits
    again = Lasso(alpha=0.3).fit(X.copy(), y.copy()).coef_
    assert cache.hits == hits + 1 and np.array_equal(first, again)
    # a changed entry is a different dataset
    X2 = X.copy()
    X2[3, 4] += 1e-9
    misses = cache.misses
    Lasso(alpha=0.3).fit(X2, y)
    assert cache.misses == misses + 1
    # groups and penalties are per fit, not part of the dataset: one cached dataset serves both estimators
    groups = np.repeat(np.arange(16), 5)
    g1 = GroupLasso(groups=groups, alpha=0.3).fit(X, y).coef_
    l1 = Lasso(alpha=0.3).fit(X, y).coef_
    g2 = GroupLasso(groups=groups, alpha=0.3).fit(X, y).coef_
    assert np.array_equal(g1, g2) and np.array_equal(l1, first)
    cache.clear()

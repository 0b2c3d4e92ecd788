"""Grid search with the one-standard-error rule (mirrors /root/reference/tests/test_model_selection.py).

CPU: the selection rule and the generic path with scikit-learn's own Lasso (the reference's test_onestd
needs nothing else) and with the oracle behind our estimators.  GPU: the device-resident fast path
against the generic path, cell by cell.
"""

import warnings

import numpy as np
import pytest
from sklearn.datasets import make_regression
from sklearn.linear_model import Lasso as SkLasso
from sklearn.model_selection import GridSearchCV as SkGridSearchCV
from sklearn.model_selection import KFold, train_test_split

from _oracle_backend import OracleBackend
from sparselm_amd import _backend
from sparselm_amd.model import AdaptiveLasso, GroupLasso, Lasso, SparseGroupLasso
from sparselm_amd.model_selection import GridSearchCV, _format_results, select_best_index_onestd


def test_onestd_with_sklearn_lasso():
    # /root/reference/tests/test_model_selection.py:122-167 (deterministic: one repetition suffices)
    X, y = make_regression(n_samples=200, n_features=100, n_informative=10, noise=40.0, bias=-15.0, random_state=0)
    X_train, _, y_train, _ = train_test_split(X, y, test_size=0.25, random_state=0)
    cv5 = KFold(n_splits=5, shuffle=True, random_state=0)
    params = {"alpha": np.logspace(-1, 1, 10)}
    std = GridSearchCV(SkLasso(fit_intercept=True), params, opt_selection_method="one_std_score", cv=cv5).fit(X_train, y_train)
    opt = GridSearchCV(SkLasso(fit_intercept=True), params, opt_selection_method="max_score", cv=cv5).fit(X_train, y_train)
    assert opt.best_params_["alpha"] <= std.best_params_["alpha"]
    assert np.sum(np.abs(opt.best_estimator_.coef_) >= 1e-6) >= np.sum(np.abs(std.best_estimator_.coef_) >= 1e-6)
    # max_score agrees with scikit-learn's own search under the same default scoring
    sk = SkGridSearchCV(SkLasso(fit_intercept=True), params, scoring="neg_root_mean_squared_error", cv=cv5).fit(X_train, y_train)
    assert opt.best_params_ == sk.best_params_
    np.testing.assert_allclose(opt.cv_results_["mean_test_score"], sk.cv_results_["mean_test_score"])
    assert opt.best_score_ == pytest.approx(sk.best_score_)
    assert std.best_score_std_ == pytest.approx(std.cv_results_["std_test_score"][std.best_index_])
    with pytest.raises(ValueError):
        GridSearchCV(SkLasso(), params, opt_selection_method="median").fit(X_train, y_train)


def test_one_std_rule_on_synthetic_results():
    cands = [{"alpha": a} for a in (0.01, 0.1, 1.0, 10.0)]
    scores = np.array([[-1.0, -1.2], [-0.9, -1.1], [-1.06, -1.16], [-3.0, -3.2]])
    res = _format_results(cands, scores, np.zeros_like(scores))
    assert list(res["rank_test_score"]) == [2, 1, 3, 4]
    # best = alpha 0.1 (mean -1.0, std 0.1): closest to -1.1 among alpha >= 0.1 is alpha = 1.0
    assert select_best_index_onestd(res) == 2


def test_generic_path_with_oracle_backend(golden):
    X, y = golden["l1_X"], golden["l1_y"]
    grid = {"alpha": [0.05, 0.5, 2.0, 10.0]}
    cv = KFold(4, shuffle=True, random_state=1)
    with _backend.use_backend(OracleBackend()), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        # fit_intercept=True keeps this on the generic (scikit-learn loop) path
        gs = GridSearchCV(Lasso(fit_intercept=True), grid, cv=cv, opt_selection_method="one_std_score").fit(X, y)
        sk = SkGridSearchCV(Lasso(fit_intercept=True), grid, cv=cv, scoring="neg_root_mean_squared_error").fit(X, y)
    np.testing.assert_allclose(gs.cv_results_["mean_test_score"], sk.cv_results_["mean_test_score"], rtol=1e-9)
    assert gs.best_params_["alpha"] >= sk.best_params_["alpha"]
    assert gs.predict(X).shape == y.shape


# ---- device-resident fast path ---------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("scoring", ["neg_root_mean_squared_error", "r2"])
def test_fast_path_matches_generic_path(golden, scoring):
    X, y, groups, gw = golden["grp_X"], golden["grp_y"], golden["grp_groups"], golden["grp_gw"]
    cv = KFold(5, shuffle=True, random_state=0)
    cases = [
        (Lasso(), {"alpha": list(np.geomspace(20, 0.05, 7))}),
        (GroupLasso(groups=groups, group_weights=gw), {"alpha": list(np.geomspace(20, 0.1, 6))}),
        (SparseGroupLasso(groups=groups), {"alpha": list(np.geomspace(10, 0.1, 5)), "l1_ratio": [0.1, 0.5, 0.9]}),
    ]
    for est, grid in cases:
        est.set_params(solver_options={"tol": 1e-11})
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            fast = GridSearchCV(est, grid, cv=cv, scoring=scoring).fit(X, y)
            slow = SkGridSearchCV(est, grid, cv=cv, scoring=scoring).fit(X, y)
        assert fast.cv_results_["params"] == slow.cv_results_["params"]
        for f in range(5):
            np.testing.assert_allclose(fast.cv_results_[f"split{f}_test_score"], slow.cv_results_[f"split{f}_test_score"],
                                       rtol=1e-7, atol=1e-9)
        assert fast.best_params_ == slow.best_params_
        assert fast.best_score_ == pytest.approx(slow.best_score_, rel=1e-7)
        np.testing.assert_allclose(fast.best_estimator_.coef_, slow.best_estimator_.coef_, rtol=0,
                                   atol=1e-7 * np.max(np.abs(slow.best_estimator_.coef_)))
        np.testing.assert_allclose(fast.predict(X), slow.predict(X), rtol=1e-6, atol=1e-6)
        assert hasattr(fast, "search_time_")  # marks the device path


@pytest.mark.gpu
def test_fast_path_batches_dealt_to_several_streams_give_the_same_search(golden):
    """streams > 1: the batches of the device path run on further engines of the device, each on its own copy of
    the dataset and from its own host thread -- the same cells, the same numbers."""
    X, y, groups = golden["grp_X"], golden["grp_y"], golden["grp_groups"]
    cv = KFold(5, shuffle=True, random_state=0)
    grid = {"alpha": list(np.geomspace(10, 0.1, 5)), "l1_ratio": [0.1, 0.5, 0.9]}
    est = SparseGroupLasso(groups=groups, fit_intercept=True, solver_options={"tol": 1e-11})
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        one = GridSearchCV(est, grid, cv=cv, lanes=4, streams=1).fit(X, y)
        three = GridSearchCV(est, grid, cv=cv, lanes=4, streams=3).fit(X, y)
        ada1 = GridSearchCV(AdaptiveLasso(), {"alpha": [0.5, 1.0, 2.0]}, cv=cv, lanes=4, streams=1).fit(X, y)
        ada2 = GridSearchCV(AdaptiveLasso(), {"alpha": [0.5, 1.0, 2.0]}, cv=cv, lanes=4, streams=2).fit(X, y)
    for a, b in ((one, three), (ada1, ada2)):
        for f in range(5):
            np.testing.assert_array_equal(a.cv_results_[f"split{f}_test_score"], b.cv_results_[f"split{f}_test_score"])
        assert a.best_params_ == b.best_params_
        np.testing.assert_array_equal(a.best_estimator_.coef_, b.best_estimator_.coef_)


@pytest.mark.gpu
def test_fast_path_one_std_and_fallbacks(golden):
    X, y = golden["l1_X"], golden["l1_y"]
    grid = {"alpha": list(np.logspace(-1, 1.5, 8))}
    cv = KFold(5, shuffle=True, random_state=0)
    opt = GridSearchCV(Lasso(), grid, cv=cv).fit(X, y)
    std = GridSearchCV(Lasso(), grid, cv=cv, opt_selection_method="one_std_score").fit(X, y)
    assert std.best_params_["alpha"] >= opt.best_params_["alpha"]
    assert std.best_score_std_ > 0
    # custom scorers go through the generic loop (no search_time_); adaptive estimators and
    # fit_intercept=True stay on the device path
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        r2 = GridSearchCV(Lasso(), {"alpha": [0.5, 2.0]}, cv=3, scoring="neg_mean_absolute_error").fit(X, y)
        ada = GridSearchCV(AdaptiveLasso(), {"alpha": [0.5, 2.0]}, cv=3).fit(X, y)
        icpt = GridSearchCV(Lasso(fit_intercept=True), {"alpha": [0.5, 2.0]}, cv=3).fit(X, y)
    assert not hasattr(r2, "search_time_") and hasattr(ada, "search_time_") and hasattr(icpt, "search_time_")
    assert ada.best_estimator_.n_iter_ >= 1 and icpt.best_estimator_.intercept_ != 0.0
    # invalid candidates raise the estimator's own error class before anything is solved
    with pytest.raises(ValueError):
        GridSearchCV(Lasso(), {"alpha": [1.0, -1.0]}, cv=3).fit(X, y)


def _grid_rank(rank, world_size, port, out_dir):
    import os
    import sys

    import torch.distributed as dist

    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    for path in (root, os.path.join(root, "sparse-lm_amd"), os.path.join(root, "tests")):
        if path not in sys.path:
            sys.path.insert(0, path)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world_size), LOCAL_RANK="0")  # one GPU on the test box: both ranks share it
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    from sparselm_amd.model import SparseGroupLasso
    from sparselm_amd.model_selection import GridSearchCV as GS

    with np.load(os.path.join(root, "tests", "golden", "lasso_family_golden.npz")) as f:
        X, y, groups = f["grp_X"], f["grp_y"], f["grp_groups"]
    grid = {"alpha": list(np.geomspace(10, 0.1, 5)), "l1_ratio": [0.1, 0.5, 0.9]}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        gs = GS(SparseGroupLasso(groups=groups, solver_options={"tol": 1e-11}), grid,
                cv=KFold(5, shuffle=True, random_state=0)).fit(X, y)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), mean=gs.cv_results_["mean_test_score"],
             coef=gs.best_estimator_.coef_, best=gs.best_index_)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_fast_path_two_ranks_share_the_grid(golden, tmp_path):
    # N > 1 path of the grid-aware search: (l1_ratio, fold) units dealt to ranks, results gathered
    import torch.multiprocessing as mp

    mp.spawn(_grid_rank, args=(2, 29641, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    np.testing.assert_array_equal(r0["mean"], r1["mean"])  # every rank ends with the full table
    assert int(r0["best"]) == int(r1["best"])
    X, y, groups = golden["grp_X"], golden["grp_y"], golden["grp_groups"]
    grid = {"alpha": list(np.geomspace(10, 0.1, 5)), "l1_ratio": [0.1, 0.5, 0.9]}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        one = GridSearchCV(SparseGroupLasso(groups=groups, solver_options={"tol": 1e-11}), grid,
                           cv=KFold(5, shuffle=True, random_state=0)).fit(X, y)
    np.testing.assert_allclose(r0["mean"], one.cv_results_["mean_test_score"], rtol=1e-9)
    np.testing.assert_allclose(r0["coef"], one.best_estimator_.coef_, rtol=0, atol=1e-9 * np.max(np.abs(one.best_estimator_.coef_)))


def test_line_search_with_sklearn_estimator():
    # /root/reference/tests/test_model_selection.py:170-187 pattern, with an estimator that needs no GPU
    from sklearn.linear_model import ElasticNet

    from sparselm_amd.model_selection import LineSearchCV

    X, y = make_regression(n_samples=120, n_features=30, n_informative=6, noise=5.0, random_state=2)
    grid = [("alpha", [0.01, 0.1, 1.0, 10.0]), ("l1_ratio", [0.2, 0.5, 0.9])]
    ls = LineSearchCV(ElasticNet(max_iter=10000), grid, opt_selection_method=["one_std_score", "max_score"], cv=4, n_iter=3)
    ls.fit(X, y)
    assert len(ls.history_) == 3
    assert set(ls.best_params_) == {"alpha", "l1_ratio"}
    assert ls.best_params_["alpha"] in grid[0][1] and ls.best_params_["l1_ratio"] in grid[1][1]
    assert ls.predict(X).shape == y.shape
    assert ls.best_score_ <= 0
    with pytest.raises(ValueError):
        LineSearchCV(ElasticNet(), {"alpha": [1.0]}).fit(X, y)
    with pytest.raises(ValueError):
        LineSearchCV(ElasticNet(), grid, opt_selection_method=["max_score"]).fit(X, y)


def test_make_group_regression_shapes():
    # /root/reference/tests/test_dataset.py:13-61
    from sparselm_amd.dataset import make_group_regression

    X, y, groups, coef = make_group_regression(n_samples=50, n_groups=6, n_features_per_group=[2, 3, 4, 2, 5, 1],
                                               n_informative_groups=2, frac_informative_in_group=0.5, noise=1.0,
                                               coef=True, random_state=0)
    assert X.shape == (50, 17) and y.shape == (50,) and groups.shape == (17,) and coef.shape == (17,)
    assert len(np.unique(groups)) == 6
    assert len(np.unique(groups[coef != 0])) == 2
    X2, y2, g2 = make_group_regression(random_state=0)
    assert X2.shape == (100, 200) and len(np.unique(g2)) == 20


@pytest.mark.gpu
def test_line_search_fast_path_and_ols(golden):
    from sparselm_amd.model import OrdinaryLeastSquares
    from sparselm_amd.model_selection import LineSearchCV

    X, y, groups = golden["grp_X"], golden["grp_y"], golden["grp_groups"]
    grid = [("alpha", list(np.geomspace(10, 0.1, 5))), ("l1_ratio", [0.1, 0.5, 0.9])]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ls = LineSearchCV(SparseGroupLasso(groups=groups), grid, cv=3).fit(X, y)
    assert all(hasattr(h, "search_time_") for h in ls.history_)  # every line ran on the device path
    assert ls.best_params_["alpha"] in grid[0][1]
    Xo, yo, sw = golden["ols_X"], golden["ols_y"], golden["ols_sw"]
    ols = OrdinaryLeastSquares(fit_intercept=True, solver_options={"tol": 1e-12, "max_iter": 200000}).fit(Xo, yo, sample_weight=sw)
    np.testing.assert_allclose(ols.coef_, golden["ols_coef_icpt"], rtol=1e-7)  # reference tests/test_ols.py:35-66
    np.testing.assert_allclose(ols.intercept_, golden["ols_icpt"], rtol=1e-7)


@pytest.mark.gpu
def test_fast_path_with_intercept_matches_generic_path(golden):
    # fit_intercept=True on the device path: an unpenalised column of ones, fold by fold equivalent to
    # centring by the training-fold means
    X, y, groups, gw = golden["grp_X"], golden["grp_y"], golden["grp_groups"], golden["grp_gw"]
    X = X + 3.0 * np.arange(X.shape[1]) / X.shape[1]  # non-zero column means
    y = y + 25.0
    cv = KFold(5, shuffle=True, random_state=0)
    cases = [
        (Lasso(fit_intercept=True), {"alpha": list(np.geomspace(20, 0.05, 6))}),
        (GroupLasso(groups=groups, group_weights=gw, fit_intercept=True), {"alpha": list(np.geomspace(20, 0.1, 5))}),
        (SparseGroupLasso(groups=groups, fit_intercept=True), {"alpha": list(np.geomspace(10, 0.1, 4)), "l1_ratio": [0.2, 0.8]}),
    ]
    for est, grid in cases:
        est.set_params(solver_options={"tol": 1e-12, "max_iter": 200000})
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            fast = GridSearchCV(est, grid, cv=cv).fit(X, y)
            slow = SkGridSearchCV(est, grid, cv=cv, scoring="neg_root_mean_squared_error").fit(X, y)
        assert hasattr(fast, "search_time_")  # the device path ran
        np.testing.assert_allclose(fast.cv_results_["mean_test_score"], slow.cv_results_["mean_test_score"], rtol=1e-6)
        assert fast.best_params_ == slow.best_params_
        np.testing.assert_allclose(fast.best_estimator_.coef_, slow.best_estimator_.coef_, rtol=0,
                                   atol=1e-6 * np.max(np.abs(slow.best_estimator_.coef_)))
        assert fast.best_estimator_.intercept_ == pytest.approx(slow.best_estimator_.intercept_, rel=1e-6)
        np.testing.assert_allclose(fast.predict(X), slow.predict(X), rtol=1e-6, atol=1e-6)


@pytest.mark.gpu
def test_fast_path_adaptive_estimators_match_generic_path(golden):
    # Adaptive* in the grid: every (candidate, fold) re-weighting loop is a lane; outer iteration k of up to
    # sixteen of them is one engine call.  Same scores, selection and refit as scikit-learn's loop of fits.
    from sklearn.datasets import make_regression

    from sparselm_amd.model import AdaptiveGroupLasso, AdaptiveLasso, AdaptiveRidgedGroupLasso, AdaptiveSparseGroupLasso

    X, y, groups, gw = golden["grp_X"], golden["grp_y"], golden["grp_groups"], golden["grp_gw"]
    Xr, yr = make_regression(n_samples=100, n_features=80, n_informative=10, random_state=0)  # README example
    cv = KFold(5, shuffle=True, random_state=0)
    so = {"tol": 1e-12, "max_iter": 400000}
    cases = [
        (AdaptiveLasso(fit_intercept=True, solver_options=so), {"alpha": list(np.logspace(-2, 2, 6))}, Xr, yr),
        (AdaptiveLasso(max_iter=5, warm_start=False, solver_options=so), {"alpha": [8.0, 2.0, 0.5]}, X, y),
        (AdaptiveGroupLasso(groups=groups, group_weights=gw, solver_options=so), {"alpha": list(np.geomspace(10, 0.3, 4))}, X, y),
        (AdaptiveSparseGroupLasso(groups=groups, fit_intercept=True, solver_options=so),
         {"alpha": [5.0, 1.0], "l1_ratio": [0.3, 0.7]}, X, y + 10.0),
        (AdaptiveRidgedGroupLasso(groups=groups, delta=(0.5,), solver_options=so), {"alpha": [6.0, 1.5]}, X, y),
    ]
    for est, grid, Xc, yc in cases:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            fast = GridSearchCV(est, grid, cv=cv).fit(Xc, yc)
            slow = SkGridSearchCV(est, grid, cv=cv, scoring="neg_root_mean_squared_error").fit(Xc, yc)
        assert hasattr(fast, "search_time_"), type(est).__name__  # the device path ran
        np.testing.assert_allclose(fast.cv_results_["mean_test_score"], slow.cv_results_["mean_test_score"],
                                   rtol=1e-5, err_msg=type(est).__name__)
        assert fast.best_params_ == slow.best_params_
        be, bs = fast.best_estimator_, slow.best_estimator_
        np.testing.assert_allclose(be.coef_, bs.coef_, rtol=0, atol=1e-5 * np.max(np.abs(bs.coef_)))
        assert be.intercept_ == pytest.approx(bs.intercept_, rel=1e-5, abs=1e-9)
        assert be.n_iter_ == bs.n_iter_
        np.testing.assert_allclose(be.adaptive_weights_, bs.adaptive_weights_, rtol=1e-3)


@pytest.mark.gpu
def test_overlap_and_standardized_searches_take_the_generic_path(golden):
    from sparselm_amd.model import OverlapGroupLasso

    X, y = golden["grp_X"][:, :24], golden["grp_y"]
    group_list = [[j // 4] + ([j // 4 + 1] if j % 4 == 3 and j // 4 + 1 < 6 else []) for j in range(24)]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        gs = GridSearchCV(OverlapGroupLasso(group_list=group_list), {"alpha": [5.0, 1.0, 0.2]}, cv=3).fit(X, y)
        ref = SkGridSearchCV(OverlapGroupLasso(group_list=group_list), {"alpha": [5.0, 1.0, 0.2]}, cv=3,
                             scoring="neg_root_mean_squared_error").fit(X, y)
    assert not hasattr(gs, "search_time_")
    np.testing.assert_allclose(gs.cv_results_["mean_test_score"], ref.cv_results_["mean_test_score"], rtol=1e-8)
    np.testing.assert_allclose(gs.best_estimator_.coef_, ref.best_estimator_.coef_, atol=1e-8)


@pytest.mark.gpu
@pytest.mark.parametrize("intercept", [False, True])
def test_device_fast_path_against_the_oracle_backend(golden, intercept):
    """The device-resident grid path checked DIRECTLY against the CPU oracle (not against the HIP generic loop):
    scikit-learn's own GridSearchCV over the same estimators with every fit routed through oracle.fista has to
    produce the same cv_results_, the same selection and the same refitted coefficients."""
    X, y, groups, gw = golden["grp_X"], golden["grp_y"], golden["grp_groups"], golden["grp_gw"]
    cv = KFold(4, shuffle=True, random_state=3)
    cases = [
        (Lasso(fit_intercept=intercept), {"alpha": list(np.geomspace(20, 0.05, 6))}),
        (GroupLasso(groups=groups, group_weights=gw, fit_intercept=intercept), {"alpha": list(np.geomspace(20, 0.1, 5))}),
        (SparseGroupLasso(groups=groups, fit_intercept=intercept), {"alpha": list(np.geomspace(10, 0.1, 4)), "l1_ratio": [0.2, 0.8]}),
        (AdaptiveLasso(fit_intercept=intercept, max_iter=3), {"alpha": list(np.geomspace(5, 0.1, 4))}),
    ]
    for est, grid in cases:
        est.set_params(solver_options={"tol": 1e-11})
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            fast = GridSearchCV(est, grid, cv=cv).fit(X, y)
            assert hasattr(fast, "search_time_")  # the device path ran
            with _backend.use_backend(OracleBackend()):
                ref_est = est.__class__(**{**est.get_params(), "solver_options": {"tol": 1e-13}})
                ref = SkGridSearchCV(ref_est, grid, cv=cv, scoring="neg_root_mean_squared_error").fit(X, y)
        assert fast.cv_results_["params"] == ref.cv_results_["params"]
        for f in range(4):
            np.testing.assert_allclose(fast.cv_results_[f"split{f}_test_score"], ref.cv_results_[f"split{f}_test_score"],
                                       rtol=1e-6, atol=1e-8)
        assert fast.best_index_ == ref.best_index_
        scale = np.max(np.abs(ref.best_estimator_.coef_))
        np.testing.assert_allclose(fast.best_estimator_.coef_, ref.best_estimator_.coef_, rtol=0, atol=1e-6 * scale)
        assert fast.best_estimator_.intercept_ == pytest.approx(ref.best_estimator_.intercept_, abs=1e-6 * max(1.0, abs(ref.best_estimator_.intercept_)))


def test_one_std_rule_applies_without_refit_on_the_generic_path():
    """reference model_selection.py:356-372: the rule picks best_index_ for every single-metric search, refitted or
    not -- LineSearchCV(refit=False) hands best_params_ from one line to the next."""
    X, y = make_regression(n_samples=120, n_features=40, n_informative=6, noise=25.0, random_state=1)
    params = {"alpha": np.logspace(-1, 1.5, 9)}
    cv = KFold(4, shuffle=True, random_state=0)
    with_refit = GridSearchCV(SkLasso(), params, opt_selection_method="one_std_score", cv=cv).fit(X, y)
    no_refit = GridSearchCV(SkLasso(), params, opt_selection_method="one_std_score", cv=cv, refit=False).fit(X, y)
    plain = GridSearchCV(SkLasso(), params, cv=cv, refit=False).fit(X, y)
    assert no_refit.best_index_ == with_refit.best_index_ == select_best_index_onestd(no_refit.cv_results_)
    assert no_refit.best_params_ == with_refit.best_params_
    assert no_refit.best_params_["alpha"] > plain.best_params_["alpha"]  # (on this data the rule does move the choice)
    assert no_refit.best_score_ == pytest.approx(no_refit.cv_results_["mean_test_score"][no_refit.best_index_])
    assert not hasattr(no_refit, "best_estimator_")


def test_world_size_in_the_environment_without_a_process_group_means_one_rank(monkeypatch):
    """torchrun / SLURM export WORLD_SIZE whether or not the script ever calls init_process_group: without a group
    there is nobody to gather from, so the search must not deal 1/world of the grid to this process."""
    from sparselm_amd import distributed as D
    from sparselm_amd.model_selection import _gather

    monkeypatch.setenv("WORLD_SIZE", "4")
    monkeypatch.setenv("RANK", "2")
    assert D.active_world() == (0, 1)
    monkeypatch.delenv("WORLD_SIZE")
    assert D.active_world() == (0, 1)
    units = [(0, 0), (0, 1), (1, 0)]
    full = {u: ([0], np.zeros(1), 0.0) for u in units}
    assert _gather(full, units) == full
    with pytest.raises(RuntimeError, match="not solved by any rank"):  # the check runs on the single-rank branch too
        _gather({units[0]: full[units[0]]}, units)


def test_grids_over_non_penalty_parameters_leave_the_device_path():
    """The device path sets the design, the groups and the preprocessing up once, from the base estimator: a grid that
    varies anything but penalty parameters (standardize, fit_intercept, groups, ...) must go through the generic loop."""
    from sparselm_amd.model import AdaptiveLasso, GroupLasso, SparseGroupLasso

    groups = np.repeat(np.arange(4), 5)
    ok = lambda est, grid: GridSearchCV(est, grid)._fast_path_ok({})  # noqa: E731
    assert ok(SparseGroupLasso(groups=groups), {"alpha": [1.0, 0.1], "l1_ratio": [0.2, 0.8]})
    assert ok(GroupLasso(groups=groups), [{"alpha": [1.0]}, {"alpha": [0.1], "group_weights": [np.ones(4)]}])
    assert not ok(SparseGroupLasso(groups=groups), {"alpha": [1.0, 0.1], "standardize": [False, True]})
    assert not ok(GroupLasso(groups=groups), {"alpha": [1.0, 0.1], "fit_intercept": [False, True]})
    assert not ok(GroupLasso(groups=groups), {"alpha": [1.0], "groups": [groups, groups[::-1]]})
    assert not ok(GroupLasso(groups=groups), [{"alpha": [1.0]}, {"alpha": [0.1], "standardize": [True]}])
    assert not ok(GroupLasso(groups=groups), {"alpha": [1.0], "max_iter": [2, 3]})  # (not a parameter of the plain class's loop)
    assert ok(AdaptiveLasso(), {"alpha": [1.0, 0.1], "max_iter": [2, 3], "eps": [1e-6, 1e-4]})
    assert not ok(AdaptiveLasso(), {"alpha": [1.0, 0.1], "fit_intercept": [True]})


@pytest.mark.gpu
def test_the_shares_of_an_eight_rank_search_reproduce_the_one_rank_table(monkeypatch):
    """Sixteen-lane searches deal path POINTS to the lane slots of all ranks (distributed.plan_lane_calls): pieces of
    paths spread over the whole alpha range, short pieces of two paths walked one after the other in one lane.  The
    eight shares of an 8-rank search, solved here one after the other, must tile the one-rank table exactly once and
    reproduce it -- and the generic path's, fit by fit."""
    monkeypatch.setenv("SLM_WS", "1")  # working set + split pass: sixteen lanes also for a problem of this size
    rng = np.random.default_rng(5)
    n, p, G = 3000, 120, 24
    groups = rng.permutation(np.repeat(np.arange(G), p // G))
    X = rng.standard_normal((n, p))
    beta = np.zeros(p)
    for g in rng.choice(G, 5, replace=False):
        beta[groups == g] = rng.uniform(1.0, 4.0, p // G) * rng.choice([-1, 1], p // G)
    y = X @ beta + 2.0 * rng.standard_normal(n)
    grid = {"alpha": list(np.geomspace(3.0, 0.01, 22)), "l1_ratio": [0.2, 0.5, 0.8]}
    cv = KFold(5, shuffle=True, random_state=0)
    est = SparseGroupLasso(groups=groups, solver_options={"tol": 1e-11})
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        search = GridSearchCV(est, grid, cv=cv, lanes=16)  # (the plan checked below is the sixteen-lane one; thirty-two: next test)
        one = search._device_cells(X, y, rank=0, world=1)
        shares = [search._device_cells(X, y, rank=r, world=8) for r in range(8)]
        fitted = GridSearchCV(est, grid, cv=cv).fit(X, y)
        slow = SkGridSearchCV(est, {"alpha": grid["alpha"][::7], "l1_ratio": [0.5]}, cv=cv,
                              scoring="neg_root_mean_squared_error").fit(X, y)
    assert one.shape == (66, 5) and not np.isnan(one).any()
    owners = np.sum([~np.isnan(s) for s in shares], axis=0)
    assert np.all(owners == 1), "every (candidate, fold) cell belongs to exactly one rank"
    assert all((~np.isnan(s)).sum() in range(39, 43) for s in shares)  # 330 fits over 8 ranks
    merged = np.nansum(shares, axis=0)
    np.testing.assert_allclose(merged, one, rtol=1e-8, atol=1e-10)
    table = np.c_[tuple(fitted.cv_results_[f"split{f}_test_score"] for f in range(5))]
    np.testing.assert_allclose(table, one, rtol=1e-8, atol=1e-10)
    # the lanes of this search do hold pieces (not whole paths), and some hold pieces of two paths
    from sparselm_amd.model_selection import _DeviceGrid

    g = _DeviceGrid(search, X, y, None)
    with g.open():
        plan = g.plan(8)
    lanes = [lane for calls in plan for call in calls for lane in call]
    assert max(len(idx) for lane in lanes for _, idx in lane) == 3 and sum(len(lane) > 1 for lane in lanes) == 5
    # an independent referee on a few cells: scikit-learn's loop around single fits
    params = fitted.cv_results_["params"]
    for q, sp in zip(slow.cv_results_["params"], range(len(slow.cv_results_["params"]))):
        i = params.index(q)
        for f in range(5):
            assert merged[i, f] == pytest.approx(slow.cv_results_[f"split{f}_test_score"][sp], rel=1e-7)


@pytest.mark.gpu
def test_an_eight_rank_search_with_the_folds_grams_built_together(monkeypatch):
    """Grid mode with the folds' Grams: the eight ranks of a search -- eight engines of an in-process communicator, a
    thread each, every one with a replica of (X, y) -- build the Grams of the five folds together (every rank the parts of
    its eighth of the rows, one sum over the ranks per fold) and solve their shares from them; the merged table is the
    one-rank table over X."""
    import threading

    from sparselm_amd import _engine

    monkeypatch.setenv("SLM_WS", "1")
    rng = np.random.default_rng(6)
    n, p, G = 2400, 96, 16
    groups = rng.permutation(np.repeat(np.arange(G), p // G))
    X = rng.standard_normal((n, p))
    beta = np.zeros(p)
    for g in rng.choice(G, 4, replace=False):
        beta[groups == g] = rng.uniform(1.0, 4.0, p // G) * rng.choice([-1, 1], p // G)
    y = X @ beta + 2.0 * rng.standard_normal(n)
    grid = {"alpha": list(np.geomspace(3.0, 0.01, 20)), "l1_ratio": [0.3, 0.7]}
    cv = KFold(5, shuffle=True, random_state=0)
    world = 8
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        one = GridSearchCV(SparseGroupLasso(groups=groups, solver_options={"tol": 1e-11, "covariance": False}), grid, cv=cv)._device_cells(X, y)
        engines = [_engine.Engine(0) for _ in range(world)]
        shares, errors, counts = [None] * world, [], [None] * world
        try:
            _engine.init_local_comm(engines, timeout_s=120.0)

            def work(r):
                try:
                    est = SparseGroupLasso(groups=groups, solver_options={"tol": 1e-11, "covariance": True})
                    shares[r] = GridSearchCV(est, grid, cv=cv)._device_cells(X, y, rank=r, world=world, engine=engines[r])
                    counts[r] = engines[r].comm_collectives()
                except BaseException as exc:  # noqa: BLE001
                    errors.append(exc)

            threads = [threading.Thread(target=work, args=(r,)) for r in range(world)]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
        finally:
            for e in engines:
                e.close()
    assert not errors, errors[:1]
    assert counts == [5] * world  # one sum per fold, nothing per pass
    owners = np.sum([~np.isnan(s) for s in shares], axis=0)
    assert np.all(owners == 1)
    np.testing.assert_allclose(np.nansum(shares, axis=0), one, rtol=1e-8, atol=1e-10)


@pytest.mark.gpu
def test_line_search_keeps_one_device_dataset_for_all_its_lines(monkeypatch):
    """LineSearchCV (reference model_selection.py:427-707) runs one GridSearchCV per line on the same (X, y): the device
    dataset is opened once and lent to every line (`_DatasetLease`), and the lines give what stand-alone searches give."""
    from sparselm_amd import model_selection as ms
    from sparselm_amd.model import SparseGroupLasso

    rng = np.random.default_rng(11)
    n, p = 600, 60
    X = rng.standard_normal((n, p))
    groups = np.arange(p) // 5
    coef = np.where(groups < 3, rng.standard_normal(p), 0.0)
    y = X @ coef + 0.5 * rng.standard_normal(n)
    grid = [("alpha", list(np.geomspace(1.0, 0.02, 6))), ("l1_ratio", [0.2, 0.5, 0.8])]
    opened = []
    real_open = ms._DeviceGrid.open
    monkeypatch.setattr(ms._DeviceGrid, "open", lambda self: (opened.append(1), real_open(self))[1])
    cached = []  # stand-alone searches go through the device dataset cache (same content: found again, not uploaded)
    real_cached = ms._DeviceGrid.open_cached
    monkeypatch.setattr(ms._DeviceGrid, "open_cached", lambda self: (cached.append(1), real_cached(self))[1])
    est = SparseGroupLasso(groups=groups, fit_intercept=True, solver_options={"tol": 1e-10})
    line = ms.LineSearchCV(est, grid, cv=3, n_iter=4).fit(X, y)
    assert len(opened) == 1 and len(line.history_) == 4
    # the same lines as stand-alone searches
    best = {"alpha": grid[0][1][0], "l1_ratio": grid[1][1][0]}
    for i, search in enumerate(line.history_):
        name, values = grid[i % 2]
        alone = ms.GridSearchCV(est, {k: (list(values) if k == name else [v]) for k, v in best.items()}, cv=3, refit=line.refit).fit(X, y)
        np.testing.assert_allclose(search.cv_results_["mean_test_score"], alone.cv_results_["mean_test_score"], rtol=1e-9)
        assert search.best_params_ == alone.best_params_
        best = dict(alone.best_params_)
    assert len(opened) == 1 and len(cached) == 4  # the line search opened ONE dataset; each stand-alone search asked the cache
    assert line.best_params_ == best


def test_covariance_auto_weighs_the_passes_against_the_grams():
    """`_DeviceGrid.covariance` (solver_options covariance="auto"): Grams are asked for when what the share's passes save
    exceeds them -- for BASELINE config 4's 2 500 fits on a K-fold split (whose test rows partition the rows: one triangle
    product in all), not for an eighth of it (a rank of eight), not for the same grid on a split that is no partition unless
    a line search has lines to come; False / True are obeyed; a dataset that cannot build them says no."""
    from types import SimpleNamespace

    from sparselm_amd import model_selection as ms

    built = []

    class FakeDataset:
        n, p = 100_000, 5_000

        def covariance(self, mask, n_eff):
            built.append(n_eff)

    def grid(option, points, adaptive=False, lease=None, lanes=16, kfold=True):
        g = ms._DeviceGrid.__new__(ms._DeviceGrid)
        g.est = SimpleNamespace(solver_options={"covariance": option}, max_iter=5)
        g.search = SimpleNamespace(**({"_lease": lease} if lease is not None else {}))
        g.adaptive, g.lanes, g.n_splits = adaptive, lanes, 5
        fold = np.arange(10) % 5
        g.test_masks = [(fold == f).astype(float) for f in range(5)] if kfold else [(np.arange(10) < 2).astype(float)] * 5
        g.train_masks = [1.0 - t for t in g.test_masks]
        calls = [[[(0, list(range(points)))]]]
        return g, calls

    for option, points, kw, want in (("auto", 2500, {}, True), ("auto", 320, {}, False), ("auto", 2500, {"kfold": False}, False),
                                     ("auto", 50_000, {"kfold": False}, True), (False, 50_000, {}, False), (True, 100, {}, True),
                                     ("auto", 2500, {"lease": SimpleNamespace(repeats=8), "kfold": False}, True),
                                     ("auto", 60, {"adaptive": True}, True), ("auto", 10, {"adaptive": True}, False)):
        built.clear()
        g, calls = grid(option, points, **kw)
        assert g.covariance(FakeDataset(), calls) is want, (option, points, kw)
        assert len(built) == (5 if want else 0)

    class Refusing(FakeDataset):
        def covariance(self, mask, n_eff):
            raise NotImplementedError("row-sharded")

    g, calls = grid(True, 100)
    assert g.covariance(Refusing(), calls) is False

    # a rank of eight: no for its eighth of config 4 on its own, yes when the ranks build the Grams together (a replica on
    # an engine with a communicator over all of them: an eighth of the products each, plus the exchange)
    class Shared(FakeDataset):
        engine = SimpleNamespace(comm_ranks=lambda: 8)

        def covariance_folds(self, masks, n_effs):
            built.extend(n_effs)

    built.clear()
    g, calls = grid("auto", 320)
    assert g.covariance(FakeDataset(), calls, 8) is False and not built
    assert g.covariance(Shared(), calls, 8) is True and len(built) == 5


def test_grid_values_are_told_apart_by_content_not_by_repr():
    """Two `group_weights` candidates of more than a thousand entries that differ only where numpy's repr elides the array
    are two units of the device grid, not one (each is solved with its own penalty)."""
    from sparselm_amd.model import GroupLasso
    from sparselm_amd.model_selection import _DeviceGrid, _value_key

    p = 1200
    w1 = np.ones(p)
    w2 = np.ones(p)
    w2[600] = 2.0
    assert repr(w1) == repr(w2) and _value_key(w1) != _value_key(w2)
    assert _value_key(0.5) == _value_key(0.5) and _value_key([1.0, 2.0]) == _value_key(np.array([1.0, 2.0]))
    rng = np.random.default_rng(0)
    X, y = rng.standard_normal((40, p)), rng.standard_normal(40)
    search = GridSearchCV(GroupLasso(groups=np.arange(p)), {"alpha": [1.0, 0.5], "group_weights": [w1, w2]}, cv=KFold(2))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        grid = _DeviceGrid(search, X, y, None)
    assert len(grid.combos) == 2 and sorted(len(c) for c in grid.combos) == [2, 2]


@pytest.mark.gpu
def test_device_search_refuses_a_design_with_nan_or_infinity(golden):
    """The device path of the search stands in for every cell's `fit`, whose validation would raise on a NaN or an infinity
    in X: the same ValueError, from a scan of the device copy (small and large designs alike), and for y from the host."""
    X, y = golden["l1_X"].copy(), golden["l1_y"].copy()
    grid = {"alpha": [1.0, 0.3, 0.1]}
    for bad, word in ((np.nan, "NaN"), (np.inf, "infinity")):
        Xb = X.copy()
        Xb[3, 1] = bad
        with pytest.raises(ValueError, match=word):
            GridSearchCV(Lasso(), grid, cv=3).fit(Xb, y)
    rng = np.random.default_rng(3)
    Xl = rng.standard_normal((1400, 900))  # (a design of the size from which `fit` itself leaves the scan to the device)
    yl = Xl[:, :4] @ np.ones(4) + rng.standard_normal(1400)
    Xl[700, 450] = np.nan
    with pytest.raises(ValueError, match="NaN"):
        GridSearchCV(Lasso(), grid, cv=3).fit(Xl, yl)
    yb = y.copy()
    yb[0] = np.inf
    with pytest.raises(ValueError):
        GridSearchCV(Lasso(), grid, cv=3).fit(X, yb)
    assert GridSearchCV(Lasso(), grid, cv=3).fit(X, y).best_params_["alpha"] in grid["alpha"]  # (clean data: as before)

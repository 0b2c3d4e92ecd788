"""Generate the committed golden fixtures under tests/golden/ (run once, in the build container).

Sources of truth, none of which is this repo's product code:
  * literal known answers transcribed from the reference's own tests
    (/root/reference/tests/test_lasso.py:29-61, tests/test_ols.py:35-66);
  * scikit-learn's coordinate-descent ``Lasso`` (identical objective 1/(2n)||y-Xb||^2 + alpha||b||_1),
    run at tol=1e-15 -- an independent solver for the l1 and weighted-l1 cases;
  * closed-form weighted least squares (alpha = 0);
  * group labels produced by the reference's ``sparselm.dataset.make_group_regression``
    (importable here from /root/reference/src: it needs only numpy + sklearn);
  * for group / sparse-group / ridged / adaptive fits (cvxpy is absent, the reference's tests hold no
    numbers): the oracle's solutions with their KKT residuals AND, next to each, the solution of
    tests/golden/second_solver.py -- an active-set method with Newton's method on the optimality
    conditions of the face (scipy.optimize.root), sharing no code with the oracle's proximal-gradient
    iteration (keys ``*_coef2``; tests/test_oracle.py holds the two to 1e-8 of each other).  Still
    "parity unpinned" against cvxpy itself, but no longer resting on one implementation.

Usage:  python tests/golden/make_golden.py     (writes tests/golden/*.npz)
"""

from __future__ import annotations

import os
import sys

import numpy as np
from sklearn.datasets import make_regression
from sklearn.linear_model import Lasso as SkLasso

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/src")

sys.path.insert(0, HERE)

import oracle  # noqa: E402
import second_solver  # noqa: E402
from sparselm.dataset import make_group_regression  # noqa: E402  (reference, numpy+sklearn only)


def sk_lasso(X, y, alpha, fit_intercept=False, sample_weight=None):
    m = SkLasso(alpha=alpha, fit_intercept=fit_intercept, tol=1e-15, max_iter=1_000_000)
    m.fit(X, y, sample_weight=sample_weight)
    return m.coef_.copy(), float(m.intercept_)


def main():
    out = {}

    # ---- 1. reference known answers (tests/test_lasso.py:29-61) ---------------------------------
    out["toy_X"] = np.array([[-1.0], [0.0], [1.0]])
    out["toy_y"] = np.array([-1.0, 0.0, 1.0])
    out["toy_T"] = np.array([[2.0], [3.0], [4.0]])
    out["toy_alpha"] = np.array([1e-8, 0.1, 0.5, 1.0])
    out["toy_coef"] = np.array([1.0, 0.85, 0.25, 0.0])
    out["toy_pred"] = np.array([[2, 3, 4], [1.7, 2.55, 3.4], [0.5, 0.75, 1.0], [0, 0, 0]], dtype=float)

    # ---- 2. l1 problems against scikit-learn CD ------------------------------------------------
    X, y = make_regression(n_samples=200, n_features=30, n_informative=8, noise=3.0, bias=2.5, random_state=7)
    rng = np.random.default_rng(11)
    sw = 0.5 + rng.uniform(size=200)
    out["l1_X"], out["l1_y"], out["l1_sw"] = X, y, sw
    alphas = np.array([0.05, 0.5, 2.0, 10.0, 40.0])
    out["l1_alpha"] = alphas
    out["l1_coef"] = np.array([sk_lasso(X, y, a)[0] for a in alphas])
    res = [sk_lasso(X, y, a, fit_intercept=True) for a in alphas]
    out["l1_coef_icpt"] = np.array([r[0] for r in res])
    out["l1_icpt"] = np.array([r[1] for r in res])
    res = [sk_lasso(X, y, a, fit_intercept=True, sample_weight=sw) for a in alphas]
    out["l1_coef_sw"] = np.array([r[0] for r in res])
    out["l1_icpt_sw"] = np.array([r[1] for r in res])

    # weighted l1 (the inner solve of AdaptiveLasso): min 1/(2n)||y - Xb||^2 + sum_j w_j|b_j|
    # == sklearn Lasso(alpha=1) on X/w with b = bt/w
    w = 0.2 + 3.0 * rng.uniform(size=30)
    bt, _ = sk_lasso(X / w, y, 1.0)
    out["wl1_w"] = w
    out["wl1_coef"] = bt / w

    # ---- 3. weighted least squares closed form (tests/test_ols.py:35-66), the alpha = 0 limit ----
    Xo = rng.normal(size=(10, 8))
    yo = rng.normal(size=10)
    swo = 1.0 + rng.uniform(size=10)
    W = np.diag(swo)
    out["ols_X"], out["ols_y"], out["ols_sw"] = Xo, yo, swo
    out["ols_coef"] = np.linalg.solve(Xo.T @ W @ Xo, Xo.T @ W @ yo)
    Xa = np.hstack([np.ones((10, 1)), Xo])
    th = np.linalg.solve(Xa.T @ W @ Xa, Xa.T @ W @ yo)
    out["ols_coef_icpt"], out["ols_icpt"] = th[1:], th[0]

    # ---- 4. grouped problem with the reference's label generator ------------------------------
    Xg, yg, groups, coef = make_group_regression(
        n_samples=150,
        n_groups=8,
        n_features_per_group=[3, 5, 2, 7, 4, 5, 1, 6],
        n_informative_groups=3,
        frac_informative_in_group=0.6,
        noise=4.0,
        coef=True,
        random_state=3,
    )
    out["grp_X"], out["grp_y"], out["grp_groups"], out["grp_true_coef"] = Xg, yg, groups, coef
    gw = 0.5 + rng.uniform(size=8)
    out["grp_gw"] = gw
    gidx, G = oracle.group_index(groups, Xg.shape[1])
    n, p = Xg.shape

    def kkt(beta, a, b, d, Xp=Xg, yp=yg):
        grad = Xp.T @ (Xp @ beta - yp) / len(yp)
        return oracle.kkt_residual(grad, beta, np.broadcast_to(a, (p,)), np.broadcast_to(b, (G,)), np.broadcast_to(d, (G,)), gidx, G)

    alpha_g = 3.0
    r = oracle.fit_group_lasso(Xg, yg, groups=groups, alpha=alpha_g, group_weights=gw)
    out["grp_gl_coef"], out["grp_gl_kkt"] = r["coef"], kkt(r["coef"], 0.0, alpha_g * gw, 0.0)
    r = oracle.fit_sparse_group_lasso(Xg, yg, groups=groups, l1_ratio=0.3, alpha=alpha_g, group_weights=gw)
    out["grp_sgl_coef"] = r["coef"]
    out["grp_sgl_kkt"] = kkt(r["coef"], 0.3 * alpha_g, 0.7 * alpha_g * gw, 0.0)
    delta = np.linspace(0.5, 2.0, 8)
    r = oracle.fit_ridged_group_lasso(Xg, yg, groups=groups, alpha=alpha_g, delta=delta, group_weights=gw)
    out["grp_delta"] = delta
    out["grp_rgl_coef"], out["grp_rgl_kkt"] = r["coef"], kkt(r["coef"], 0.0, alpha_g * gw, delta)
    out["grp_alpha"] = np.array(alpha_g)

    # adaptive variants (fit_intercept=True like the reference tests, tests/test_lasso.py:92-99)
    r = oracle.fit_adaptive_lasso(Xg, yg, alpha=1.5, fit_intercept=True)
    out["ada_l_coef"], out["ada_l_icpt"], out["ada_l_w"], out["ada_l_niter"] = r["coef"], r["intercept"], r["weights"], r["n_iter"]
    r = oracle.fit_adaptive_group_lasso(Xg, yg, groups=groups, alpha=1.5, group_weights=gw, fit_intercept=True)
    out["ada_gl_coef"], out["ada_gl_icpt"], out["ada_gl_w"], out["ada_gl_niter"] = r["coef"], r["intercept"], r["weights"], r["n_iter"]
    r = oracle.fit_adaptive_sparse_group_lasso(
        Xg, yg, groups=groups, l1_ratio=0.4, alpha=1.5, group_weights=gw, fit_intercept=True
    )
    out["ada_sgl_coef"], out["ada_sgl_icpt"], out["ada_sgl_w"], out["ada_sgl_niter"] = r["coef"], r["intercept"], r["weights"], r["n_iter"]
    r = oracle.fit_adaptive_ridged_group_lasso(
        Xg, yg, groups=groups, alpha=1.5, delta=(0.7,), group_weights=gw, fit_intercept=True
    )
    out["ada_rgl_coef"], out["ada_rgl_icpt"], out["ada_rgl_w"], out["ada_rgl_niter"] = r["coef"], r["intercept"], r["weights"], r["n_iter"]

    # the same fits by the second, independent solver (active set + Newton on the face)
    out["grp_gl_coef2"] = second_solver.group_lasso(Xg, yg, groups, alpha_g, gw)
    out["grp_sgl_coef2"] = second_solver.group_lasso(Xg, yg, groups, alpha_g, gw, l1_ratio=0.3)
    out["grp_rgl_coef2"] = second_solver.group_lasso(Xg, yg, groups, alpha_g, gw, delta=delta)
    for key, kw in (("ada_gl", {}), ("ada_sgl", {"l1_ratio": 0.4}), ("ada_rgl", {"delta": (0.7,)})):
        r2 = second_solver.adaptive(Xg, yg, groups, 1.5, gw, fit_intercept=True, **kw)
        out[f"{key}_coef2"], out[f"{key}_icpt2"], out[f"{key}_niter2"] = r2["coef"], r2["intercept"], r2["n_iter"]

    # ---- 4b. standardised sparse-group penalty (reference model/_lasso.py:627-639 with the group norms of
    # :249-252): the oracle's primal-dual solution with its optimality residual in the original problem
    # (oracle.kkt_standardized, certificate computed without the solver's dual iterate)
    r = oracle.fit_sparse_group_lasso(Xg, yg, groups=groups, l1_ratio=0.5, alpha=0.4, group_weights=gw, standardize=True)
    out["std_sgl_coef"] = r["coef"]
    out["std_sgl_kkt"] = oracle.kkt_standardized(Xg, yg, 0.2 * np.ones(p), 0.2 * gw, gidx, G, r["coef"])
    r = oracle.fit_adaptive_sparse_group_lasso(
        Xg, yg, groups=groups, l1_ratio=0.4, alpha=0.8, group_weights=gw, fit_intercept=True, standardize=True
    )
    out["std_ada_sgl_coef"], out["std_ada_sgl_icpt"], out["std_ada_sgl_w"], out["std_ada_sgl_niter"] = (
        r["coef"], r["intercept"], r["weights"], r["n_iter"])

    # ---- 5. AdaptiveLasso inner-solve sequence pinned by sklearn CD (weights = alpha^2/(|b|+eps)) -
    alpha_a, eps = 1.5, 1e-6
    wseq = [alpha_a * np.ones(30)]
    bseq = []
    for _ in range(3):
        wj = wseq[-1]
        bt, _ = sk_lasso(X / wj, y, 1.0)
        bseq.append(bt / wj)
        wseq.append(alpha_a * (alpha_a / (np.abs(bseq[-1]) + eps)))
    out["ada_sk_alpha"] = np.array(alpha_a)
    out["ada_sk_coefs"] = np.array(bseq)
    out["ada_sk_weights"] = np.array(wseq)

    path = os.path.join(HERE, "lasso_family_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")
    for k in ("grp_gl_kkt", "grp_sgl_kkt", "grp_rgl_kkt", "std_sgl_kkt"):
        print(k, out[k])


if __name__ == "__main__":
    main()

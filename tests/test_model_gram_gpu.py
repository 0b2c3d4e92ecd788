"""The model Gram (csrc/mg_kernels.hpp): what lanes beyond the working set's 512 columns iterate on between two passes
over X -- an fp16 product of the whole of X^T W X / n that only ever proposes points.  (1) its entries against numpy, tile
edges, row weights and columns of very different scale included; (2) dense-ended paths with the rounds forced on at sizes
the oracle finishes (SLM_MG=2), against the same calls without (FLAG_NO_MODEL_GRAM) and against oracle.fista -- plain and
grouped penalties, lanes with their own row masks; (3) the same bits run to run.  The full-size case is
tests/test_baseline_configs_gpu.py::test_headline_path_with_a_dense_end_*.  Reference: the cvxpy solve has no density
regime (/root/reference/src/sparselm/model/_base.py:512-519)."""

import numpy as np
import pytest

import oracle
from sparselm_amd import _engine

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    return _engine.get_engine(0)


@pytest.mark.parametrize("n,p,weighted,scales", [(3000, 200, False, False), (1037, 77, True, False), (6500, 300, False, True),
                                                 (12900, 129, True, True), (64, 16, False, False), (257, 130, False, False)])
def test_model_gram_entries_against_numpy(eng, n, p, weighted, scales):
    """fp16 operands, fp32 accumulation over chunks of rows, chunks summed in fp64: every entry within 2e-3 of the columns'
    scales (measured: 2e-5 ... 2e-4), the spectral error of the scaled matrix below 1e-3, the matrix symmetric to the bit,
    pad rows and columns exactly zero."""
    rng = np.random.default_rng(n + p)
    X = rng.standard_normal((n, p))
    if scales:
        X *= 10.0 ** rng.uniform(-6, 6, p)
    y = rng.standard_normal(n)
    w = rng.uniform(0.2, 3.0, n) if weighted else None
    with eng.dataset(X, y, row_weight=w) as ds:
        G = ds.model_gram(download=True)
    ld = (p + 15) // 16 * 16
    assert G.shape == (ld, ld)
    W = np.ones(n) if w is None else w
    ref = (X * W[:, None]).T @ X / n
    d = np.sqrt(np.diag(ref))
    assert np.max(np.abs(G[:p, :p] - ref) / np.outer(d, d)) < 2e-3
    assert np.linalg.norm((G[:p, :p] - ref) / np.outer(d, d), 2) < 2e-3
    assert np.array_equal(G, G.T)
    assert not G[p:, :].any()


def _dense_problem(rng, n, p, k, noise):
    X = rng.standard_normal((n, p))
    bt = np.zeros(p)
    bt[rng.choice(p, k, replace=False)] = 10 * rng.uniform(0.2, 1.0, k)
    return X, X @ bt + noise * rng.standard_normal(n)


@pytest.mark.parametrize("n,p,k,noise,lo,grouped", [(4000, 640, 30, 100.0, 1e-3, False), (6000, 800, 60, 30.0, 1e-2, False),
                                                    (5000, 600, 12, 100.0, 1e-2, True)])
def test_dense_ended_paths_on_the_model_gram_match_plain_steps_and_the_oracle(eng, monkeypatch, n, p, k, noise, lo, grouped):
    monkeypatch.setenv("SLM_MG", "2")  # (rounds at any size, from the first snapshot on)
    rng = np.random.default_rng(p)
    X, y = _dense_problem(rng, n, p, k, noise)
    gid = None
    with eng.dataset(X, y) as ds:
        if grouped:
            G = p // 10
            gid = rng.permutation(np.repeat(np.arange(G), 10)).astype(np.int32)
            ds.set_groups(gid, G)
        g0, _ = ds.gradient(None)
        if grouped:
            amax = float(np.max(np.sqrt(np.bincount(gid, weights=g0 * g0, minlength=G))))
            pts = [(0.0, a, 0.0) for a in np.geomspace(amax, lo * amax, 40)]
        else:
            amax = float(np.max(np.abs(g0)))
            pts = [(a, 0.0, 0.0) for a in np.geomspace(amax, lo * amax, 40)]
        r = ds.solve_path(pts, lanes=16, flags=_engine.FLAG_WORKING_SET, tol=1e-9)
        r2 = ds.solve_path(pts, lanes=16, flags=_engine.FLAG_WORKING_SET, tol=1e-9)
        q = ds.solve_path(pts, lanes=16, flags=_engine.FLAG_WORKING_SET | _engine.FLAG_NO_MODEL_GRAM, tol=1e-9)
    assert r.converged and q.converged
    assert r.mg_rounds > 0 and r.mg_inner_iters > 0 and q.mg_rounds == 0
    assert r.grad_launches < q.grad_launches  # (measured: 14 / 10 / 11 passes against 76 / 31 / 49)
    assert np.count_nonzero(r.betas[-1]) > 512
    assert np.max(np.abs(r.betas - q.betas)) < 1e-6 * np.max(np.abs(q.betas))
    assert np.array_equal(r.betas, r2.betas)  # (fixed-order sums in the build and in every round)
    gidx, Gn = oracle.group_index(gid, p)
    for kk in (len(pts) // 2, len(pts) - 1):
        a, b, d = pts[kk]
        bo, _ = oracle.fista(X, y, a, b, d, gidx, Gn, beta0=q.betas[kk], tol=1e-13)
        assert np.max(np.abs(r.betas[kk] - bo)) < 1e-6 * np.max(np.abs(bo))


def test_lanes_with_their_own_row_masks_take_the_model_gram_of_their_fold(eng, monkeypatch):
    """The lanes of a grid search: three folds' training masks with 1 / n_train scaling, three paths per fold --
    every row set of the call gets its own model Gram (found again by the fingerprint of the mask on a second call)."""
    monkeypatch.setenv("SLM_MG", "2")
    rng = np.random.default_rng(5)
    n, p = 4500, 560
    X, y = _dense_problem(rng, n, p, 25, 60.0)
    fold = rng.integers(0, 3, n)
    masks = [(fold != f).astype(np.float64) for f in range(3)]
    with eng.dataset(X, y) as ds:
        lanes = []
        for f in range(3):
            m = masks[f]
            amax = float(np.max(np.abs(X.T @ (m * y)))) / m.sum()
            for lo in (1e-2, 3e-3, 1e-3):  # (nine lanes: more than the fused kernels serve -- the split pass takes the call)
                lanes.append({"points": [(a, 0.0, 0.0) for a in np.geomspace(amax, lo * amax, 12)], "row_weight": m, "n_eff": int(m.sum())})
        r = ds.solve_lanes(lanes, tol=1e-9, flags=_engine.FLAG_WORKING_SET)
        again = ds.solve_lanes(lanes, tol=1e-9, flags=_engine.FLAG_WORKING_SET)
        q = ds.solve_lanes(lanes, tol=1e-9, flags=_engine.FLAG_WORKING_SET | _engine.FLAG_NO_MODEL_GRAM)
    assert r[0].mg_rounds > 0 and again[0].mg_rounds > 0 and again[0].mg_build_ms == 0.0  # (the Grams were found again)
    gidx, Gn = oracle.group_index(None, p)
    for l, (a_lane, b_lane) in enumerate(zip(r, q)):
        assert a_lane.converged and b_lane.converged
        assert np.max(np.abs(a_lane.betas - b_lane.betas)) < 1e-6 * np.max(np.abs(b_lane.betas))
        assert np.array_equal(a_lane.betas, again[l].betas)
    m = masks[1]
    a_last = lanes[3]["points"][-1][0]
    bo, _ = oracle.fista(X[m > 0], y[m > 0], a_last, 0.0, 0.0, gidx, Gn, beta0=q[3].betas[-1], tol=1e-13)
    assert np.max(np.abs(r[3].betas[-1] - bo)) < 1e-6 * np.max(np.abs(bo))


@pytest.mark.parametrize("lanes,K", [(18, 50), (16, 50), (20, 57), (7, 31)])
def test_tail_points_change_hands_at_the_dense_end_of_an_interleaved_path(eng, monkeypatch, lanes, K):
    """A shared path whose points beyond the last full band belong to the LAST lanes (interleaved_walk): once the rounds on
    the model Gram are on, a lane that has finished takes over a tail point its owner has not started
    (tail_handover_kernel).  Which lane solves a point changes nothing that is reported: against the same path without the
    hand-over, against one plain lane, and the same bits run to run."""
    monkeypatch.setenv("SLM_MG", "2")
    rng = np.random.default_rng(lanes * 100 + K)
    n, p = 4200, 660
    X, y = _dense_problem(rng, n, p, 25, 80.0)
    WS, PLAIN = _engine.FLAG_WORKING_SET, _engine.FLAG_NO_WORKING_SET
    with eng.dataset(X, y) as ds:
        g0, _ = ds.gradient(None)
        amax = float(np.max(np.abs(g0)))
        pts = [(a, 0.0, 0.0) for a in np.geomspace(amax, 1e-3 * amax, K)]
        with_h = ds.solve_path(pts, lanes=lanes, flags=WS, tol=1e-9)
        again = ds.solve_path(pts, lanes=lanes, flags=WS, tol=1e-9)
        monkeypatch.setenv("SLM_NO_HANDOVER", "1")
        without = ds.solve_path(pts, lanes=lanes, flags=WS, tol=1e-9)
        monkeypatch.delenv("SLM_NO_HANDOVER")
        one = ds.solve_path(pts, lanes=1, flags=PLAIN, tol=1e-10)
    assert with_h.converged and without.converged and one.converged and with_h.mg_rounds > 0
    assert np.count_nonzero(with_h.betas[-1]) > 512  # (the end of the path lies beyond the working set)
    assert with_h.grad_launches <= without.grad_launches + 1
    scale = np.max(np.abs(one.betas))
    assert np.max(np.abs(with_h.betas - without.betas)) < 1e-6 * scale
    assert np.max(np.abs(with_h.betas - one.betas)) < 1e-6 * scale
    assert np.array_equal(with_h.betas, again.betas)


def test_randomised_cross_check_of_the_rounds():
    """tools/mg_fuzz.py: the rounds forced on over random penalty kinds, group sizes, 9-16 lanes, shared paths and lanes with
    fold masks, iid / correlated / duplicated / badly scaled designs, more columns than rows, dataset row weights: every
    call against the same call without the rounds and against one plain lane (1e-6, or the same objective where the
    minimiser is not unique)."""
    import os
    import subprocess
    import sys

    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "mg_fuzz.py"), "60", "11"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "MG FUZZ cases 60" in out.stdout and "flagged 0" in out.stdout
    assert "with rounds 0 " not in out.stdout  # (the rounds did run)

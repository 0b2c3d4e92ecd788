"""The Grams behind covariance passes against numpy, entry by entry (csrc/cov_kernels.hpp: cov_syrk_kernel<3> / <4>,
cov_rows / cov_combine / cov_sum, the folds' linear terms through xtr_mfma_kernel), the covariance route against the
oracle directly, and the Grams of a replicated dataset summed from the row blocks of eight ranks (grid mode's one
collective; the reference's counterpart is the joblib dispatch of /root/reference/src/sparselm/model_selection.py:273,
304-323).  ``slm_dataset_covariance_download`` is the window."""

import os
import threading

import numpy as np
import pytest

import oracle
from sparselm_amd import _engine

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    return _engine.get_engine(0)


def _expect(X, y, w, n_eff):
    w = np.ones(len(y)) if w is None else w
    return X.T @ (w[:, None] * X) / n_eff, X.T @ (w * y) / n_eff, float(np.sum(w * y * y) / n_eff)


def _check_entry(ds, index, X, y, w, n_eff, tol=2e-13):
    G, c, sc = ds.covariance_download(index)
    Ge, ce, yye = _expect(X, y, w, n_eff)
    scale = max(float(np.max(np.abs(Ge))), 1e-300)
    assert np.max(np.abs(G - Ge)) <= tol * scale * max(1.0, np.sqrt(X.shape[0]) / 8), (index, np.max(np.abs(G - Ge)) / scale)
    assert np.array_equal(G, G.T)  # (the mirror is written with the triangle)
    assert np.max(np.abs(c - ce)) <= 1e-12 * max(float(np.max(np.abs(ce))), 1e-300)
    assert abs(sc["yy"] - yye) <= 1e-12 * yye
    assert sc["n_eff"] == n_eff


# rows below four, rows % 4 != 0, ld not a multiple of 96 or 128 (p = 50: ld = 64; p = 200: ld = 208; p = 100: ld = 112),
# one wider than a tile of either size
SHAPES = [(3, 7), (5, 50), (37, 50), (1001, 200), (643, 100), (2000, 300)]


@pytest.mark.parametrize("tile", ["3", "4"])
@pytest.mark.parametrize("n,p", SHAPES)
def test_gram_entries_against_numpy(eng, n, p, tile, monkeypatch):
    """Every way a Gram is built -- all rows, a 0/1 mask that leaves out fewer than half of the rows (all rows minus the rows
    left out), one that leaves out more (the kept rows scaled into a copy), general weights -- on both tile sizes of the
    triangle product, against X^T W X / n_eff, X^T W y / n_eff and y^T W y / n_eff in numpy."""
    monkeypatch.setenv("SLM_COV_TILE", tile)
    rng = np.random.default_rng(n * 1000 + p)
    X = rng.standard_normal((n, p)) * rng.uniform(0.2, 3.0, p)
    y = rng.standard_normal(n) * 3.0 + 1.0
    few = (rng.uniform(size=n) > 0.2).astype(float)
    if few.sum() == 0:
        few[0] = 1.0
    most_out = np.zeros(n)
    most_out[rng.choice(n, max(1, n // 4), replace=False)] = 1.0
    w = rng.uniform(0.1, 2.5, n)
    with eng.dataset(X, y) as ds:
        ds.covariance(None, 0)
        _check_entry(ds, 0, X, y, None, float(n))
        ds.covariance(few, int(few.sum()))
        _check_entry(ds, 1, X, y, few, float(few.sum()))
        ds.covariance(most_out, int(most_out.sum()))
        _check_entry(ds, 2, X, y, most_out, float(most_out.sum()))
        ds.covariance(w, 0)
        _check_entry(ds, 3, X, y, w, float(n))
        ds.covariance(w, 17)  # (the same weights under another scaling are another row set)
        _check_entry(ds, 4, X, y, w, 17.0)
        assert ds.covariance_count() == 5


@pytest.mark.parametrize("n,p,k", [(11, 7, 3), (643, 100, 5), (2000, 300, 5), (1001, 200, 4)])
def test_fold_grams_against_numpy_and_their_partition_sum(eng, n, p, k):
    """The folds of a K-fold split at once (no Gram of all rows is formed from X: it is the sum of the test rows' Grams): every
    fold's training Gram, linear term and y^T W y against numpy; the folds' test-row Grams add up to X^T X."""
    rng = np.random.default_rng(7 * n + p)
    X = rng.standard_normal((n, p))
    y = rng.standard_normal(n) + 0.5 * X[:, 0]
    folds = rng.permutation(n) % k
    masks = [(folds != f).astype(float) for f in range(k)]
    with eng.dataset(X, y) as ds:
        ds.covariance_folds(masks, [int(m.sum()) for m in masks])
        assert ds.covariance_count() == k
        total = np.zeros((p, p))
        for f, m in enumerate(masks):
            _check_entry(ds, f, X, y, m, float(m.sum()))
            G, _, _ = ds.covariance_download(f)
            total += X.T @ X - G * m.sum()  # = the test rows' Gram
        np.testing.assert_allclose(total, X.T @ X, rtol=0, atol=1e-11 * np.max(np.abs(X.T @ X)))
        # in two steps: the same entries again on a fresh dataset
        with eng.dataset(X, y) as ds2:
            assert ds2.covariance_folds_begin(masks, [int(m.sum()) for m in masks])
            ds2.covariance_folds_finish()
            for f in range(k):
                a, b = ds.covariance_download(f), ds2.covariance_download(f)
                assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]
            # masks that are no partition: nothing is started, and the one-call form builds them mask by mask
        overlapping = [masks[0], masks[0] * masks[1]]
        with eng.dataset(X, y) as ds3:
            assert not ds3.covariance_folds_begin(overlapping, [int(m.sum()) for m in overlapping])
            with pytest.raises(ValueError):
                ds3.covariance_folds_finish()
            ds3.covariance_folds(overlapping, [int(m.sum()) for m in overlapping])
            for f, m in enumerate(overlapping):
                _check_entry(ds3, f, X, y, m, float(m.sum()))


def test_covariance_route_against_the_oracle(eng):
    """Sparse-group paths on the folds of a 5-fold split, solved from the folds' Grams, against the oracle's FISTA on the
    training rows themselves (not against the engine's route over X)."""
    n, p = 3000, 120
    rng = np.random.default_rng(3)
    G = p // 6
    groups = rng.permutation(np.repeat(np.arange(G), 6))
    coef = np.zeros(p)
    for g in rng.choice(G, 4, replace=False):
        coef[groups == g] = 5.0 * rng.uniform(0.2, 1.0, 6)
    X = rng.standard_normal((n, p))
    y = X @ coef + 2.0 * rng.standard_normal(n)
    folds = rng.permutation(n) % 5
    masks = [(folds != f).astype(float) for f in range(5)]
    gidx, Gn = oracle.group_index(groups, p)
    with eng.dataset(X, y) as ds:
        ds.set_groups(gidx, Gn)
        g0, _ = ds.gradient(None)
        amax = float(np.max(np.abs(g0)))
        al = np.geomspace(0.7 * amax, 0.02 * amax, 6)
        pts = np.c_[0.4 * al, 0.6 * al, 0 * al]
        ds.covariance_folds(masks, [int(m.sum()) for m in masks])
        specs = [dict(points=pts, row_weight=m, n_eff=int(m.sum())) for m in masks]
        res = ds.solve_lanes(specs, tol=1e-11, flags=_engine.FLAG_WORKING_SET | _engine.FLAG_COVARIANCE)
    for f, (m, r) in enumerate(zip(masks, res)):
        assert r.converged
        tr = m > 0
        b = None
        for k, a in enumerate(al):
            b, info = oracle.fista(X[tr], y[tr], 0.4 * a, 0.6 * a, 0.0, gidx, Gn, beta0=b, tol=1e-13)
            assert info["converged"]
            assert np.max(np.abs(r.betas[k] - b)) <= 1e-8 * max(float(np.max(np.abs(b))), 1e-12), (f, k)


def _rank_threads(n_ranks, work):
    out, errors = [None] * n_ranks, []

    def run(r):
        try:
            out[r] = work(r)
        except BaseException as exc:  # noqa: BLE001 -- re-raised below
            errors.append(exc)

    threads = [threading.Thread(target=run, args=(r,)) for r in range(n_ranks)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    return out


@pytest.mark.parametrize("n,p,world", [(2003, 150, 8), (157, 40, 3)])
def test_grams_summed_from_the_row_blocks_of_the_ranks(eng, n, p, world):
    """Grid mode at `world` ranks: every rank holds all rows (a replica) and builds the folds' parts of ITS row block only; the
    parts are summed through the communicator (here the in-process one: `world` engines on this device, a thread each).
    Every rank ends up with the same Grams, bit for bit; they equal the single-rank Grams to rounding, and numpy's."""
    rng = np.random.default_rng(n + p)
    X = rng.standard_normal((n, p))
    y = rng.standard_normal(n) - X[:, 1]
    folds = rng.permutation(n) % 5
    masks = [(folds != f).astype(float) for f in range(5)]
    n_effs = [int(m.sum()) for m in masks]
    with eng.dataset(X, y) as one:
        one.covariance_folds(masks, n_effs)
        single = [one.covariance_download(f) for f in range(5)]
    engines = [_engine.Engine(0) for _ in range(world)]
    try:
        _engine.init_local_comm(engines, timeout_s=60.0)
        sets = [e.dataset(X, y) for e in engines]
        for d in sets:
            d.set_replicated(True)

        def work(r):
            assert sets[r].covariance_folds_begin(masks, n_effs)
            sets[r].covariance_folds_finish()
            return [sets[r].covariance_download(f) for f in range(5)]

        got = _rank_threads(world, work)
        for r in range(world):
            for f in range(5):
                G, c, sc = got[r][f]
                assert np.array_equal(G, got[0][f][0]) and np.array_equal(c, got[0][f][1]) and sc == got[0][f][2]
                scale = np.max(np.abs(single[f][0]))
                assert np.max(np.abs(G - single[f][0])) <= 1e-13 * scale
                assert np.max(np.abs(c - single[f][1])) <= 1e-13 * np.max(np.abs(single[f][1]))
                assert abs(sc["yy"] - single[f][2]["yy"]) <= 1e-13 * single[f][2]["yy"]
        for f in range(5):
            _check_entry(sets[0], f, X, y, masks[f], float(n_effs[f]))
        # a replica solves without any per-pass collective: the same path on two ranks at different times, same answer
        before = [e.comm_collectives() for e in engines]
        g0, _ = sets[0].gradient(None)
        al = np.geomspace(float(np.max(np.abs(g0))), 0.05 * float(np.max(np.abs(g0))), 5)
        spec = [dict(points=np.c_[al, 0 * al, 0 * al], row_weight=masks[2], n_eff=n_effs[2])]
        a = sets[0].solve_lanes(spec, tol=1e-10, flags=_engine.FLAG_WORKING_SET | _engine.FLAG_COVARIANCE)[0]
        b = sets[1].solve_lanes(spec, tol=1e-10, flags=_engine.FLAG_WORKING_SET | _engine.FLAG_COVARIANCE)[0]
        assert a.converged and np.array_equal(a.betas, b.betas)
        assert [e.comm_collectives() for e in engines] == before
        for d in sets:
            d.close()
    finally:
        for e in engines:
            e.close()


def _run_tool(name, *args):
    import subprocess
    import sys

    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    return subprocess.run([sys.executable, os.path.join(root, "tools", name), *args], capture_output=True, text=True, timeout=900)


def test_randomised_cross_check_of_covariance_passes():
    """tools/covariance_fuzz.py: 60 random calls (penalty kinds, group sizes, 1-16 lanes, folds / general weights / no
    weights as row sets, per-lane 1/n, warm starts, p > n, with and without the working set) solved over X and from the
    Grams of their row sets."""
    out = _run_tool("covariance_fuzz.py", "60", "1")
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "flagged 0" in out.stdout.splitlines()[-1]


def test_odd_shapes_through_the_on_chip_solvers_and_the_grams():
    """tools/edge_cases.py: one to three features, two rows, one group over everything, p = 128, one-point paths on sixteen
    lanes -- on chip, through the splitting, and from Grams -- each against the general path."""
    out = _run_tool("edge_cases.py")
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "edge cases done" in out.stdout.splitlines()[-1]

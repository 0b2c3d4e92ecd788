"""Row-sharded mode with MORE THAN ONE rank, on one GPU: the in-process communicator (slm_comm_init_local)
makes two or three engines of this process the ranks of one job, each driven by its own host thread, so the
multi-rank state machine of csrc/engine.hip -- gradient all-reduce, rank-agreed stop word, Gram exchange of
the working set, sharded centring -- runs for real where RCCL cannot form a group (it refuses several ranks
on one device).  The RCCL leg itself is measured by `bench.py --gpus N` (extra_legs.rowshard).

Reference semantics being reproduced: one `_solve` on the whole (X, y) (src/sparselm/model/_base.py:512-519)
and `_preprocess_data` centring on the whole matrix (:216-222); the reference has no distributed code.
"""

import threading

import numpy as np
import pytest

from sparselm_amd import _engine
from sparselm_amd.distributed import row_range

pytestmark = pytest.mark.gpu


def _run_ranks(n_ranks, work, timeout_s=30.0):
    """work(rank, engine) on n_ranks fresh engines joined by an in-process communicator; returns the results
    and the collective counts; re-raises the first exception of any rank."""
    engines = [_engine.Engine(0) for _ in range(n_ranks)]
    _engine.init_local_comm(engines, timeout_s=timeout_s)
    out = [None] * n_ranks
    err = [None] * n_ranks

    def body(r):
        try:
            out[r] = work(r, engines[r])
        except BaseException as exc:  # noqa: BLE001 - reported below
            err[r] = exc

    threads = [threading.Thread(target=body, args=(r,)) for r in range(n_ranks)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    counts = [e.comm_collectives() for e in engines]
    infos = [e.comm_info() for e in engines]
    for e in engines:
        e.comm_destroy()
        e.close()
    for exc in err:
        if exc is not None:
            raise exc
    return out, counts, infos


def _problem(n, p, seed, n_inf=10, noise=0.5):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, p)) + 0.3
    beta = np.zeros(p)
    beta[rng.choice(p, n_inf, replace=False)] = rng.uniform(1, 4, n_inf) * rng.choice([-1, 1], n_inf)
    y = X @ beta + noise * rng.standard_normal(n) + 2.0
    return X, y


def _alphas(X, y, k=8, lo=2e-2):
    amax = np.max(np.abs(X.T @ y)) / len(y)
    return np.geomspace(amax, lo * amax, k)


@pytest.mark.parametrize("n_ranks,flags", [(2, 0), (3, 0), (2, _engine.FLAG_WORKING_SET)])
def test_local_ranks_reproduce_the_single_rank_path(n_ranks, flags):
    X, y = _problem(3001, 257, seed=n_ranks)
    n = len(y)
    pts = [(a, 0.0, 0.0) for a in _alphas(X, y)]

    def work(r, eng):
        lo, hi = row_range(n, r, n_ranks)
        with eng.dataset(X[lo:hi], y[lo:hi]) as ds:
            ds.set_global_rows(n)
            g0, _ = ds.gradient(None)
            res = ds.solve_path(pts, tol=1e-10, flags=flags, lanes=2)
        return g0, res

    out, counts, infos = _run_ranks(n_ranks, work)
    assert infos == [(r, n_ranks) for r in range(n_ranks)]
    assert len(set(counts)) == 1 and counts[0] > 0  # every rank entered the same number of collectives
    # two collectives per pass -- the gradients, then the stop vector (in working-set solves behind the staged Gram parts
    # in one buffer) -- plus the gradient call above and the power steps of the step bound.  NOT one (round-5 verdict, item 7):
    # which passes grow the working set is decided on the device, from that pass's all-reduced gradient, and a collective's size
    # is fixed on the host when it is queued -- one collective per non-growing pass would need the host to look at every pass
    # before queueing the next (a round trip per pass, as long as the 2 MB all-reduce it saves), or the Gram parts a pass late
    # (a pass of X wherever the set has to grow before the model can move); the stop words cannot ride with the gradients of
    # their own pass, whose tail kernel decides them.  DESIGN.md section 6.  The count is pinned here from both sides.
    passes = out[0][1].grad_launches
    assert 2 * passes <= counts[0] <= 2 * (passes + 2) + 8, (counts[0], passes)
    with _engine.get_engine(0).dataset(X, y) as ds:
        g_ref, _ = ds.gradient(None)
        ref = ds.solve_path(pts, tol=1e-10, flags=flags, lanes=2)
    for g0, res in out:
        np.testing.assert_allclose(g0, g_ref, rtol=0, atol=1e-12 * np.max(np.abs(g_ref)))
        assert res.converged
        assert np.max(np.abs(res.betas - ref.betas)) <= 1e-8 * np.max(np.abs(ref.betas))
        # identical bits on every rank: the exchange adds the staged parts in rank order everywhere
        assert np.array_equal(res.betas, out[0][1].betas)
        assert res.grad_launches == out[0][1].grad_launches
    if flags:
        assert out[0][1].ws_refined > 0


def test_group_penalty_with_working_set_across_two_ranks():
    X, y = _problem(2400, 120, seed=7)
    n = len(y)
    groups = np.repeat(np.arange(24), 5)
    c = X.T @ y / n
    bmax = np.max(np.sqrt(np.bincount(groups, weights=c * c)))
    pts = [(0.3 * a, 0.7 * a, 0.0) for a in np.geomspace(bmax, 0.03 * bmax, 6)]

    def work(r, eng):
        lo, hi = row_range(n, r, 2)
        with eng.dataset(X[lo:hi], y[lo:hi]) as ds:
            ds.set_global_rows(n)
            ds.set_groups(groups, 24)
            return ds.solve_path(pts, tol=1e-10, flags=_engine.FLAG_WORKING_SET, want_group_norms=True)

    out, counts, _ = _run_ranks(2, work)
    assert counts[0] == counts[1]
    with _engine.get_engine(0).dataset(X, y) as ds:
        ds.set_groups(groups, 24)
        ref = ds.solve_path(pts, tol=1e-10, flags=_engine.FLAG_WORKING_SET, want_group_norms=True)
    for res in out:
        assert res.converged and res.ws_refined > 0
        assert np.max(np.abs(res.betas - ref.betas)) <= 1e-8 * np.max(np.abs(ref.betas))
        assert np.array_equal(res.betas, out[0].betas)
        np.testing.assert_allclose(res.group_norms, ref.group_norms, rtol=0, atol=1e-8 * np.max(ref.group_norms))


@pytest.mark.parametrize("flags", [0, _engine.FLAG_WORKING_SET])
def test_re_weighted_solves_across_two_ranks_carry_their_start(flags, monkeypatch):
    """The rounds of an adaptive group estimator on a row-sharded dataset: each re-weighted solve starts where the one
    before ended and leaves out its first pass -- and the all-reduce that goes with it -- on every rank alike (the
    decision is made from the caller's arrays, which the ranks share)."""
    X, y = _problem(2600, 150, seed=11)
    n = len(y)
    groups = np.repeat(np.arange(30), 5)
    c = X.T @ y / n
    alpha = 0.05 * np.max(np.sqrt(np.bincount(groups, weights=c * c)))

    def rounds(ds):
        ds.set_groups(groups, 30)
        b, beta, out, passes = alpha * np.ones(30), None, [], []
        for _ in range(3):
            res = ds.solve_path([(0.0, 1.0, 0.0)], b=b, beta0=beta, tol=1e-10, flags=flags, want_group_norms=True)
            assert res.converged
            beta = res.betas[0].copy()
            b = alpha * (alpha / (res.group_norms[0] + 1e-6))
            out.append(beta)
            passes.append(res.grad_launches)
        return out, passes

    def work(r, eng):
        lo, hi = row_range(n, r, 2)
        with eng.dataset(X[lo:hi], y[lo:hi]) as ds:
            ds.set_global_rows(n)
            return rounds(ds)

    out, counts, _ = _run_ranks(2, work)
    assert counts[0] == counts[1]
    monkeypatch.setenv("SLM_NO_CARRY", "1")
    plain, plain_counts, _ = _run_ranks(2, work)
    monkeypatch.delenv("SLM_NO_CARRY")
    with _engine.get_engine(0).dataset(X, y) as ds:
        ref, _ = rounds(ds)
    (b0, k0), (b1, k1) = out
    assert k0 == k1 and k0[0] == plain[0][1][0]
    assert all(a == b - 1 for a, b in zip(k0[1:], plain[0][1][1:])), (k0, plain[0][1])
    assert counts[0] < plain_counts[0]  # (fewer collectives, too)
    for k in range(3):
        assert np.array_equal(b0[k], b1[k])
        assert np.max(np.abs(b0[k] - ref[k])) <= 1e-7 * np.max(np.abs(ref[k]))


def test_centring_a_row_sharded_dataset_subtracts_the_global_means():
    X, y = _problem(1999, 61, seed=3)
    n = len(y)
    w = np.random.default_rng(0).uniform(0.2, 2.0, n)
    w *= n / w.sum()
    alpha = 0.05 * np.max(np.abs(X.T @ y)) / n

    def work(r, eng):
        lo, hi = row_range(n, r, 2)
        with eng.dataset(X[lo:hi], y[lo:hi], row_weight=w[lo:hi]) as ds:
            ds.set_global_rows(n)
            xm, ym = ds.center()
            Xc, yc = ds.download()
            res = ds.solve_path([(alpha, 0.0, 0.0)], tol=1e-11)
        return xm, ym, Xc, yc, res.betas[0]

    out, counts, _ = _run_ranks(2, work)
    assert counts[0] == counts[1]
    xm_ref = w @ X / w.sum()
    ym_ref = w @ y / w.sum()
    for r, (xm, ym, Xc, yc, beta) in enumerate(out):
        lo, hi = row_range(n, r, 2)
        np.testing.assert_allclose(xm, xm_ref, rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(ym, ym_ref, rtol=1e-12)
        np.testing.assert_allclose(Xc, X[lo:hi] - xm_ref, rtol=0, atol=1e-12)
        np.testing.assert_allclose(yc, y[lo:hi] - ym_ref, rtol=0, atol=1e-12)
    with _engine.get_engine(0).dataset(X, y, row_weight=w) as ds:
        ds.center()
        ref = ds.solve_path([(alpha, 0.0, 0.0)], tol=1e-11).betas[0]
    assert np.max(np.abs(out[0][4] - ref)) <= 1e-8 * np.max(np.abs(ref))
    assert np.array_equal(out[0][4], out[1][4])


@pytest.mark.parametrize("flags", [0, _engine.FLAG_WORKING_SET])
def test_ranks_whose_states_differ_abort_together_instead_of_hanging(flags):
    """Rank 1 is given a looser tolerance: its own state says "point finished" passes before rank 0's does --
    the situation a not-bit-identical all-reduce would create.  Every pass the ranks all-reduce, next to the
    gradients, a stop vector with a digest of their control blocks: the first pass after which the digests
    differ ends the solve on BOTH ranks with an error, after the same number of collectives -- no hang, no
    silently wrong coefficients."""
    X, y = _problem(2000, 150, seed=11)
    n = len(y)
    pts = [(a, 0.0, 0.0) for a in _alphas(X, y, k=6)]
    errors = [None, None]

    def work(r, eng):
        lo, hi = row_range(n, r, 2)
        with eng.dataset(X[lo:hi], y[lo:hi]) as ds:
            ds.set_global_rows(n)
            try:
                ds.solve_path(pts, tol=1e-11 if r == 0 else 1e-5, flags=flags)
            except _engine.EngineError as exc:
                errors[r] = str(exc)

    _, counts, _ = _run_ranks(2, work, timeout_s=20.0)
    assert counts[0] == counts[1] and counts[0] > 0
    if flags == 0:
        assert all(e is not None and "states differ" in e for e in errors), errors
    else:
        # (refined points are exact on the model: both tolerances may well be met in the same pass every time,
        #  and then nothing differs; what must never happen is one rank failing alone)
        assert (errors[0] is None) == (errors[1] is None), errors


def test_a_missing_rank_fails_the_collective_instead_of_hanging():
    X, y = _problem(500, 40, seed=5)

    def work(r, eng):
        if r == 1:
            return None  # never enters a collective
        with eng.dataset(X[:250], y[:250]) as ds:
            ds.set_global_rows(500)
            return ds.gradient(None)

    with pytest.raises(_engine.EngineError, match="did not arrive"):
        _run_ranks(2, work, timeout_s=2.0)


def test_single_local_rank_is_the_plain_engine():
    X, y = _problem(800, 50, seed=9)
    pts = [(a, 0.0, 0.0) for a in _alphas(X, y, k=4)]

    # (a sharded solve never switches the working set on mid-way -- that decision would hang on one rank's
    #  state -- so the comparison is between two plain iterations)
    plain = _engine.FLAG_NO_WORKING_SET

    def work(r, eng):
        with eng.dataset(X, y) as ds:
            return ds.solve_path(pts, tol=1e-10, flags=plain)

    out, counts, infos = _run_ranks(1, work)
    assert infos == [(0, 1)] and counts[0] > 0
    with _engine.get_engine(0).dataset(X, y) as ds:
        ref = ds.solve_path(pts, tol=1e-10, flags=plain)
    assert np.array_equal(out[0].betas, ref.betas)
    assert np.array_equal(out[0].n_iter, ref.n_iter)

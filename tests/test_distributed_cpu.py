"""N > 1 host logic on CPU: world_size-2 gloo process groups (no GPU).

Grid mode: units dealt to ranks, each rank solves its share (the oracle stands in for the engine
here -- tests only), results gathered and compared with the single-process answer.
Row-sharded mode: the id broadcast + communicator wiring against a recording stand-in engine, and the
identity the in-engine all-reduce relies on (sum of per-shard X_r^T r equals the full gradient).
"""

import os
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _init(rank, world_size, port):
    for path in (ROOT, os.path.join(ROOT, "sparse-lm_amd"), os.path.join(ROOT, "tests")):
        if path not in sys.path:
            sys.path.insert(0, path)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world_size), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world_size)


def _problem():
    rng = np.random.default_rng(3)
    X = rng.standard_normal((120, 15))
    beta = np.zeros(15)
    beta[:4] = [3.0, -2.0, 1.0, 0.5]
    y = X @ beta + 0.1 * rng.standard_normal(120)
    return X, y


def _grid_worker(rank, world_size, port, out_dir):
    _init(rank, world_size, port)
    import oracle
    from sparselm_amd import distributed as D

    X, y = _problem()
    n, p = X.shape
    gidx, G = oracle.group_index(None, p)
    folds = np.arange(n) % 3
    units = [(f, a) for f in range(3) for a in (0.5, 0.1, 0.02)]

    def solve(unit):
        f, a = unit
        tr = folds != f
        beta, _ = oracle.fista(X[tr], y[tr], a, 0.0, 0.0, gidx, G)
        return beta, float(np.mean((X[~tr] @ beta - y[~tr]) ** 2))

    local = D.run_units(units, solve, costs=[1.0 / a for _, a in units])
    assert sorted(local) == D.shard_units(len(units), rank, world_size, [1.0 / a for _, a in units])
    allres = D.gather_results(local, len(units))
    if rank == 0:
        np.savez(os.path.join(out_dir, "grid.npz"), betas=np.array([r[0] for r in allres]),
                 mse=np.array([r[1] for r in allres]))
    dist.barrier()
    dist.destroy_process_group()


class _RecordingEngine:
    def __init__(self):
        self.calls = []

    def comm_unique_id(self):
        return bytes(range(128))

    def comm_init(self, rank, world_size, uid):
        self.calls.append((rank, world_size, uid))


def _shard_worker(rank, world_size, port, out_dir):
    _init(rank, world_size, port)
    import torch

    from sparselm_amd import distributed as D

    eng = _RecordingEngine()
    D.init_row_sharding(eng)
    assert eng.calls == [(rank, world_size, bytes(range(128)))]
    # the exchange step of the row-sharded iteration: sum_r X_r^T (X_r z - y_r) == X^T (X z - y)
    X, y = _problem()
    lo, hi = D.row_range(len(y), rank, world_size)
    z = np.linspace(-1, 1, X.shape[1])
    part = torch.from_numpy(X[lo:hi].T @ (X[lo:hi] @ z - y[lo:hi]))
    dist.all_reduce(part)
    np.testing.assert_allclose(part.numpy(), X.T @ (X @ z - y), rtol=1e-12)
    # the lane cap the ranks plan a search with: the smallest any rank serves (one rank without the memory for the
    # column-major copy serves sixteen where its peers serve thirty-two)
    assert D.min_over_ranks(32 if rank == 0 else 16) == 16
    assert D.min_over_ranks(7) == 7
    dist.barrier()
    dist.destroy_process_group()


def _spawn(fn, tmp_path, port):
    mp.spawn(fn, args=(2, port, str(tmp_path)), nprocs=2, join=True)


def test_grid_mode_two_ranks_matches_single_process(tmp_path):
    _spawn(_grid_worker, tmp_path, 29611)
    got = np.load(tmp_path / "grid.npz")
    import oracle

    X, y = _problem()
    gidx, G = oracle.group_index(None, X.shape[1])
    folds = np.arange(len(y)) % 3
    k = 0
    for f in range(3):
        for a in (0.5, 0.1, 0.02):
            tr = folds != f
            beta, _ = oracle.fista(X[tr], y[tr], a, 0.0, 0.0, gidx, G)
            np.testing.assert_allclose(got["betas"][k], beta, rtol=0, atol=1e-12)
            np.testing.assert_allclose(got["mse"][k], np.mean((X[~tr] @ beta - y[~tr]) ** 2), rtol=1e-12)
            k += 1


def test_row_sharding_wiring_two_ranks(tmp_path):
    _spawn(_shard_worker, tmp_path, 29612)


def test_shard_units_partitions_and_balances():
    from sparselm_amd.distributed import row_range, shard_units

    for world in (1, 2, 3, 8):
        owned = [shard_units(50, r, world) for r in range(world)]
        assert sorted(i for o in owned for i in o) == list(range(50))
        assert max(map(len, owned)) - min(map(len, owned)) <= 1
    costs = list(np.geomspace(1, 30, 50))
    loads = [sum(costs[i] for i in shard_units(50, r, 8, costs)) for r in range(8)]
    assert max(loads) / (sum(costs) / 8) < 1.08  # LPT greedy: within 8 % of perfect balance
    with pytest.raises(ValueError):
        shard_units(5, 3, 2)
    spans = [row_range(1_000_003, r, 8) for r in range(8)]
    assert spans[0][0] == 0 and spans[-1][1] == 1_000_003
    assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
    assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def _check_plan(unit_points, keys, world, lanes, fine=True):
    from sparselm_amd.distributed import plan_lane_calls

    plan = plan_lane_calls(unit_points, keys, world, lanes, fine=fine)
    assert len(plan) == world
    seen = {}
    depth = []
    for r, calls in enumerate(plan):
        for call in calls:
            assert 1 <= len(call) <= lanes
            for lane in call:
                assert len({keys[u] for u, _ in lane}) == 1  # pieces that share a lane share the key (row mask)
                for u, idx in lane:
                    assert len(idx) > 0
                    for i in idx:
                        assert (u, i) not in seen, "a path point was dealt twice"
                        seen[(u, i)] = r
            depth.append((r, max(sum(len(idx) for _, idx in lane) for lane in call)))
    assert sorted(seen) == [(u, i) for u in range(len(unit_points)) for i in range(unit_points[u])]
    assert plan == plan_lane_calls(unit_points, keys, world, lanes, fine=fine)  # deterministic: every rank computes the same
    return plan, depth


def test_lane_call_plan_covers_every_path_point_once():
    # BASELINE config 4: 5 folds x 10 l1_ratio, 50 alphas each, sixteen lanes per call
    keys = [f for f in range(5) for _ in range(10)]
    passes = {}
    for world in (1, 2, 3, 4, 8):
        plan, depth = _check_plan([50] * 50, keys, world, 16)
        per_rank = [sum(1 + d for r, d in depth if r == rank) for rank in range(world)]
        passes[world] = max(per_rank)
        assert max(per_rank) - min(per_rank) <= 1
    # a call costs 1 + (points of its longest lane) passes: 8 ranks get within 5 % of an eighth of one rank's passes
    assert passes[1] <= 163 and passes[8] <= 21 and passes[1] / passes[8] >= 7.6
    assert passes[2] <= 86 and passes[4] <= 43
    # on 8 ranks every lane holds 20 points: two pieces of 20 per path, and the 10 left over pair up inside their fold
    plan, _ = _check_plan([50] * 50, keys, 8, 16)
    sizes = sorted(sum(len(idx) for _, idx in lane) for calls in plan for call in calls for lane in call)
    assert sizes == [20] * 125
    joint = [lane for calls in plan for call in calls for lane in call if len(lane) > 1]
    assert len(joint) == 25
    for lane in joint:  # walked down one path and up the next: consecutive points stay neighbours
        (u0, i0), (u1, i1) = lane
        assert i0 == sorted(i0) and i1 == sorted(i1, reverse=True) and u0 != u1
    # every piece is spread over the whole path: it starts near the top and ends near the bottom
    for calls in plan:
        for call in calls:
            for lane in call:
                for _, idx in lane:
                    assert min(idx) <= 4 and max(idx) >= 45


def test_lane_call_plan_odd_shapes():
    rng = np.random.default_rng(0)
    for _ in range(40):
        n_units = int(rng.integers(1, 40))
        pts = [int(k) for k in rng.integers(1, 60, n_units)]
        keys = [int(k) for k in rng.integers(0, 3, n_units)]
        world, lanes = int(rng.integers(1, 9)), int(rng.choice([1, 4, 6, 16]))
        _check_plan(pts, keys, world, lanes)
        plan, _ = _check_plan(pts, keys, world, lanes, fine=False)
        assert all(len(lane) == 1 and lane[0][1] == list(range(pts[lane[0][0]]))
                   for calls in plan for call in calls for lane in call)  # fine=False: whole paths only
    with pytest.raises(ValueError):
        _check_plan([3, 4], [0], 2, 4)


def test_lane_points_brings_per_piece_secant_factors():
    from sparselm_amd._engine import lane_points, path_extrapolation

    al = np.geomspace(10.0, 0.01, 50)
    up = np.c_[0.3 * al, 0.7 * al, 0 * al]
    a, b = up[[0, 3, 5, 8, 10]], up[[44, 39, 34, 29]]
    pts, gam = lane_points([a, b])
    assert pts.shape == (9, 3) and gam.shape == (9,)
    np.testing.assert_array_equal(pts, np.vstack([a, b]))
    np.testing.assert_array_equal(gam[:5], path_extrapolation(a))
    np.testing.assert_array_equal(gam[5:], path_extrapolation(b))
    assert gam[0] == gam[1] == gam[5] == gam[6] == 0.0 and np.all(gam[[2, 3, 4, 7, 8]] > 0.0)
    # exact for a piecewise-linear path: s_k = s_{k-1} + gamma (s_{k-1} - s_{k-2}) along the piece's own points
    s = b[:, 0] / 0.3
    np.testing.assert_allclose(s[2:], s[1:-1] + gam[7:] * (s[1:-1] - s[:-2]), rtol=1e-12)

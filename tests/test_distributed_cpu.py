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

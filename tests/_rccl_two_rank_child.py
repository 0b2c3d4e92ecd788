"""Child of tests/test_rccl_ranks_gpu.py, one process per GPU under `python -m torch.distributed.run`: a row-sharded Lasso
path and a grouped working-set solve over a REAL RCCL communicator (slm_comm_init), then the folds' Grams of a replicated
dataset summed over the ranks.  Every rank writes what it got to <outdir>/rank<r>.npz; the parent compares."""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sparse-lm_amd")]

import torch.distributed as dist  # noqa: E402

from sparselm_amd import _engine  # noqa: E402
from sparselm_amd import distributed as D  # noqa: E402


def main(outdir):
    rank, world, local = D.world()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend="gloo")
    eng = _engine.Engine(local % max(1, _engine.device_count()))
    D.init_row_sharding(eng, rank, world)
    out = {"rccl_ranks": eng.comm_ranks(), "rccl_rank": eng.comm_info()[0]}
    n, p = 4096, 320
    X = np.random.default_rng(0).standard_normal((n, p)) + 0.2
    beta = np.zeros(p)
    beta[::17] = 3.0
    y = X @ beta + 0.5 * np.random.default_rng(1).standard_normal(n)
    lo, hi = D.row_range(n, rank, world)
    amax = float(np.max(np.abs(X.T @ y)) / n)
    pts = [(a, 0.0, 0.0) for a in np.geomspace(amax, 0.02 * amax, 8)]
    with eng.dataset(X[lo:hi], y[lo:hi]) as ds:
        ds.set_global_rows(n)
        res = ds.solve_path(pts, tol=1e-10, lanes=2)
        out["lasso"] = res.betas
        out["lasso_ok"] = res.converged
        groups = np.arange(p) // 8
        ds.set_groups(groups, p // 8)
        gpts = [(0.0, 4.0 * a, 0.0) for a in np.geomspace(amax, 0.05 * amax, 6)]
        res = ds.solve_path(gpts, tol=1e-10, flags=_engine.FLAG_WORKING_SET, lanes=2)
        out["group"] = res.betas
        out["group_ok"] = res.converged and res.ws_refined > 0
    out["collectives_row_sharded"] = eng.comm_collectives()
    # grid mode: a replica per rank, the folds' Grams summed from the ranks' row blocks
    folds = np.random.default_rng(2).permutation(n) % 4
    masks = [(folds != f).astype(float) for f in range(4)]
    with eng.dataset(X, y) as ds:
        ds.set_replicated(True)
        ds.covariance_folds(masks, [int(m.sum()) for m in masks])
        G, c, sc = ds.covariance_download(2)
        out["gram"], out["gram_c"], out["gram_yy"] = G, c, sc["yy"]
    out["collectives"] = eng.comm_collectives()
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), **out)
    dist.barrier()
    eng.comm_destroy()
    eng.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])

"""Pieces of an 8-rank config-4 search with the folds' Grams built together, for `rocprofv3 --kernel-trace`:
  build [rank]  -- the parts of rank `rank`'s eighth of the rows (covariance_folds_begin), three times
  solve [rank]  -- rank `rank`'s share solved from the (complete) Grams, three times
Host-side times go to stderr.  tools/kernel_table.py condenses the trace."""
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "sparse-lm_amd")]
import bench  # noqa: E402
from sparselm_amd import _engine  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "build"
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 0
world, n, p = 8, 100_000, 5_000
if what == "build":
    engines = [_engine.Engine(0) for _ in range(world)]
    c4 = bench.Config4(engines[rank], n, p)
    _engine.init_local_comm(engines, timeout_s=5.0)
    c4.ds.set_replicated(True)
    n_effs = [int(m.sum()) for m in c4.masks]
    for rep in range(4):
        c4.ds.covariance_clear()
        engines[rank].synchronize()
        t0 = time.perf_counter()
        assert c4.ds.covariance_folds_begin(c4.masks, n_effs)
        t1 = time.perf_counter()
        engines[rank].synchronize()
        print(f"begin: returned after {1e3 * (t1 - t0):.2f} ms, device done after {1e3 * (time.perf_counter() - t0):.2f} ms", file=sys.stderr)
    c4.ds.covariance_clear()
    c4.close()
else:
    c4 = bench.Config4(_engine.get_engine(0), n, p)
    c4.build_covariance()
    calls = c4.calls_of(world, rank)
    for rep in range(4):
        sec, passes = c4.run(calls)
        print(f"rank {rank} of {world} from the Grams: {passes} passes, {1e3 * sec:.2f} ms", file=sys.stderr)
    c4.close()

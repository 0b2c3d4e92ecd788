"""Multi-GPU plumbing: one process per GPU, launched by ``python -m torch.distributed.run``.

Two modes:

* **grid mode** -- independent units (warm-started alpha paths per (fold, l1_ratio, ...)) are dealt to
  ranks; X is replicated per GPU; there is NO data-path collective.  ``shard_units`` /
  ``gather_results`` below; torch.distributed (any backend, gloo is enough) only moves the small
  result objects.
* **row-sharded mode** -- very tall X split by rows; every FISTA iteration all-reduces the p-vector
  ``X_r^T r`` with RCCL inside the engine (``slm_comm_init``).  ``init_row_sharding`` wires the
  communicator: rank 0 creates the RCCL unique id and torch.distributed broadcasts its 128 bytes.

The reference has no distributed code at all (its only parallelism is joblib over (candidate, fold),
src/sparselm/model_selection.py:273,304-323); this is the MI355X-node replacement for that.
"""

from __future__ import annotations

import os
from typing import Any, Callable, Sequence


def world():
    """(rank, world_size, local_rank) from the torch.distributed.run environment (1 process => 0, 1, 0)."""
    return (
        int(os.environ.get("RANK", "0")),
        int(os.environ.get("WORLD_SIZE", "1")),
        int(os.environ.get("LOCAL_RANK", "0")),
    )


def active_world():
    """(rank, world_size) the grid search shards over: torch.distributed's, when a process group is
    INITIALISED; (0, 1) otherwise -- whatever RANK / WORLD_SIZE say.  A launcher (torchrun, SLURM) that sets
    those variables for a script which never calls ``init_process_group`` must not make each process solve
    1/world of the grid and keep the rest as NaN: without a process group there is nobody to gather from, so
    every process solves everything.  torch is only imported when the environment hints at more than one rank."""
    if int(os.environ.get("WORLD_SIZE", "1")) <= 1:
        return 0, 1
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_units(n_units: int, rank: int, world_size: int, costs: Sequence[float] | None = None) -> list[int]:
    """Indices of the units rank ``rank`` owns.

    Without ``costs``: round-robin (unit i -> rank i % world).  With ``costs``: longest-processing-
    time-first greedy assignment (deterministic, identical on every rank), which keeps the 50
    (fold x l1_ratio) paths of BASELINE config 4 within one unit of balance on 8 GPUs.
    """
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError(f"bad rank {rank} of {world_size}")
    if costs is None:
        return list(range(rank, n_units, world_size))
    if len(costs) != n_units:
        raise ValueError("costs must have one entry per unit")
    order = sorted(range(n_units), key=lambda i: (-float(costs[i]), i))
    load = [0.0] * world_size
    owner = [0] * n_units
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        owner[i] = r
        load[r] += float(costs[i])
    return [i for i in range(n_units) if owner[i] == rank]


def row_range(n_rows: int, rank: int, world_size: int) -> tuple[int, int]:
    """Contiguous balanced row block [lo, hi) of rank ``rank`` in row-sharded mode."""
    base, rem = divmod(n_rows, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def run_units(units: Sequence[Any], solve: Callable[[Any], Any], rank: int | None = None,
              world_size: int | None = None, costs: Sequence[float] | None = None) -> dict[int, Any]:
    """Solve this rank's share of ``units``; returns {unit index: result}."""
    r, w, _ = world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    return {i: solve(units[i]) for i in shard_units(len(units), rank, world_size, costs)}


def gather_results(local: dict[int, Any], n_units: int) -> list[Any] | None:
    """All ranks' ``{index: result}`` merged into a list on every rank (torch.distributed
    all_gather_object; a plain pass-through when not initialised)."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        merged = dict(local)
    else:
        parts: list[Any] = [None] * dist.get_world_size()
        dist.all_gather_object(parts, local)
        merged = {}
        for part in parts:
            merged.update(part)
    missing = [i for i in range(n_units) if i not in merged]
    if missing:
        raise RuntimeError(f"units {missing[:8]} were not solved by any rank")
    return [merged[i] for i in range(n_units)]


def init_row_sharding(engine, rank: int | None = None, world_size: int | None = None) -> None:
    """Create the engine's RCCL communicator: rank 0 makes the unique id, everybody receives it
    through torch.distributed, then all ranks enter ``slm_comm_init`` together."""
    import torch
    import torch.distributed as dist

    r, w, _ = world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    if world_size == 1:
        uid = engine.comm_unique_id()
    else:
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed must be initialised before init_row_sharding")
        buf = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            buf = torch.frombuffer(bytearray(engine.comm_unique_id()), dtype=torch.uint8).clone()
        dist.broadcast(buf, src=0)
        uid = bytes(buf.tolist())
    engine.comm_init(rank, world_size, uid)

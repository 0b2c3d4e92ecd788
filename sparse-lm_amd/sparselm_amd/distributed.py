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


_grid_engines: dict = {}


def min_over_ranks(value: int) -> int:
    """The smallest `value` among the ranks of the process group (the value itself without one): what every rank must use
    where a per-rank answer -- the lanes a dataset's kernels serve, which depends on the device memory of THAT rank (the
    column-major copy behind more than sixteen lanes) -- decides how work is dealt among the ranks."""
    try:
        import torch
        import torch.distributed as dist
    except ImportError:
        return int(value)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return int(value)
    t = torch.tensor([int(value)], dtype=torch.int64)
    if dist.get_backend() == "nccl" and torch.cuda.is_available():
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t.item())


def grid_engine(rank: int, world_size: int):
    """The engine grid-mode searches of this process use among ``world_size`` ranks: a further engine (stream) on the
    rank's device with an RCCL communicator over all ranks, made once per process.  Datasets opened on it are marked as
    REPLICAS (``Dataset.set_replicated``): every rank holds all rows and solves its own lanes with no per-pass
    collective; the communicator carries one thing, the folds' Grams summed from the ranks' row blocks
    (``Dataset.covariance_folds``).  The per-process default engine stays without a communicator, so ordinary fits are
    untouched.  Returns ``None`` when the communicator cannot be formed on every rank (several ranks on one GPU: RCCL
    refuses) -- the ranks agree on that through the process group -- and the search then runs without sharded Grams."""
    import torch
    import torch.distributed as dist

    from . import _engine

    key = (os.getpid(), int(world_size))
    if key in _grid_engines:
        return _grid_engines[key]
    eng = _engine.Engine(_engine.get_engine().device_id)
    ok = 1
    try:
        init_row_sharding(eng, rank, world_size)
        ok = int(eng.comm_ranks() == world_size)
    except Exception:  # noqa: BLE001 -- whatever went wrong, the ranks settle it together below
        ok = 0
    flag = torch.tensor([ok], dtype=torch.int32)
    if dist.get_backend() == "nccl" and torch.cuda.is_available():
        flag = flag.cuda()
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) != 1:
        try:
            eng.close()
        except Exception:  # noqa: BLE001
            pass
        eng = None
    _grid_engines[key] = eng
    return eng


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
        on_gpu = dist.get_backend() == "nccl" and torch.cuda.is_available()  # (that backend moves device tensors only)
        if on_gpu:
            buf = buf.cuda()
        dist.broadcast(buf, src=0)
        uid = bytes(buf.cpu().tolist())
    engine.comm_init(rank, world_size, uid)


# ---------------------------------------------------------------------------------------------------
# grid mode, fine-grained: path POINTS, not whole paths, are what gets dealt
# ---------------------------------------------------------------------------------------------------
def _spread(sizes: Sequence[int]) -> list[list[int]]:
    """Deal the points 0..K-1 (K = sum(sizes)) of one path to pieces of the given sizes so that every piece is
    spread evenly over the whole path (largest-deficit dealing): a piece of 20 and a piece of 10 of a 50-point
    path take two of every five and one of every five points.  Every piece then starts near the top of the path
    -- where the first working set of a solve suffices -- and all pieces move down the path together."""
    K = int(sum(sizes))
    got = [[] for _ in sizes]
    for i in range(K):
        # the piece that is furthest behind its share of the first i + 1 points (ties: the larger piece, then the first)
        j = max(range(len(sizes)), key=lambda q: (sizes[q] * (i + 1) / K - len(got[q]), sizes[q], -q)
                if len(got[q]) < sizes[q] else (-1e300, 0, 0))
        got[j].append(i)
    return got


def _chunks(sizes: Sequence[int]) -> list[list[int]]:
    """The same pieces as contiguous ranges of the path (the first piece its top): every lane moves down the path one
    point per pass, like a whole path would, but all pieces after the first start cold in the middle of the path."""
    out, at = [], 0
    for size in sizes:
        out.append(list(range(at, at + size)))
        at += size
    return out


def plan_lane_calls(unit_points: Sequence[int], unit_keys: Sequence[Any], world_size: int, lanes: int,
                    fine: bool = True, spread: bool = True, snake: bool = True) -> list[list[list[list[tuple[int, list[int]]]]]]:
    """The calls every rank makes to solve a grid of warm-started paths, ``lanes`` lanes per call.

    ``unit_points[u]`` is the number of points of unit u's path (a unit: one (fold, other-parameters) pair of a
    grid search; the reference hands every single (candidate, fold) fit to joblib, src/sparselm/model_selection.py:273,
    304-323), ``unit_keys[u]`` what two units must have in common for pieces of them to follow each other in ONE lane
    (same row mask, proportional penalty vectors).  Returns ``plan[rank]`` = list of calls, a call = list of lanes, a
    lane = list of segments ``(u, point indices in the order the lane walks them)``.

    A pass over X advances every lane of a call by one point, so a call costs 1 + (points of its longest lane)
    passes and the cheapest plan fills ``world_size * lanes`` lanes evenly.  Whole rounds of ``world_size * lanes``
    units go out as whole paths, contiguous runs of the given order per call (callers order units fold-major: the
    lanes of a call then share row masks and their Grams).  What is left -- everything, when there are fewer units
    than lane slots, as for the 50 units of BASELINE config 4 on 8 x 16 slots -- is cut: with d = the fewest points per
    lane that fits, unit u becomes floor(K_u / d) pieces of d points and one shorter piece, each spread evenly over
    the path (`_spread`); the short pieces of units with the same key share lanes (first fit, decreasing), walked in
    alternating direction -- down one path, up the next -- so that consecutive points stay neighbours.  ``fine=False``
    (few lanes: the fused kernels without the working set, where a lane's cold start costs tens of passes) leaves
    units whole; ``spread=False`` cuts paths into contiguous ranges instead (`_chunks`), ``snake=False`` walks every
    piece of a shared lane from the top of its path (both: measurements).  Deterministic: every rank computes the same plan."""
    n_units = len(unit_points)
    if len(unit_keys) != n_units:
        raise ValueError("one key per unit")
    if world_size < 1 or lanes < 1:
        raise ValueError("world_size and lanes must be positive")
    slots = world_size * lanes
    plan: list[list[list]] = [[] for _ in range(world_size)]
    whole = (n_units // slots) * slots if fine else n_units
    for i0 in range(0, whole, lanes):  # call after call, rank after rank
        call = [[(u, list(range(unit_points[u])))] for u in range(i0, min(i0 + lanes, whole))]
        plan[(i0 // lanes) % world_size].append(call)
    rest = list(range(whole, n_units))
    if not rest:
        return plan
    total = sum(unit_points[u] for u in rest)
    kmax = max(unit_points[u] for u in rest)
    d = max(1, -(-total // slots))
    while True:
        long_lanes, short = [], {}
        for u in rest:
            K = unit_points[u]
            sizes = [d] * (K // d) + ([K % d] if K % d else [])
            for size, idx in zip(sizes, _spread(sizes) if spread else _chunks(sizes)):
                if size == d:
                    long_lanes.append((unit_keys[u], [(u, idx)]))
                else:
                    short.setdefault(unit_keys[u], []).append((size, u, idx))
        packed = []
        for key, pieces in short.items():  # first fit, decreasing, inside a key
            bins: list[tuple[int, list]] = []
            for size, u, idx in sorted(pieces, key=lambda t: (-t[0], t[1])):
                for b, (fill, segs) in enumerate(bins):
                    if fill + size <= d:
                        bins[b] = (fill + size, segs + [(u, idx)])
                        break
                else:
                    bins.append((size, [(u, idx)]))
            for _, segs in bins:
                segs = [(u, idx if (k % 2 == 0 or not snake) else idx[::-1]) for k, (u, idx) in enumerate(segs)]
                packed.append((key, segs))
        if len(long_lanes) + len(packed) <= slots or d >= kmax:
            break
        d += 1
    # lanes of one key next to each other (a call then holds few distinct row masks); contiguous, even chunks per rank
    all_lanes = long_lanes + packed
    order = {}
    for key, _ in all_lanes:
        order.setdefault(key, len(order))
    all_lanes.sort(key=lambda t: order[t[0]])  # (stable)
    lane_lists = [segs for _, segs in all_lanes]
    n_calls = -(-len(lane_lists) // slots)  # (more than one only when d == kmax still does not fit: cannot happen for rest < slots)
    per_round = -(-len(lane_lists) // n_calls)
    for c in range(n_calls):
        chunk = lane_lists[c * per_round : (c + 1) * per_round]
        base, rem = divmod(len(chunk), world_size)
        at = 0
        for r in range(world_size):
            take = base + (1 if r < rem else 0)
            if take:
                plan[r].append(chunk[at : at + take])
            at += take
    return plan

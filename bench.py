#!/usr/bin/env python3
"""Headline benchmark: fits/s over a 50-alpha warm-started Lasso path at n=100k, p=5k (fp64).

A "step" is one complete path solve on one GPU: seed Lipschitz estimate (power iteration, re-done
every step) + 50 converged alpha points (tol 1e-8), with (X, y) already resident in HBM.  The path
is walked by lanes that share every pass over X -- as many as the engine chooses (`--lanes 0`, the default: eighteen
for fifty points, sixteen on the matrix cores and two on the vector units beside them, three passes instead of
four), or `--lanes N`; `--lanes 1` is the strictly sequential warm-started path.  Set-up, before the W warm-up steps: the dataset is generated on the device, fifty
gradient launches bring the clocks up, and three untimed paths pay the dataset's one-off costs (column-major copy
of X, work-space allocations, first block of the page-locked result pool; reported under
`config.one_off_costs_outside_value_ms`).

Ranks.  ``python bench.py --gpus N`` with N > 1 and no WORLD_SIZE in the environment starts the N ranks
itself: the parent process -- before it has touched HIP in any way -- runs
``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`` on this
file as a CHILD process, relays its output and exits with its code.  Launched under torchrun (the
driver's form) WORLD_SIZE is taken from the environment.  At N > 1 every rank owns an independent unit of
the (alpha x CV-fold) grid -- its own synthetic fold, same law, different seed -- so the headline has no
data-path collective ("weak" scaling); torch.distributed (gloo) carries the barrier, the max-over-ranks
of the timings and the per-rank device report.

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline     -- achieved HBM GB/s of the kernel that streams X, from HIP events recorded on the
                  engine's own stream inside the timed region, against the 8 TB/s peak;
  cpu_baseline -- the oracle's C twin (OpenMP, as many threads as the cgroup grants CPUs) timed on a bounded prefix of the same
                  path on rank 0 at N = 1 (a reported baseline, not the target);
  ranks        -- rank -> device of every rank and the imbalance (slowest rank / mean rank);
  extra_legs   -- measured AFTER the timed region, never part of `value`:
                  "config4_grid": BASELINE config 4 (SparseGroupLasso, 5 folds x 10 l1_ratio x 50 alpha =
                  2500 fits) with its 50 (fold, l1_ratio) units dealt to the ranks (strong scaling);
                  "rowshard": BASELINE config 5's shape (AdaptiveGroupLasso, 3 re-weighting solves,
                  125 000 rows x 10 000 columns PER RANK) through the engine's RCCL communicator over all
                  ranks (weak scaling in rows; one all-reduce of the gradients per pass);
                  "concurrent_paths": the headline path on three engines of one GPU at once (per rank).
                  `--no-extra` skips them.
"""

from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time



def _host_cpu_share() -> int:
    """CPUs this process may actually use: its affinity mask cut down to the cgroup's quota.  (A one-GPU box of the pool
    shows all 256 hardware threads of the host but grants 16 CPUs' worth of time: 256 OpenMP threads on such a share are
    throttled into a third of what 16 deliver, and noisily so.)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]  # cgroup v2
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        try:  # cgroup v1
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0 and period > 0:
                n = min(n, max(1, (quota + period // 2) // period))
        except (OSError, ValueError):
            pass
    return n


HOST_CPUS = _host_cpu_share()
# (before numpy / the OpenMP runtime of the oracle's C twin are loaded: they size their pools once)
os.environ.setdefault("OMP_NUM_THREADS", str(HOST_CPUS))

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "sparse-lm_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
PRIMING_PATHS = 3  # untimed solves of the path right after the dataset is made (set-up, see main)


def measured_traffic(n, p, lanes, kernel):
    """(HBM bytes per gradient launch, note) from the committed PMC passes (profiles/roofline_traffic.json, written by
    tools/update_roofline_traffic.py from `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`).  The figure is a committed
    measurement, not one of this run: it is quoted only for the (n, p, lanes) and the kernel it was taken on, and only while
    the kernel's source file is the one the counters were taken on (its SHA-256 is recorded with them) -- otherwise None
    and the reason."""
    import hashlib

    try:
        with open(os.path.join(ROOT, "profiles", "roofline_traffic.json")) as f:
            t = json.load(f)
        if t["workload"] != {"n": n, "p": p, "lanes": lanes} or kernel not in t["kernel"]:
            return None, "no counter run for this workload and kernel"
        on = t.get("taken_on") or {}
        src = os.path.join(ROOT, on.get("kernel_source", ""))
        if not on.get("kernel_source_sha256") or not os.path.isfile(src):
            return None, "the counter run does not record the kernel source it was taken on"
        if hashlib.sha256(open(src, "rb").read()).hexdigest() != on["kernel_source_sha256"]:
            return None, f"{on['kernel_source']} has changed since the counters were taken (commit {on.get('commit', '?')[:12]}): re-collect"
        return t["hbm_bytes_per_launch"], f"PMC passes of commit {on.get('commit', '?')[:12]}, {t.get('source')}"
    except (OSError, KeyError, ValueError) as exc:
        return None, f"profiles/roofline_traffic.json unreadable: {exc!r}"


def make_coef(p, n_informative, seed):
    """Ground truth of sklearn.datasets.make_regression: n_informative coefficients 100*U(0,1) at
    random positions, the rest zero."""
    rng = np.random.default_rng(seed)
    coef = np.zeros(p)
    idx = rng.choice(p, size=n_informative, replace=False)
    coef[idx] = 100.0 * rng.uniform(size=n_informative)
    return coef


# ---------------------------------------------------------------------------------------------------
# CPU legs (rank 0, N = 1 only; after the timed region)
# ---------------------------------------------------------------------------------------------------
def cpu_baseline(ds, alphas, L, tol, gpu_betas, budget_s):
    """Time the C oracle on the same path (full n x p) on the host cores: the whole path when it fits the
    budget, otherwise a prefix (the sample string says which)."""
    import oracle
    from oracle import cref

    X0, y = ds.download()
    n, p = X0.shape
    gidx, G = oracle.group_index(None, p)
    z = np.zeros(p)
    numa = cref.NumaMatrix(X0)  # pages first-touched by the threads that stream them
    del X0
    X = numa.array
    t_grad = float("inf")
    for _ in range(3):
        t0 = time.perf_counter()
        cref.gradient(X, y, z)
        t_grad = min(t_grad, time.perf_counter() - t0)
    beta = None
    done = 0
    iters = 0
    worst = 0.0
    t_start = time.perf_counter()
    for k, alpha in enumerate(alphas):
        if done >= 2 and time.perf_counter() - t_start > budget_s:
            break
        beta, it = cref.fista(X, y, alpha, 0.0, 0.0, gidx, G, beta0=beta, L=L, tol=tol, max_iter=10000)
        iters += abs(it)
        done += 1
        ref_max = np.max(np.abs(beta))
        if k > 0 and ref_max > 0:  # k = 0 is alpha_max: the solution is 0 up to rounding of alpha_max
            worst = max(worst, float(np.max(np.abs(gpu_betas[k] - beta)) / ref_max))
    elapsed = time.perf_counter() - t_start
    whole = done == len(alphas)
    out = {
        "value": done / elapsed,
        "unit": "fits/s",
        "cores": cref.num_threads(),
        "kind": "port",
        "sample": ("the whole" if whole else f"first {done} of the") + f" {len(alphas)}-alpha warm-started path, full "
        f"{n}x{p} fp64, tol {tol:g}, {iters} fused one-pass gradients "
        f"({t_grad * 1e3:.0f} ms each = {8e-9 * n * p / t_grad:.0f} GB/s) by oracle/fista_ref.c with OpenMP"
        + ("" if whole else "; the later (denser) points cost more gradients each, so the whole path would rate lower"),
        "beta_rel_inf_err_gpu_vs_oracle": worst,
    }
    # Second CPU line: scikit-learn's coordinate-descent lasso_path with a precomputed
    # Gram, the strongest stock CPU solver for this objective, on the WHOLE 50-alpha path.  It also
    # checks the GPU coefficients at full size against an independent implementation.  Skipped when
    # the host is so loaded that the oracle's gradient already crawls (keeps the default run short).
    if t_grad < 0.3:
        try:
            from sklearn.linear_model import lasso_path
            from threadpoolctl import threadpool_limits

            threads = min(64, HOST_CPUS)
            Xf = np.asfortranarray(X)
            with threadpool_limits(limits=threads):
                t0 = time.perf_counter()
                _, coefs, _ = lasso_path(Xf, y, alphas=alphas, precompute=True, tol=1e-10, max_iter=100000)
                dt = time.perf_counter() - t0
            ref = coefs.T
            out["sklearn_lasso_path"] = {
                "value": len(alphas) / dt,
                "unit": "fits/s",
                "cores": threads,
                "seconds_per_path": dt,
                "what": "sklearn.linear_model.lasso_path(precompute=True, tol=1e-10), all alphas, BLAS capped at "
                f"{threads} threads",
                "beta_rel_inf_err_gpu_vs_sklearn": float(np.max(np.abs(gpu_betas - ref)) / np.max(np.abs(ref))),
            }
            del Xf
        except Exception as exc:  # never let the extra line break the contract line
            out["sklearn_lasso_path"] = {"error": repr(exc)}
    out["cvxpy"] = cvxpy_leg(X, y, alphas, gpu_betas)
    numa.__exit__()
    return stock_first(out, len(alphas), n, p)


def stock_first(out, n_alphas, n, p):
    """The contract's `cpu_baseline.value` is the STRONGEST stock CPU solver for this objective (BASELINE.md section 3, B2:
    scikit-learn's coordinate descent with a precomputed Gram, the whole path); the oracle's C twin -- the same FISTA as the
    engine, the differential-test target (B1) -- stands beside it as `port`.  Without a scikit-learn figure the port is the value."""
    sk = out.get("sklearn_lasso_path") or {}
    if not sk.get("value"):
        return out
    port = {k: out[k] for k in ("value", "unit", "cores", "kind", "sample", "beta_rel_inf_err_gpu_vs_oracle") if k in out}
    return {"value": sk["value"], "unit": "fits/s", "cores": sk["cores"], "kind": "stock",
            "sample": f"the whole {n_alphas}-alpha path, full {n}x{p} fp64, {sk['seconds_per_path']:.1f} s: " + sk["what"],
            "beta_rel_inf_err_gpu_vs_sklearn": sk["beta_rel_inf_err_gpu_vs_sklearn"], "port": port, "cvxpy": out.get("cvxpy")}


def cvxpy_leg(X, y, alphas, gpu_betas, n_red=10_000, p_red=500, cap_s=60.0):
    """the reference's own route -- a cvxpy problem handed to a conic solver
    (reference src/sparselm/model/_base.py:512-519, objective _lasso.py:109-121) -- written here from the
    math, at the reduced size the plan names (n = 10 000, p = 500: the full 100 000 x 5 000 problem is a
    5e8-nonzero conic matrix) on the first rows / columns of the same data.  Recorded as unavailable when
    cvxpy cannot be imported on the box."""
    try:
        import cvxpy as cp
    except Exception as exc:
        return {"status": "cvxpy unavailable on box", "detail": repr(exc)[:120]}
    try:
        Xr = np.ascontiguousarray(X[:n_red, :p_red])
        yr = y[:n_red]
        alpha = float(alphas[len(alphas) // 2])
        beta = cp.Variable(p_red)
        prob = cp.Problem(cp.Minimize(cp.sum_squares(Xr @ beta - yr) / (2 * n_red) + alpha * cp.norm1(beta)))
        t0 = time.perf_counter()
        prob.solve()
        dt = time.perf_counter() - t0
        return {"status": "ok", "value": 1.0 / dt, "unit": "fits/s", "seconds_per_fit": dt, "n": n_red, "p": p_red,
                "alpha": alpha, "solver": str(prob.solver_stats.solver_name), "capped_at_s": cap_s,
                "what": "one cold cvxpy solve at the path's middle alpha, reduced size"}
    except Exception as exc:
        return {"status": "cvxpy failed", "detail": repr(exc)[:200]}


# ---------------------------------------------------------------------------------------------------
# extra legs (all ranks; after the timed region; never part of `value`)
# ---------------------------------------------------------------------------------------------------
class StdoutToStderr:
    """File descriptor 1 points at stderr for the duration: RCCL prints a version banner to stdout when a
    communicator is created, and stdout carries the ONE contract line."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        try:  # the banner sits in the C library's buffer (stdout is a pipe): push it out while fd 1 is still stderr
            import ctypes

            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        os.dup2(self._saved, 1)
        os.close(self._saved)


class Watchdog:
    """Hard stop for a leg that might hang in a collective: past the limit, rank 0 prints the line it
    already holds (with the leg marked as timed out) and every rank leaves through os._exit, so a stuck
    exchange never takes the contract line or the driver's run with it."""

    def __init__(self, seconds, on_expire):
        self._t = threading.Timer(seconds, on_expire)
        self._t.daemon = True

    def __enter__(self):
        self._t.start()
        return self

    def __exit__(self, *exc):
        self._t.cancel()


class Config4:
    """BASELINE config 4 as a device-resident problem: 5 folds x 10 l1_ratio, each a 50-alpha SparseGroupLasso path =
    2500 fits on make_regression-law data with 500 shuffled groups of 10 (25 informative).  `calls_of(world, rank)` is
    the share of one rank: the path POINTS of the 50 (fold, l1_ratio) units are dealt to the lane slots of all ranks by
    `distributed.plan_lane_calls` (thirty-two lanes per call -- over X and from the Grams alike; whole paths while there are more units than slots, evenly
    spread pieces of paths otherwise).  Every rank generates the SAME (X, y) (same seed): X replicated per GPU, no
    data-path collective."""

    K = 50

    def __init__(self, eng, n, p, noise_sd=10.0):
        from sparselm_amd import _engine

        self.G = G = p // 10
        rng = np.random.default_rng(1)
        self.groups = groups = rng.permutation(np.repeat(np.arange(G), 10))
        coef = np.zeros(p)
        for g in rng.choice(G, 25, replace=False):
            coef[groups == g] = 100.0 * rng.uniform(size=10)
        self.ds = ds = eng.synthetic_dataset(n, p, seed=11, coef=coef, noise_sd=noise_sd)
        self.copies = [ds]
        ds.set_groups(groups, G)
        g0, _ = ds.gradient(None)
        gnorm = np.sqrt(np.bincount(groups, weights=g0 * g0, minlength=G))
        bmax, amax1 = float(gnorm.max()), float(np.max(np.abs(g0)))
        folds = np.random.default_rng(0).permutation(n) % 5  # KFold(5, shuffle=True)
        self.masks = [(folds != f).astype(float) for f in range(5)]
        # fold-major: a call shares row masks and their Grams
        self.units = [(f, r) for f in range(5) for r in np.linspace(0.05, 0.95, 10)]
        self.unit_pts = []
        for f, r in self.units:
            amax = min(bmax / (1 - r), amax1 / r)
            al = np.geomspace(amax, 1e-3 * amax, self.K)
            self.unit_pts.append(np.c_[r * al, (1 - r) * al, 0 * al])
        self.flags = 0  # solve flags of every call (build_covariance() adds FLAG_COVARIANCE)

    @property
    def lanes(self):
        """lanes per call, as the engine serves them under the current flags: thirty-two over X (two halves on one read of X)
        and from the folds' Grams (the product of a Gram and the lanes' points: a launch per half)"""
        return getattr(self, "_lanes", None) or self.ds.max_lanes(self.flags)

    @lanes.setter
    def lanes(self, value):
        self._lanes = value

    def build_covariance(self):
        """One Gram per fold on the first engine's dataset (slm_dataset_covariance); seconds spent."""
        from sparselm_amd import _engine

        t0 = time.perf_counter()
        self.ds.covariance_folds(self.masks, [int(m.sum()) for m in self.masks])
        self.flags |= _engine.FLAG_COVARIANCE
        return time.perf_counter() - t0

    def calls_of(self, world, rank, **plan_options):
        from sparselm_amd import distributed as D

        return D.plan_lane_calls([self.K] * len(self.units), [f for f, _ in self.units], world, self.lanes, **plan_options)[rank]

    def run_call(self, d, call, keep=None):
        from sparselm_amd import _engine

        specs = []
        for lane in call:
            pts, gam = _engine.lane_points([self.unit_pts[u][idx] for u, idx in lane])
            f = self.units[lane[0][0]][0]
            specs.append(dict(points=pts, extrap=gam, row_weight=self.masks[f], n_eff=int(self.masks[f].sum())))
        out = d.solve_lanes(specs, flags=self.flags)
        if not all(o.converged for o in out):
            raise RuntimeError("config 4: a path did not converge")
        if keep is not None:  # (unit, point) -> coefficients, for the checks
            for lane, o in zip(call, out):
                at = 0
                for u, idx in lane:
                    for k, i in enumerate(idx):
                        keep[(u, i)] = o.betas[at + k].copy()
                    at += len(idx)
        return out[0].grad_launches

    def run(self, calls, n_streams=1):
        """(seconds, passes) of `calls`, one after the other on one engine or dealt to `n_streams` standing ones"""
        todo, lock, passes, errors = list(range(len(calls))), threading.Lock(), [0], []

        def work(i):
            try:
                while True:
                    with lock:
                        if not todo or errors:
                            return
                        b = todo.pop(0)
                    k = self.run_call(self.copies[i], calls[b])
                    with lock:
                        passes[0] += k
            except BaseException as exc:
                with lock:
                    errors.append(exc)

        threads = [threading.Thread(target=work, args=(i,)) for i in range(1, n_streams)]
        for c in self.copies[:n_streams]:
            c.engine.synchronize()
        t0 = time.perf_counter()
        for t in threads:
            t.start()
        work(0)
        for t in threads:
            t.join()
        for c in self.copies[:n_streams]:
            c.engine.synchronize()
        if errors:
            raise errors[0]
        return time.perf_counter() - t0, passes[0]

    def add_streams(self, n_streams):
        while len(self.copies) < n_streams:
            c = self.ds.clone()
            c.set_groups(self.groups, self.G)
            self.copies.append(c)

    def drop_streams(self):
        for c in self.copies[1:]:
            e = c.engine
            c.close()
            e.close()
        self.copies = self.copies[:1]

    def close(self):
        self.drop_streams()
        self.ds.close()


def leg_config4_grid(eng, rank, world, n, p, device_id=0, streams=3, emulate_world=8):
    """BASELINE config 4 on the grid mode (`Config4`).  Timed: the rank's calls one after the other on one engine; dealt
    to `streams` standing engines (HIP streams) of its GPU when it has more than one call to make; and, at world == 1,
    the share of every rank of an `emulate_world`-rank job, one after the other on this GPU (what each of 8 GPUs would
    be doing side by side)."""
    c4 = Config4(eng, n, p)
    try:
        def points(calls):
            return sum(len(idx) for call in calls for lane in call for _, idx in lane)

        mine = c4.calls_of(world, rank)
        c4.run(mine)  # warm (column-major copy, buffers)
        seconds, passes = min(c4.run(mine) for _ in range(2))
        out = {"seconds": seconds, "passes": passes, "calls": len(mine), "points": points(mine)}
        if world == 1 and emulate_world > 1:
            shares = []
            for r in range(emulate_world):
                calls = c4.calls_of(emulate_world, r)
                c4.run(calls)
                sec, pas = min(c4.run(calls) for _ in range(2))
                shares.append({"seconds": sec, "passes": pas, "lanes": [len(c) for c in calls], "points": points(calls),
                               "row_masks": [len({c4.units[u][0] for lane in c for u, _ in lane}) for c in calls]})
            out["emulated"] = {"world": emulate_world, "shares": shares}
            try:
                out["emulated"]["shared_grams"] = emulate_shared_grams(device_id, n, p, emulate_world)
            except Exception as exc:  # noqa: BLE001 -- the leg over X stands on its own
                out["emulated"]["shared_grams"] = {"error": repr(exc)[:300]}
        n_streams = min(streams, len(mine))
        if n_streams > 1:
            c4.add_streams(n_streams)
            c4.run(mine, n_streams)  # warm the copies
            out["seconds_streams"], _ = c4.run(mine, n_streams)
            out["streams"] = n_streams
        # the rank's share from the Grams of the five folds (SLM_FLAG_COVARIANCE), the Grams' cost beside it
        try:
            out["covariance_build_s"] = c4.build_covariance()
            mine = c4.calls_of(world, rank)  # (planned again: the flags have changed)
            c4.run(mine)
            out["seconds_covariance"], out["passes_covariance"] = min(c4.run(mine) for _ in range(2))
            if n_streams > 1:  # copies made now share the Grams (slm_dataset_clone)
                c4.drop_streams()
                c4.add_streams(n_streams)
                c4.run(mine, n_streams)
                out["seconds_covariance_streams"] = min(c4.run(mine, n_streams)[0] for _ in range(2))
        except NotImplementedError as exc:
            out["covariance_error"] = repr(exc)[:200]
        return out
    finally:
        c4.close()


def emulate_shared_grams(device_id, n, p, world):
    """The grid of config 4 as `world` ranks run it WITH the folds' Grams built together: every rank a
    replica of (X, y) on an engine of an in-process communicator of this GPU.  Timed per rank, one rank at a time (on a
    node every rank has its own GPU): (A) the parts of its `world`-th of the rows (`covariance_folds_begin`), (C) its share
    solved from the Grams; (B) the exchange and the folds' Grams (`covariance_folds_finish`) run on all ranks at once --
    the collective needs them all -- and are charged 1 / world of their wall time each (bandwidth-bound work of `world`
    ranks sharing one HBM).  An xGMI ring is modelled beside the in-process exchange, never instead of it."""
    from sparselm_amd import _engine

    engines = [_engine.Engine(device_id) for _ in range(world)]
    c4s = []
    try:
        for e in engines:  # (before the communicator exists: Config4 evaluates a gradient, which a row block would all-reduce)
            c4s.append(Config4(e, n, p))
        _engine.init_local_comm(engines, timeout_s=60.0)
        for c in c4s:
            c.ds.set_replicated(True)
            c.flags |= _engine.FLAG_COVARIANCE
        masks, n_effs = c4s[0].masks, [int(m.sum()) for m in c4s[0].masks]

        def all_ranks(fn):
            errors = []

            def run(r):
                try:
                    fn(r)
                except BaseException as exc:  # noqa: BLE001
                    errors.append(exc)

            threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
            for e in engines:
                e.synchronize()
            t0 = time.perf_counter()
            for t in threads:
                t.start()
            for t in threads:
                t.join()
            for e in engines:
                e.synchronize()
            if errors:
                raise errors[0]
            return time.perf_counter() - t0

        build, finish_wall = [float("inf")] * world, float("inf")
        for rep in range(3):  # (the first round pays the allocations -- a fresh process's one-off -- and is not timed; of the
            for c in c4s:     #  other two every rank's faster build counts: a driver allocation now and then stalls for tens of ms)
                c.ds.covariance_clear()
            for r, c in enumerate(c4s):
                engines[r].synchronize()
                t0 = time.perf_counter()
                if not c.ds.covariance_folds_begin(masks, n_effs):
                    raise RuntimeError("the folds of config 4 are a partition")
                engines[r].synchronize()
                if rep > 0:
                    build[r] = min(build[r], time.perf_counter() - t0)
            wall = all_ranks(lambda r: c4s[r].ds.covariance_folds_finish())
            if rep > 0:
                finish_wall = min(finish_wall, wall)
        shares = []
        for r, c in enumerate(c4s):
            calls = c.calls_of(world, r)
            c.run(calls)
            sec, pas = min(c.run(calls) for _ in range(3))
            shares.append({"build_s": build[r], "solve_s": sec, "passes": pas})
        ld = c4s[0].ds.ld
        # per fold: the packed lower triangle (rounded up to 16 doubles), X_f^T y_f and y_f . y_f (cov_folds_begin)
        exchange_bytes = 8.0 * len(masks) * ((ld * (ld + 1) // 2 + 15) // 16 * 16 + ld + 16)
        return {"world": world, "shares": shares, "finish_wall_s_all_ranks_on_one_gpu": finish_wall,
                "exchange_bytes_per_rank": exchange_bytes, "collectives_per_rank": engines[0].comm_collectives() // 3}
    finally:
        for c in c4s:
            c.close()
        for e in engines:
            e.close()


def leg_config4_dense(eng, n, p, noise_sd=100.0):
    """Config 4's grid with noise 100 instead of 10: every path ends at thousands of non-zeros, the working set gives up
    on a quarter of the points.  Over X (plain sixteen-lane passes there) and from the folds' Grams; the coefficient
    vectors of the two runs against each other."""
    c4 = Config4(eng, n, p, noise_sd=noise_sd)
    try:
        calls = c4.calls_of(1, 0)
        keep_x, keep_c = {}, {}
        for call in calls:
            c4.run_call(c4.ds, call, keep_x)
        seconds, passes = min(c4.run(calls) for _ in range(2))
        from sparselm_amd import _engine
        c4.flags |= _engine.FLAG_NO_MODEL_GRAM  # round 4's route: plain sixteen-lane passes beyond the working set
        seconds_plain, passes_plain = c4.run(calls)
        c4.flags &= ~_engine.FLAG_NO_MODEL_GRAM
        nnz = sorted(int(np.count_nonzero(keep_x[(u, c4.K - 1)])) for u in range(len(c4.units)))
        above = sum(int(np.count_nonzero(b)) > 512 for b in keep_x.values())
        build = c4.build_covariance()
        calls = c4.calls_of(1, 0)  # (planned again: the flags have changed)
        for call in calls:
            c4.run_call(c4.ds, call, keep_c)
        seconds_c, passes_c = min(c4.run(calls) for _ in range(2))
        worst = max(float(np.max(np.abs(keep_c[k] - keep_x[k])) / max(float(np.max(np.abs(keep_x[k]))), 1e-300)) for k in keep_x)
        return {"seconds": seconds, "passes": passes, "seconds_without_model_gram": seconds_plain, "passes_without_model_gram": passes_plain,
                "seconds_covariance": seconds_c, "passes_covariance": passes_c,
                "covariance_build_s": build, "nnz_last_min_median_max": [nnz[0], nnz[len(nnz) // 2], nnz[-1]],
                "points_above_512_nonzeros": above, "worst_rel_inf_diff": worst, "noise_sd": noise_sd}
    finally:
        c4.close()


def leg_config1_small():
    """BASELINE config 1 and the reference's own problem sizes through the estimators (what its users run: README.md:42-55,
    tests/conftest.py:17-19): the on-chip solvers' territory.  Times include everything a user's call includes."""
    import warnings

    from sklearn.datasets import make_regression
    from sparselm_amd.model import AdaptiveLasso, Lasso, SparseGroupLasso
    from sparselm_amd.model_selection import GridSearchCV

    out = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        X, y = make_regression(n_samples=100, n_features=80, n_informative=10, random_state=0)
        grid = {"alpha": np.logspace(-8, 2, 10)}
        GridSearchCV(AdaptiveLasso(fit_intercept=False), grid).fit(X, y)
        t = []
        for _ in range(5):
            t0 = time.perf_counter()
            gs = GridSearchCV(AdaptiveLasso(fit_intercept=False), grid).fit(X, y)
            t.append(time.perf_counter() - t0)
        out["readme_grid_ms"] = 1e3 * sorted(t)[len(t) // 2]
        out["readme_grid_best_alpha"] = float(gs.best_params_["alpha"])
        Xs, ys = make_regression(n_samples=25, n_features=30, n_informative=10, random_state=1)
        Lasso(alpha=0.1).fit(Xs, ys)
        t0 = time.perf_counter()
        for _ in range(50):
            m = Lasso(alpha=0.1).fit(Xs, ys)
        out["lasso_fit_25x30_ms"] = 1e3 * (time.perf_counter() - t0) / 50
        out["lasso_fit_converged"] = bool(m.solver_info_["converged"])
        groups = np.arange(80) // 8
        SparseGroupLasso(groups=groups, alpha=0.5, standardize=True, fit_intercept=True).fit(X, y)
        t0 = time.perf_counter()
        for _ in range(5):
            m = SparseGroupLasso(groups=groups, alpha=0.5, standardize=True, fit_intercept=True).fit(X, y)
        out["standardized_sgl_fit_100x80_ms"] = 1e3 * (time.perf_counter() - t0) / 5
        out["standardized_sgl_sweeps"] = int(m.solver_info_["n_iter"])
    return out


def soak_case(seed, p):
    """The law of tools/headline_soak.py: (coef, noise_sd, path floor as a fraction of alpha_max) of random dataset `seed`
    -- 5...199 informative features of scale 1 or 100, noise 0.1 / 10 / 100, path down to 1e-3 / 1e-2 / 0.1 alpha_max."""
    rng = np.random.default_rng(seed)
    k = int(rng.integers(5, 200))
    coef = np.zeros(p)
    coef[rng.choice(p, k, replace=False)] = rng.choice([1.0, 100.0]) * rng.standard_normal(k)
    noise = float(rng.choice([0.1, 10.0, 100.0]))
    lo = float(rng.choice([1e-3, 1e-2, 0.1]))
    return coef, noise, lo, k


def leg_soak(eng, n, p, K, tol, lanes, seeds=range(4, 16)):
    """The headline path on a DISTRIBUTION of datasets of the headline shape, not on the one the value is quoted on:
    twelve seeds of `soak_case` (five of them noise-fitting paths whose ends outgrow the 512-column working set and
    take their points from rounds on the model Gram; round 4: plain steps on the sixteen-lane split pass).  Every path is
    checked against the plain four-lane iteration of the same data (no working set, tol 1e-9)."""
    from sparselm_amd import _engine

    rows = []
    for seed in seeds:
        coef, noise, lo, k = soak_case(seed, p)
        with eng.synthetic_dataset(n, p, seed=100 + seed, coef=coef, noise_sd=noise) as ds:
            g0, _ = ds.gradient(None)
            amax = float(np.max(np.abs(g0)))
            pts = [(a, 0.0, 0.0) for a in np.geomspace(amax, lo * amax, K)]
            eng.synchronize()
            t0 = time.perf_counter()
            first = ds.solve_path(pts, tol=tol, lanes=lanes, flags=_engine.FLAG_FRESH_L)  # (one-off costs of a fresh dataset:
            first_ms = 1e3 * (time.perf_counter() - t0)                                   #  column-major copy, work space, model Gram)
            best = None
            for _ in range(2):
                eng.synchronize()
                t0 = time.perf_counter()
                r = ds.solve_path(pts, tol=tol, lanes=lanes, flags=_engine.FLAG_FRESH_L)
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            q = ds.solve_path(pts, lanes=4, flags=_engine.FLAG_NO_WORKING_SET, tol=1e-9)
            err = float(np.max(np.abs(r.betas - q.betas)) / max(float(np.max(np.abs(q.betas))), 1e-300))
            plain_ms = plain_passes = None
            if r.mg_rounds > 0:  # the same path with plain steps beyond the working set (round 4's route)
                t0 = time.perf_counter()
                w = ds.solve_path(pts, tol=tol, lanes=lanes, flags=_engine.FLAG_FRESH_L | _engine.FLAG_NO_MODEL_GRAM)
                plain_ms, plain_passes = 1e3 * (time.perf_counter() - t0), int(w.grad_launches)
            rows.append({"seed": int(seed), "informative": k, "noise": noise, "floor": lo, "ms": 1e3 * best,
                         "first_solve_ms": first_ms, "model_gram_build_ms": float(first.mg_build_ms), "model_gram_rounds": int(r.mg_rounds),
                         "model_gram_inner_iters": int(r.mg_inner_iters), "model_gram_rejected": int(r.mg_rejected),
                         "ms_without_model_gram": plain_ms, "passes_without_model_gram": plain_passes,
                         "fits_per_s": K / best, "passes": int(r.grad_launches), "plain_passes": int(q.grad_launches),
                         "nnz_last": int(np.count_nonzero(r.betas[-1])), "ws_columns": int(r.ws_columns),
                         "converged": bool(r.converged and q.converged), "rel_inf_err_vs_plain": err})
    return {"cases": rows}


def leg_headline_draws(eng, n, p, K, tol, lanes, seeds=(7, 1001, 1002, 1003, 1004, 1005, 1006, 1007)):
    """The headline path on OTHER draws of the headline's own law (the same fifty coefficients, other X and noise): the
    dataset `value` is quoted on is one draw, and how many passes a path takes depends on the draw -- a lane's point that
    meets a feature the working set did not hold is verified a pass later.  Per draw: passes and ms (mean of five calls behind two warm-up calls) on the engine's choice
    of lanes (or --lanes) and on sixteen."""
    from sparselm_amd import _engine

    coef = make_coef(p, 50, seed=0)
    rows = []
    for dseed in seeds:
        with eng.synthetic_dataset(n, p, seed=dseed, coef=coef, noise_sd=10.0) as ds:
            g0, _ = ds.gradient(None)
            amax = float(np.max(np.abs(g0)))
            pts = [(a, 0.0, 0.0) for a in np.geomspace(amax, 1e-3 * amax, K)]
            row = {"data_seed": int(dseed)}
            for tag, ln in (("", lanes), ("_16_lanes", 16)):
                # (steady state, like `value`, which has its warm-up steps: the first call on a fresh dataset makes the
                #  column-major copy, the second still allocates what the first did not reach -- 5-14 ms and +0.03-0.6 ms
                #  measured; from the third on the calls repeat to 0.02 ms)
                for _ in range(2):
                    ds.solve_path(pts, tol=tol, lanes=ln, flags=_engine.FLAG_FRESH_L)
                eng.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    r = ds.solve_path(pts, tol=tol, lanes=ln, flags=_engine.FLAG_FRESH_L)
                eng.synchronize()
                row["ms" + tag] = 1e3 * (time.perf_counter() - t0) / 5
                row["passes" + tag] = int(r.grad_launches)
                row["converged" + tag] = bool(r.converged)
            rows.append(row)
    return {"cases": rows}


def leg_literal_config2(eng, K, tol, lanes, twin=True):
    """BASELINE.md section 2, row C2, LITERALLY: sklearn.datasets.make_regression(100_000, 5_000, n_informative=50, noise=10.0,
    random_state=0), generated on the host by scikit-learn and uploaded -- generation and upload outside every timing.  The
    headline's `value` is quoted on the engine's on-device generator of the same law (SURVEY 8d allows that); this leg runs the
    dataset itself: passes and ms per 50-alpha path, and three of its points against the oracle's C twin (started at the
    engine's point and run on to 1e-10, the twin may not move it by more than 1e-6 rel-inf)."""
    from sklearn.datasets import make_regression

    from sparselm_amd import _engine

    t0 = time.perf_counter()
    X0, y = make_regression(n_samples=100_000, n_features=5_000, n_informative=50, noise=10.0, random_state=0)
    X0 = np.ascontiguousarray(X0)
    gen_s = time.perf_counter() - t0
    n, p = X0.shape
    t0 = time.perf_counter()
    with eng.dataset(X0, y) as ds:
        eng.synchronize()
        upload_s = time.perf_counter() - t0
        g0, _ = ds.gradient(None)
        amax = float(np.max(np.abs(g0)))
        alphas = np.geomspace(amax, 1e-3 * amax, K)
        pts = [(a, 0.0, 0.0) for a in alphas]
        for _ in range(3):
            res = ds.solve_path(pts, tol=tol, lanes=lanes, flags=_engine.FLAG_FRESH_L)
        eng.synchronize()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            res = ds.solve_path(pts, tol=tol, lanes=lanes, flags=_engine.FLAG_FRESH_L)
        eng.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / reps
        lanes_run = ds.path_lanes(K, _engine.FLAG_FRESH_L) if lanes == 0 else min(lanes, K)
    out = {"dataset": "sklearn.datasets.make_regression(n_samples=100000, n_features=5000, n_informative=50, noise=10.0, random_state=0)",
           "generated": "host (scikit-learn), uploaded; outside the timing", "generation_s": gen_s, "upload_s": upload_s,
           "ms_per_path": ms, "fits_per_s": K / (1e-3 * ms), "passes": int(res.grad_launches), "lanes": int(lanes_run),
           "converged": bool(res.converged), "nnz_last": int(np.count_nonzero(res.betas[-1])), "alpha_max": amax}
    if twin:
        from oracle import cref

        with cref.NumaMatrix(X0) as X:
            del X0
            v = np.random.default_rng(0).standard_normal(p)
            lam = 1.0
            for _ in range(8):
                v /= np.linalg.norm(v)
                gv, _ = cref.gradient(X, 0.0 * y, v)
                lam = float(np.linalg.norm(gv))
                v = gv
            single = np.arange(p, dtype=np.int32)
            errs = {}
            for k in (10, 30, K - 1):
                b, _ = cref.fista(X, y, alphas[k], 0.0, 0.0, single, p, beta0=res.betas[k], L=1.1 * lam, tol=1e-10, max_iter=200)
                errs[str(k)] = float(np.max(np.abs(b - res.betas[k])) / np.max(np.abs(b)))
        out["rel_inf_move_of_the_c_twin_from_the_engines_point"] = errs
        out["worst_rel_inf_vs_c_twin"] = max(errs.values())
    return out


def leg_wide_rows(eng, n=50_000, p=20_000, K=32, tol=1e-8):
    """Rows beyond 10 240 columns (round-5 verdict, item 4): a Lasso path at n = 50 000, p = 20 000 -- 8 GB of X -- on the split
    pass: sixteen lanes and the working set like any other width (until round 6: one lane on the two-pass kernels, two reads
    of X per gradient).  Per pass of the path: the X^T R kernel under HIP events against the algorithmic bytes of one
    gradient of sixteen lane slots."""
    from sparselm_amd import _engine

    coef = make_coef(p, 50, seed=0)
    with eng.synthetic_dataset(n, p, seed=77, coef=coef, noise_sd=10.0) as ds:
        g0, _ = ds.gradient(None)
        amax = float(np.max(np.abs(g0)))
        pts = [(a, 0.0, 0.0) for a in np.geomspace(amax, 1e-2 * amax, K)]
        flags = _engine.FLAG_PROFILE | _engine.FLAG_FRESH_L
        lanes = ds.path_lanes(K, flags)
        for _ in range(2):
            r = ds.solve_path(pts, tol=tol, lanes=0, flags=flags)
        eng.synchronize()
        t0 = time.perf_counter()
        reps = 3
        ms_k, n_k = 0.0, 0
        for _ in range(reps):
            r = ds.solve_path(pts, tol=tol, lanes=0, flags=flags)
            ms_k += r.grad_ms_total
            n_k += r.grad_timed
        eng.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / reps
        one = ds.solve_path(pts[:4], tol=tol, lanes=1, flags=flags)  # (one lane: the same route -- the split pass -- at this width)
    slots = max(16, lanes) if lanes <= 20 else 32
    bytes_per_pass = 8.0 * (n * p + slots * n + slots * p)
    t_k = ms_k / max(1, n_k)
    return {"n": n, "p": p, "alphas": K, "lanes": int(lanes), "ms_per_path": ms, "fits_per_s": K / (1e-3 * ms), "passes": int(r.grad_launches),
            "converged": bool(r.converged and one.converged), "working_set_refinements": int(r.ws_refined),
            "xtr_kernel_ms": t_k, "algorithmic_bytes_per_pass": bytes_per_pass,
            "roofline_frac_per_pass": (bytes_per_pass / (t_k * 1e-3) / 1e9 / HBM_PEAK_GBS) if t_k > 0 else None,
            "one_lane_passes_4_points": int(one.grad_launches)}


def leg_plain(eng, rank, n, p, K, tol, steps=3):
    """The headline path WITHOUT the working set, so that the kernel's contribution can be told from the algorithm's:
    `plain_fista` -- accelerated proximal gradient with restarts (the iteration the north star names), four lanes on the fused
    one-read kernel; `plain_spectral_16` -- the engine's spectral steps on sixteen lanes, whose pass is two reads of X
    (rowdot_mfma_kernel, then xtr_mfma_kernel).  Per gradient unit (one pass for all its lanes) the roofline fraction is
    charged ONE W = 8 (n p + 2 n + 2 p lanes) bytes, as SURVEY 8(d) charges it, over the whole time between two passes."""
    from sparselm_amd import _engine

    coef = make_coef(p, 50, seed=0)
    out = {}
    with eng.synthetic_dataset(n, p, seed=1000 + rank, coef=coef, noise_sd=10.0) as ds:
        g0, _ = ds.gradient(None)
        amax = float(np.max(np.abs(g0)))
        points = [(a, 0.0, 0.0) for a in np.geomspace(amax, 1e-3 * amax, K)]
        for name, lanes, flags in (("plain_fista", 4, _engine.FLAG_FISTA_ONLY | _engine.FLAG_NO_WORKING_SET),
                                   ("plain_spectral_16", 16, _engine.FLAG_NO_WORKING_SET)):
            flags |= _engine.FLAG_FRESH_L | _engine.FLAG_PROFILE
            ds.solve_path(points, tol=tol, flags=flags, lanes=lanes)
            eng.synchronize()
            t0 = time.perf_counter()
            kernel_ms, timed, passes = 0.0, 0, 0
            for _ in range(steps):
                res = ds.solve_path(points, tol=tol, flags=flags, lanes=lanes)
                kernel_ms += res.grad_ms_total
                timed += res.grad_timed
                passes += res.grad_launches
            eng.synchronize()
            dt = (time.perf_counter() - t0) / steps
            W = 8.0 * (n * p + 2 * n + 2 * p * lanes)
            unit_ms = 1e3 * dt / (passes / steps)
            out[name] = {"fits_per_s": K / dt, "ms_per_path": 1e3 * dt, "passes": passes // steps, "lanes": lanes,
                         "converged": bool(res.converged), "gradient_unit_ms": unit_ms,
                         "roofline_frac_per_unit_one_W": W / (unit_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "kernel_ms_by_hip_events": kernel_ms / max(1, timed),
                         "kernel_timed": "the fused one-read kernel" if lanes <= 4 else "xtr_mfma_kernel only (the second of the pass's two reads)"}
    return out


def leg_config3(eng, rank, world, n, p, tol, cpu_budget_s, steps=5):
    """BASELINE config 3 (GroupLasso, 500 shuffled groups of 10, 50-alpha path) on the data law of `Config4`, timed on
    the GPU; and -- rank 0 of a one-GPU run with a CPU budget -- an independent full-size referee for the group family:
    the oracle's C twin (oracle/fista_ref.c, OpenMP) solves a prefix of that path, and one (fold, l1_ratio, alpha) cell
    of config 4 from zero, on the downloaded (X, y); nothing of the engine enters its answer (its step size comes from
    its own power iteration)."""
    from sparselm_amd import _engine

    c4 = Config4(eng, n, p)
    try:
        ds, G, groups, K = c4.ds, c4.G, c4.groups, c4.K
        g0, _ = ds.gradient(None)
        bmax = float(np.sqrt(np.bincount(groups, weights=g0 * g0, minlength=G)).max())
        alphas = np.geomspace(bmax, 1e-3 * bmax, K)
        pts = np.c_[0 * alphas, alphas, 0 * alphas]
        for _ in range(2):
            res = ds.solve_path(pts, tol=tol, lanes=0, flags=_engine.FLAG_FRESH_L)  # (0: the engine's choice -- twenty-five for these fifty points in contiguous ranges)
        eng.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            res = ds.solve_path(pts, tol=tol, lanes=0, flags=_engine.FLAG_FRESH_L)  # (0: the engine's choice -- twenty-five for these fifty points in contiguous ranges)
        eng.synchronize()
        dt = (time.perf_counter() - t0) / steps
        out = {"fits_per_s": K / dt, "ms_per_path": 1e3 * dt, "passes": int(res.grad_launches), "converged": bool(res.converged),
               "active_groups_last": int(np.sum(np.bincount(groups, weights=res.betas[-1] ** 2, minlength=G) > 0))}
        if not (rank == 0 and world == 1 and cpu_budget_s > 0):
            return out
        # one cell of config 4 on the GPU: fold 2, l1_ratio 0.55, the path down to its 31st alpha
        u = 2 * 10 + 5
        f, ratio = c4.units[u]
        cell_k = 30
        cell = ds.solve_lanes([dict(points=c4.unit_pts[u][: cell_k + 1], row_weight=c4.masks[f], n_eff=int(c4.masks[f].sum()))],
                              tol=tol)[0]
        import oracle
        from oracle import cref

        X0, y = ds.download()
        numa = cref.NumaMatrix(X0)
        del X0
        X = numa.array
        gi = groups.astype(np.int32)

        def twin_L(w):  # lambda_max(X^T W X) / n by the twin's own gradient: 8 power steps, 10 % margin
            v = np.random.default_rng(0).standard_normal(p)
            lam = 1.0
            for _ in range(8):
                v /= np.linalg.norm(v)
                gv, _ = cref.gradient(X, 0.0 * y, v, w=w)
                lam = float(np.linalg.norm(gv))
                v = gv
            return 1.1 * lam

        L = twin_L(None)
        beta, done, worst, iters = None, 0, 0.0, 0
        t_start = time.perf_counter()
        for k, alpha in enumerate(alphas):
            if done >= 3 and time.perf_counter() - t_start > 0.6 * cpu_budget_s:
                break
            beta, it = cref.fista(X, y, 0.0, alpha, 0.0, gi, G, beta0=beta, L=L, tol=1e-9, max_iter=5000)
            iters += abs(it)
            done += 1
            top = float(np.max(np.abs(beta)))
            if k > 0 and top > 0:
                worst = max(worst, float(np.max(np.abs(res.betas[k] - beta)) / top))
        out["referee"] = {
            "what": f"oracle/fista_ref.c (OpenMP, {cref.num_threads()} threads) on the downloaded data: the first {done} points of the "
            f"GroupLasso path, warm-started, tol 1e-9 ({iters} gradients)",
            "points_checked": done, "beta_rel_inf_err_gpu_vs_oracle": worst, "seconds": time.perf_counter() - t_start,
        }
        # the config-4 cell: minimise 1/(2 n_f) sum_i w_i e_i^2 + pen  ==  1/(2 n) sum_i w_i e_i^2 + (n_f / n) pen
        w = c4.masks[f]
        scale = float(w.sum()) / n
        sa, sb, _ = c4.unit_pts[u][cell_k]
        t_cell = time.perf_counter()
        bc, it = cref.fista(X, y, scale * sa, scale * sb, 0.0, gi, G, L=twin_L(w), tol=1e-9, max_iter=3000, w=w)
        top = float(np.max(np.abs(bc)))
        out["referee_config4_cell"] = {
            "what": f"SparseGroupLasso cell (fold {f}, l1_ratio {ratio:.2f}, alpha index {cell_k}) of config 4: the twin from zero "
            f"on the fold's training rows ({abs(it)} gradients{'' if it > 0 else ', NOT converged'})",
            "beta_rel_inf_err_gpu_vs_oracle": float(np.max(np.abs(cell.betas[cell_k] - bc)) / top) if top > 0 else None,
            "nonzeros": int(np.count_nonzero(bc)), "seconds": time.perf_counter() - t_cell,
        }
        numa.__exit__()
        return out
    finally:
        c4.close()


def leg_concurrent_paths(eng, device_id, rank, n, p, K, tol, lanes, streams=3, steps=10):
    """The headline path on `streams` engines of ONE GPU at once (each with its own dataset of the same law and its
    own host thread): the launches between the passes of one path run beside the passes of the other.  Not the
    headline -- that is one path per GPU per step -- but what a grid search with more units than GPUs can do."""
    from sparselm_amd import _engine

    engines = [eng] + [_engine.Engine(device_id) for _ in range(streams - 1)]
    coef = make_coef(p, 50, seed=0)
    sets = []
    try:
        for i, e in enumerate(engines):
            ds = e.synthetic_dataset(n, p, seed=5000 + 10 * rank + i, coef=coef, noise_sd=10.0)
            g0, _ = ds.gradient(None)
            amax = float(np.max(np.abs(g0)))
            pts = [(a, 0.0, 0.0) for a in np.geomspace(amax, 1e-3 * amax, K)]
            ds.solve_path(pts, tol=tol, flags=_engine.FLAG_FRESH_L, lanes=lanes)  # warm (column-major copy, buffers)
            sets.append((ds, pts))
        ok = [True] * len(sets)

        def work(i):
            ds, pts = sets[i]
            for _ in range(steps):
                ok[i] = ok[i] and ds.solve_path(pts, tol=tol, flags=_engine.FLAG_FRESH_L, lanes=lanes).converged

        def run(which):
            threads = [threading.Thread(target=work, args=(i,)) for i in which]
            for e in engines:
                e.synchronize()
            t0 = time.perf_counter()
            for t in threads:
                t.start()
            for t in threads:
                t.join()
            for e in engines:
                e.synchronize()
            return time.perf_counter() - t0

        alone = run([0])
        together = run(list(range(len(sets))))
        return {"streams": len(sets), "converged": all(ok), "fits_per_s_one_stream": steps * K / alone,
                "fits_per_s_all_streams": len(sets) * steps * K / together, "ms_per_path_one_stream": 1e3 * alone / steps,
                "ms_per_path_pair": 1e3 * together / steps}
    finally:
        for ds, _ in sets:
            ds.close()
        for e in engines[1:]:
            e.close()


def leg_rowshard(eng, rank, world, n_rank, p, reps=2):
    """BASELINE config 5's shape through the engine's RCCL communicator: rank r owns rows
    [r n_rank, (r + 1) n_rank) of one synthetic matrix (generated on the device), 1000 groups x 10,
    AdaptiveGroupLasso with 3 re-weighting solves at alpha = 0.1 alpha_max; every pass all-reduces the
    lanes' gradients (and stop words) over the ranks."""
    from sparselm_amd import distributed as D

    G = p // 10
    groups = np.repeat(np.arange(G), 10)
    rng = np.random.default_rng(0)
    coef = np.zeros(p)
    for g in rng.choice(G, 30, replace=False):
        coef[groups == g] = rng.uniform(1, 5, 10)
    D.init_row_sharding(eng, rank=rank, world_size=world)
    ds = None
    try:
        comm_ranks = eng.comm_ranks()
        ds = eng.synthetic_dataset(n_rank, p, seed=1000, coef=coef, noise_sd=5.0, row_offset=rank * n_rank)
        ds.set_global_rows(n_rank * world)
        ds.set_groups(groups, G)
        g0, _ = ds.gradient(None)  # (all-reduced: identical on every rank)
        amax = float(np.max(np.sqrt(np.bincount(groups, weights=g0 * g0, minlength=G))))
        alpha, eps = 0.1 * amax, 1e-6
        out = {}
        for rep in range(reps + 1):  # first repetition warms up
            w = alpha * np.ones(G)
            beta = None
            passes = 0
            eng.synchronize()
            c_before = eng.comm_collectives()
            t0 = time.perf_counter()
            for _ in range(3):
                res = ds.solve_path([(0.0, 1.0, 0.0)], b=w, beta0=beta, tol=1e-8, want_group_norms=True)
                beta = res.betas[0]
                passes += res.grad_launches
                w = alpha * (alpha / (res.group_norms[0] + eps))
            eng.synchronize()
            dt = time.perf_counter() - t0
            out = {"seconds_per_fit": dt, "passes": passes, "converged": bool(res.converged),
                   "active_groups": int(np.sum(res.group_norms[0] > 0)), "rccl_ranks": comm_ranks, "device": eng.device_id,
                   "beta_checksum": float(np.sum(beta * np.arange(1, p + 1))),
                   "collectives_per_pass_incl_first": (eng.comm_collectives() - c_before) / max(1, passes)}
        # the two collectives of a pass on their own, on this communicator: the lanes' gradients (one lane: ld + 16 doubles) and
        # the working set's Gram parts with the stop words behind them (512 x 512 + 16); per collective, back to back
        ld = (p + 15) // 16 * 16
        out["collective_us"] = {"gradient_1_lane": eng.comm_all_reduce_us(ld + 16), "gram_parts_and_stop_words": eng.comm_all_reduce_us(512 * 512 + 16),
                                "packed_in_one": eng.comm_all_reduce_us(ld + 16 + 512 * 512 + 16), "stop_words_alone": eng.comm_all_reduce_us(16)}
        return out
    finally:
        if ds is not None:
            ds.close()
        eng.comm_destroy()


# ---------------------------------------------------------------------------------------------------
def spawn_ranks(n_ranks):
    """Start the ranks as a CHILD torchrun (this process has made no HIP call) and leave with its code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this pool
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.run(cmd, env=env).returncode


def strong_scaling_object(c4_s, c4_cov_s, world):
    """THE strong-scaling figure of a multi-GPU run: BASELINE config 4's grid, total work fixed, dealt to the ranks (the weak
    headline `value` is independent paths per GPU and scales trivially).  The driver computes efficiency from the per-N values."""
    return {
        "metric": "fits/sec over the 2500-fit SparseGroupLasso grid of BASELINE config 4 (5 folds x 10 l1_ratio x 50 alpha, n=100k p=5k)",
        "scaling": "strong", "n_gpus": world, "unit": "fits/s", "higher_is_better": True,
        "value": 2500.0 / c4_s, "seconds_per_grid": c4_s, "route": "over X (path points dealt to the lane slots of all ranks, no collective)",
        "value_from_fold_grams": (2500.0 / c4_cov_s) if c4_cov_s else None, "seconds_per_grid_from_fold_grams": c4_cov_s,
        "note": "measured on this run's ranks; with one rank, extra_legs.config4_grid_emulated_world8 holds the eight-rank shares "
        "as timed one after the other on this GPU -- an emulation, not a measurement on eight devices",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", type=int, default=100_000)
    ap.add_argument("--p", type=int, default=5_000)
    ap.add_argument("--alphas", type=int, default=50)
    ap.add_argument("--tol", type=float, default=1e-8)
    ap.add_argument("--lanes", type=int, default=0, help="lanes of the path advancing together on one pass over X (0: the engine's choice, "
                    "slm_solve_path_lanes with n_lanes = 0 -- eighteen for 50 points: three passes instead of four)")
    ap.add_argument("--no-ws", action="store_true", help="disable the working-set refinement (A/B runs)")
    ap.add_argument("--cpu-budget", type=float, default=20.0, help="seconds of CPU baseline work (0 = skip)")
    ap.add_argument("--no-extra", action="store_true", help="skip the config-4 grid and row-sharded legs")
    ap.add_argument("--legs", default="", help="comma-separated names of the extra legs to run (default: all)")
    ap.add_argument("--extra-timeout", type=float, default=360.0, help="hard limit (s) on the extra legs")
    ap.add_argument("--rowshard-rows", type=int, default=125_000, help="rows per rank of the row-sharded leg")
    ap.add_argument("--rowshard-cols", type=int, default=10_000)
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if rank == 0 and world != args.gpus:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: reporting n_gpus = {world}", file=sys.stderr)

    import torch
    import torch.distributed as dist

    use_dist = world > 1
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        with StdoutToStderr():  # (gloo announces its connections on stdout)
            dist.init_process_group(backend="gloo")  # control plane only: barrier + max of the timings
            dist.barrier()

    from sparselm_amd import _engine

    n_dev = max(1, _engine.device_count())
    device_id = local_rank % n_dev  # one rank per GPU; on a box with fewer GPUs than ranks (tests) ranks share devices
    have_torch_gpu = torch.cuda.is_available()
    if have_torch_gpu:
        torch.cuda.set_device(device_id % max(1, torch.cuda.device_count()))

    def sync_all():
        if use_dist:
            dist.barrier()
        if have_torch_gpu:
            torch.cuda.synchronize()
        eng.synchronize()

    def all_gather(obj):
        if not use_dist:
            return [obj]
        parts = [None] * world
        dist.all_gather_object(parts, obj)
        return parts

    eng = _engine.get_engine(device_id)
    n, p, K = args.n, args.p, args.alphas
    coef = make_coef(p, 50, seed=0)
    # independent unit per rank: fold/seed differs, law identical
    t_c = time.perf_counter()
    ds = eng.synthetic_dataset(n, p, seed=1000 + rank, coef=coef, noise_sd=10.0)
    eng.synchronize()
    create_ms = 1e3 * (time.perf_counter() - t_c)
    t_c = time.perf_counter()
    g0, _, _ = ds.gradient(None, reps=50)  # alpha_max; the extra launches bring the clocks up (setup)
    clock_warmup_ms = 1e3 * (time.perf_counter() - t_c)
    amax = float(np.max(np.abs(g0)))
    alphas = np.geomspace(amax, 1e-3 * amax, K)
    points = [(a, 0.0, 0.0) for a in alphas]
    flags = _engine.FLAG_PROFILE | _engine.FLAG_FRESH_L
    if args.no_ws:
        flags |= _engine.FLAG_NO_WORKING_SET

    # set-up, whatever --warmup says: the first path on a dataset pays for its column-major copy, the work-space
    # allocations and the first block of the result pool (reported below, outside `value`); two more bring the
    # allocator and the caches to rest.  Then the W warm-up steps of the contract, then the K timed ones.
    first_ms = None
    for _ in range(PRIMING_PATHS):
        t_c = time.perf_counter()
        ds.solve_path(points, tol=args.tol, flags=flags, lanes=args.lanes)
        if first_ms is None:
            first_ms = 1e3 * (time.perf_counter() - t_c)
    for _ in range(args.warmup):
        ds.solve_path(points, tol=args.tol, flags=flags, lanes=args.lanes)

    sync_all()
    t0 = time.perf_counter()
    grad_ms = 0.0
    grad_launches = 0
    grad_timed = 0
    res = None
    for _ in range(args.steps):
        res = ds.solve_path(points, tol=args.tol, flags=flags, lanes=args.lanes)
        grad_ms += res.grad_ms_total
        grad_launches += res.grad_launches
        grad_timed += res.grad_timed
    eng.synchronize()
    my_elapsed = time.perf_counter() - t0
    sync_all()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert res is not None
    info = eng.device_info()
    per_rank = all_gather({"rank": rank, "local_rank": local_rank, "device": device_id, "name": info["name"],
                           "seconds": my_elapsed, "converged": bool(res.converged),
                           "host": socket.gethostname(), "pid": os.getpid()})

    out = None
    if rank == 0:
        assert all(r["converged"] for r in per_rank), "a path did not converge"
        # algorithmic bytes of one launch: X once, y once, per lane z read and g written
        # p = 5000: the fused kernels stop at four lanes; the split pass of working-set solves has sixteen
        # (and plain solves of more than four lanes on large X: the same two matrix-core halves)
        lanes_run = ds.path_lanes(K, flags) if args.lanes == 0 else min(args.lanes, K)  # (0: what the engine chose)
        split = (not args.no_ws and res.ws_builds > 0) or (lanes_run > 4 and n * p >= 2**26 and p <= 10240)
        lanes_used = max(1, min(lanes_run, 32 if split else 4))
        if split:  # X once, the row residuals of the lane slots in use (sixteen on the matrix cores, the others beside them), as many gradient rows out
            slots = max(16, lanes_used) if lanes_used <= 20 else 32
            bytes_per_grad = 8.0 * (n * p + slots * n + slots * p)
        else:
            bytes_per_grad = 8.0 * (n * p + 2 * n + 2 * p * lanes_used)
        xtr_name = ("xtr_mfma_kernel" if lanes_used <= 16 else "xtr18_mfma_kernel" if lanes_used <= 18 else
                    "xtr20_mfma_kernel" if lanes_used <= 20 else "xtr32_mfma_kernel")
        t_grad_ms = grad_ms / max(1, grad_timed)
        achieved = bytes_per_grad / (t_grad_ms * 1e-3) / 1e9 if t_grad_ms > 0 else 0.0
        # after the timed region, same dataset, same run: (i) the whole gradient unit of the split pass under the same HIP
        # events -- residuals of the sixteen lane slots (resid_mfma_kernel, and rowdot_mfma_kernel's empty launch) + X^T R --
        # charged ONE W; (ii) the read-only stream ceiling of THIS device on THIS copy of X (plain 16-byte loads, summed up)
        unit_ms = ceiling_gbs = ceiling_ms = None
        if split:
            u_ms, u_n = 0.0, 0
            for _ in range(2):
                ru = ds.solve_path(points, tol=args.tol, flags=flags | _engine.FLAG_PROFILE_UNIT, lanes=args.lanes)
                u_ms += ru.grad_ms_total
                u_n += ru.grad_timed
            unit_ms = u_ms / max(1, u_n)
        try:
            ceiling_gbs, ceiling_ms = ds.read_ceiling(reps=5)
        except Exception:  # noqa: BLE001 -- a measurement beside the line, never its condition
            pass
        unit_achieved = (bytes_per_grad / (unit_ms * 1e-3) / 1e9) if unit_ms else None
        secs = [r["seconds"] for r in per_rank]
        out = {
            "metric": "fits/sec over 50-alpha Lasso path at n=100k p=5k",
            "value": world * args.steps * K / elapsed,
            "unit": "fits/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"Lasso FISTA fp64, n={n} p={p}, {K}-alpha warm-started path "
                "(geomspace alpha_max..1e-3 alpha_max), one path per GPU per step",
                "n": n,
                "p": p,
                "n_alphas": K,
                "lanes": lanes_run,
                "lanes_chosen_by": "engine (slm_solve_path_lanes, n_lanes = 0)" if args.lanes == 0 else "--lanes",
                "tol": args.tol,
                "law": "make_regression(n_informative=50, noise=10): X~N(0,1) generated on device",
                "parallelism": f"grid x{world} (independent paths, no collective)",
                "grad_evals_per_path": grad_launches / args.steps,
                "lipschitz_ms_per_path": res.lipschitz_ms,
                "working_set": {"builds": res.ws_builds, "appends": res.ws_appends, "refined": res.ws_refined, "misses": res.ws_misses, "columns": res.ws_columns},
                "one_off_costs_outside_value_ms": {
                    "what": "paid once per dataset, before the timed region",
                    "dataset_generation": create_ms,
                    "clock_warmup_50_gradient_launches": clock_warmup_ms,
                    "first_path_incl_column_major_copy_and_workspace": first_ms,
                    "priming_paths_before_the_warmup_steps": PRIMING_PATHS,
                },
            },
            "ranks": {
                "devices": {str(r["rank"]): r["device"] for r in per_rank},
                "distinct_devices": len({(r["host"], r["device"]) for r in per_rank}),
                "device_names": sorted({r["name"] for r in per_rank}),
                "seconds_per_rank": secs,
                "imbalance_max_over_mean": max(secs) / (sum(secs) / len(secs)),
            },
            "roofline": {
                "bound": "hbm",
                # the unit is one gradient X^T (X z - y) / n.  `achieved` / `frac` are the WHOLE unit of the
                # split pass -- the residuals of the lane slots and the X^T R kernel under one bracket of HIP events -- charged
                # the algorithmic bytes of one gradient; the X^T R kernel alone (the launch that streams X) is kernel_*.
                "achieved": unit_achieved if unit_achieved else achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": (unit_achieved if unit_achieved else achieved) / HBM_PEAK_GBS,
                "frac_is": "gradient unit (residuals + X^T R)" if unit_achieved else "the kernel that streams X (no split pass: the fused kernel is the whole unit)",
                "kernel_achieved": achieved,
                "kernel_frac": achieved / HBM_PEAK_GBS,
                "traffic": measured_traffic(n, p, lanes_used, xtr_name if split else "grad_fused_kernel")[0],
                "traffic_unit": "HBM bytes per launch (PMC, profiles/roofline_traffic.json)",
                "traffic_note": measured_traffic(n, p, lanes_used, xtr_name if split else "grad_fused_kernel")[1],
                "kernel": (f"{xtr_name} (X^T R of the split pass: sixteen lanes on the matrix cores"
                           f"{', the others on the vector units beside them' if 16 < lanes_used <= 20 else ''}; lanes={lanes_used})" if split
                           else f"grad_fused_kernel (lanes={lanes_used})"),
                "avg_kernel_ms": t_grad_ms,
                "gradient_unit_ms": unit_ms,
                "gradient_unit_frac": (bytes_per_grad / (unit_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if unit_ms else None,
                "gradient_unit_what": "residuals of the lane slots (resid_mfma_kernel on the gathered columns) + the X^T R kernel, "
                "one bracket of HIP events per pass (SLM_FLAG_PROFILE_UNIT), charged the same algorithmic bytes: what a 16-lane gradient costs",
                "read_stream_ceiling_gbs": ceiling_gbs,
                "read_stream_ceiling_ms_per_sweep": ceiling_ms,
                "kernel_frac_of_read_stream_ceiling": (achieved / ceiling_gbs) if ceiling_gbs else None,
                "read_stream_ceiling_what": "slm_dataset_read_ceiling: the device copy of X swept by plain 16-byte loads that are only "
                "summed up, same device, same run (the guide's 6.29 TB/s is a float4 COPY; `peak` stays the 8 TB/s of the data sheet)",
                "launches": grad_launches,
                "launches_timed_with_hip_events": grad_timed,
                "algorithmic_bytes_per_launch": bytes_per_grad,
                "path_level": {
                    "what": "algorithmic bytes of all passes of a path / whole path time (everything between the passes counted)",
                    "achieved": (grad_launches / args.steps) * bytes_per_grad / (1e-3 * 1e3 * elapsed / args.steps) / 1e9,
                    "frac": (grad_launches / args.steps) * bytes_per_grad / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS,
                },
            },
        }
        if world == 1 and args.cpu_budget > 0:
            out["cpu_baseline"] = cpu_baseline(ds, alphas, res.L, args.tol, res.betas, args.cpu_budget)
        else:
            out["cpu_baseline"] = None
    ds.close()

    # ---- extra legs: after the timed region, guarded by a hard limit ------------------------------------
    if not args.no_extra:
        legs = {}
        if out is not None:
            out["extra_legs"] = legs

        real_stdout = os.dup(1)  # (a leg may have fd 1 pointing at stderr when the limit strikes)

        def expire():
            # the contract line goes out (the timed region is long over), then every rank leaves with a code of its own:
            # a stuck leg must not read as success to torchrun / the driver
            if out is not None:
                legs["timed_out_after_s"] = args.extra_timeout
                os.write(real_stdout, (json.dumps(out) + "\n").encode())
            os.write(2, f"[bench] rank {rank}: extra legs exceeded {args.extra_timeout:.0f} s, leaving with status 3\n".encode())
            os._exit(3)

        with Watchdog(args.extra_timeout, expire):
            for name, fn in (("config4_grid", lambda: leg_config4_grid(eng, rank, world, n, p, device_id)),
                             ("config4_grid_dense_regime", lambda: leg_config4_dense(eng, n, p) if rank == 0 and world == 1 else {}),
                             ("config3_path", lambda: leg_config3(eng, rank, world, n, p, args.tol, args.cpu_budget)),
                             ("headline_draws", lambda: leg_headline_draws(eng, n, p, K, args.tol, args.lanes) if rank == 0 else {}),
                             ("soak", lambda: leg_soak(eng, n, p, K, args.tol, args.lanes) if rank == 0 else {}),
                             ("plain_iteration", lambda: leg_plain(eng, rank, n, p, K, args.tol) if rank == 0 else {}),
                             # (every rank: with a process group up, GridSearchCV shards the search over the ranks and gathers)
                             ("config1_small", lambda: leg_config1_small()),
                             ("concurrent_paths", lambda: leg_concurrent_paths(eng, device_id, rank, n, p, K, args.tol, args.lanes)),
                             ("wide_rows", lambda: leg_wide_rows(eng) if rank == 0 and world == 1 else {}),
                             ("literal_config2", lambda: leg_literal_config2(eng, K, args.tol, args.lanes, twin=args.cpu_budget > 0) if rank == 0 and world == 1 else {}),
                             ("rowshard", lambda: leg_rowshard(eng, rank, world, args.rowshard_rows, args.rowshard_cols))):
                if args.legs and name not in args.legs.split(","):
                    continue
                try:
                    with StdoutToStderr():
                        mine = {"ok": True, **fn()}
                except Exception as exc:  # a leg never breaks the contract line
                    mine = {"ok": False, "error": repr(exc)[:300]}
                if use_dist:
                    dist.barrier()
                parts = all_gather(mine)
                if out is None:
                    continue
                if not all(q["ok"] for q in parts):
                    legs[name] = {"error": [q.get("error") for q in parts if not q["ok"]][:2]}
                elif name == "config4_grid":
                    secs = [q["seconds"] for q in parts]
                    pts = [q["points"] for q in parts]
                    legs[name] = {
                        "what": "SparseGroupLasso 5 folds x 10 l1_ratio x 50 alpha = 2500 fits at n=100k p=5k; the path points "
                        "of the 50 (fold, l1_ratio) units dealt to the lane slots of the ranks (distributed.plan_lane_calls), 16 "
                        "lanes per call; strong scaling; *_streams: a rank's calls dealt to standing engines (HIP streams) of "
                        "its GPU where it has more than one to make",
                        "fits_per_s": 2500.0 / max(secs), "seconds_per_grid": max(secs), "seconds_per_rank": secs,
                        "seconds_per_grid_streams": (max(q["seconds_streams"] for q in parts)
                                                     if all("seconds_streams" in q for q in parts) else None),
                        "streams": parts[0].get("streams", 1), "calls_per_rank": [q["calls"] for q in parts],
                        "points_per_rank": pts, "passes_per_rank": [q["passes"] for q in parts],
                        "imbalance_max_over_mean": max(secs) / (sum(secs) / len(secs)),
                        "covariance": ({
                            "what": "the same share with SLM_FLAG_COVARIANCE: every pass reads the Gram of its fold (200 MB) "
                            "instead of X; the five Grams are built once per dataset (slm_dataset_covariance: cov_syrk_kernel on the "
                            "matrix cores; the test rows of the folds partition the rows, so all five cost one triangle product) -- worth it when "
                            "build_s + seconds_per_grid beats the grid over X, as here, or the search is "
                            "repeated on the dataset, or its paths end dense (config4_grid_dense_regime)",
                            "seconds_per_grid": max(q["seconds_covariance"] for q in parts),
                            "fits_per_s": 2500.0 / max(q["seconds_covariance"] for q in parts),
                            "build_s": max(q["covariance_build_s"] for q in parts),
                            "seconds_build_plus_one_grid": max(q["covariance_build_s"] + q["seconds_covariance"] for q in parts),
                            "passes_per_rank": [q["passes_covariance"] for q in parts],
                            "seconds_per_grid_streams": (max(q["seconds_covariance_streams"] for q in parts)
                                                         if all("seconds_covariance_streams" in q for q in parts) else None),
                        } if all("seconds_covariance" in q for q in parts) else {"error": parts[0].get("covariance_error")}),
                    }
                    if "emulated" in parts[0]:
                        em = parts[0]["emulated"]
                        sh = em["shares"]
                        worst = max(q["seconds"] for q in sh)
                        legs["config4_grid_emulated_world8"] = {
                            "what": f"the same grid as dealt to {em['world']} ranks: every rank's share timed on THIS GPU, one after "
                            "the other (on a node each runs on its own GPU, X replicated, no collective); speed-up = the "
                            "one-GPU grid time above / the slowest share",
                            "world": em["world"], "full_grid_s": max(secs), "max_share_s": worst,
                            "speedup_full_over_max_share": max(secs) / worst,
                            "share_seconds": [q["seconds"] for q in sh], "share_passes": [q["passes"] for q in sh],
                            "share_points": [q["points"] for q in sh], "share_lanes": [q["lanes"] for q in sh],
                            "share_row_masks": [q["row_masks"] for q in sh], "full_grid_passes": parts[0]["passes"],
                            "points_imbalance_max_over_mean": max(q["points"] for q in sh) / (sum(q["points"] for q in sh) / len(sh)),
                            "seconds_imbalance_max_over_mean": worst / (sum(q["seconds"] for q in sh) / len(sh)),
                        }
                        sg = em.get("shared_grams") or {}
                        if "shares" in sg:
                            e8 = legs["config4_grid_emulated_world8"]
                            w8 = em["world"]
                            exch = sg["finish_wall_s_all_ranks_on_one_gpu"] / w8
                            per = [q["build_s"] + exch + q["solve_s"] for q in sg["shares"]]
                            # an xGMI ring all-reduce of the same bytes: 2 (N - 1) / N x bytes / bus bandwidth
                            xgmi_busbw = 300e9
                            ring = 2.0 * (w8 - 1) / w8 * sg["exchange_bytes_per_rank"] / xgmi_busbw
                            per_ring = [q["build_s"] + max(exch, ring) + q["solve_s"] for q in sg["shares"]]
                            cov = legs[name]["covariance"]
                            best_one = min(max(secs), cov.get("seconds_build_plus_one_grid", float("inf")))
                            e8["shared_grams"] = {
                                "what": "the same 8 ranks with the folds' Grams built TOGETHER: every rank (a replica on an engine of an "
                                "in-process communicator of this GPU) builds the parts of its eighth of the rows (build_s, timed alone), the "
                                "ranks sum them and form the Grams (all at once on this one GPU: charged wall / 8 each = exchange_s), and "
                                "solves its share from the Grams (solve_s, timed alone); share = build + exchange + solve.  "
                                "*_xgmi_ring: the exchange replaced by max(measured, a modelled ring all-reduce of the same bytes at "
                                f"{xgmi_busbw / 1e9:.0f} GB/s bus bandwidth) -- a model, stated, not a measurement",
                                "build_s": [q["build_s"] for q in sg["shares"]], "solve_s": [q["solve_s"] for q in sg["shares"]],
                                "share_passes": [q["passes"] for q in sg["shares"]], "exchange_s": exch,
                                "exchange_bytes_per_rank": sg["exchange_bytes_per_rank"], "collectives_per_rank": sg["collectives_per_rank"],
                                "share_seconds": per, "max_share_s": max(per),
                                "speedup_vs_one_gpu_over_x": max(secs) / max(per),
                                "speedup_vs_best_one_gpu": best_one / max(per),
                                "best_one_gpu_s": best_one,
                                "best_one_gpu_is": "Grams built + one grid from them" if best_one < max(secs) else "the grid over X",
                                "xgmi_ring_model_s": ring, "max_share_s_xgmi_ring": max(per_ring),
                                "speedup_vs_one_gpu_over_x_xgmi_ring": max(secs) / max(per_ring),
                                "speedup_vs_best_one_gpu_xgmi_ring": best_one / max(per_ring),
                            }
                        elif sg:
                            legs["config4_grid_emulated_world8"]["shared_grams"] = sg
                elif name == "config4_grid_dense_regime":
                    q = parts[0]
                    if q.get("seconds"):
                        legs[name] = {
                            "what": "config 4's grid with noise 100 (every path ends at thousands of non-zeros; the 512-column "
                            "working set gives up on a quarter of the points): over X -- those points on rounds on the folds' model "
                            "Grams (five fp16 products, built inside the first call that needs them and kept; *_without_model_gram: "
                            "plain sixteen-lane passes of two reads of X, round 4's route) -- and from the folds' fp64 Grams "
                            "(SLM_FLAG_COVARIANCE); one GPU, rank 0",
                            "seconds_per_grid": q["seconds"], "fits_per_s": 2500.0 / q["seconds"], "passes": q["passes"],
                            "seconds_per_grid_without_model_gram": q.get("seconds_without_model_gram"),
                            "passes_without_model_gram": q.get("passes_without_model_gram"),
                            "covariance": {"seconds_per_grid": q["seconds_covariance"], "fits_per_s": 2500.0 / q["seconds_covariance"],
                                           "passes": q["passes_covariance"], "build_s": q["covariance_build_s"],
                                           "speedup": q["seconds"] / q["seconds_covariance"]},
                            "nnz_last_min_median_max": q["nnz_last_min_median_max"],
                            "points_above_512_nonzeros": q["points_above_512_nonzeros"],
                            "worst_rel_inf_diff_covariance_vs_x": q["worst_rel_inf_diff"],
                        }
                elif name == "config3_path":
                    legs[name] = {
                        "what": "BASELINE config 3: GroupLasso, 500 shuffled groups x 10 features, 50-alpha warm-started path at "
                        "n=100k p=5k, sixteen lanes, per rank; referee*: the oracle's C twin on the host cores (rank 0, one GPU)",
                        "fits_per_s": [q["fits_per_s"] for q in parts], "ms_per_path": [q["ms_per_path"] for q in parts],
                        "passes": parts[0]["passes"], "converged": all(q["converged"] for q in parts),
                        "active_groups_last": parts[0]["active_groups_last"],
                        **{k: v for k, v in parts[0].items() if k.startswith("referee")},
                    }
                elif name == "soak":
                    cases = parts[0]["cases"]
                    rate = sorted(c["fits_per_s"] for c in cases)
                    dense = [c for c in cases if c["nnz_last"] > 512]
                    legs[name] = {
                        "what": "the headline path on twelve random datasets of the headline shape (seeds 4..15 of "
                        "tools/headline_soak.py's law: 5..199 informative features, noise 0.1..100, path floor 1e-3..0.1 "
                        "alpha_max), each checked against the plain four-lane iteration; never part of `value`",
                        "median_fits_per_s": rate[len(rate) // 2], "worst_fits_per_s": rate[0], "best_fits_per_s": rate[-1],
                        "paths_ending_above_512_nonzeros": len(dense),
                        "passes": [c["passes"] for c in cases], "ms": [round(c["ms"], 2) for c in cases],
                        "nnz_last": [c["nnz_last"] for c in cases],
                        "worst_rel_inf_err_vs_plain_iteration": max(c["rel_inf_err_vs_plain"] for c in cases),
                        "all_converged": all(c["converged"] for c in cases),
                        "median_ms_dense_end": (sorted(c["ms"] for c in dense)[len(dense) // 2] if dense else None),
                        "model_gram": {
                            "what": "paths whose ends outgrow the working set take their points from rounds on the model Gram "
                            "(csrc/mg_kernels.hpp: an fp16 MFMA product of all of X^T X / n, built once per dataset inside the first "
                            "solve that needs it; every proposal verified by a pass over X in fp64).  ms / passes: the steady state "
                            "(Gram there); first_solve_ms: the fresh dataset's first path, build included; *_without: the same "
                            "paths with SLM_FLAG_NO_MODEL_GRAM (plain sixteen-lane passes of two reads, round 4's route)",
                            "seeds": [c["seed"] for c in cases if c["model_gram_rounds"] > 0],
                            "ms": [round(c["ms"], 2) for c in cases if c["model_gram_rounds"] > 0],
                            "passes": [c["passes"] for c in cases if c["model_gram_rounds"] > 0],
                            "first_solve_ms": [round(c["first_solve_ms"], 2) for c in cases if c["model_gram_rounds"] > 0],
                            "build_ms": [round(c["model_gram_build_ms"], 2) for c in cases if c["model_gram_rounds"] > 0],
                            "ms_without": [round(c["ms_without_model_gram"], 2) for c in cases if c["model_gram_rounds"] > 0],
                            "passes_without": [c["passes_without_model_gram"] for c in cases if c["model_gram_rounds"] > 0],
                            "rejected_proposals": sum(c["model_gram_rejected"] for c in cases),
                        },
                        "worst_first_solve_fits_per_s": min(K / (1e-3 * c["first_solve_ms"]) for c in cases),
                        "worst_fits_per_s_without_model_gram": min([K / (1e-3 * c["ms_without_model_gram"]) for c in cases if c["ms_without_model_gram"]] or [0.0]) or None,
                        "median_ms_sparse_end": sorted(c["ms"] for c in cases if c["nnz_last"] <= 512)[(len(cases) - len(dense)) // 2]
                        if len(dense) < len(cases) else None,
                    }
                elif name == "headline_draws":
                    cases = parts[0].get("cases", [])
                    if cases:
                        ms = sorted(c["ms"] for c in cases)
                        ms16 = sorted(c["ms_16_lanes"] for c in cases)
                        legs[name] = {
                            "what": "the headline path on eight OTHER draws of the headline's law (same coefficients, other X and noise; "
                            "rank 0): passes and ms per path on the engine's choice of lanes (or --lanes) and on sixteen lanes -- the "
                            "number of passes depends on the draw (a point that meets a feature outside the working set is verified a "
                            "pass later), `value` is quoted on ONE draw; never part of `value`",
                            "data_seeds": [c["data_seed"] for c in cases],
                            "passes": [c["passes"] for c in cases], "ms": [round(c["ms"], 3) for c in cases],
                            "passes_16_lanes": [c["passes_16_lanes"] for c in cases], "ms_16_lanes": [round(c["ms_16_lanes"], 3) for c in cases],
                            "mean_ms": sum(ms) / len(ms), "median_ms": ms[len(ms) // 2], "worst_ms": ms[-1],
                            "mean_fits_per_s": K * len(ms) / (1e-3 * sum(ms)),
                            "mean_ms_16_lanes": sum(ms16) / len(ms16),
                            "all_converged": all(c["converged"] and c["converged_16_lanes"] for c in cases),
                        }
                elif name == "wide_rows":
                    if parts[0].get("ms_per_path"):
                        legs[name] = {"what": "a 32-alpha Lasso path at n = 50 000, p = 20 000 (rows beyond the fused kernels' 10 240 columns: the split "
                                      "pass at any width, sixteen lanes and the working set; rank 0, one GPU); never part of `value`",
                                      **{k: v for k, v in parts[0].items() if k != "ok"}}
                elif name == "literal_config2":
                    if parts[0].get("ms_per_path"):
                        legs[name] = {k: v for k, v in parts[0].items() if k != "ok"}
                elif name == "plain_iteration":
                    legs["plain_fista"] = {"what": "the headline path by plain FISTA with restarts, no working set, four lanes on the "
                                           "fused one-read kernel: what the kernel alone buys", **parts[0].get("plain_fista", {})}
                    legs["plain_spectral_16"] = {"what": "the headline path by the engine's spectral steps without the working set, "
                                                 "sixteen lanes, two reads of X per pass (the regime of paths that end above 512 "
                                                 "non-zeros)", **parts[0].get("plain_spectral_16", {})}
                elif name == "config1_small":
                    legs[name] = {
                        "what": "BASELINE config 1 (the reference's README example: GridSearchCV(AdaptiveLasso), 10 alphas x 5 folds + "
                        "refit, make_regression(100, 80)) and two fits of the reference's own sizes through the estimators, on the "
                        "one-workgroup solvers (small_kernels.hpp, small_split_kernels.hpp); wall time of the user's call on rank 0 (with "
                        "more than one rank the grid search is sharded over the ranks and gathered through torch.distributed)",
                        **{k: v for k, v in parts[0].items() if k != "ok"},
                    }
                elif name == "concurrent_paths":
                    legs[name] = {
                        "what": "the headline path on three engines (streams) of ONE GPU at once, a dataset and a host thread "
                        "each: the launches between the passes of one path run beside the passes of the others; per rank",
                        "streams": parts[0]["streams"],
                        "fits_per_s_one_stream": [q["fits_per_s_one_stream"] for q in parts],
                        "fits_per_s_all_streams": [q["fits_per_s_all_streams"] for q in parts],
                        "gain": [q["fits_per_s_all_streams"] / q["fits_per_s_one_stream"] for q in parts],
                        "converged": all(q["converged"] for q in parts),
                    }
                else:
                    secs = [q["seconds_per_fit"] for q in parts]
                    sums = [q["beta_checksum"] for q in parts]
                    legs[name] = {
                        "what": f"AdaptiveGroupLasso, 3 re-weighting solves, {args.rowshard_rows} rows x {args.rowshard_cols} "
                        f"columns per rank ({world * args.rowshard_rows} rows in all), RCCL all-reduce of the lanes' "
                        "gradients every pass; weak scaling in rows",
                        "fits_per_s": 1.0 / max(secs), "seconds_per_fit": max(secs), "seconds_per_rank": secs,
                        "passes": parts[0]["passes"], "passes_agree": len({q["passes"] for q in parts}) == 1,
                        "rccl_ranks": parts[0]["rccl_ranks"], "converged": all(q["converged"] for q in parts),
                        # non-zero: the ranks sit on distinct devices and RCCL still did not join them all (a failure, loud)
                        "status": int(len({q["device"] for q in parts}) == world and any(q["rccl_ranks"] != world for q in parts)),
                        "active_groups": parts[0]["active_groups"],
                        "ranks_hold_identical_coefficients": max(sums) == min(sums),
                        "rows_per_s": world * args.rowshard_rows * parts[0]["passes"] / max(secs),
                        "collective_us": parts[0].get("collective_us"),
                        "collectives_per_pass_incl_first": parts[0].get("collectives_per_pass_incl_first"),
                    }
    if out is not None:
        # the legs' headline figures once more as plain scalars (a reader that keeps only the shallow part of the line
        # still gets them); every figure is measured by the leg named in its key
        legs = out.get("extra_legs", {})

        def pick(*path):
            v = legs
            for k in path:
                v = v.get(k) if isinstance(v, dict) else None
            return v if isinstance(v, (int, float, str, bool)) else None

        e8 = ("config4_grid_emulated_world8", "shared_grams")
        # THE strong-scaling figure of a multi-GPU run: BASELINE config 4's grid, total work fixed, dealt to the ranks (the weak
        # headline `value` is independent paths per GPU and scales trivially).  The driver computes efficiency from the
        # per-N values of this object; at N = 1 it is the one-GPU time the emulated shares are compared with.
        c4_s, c4_cov_s = pick("config4_grid", "seconds_per_grid"), pick("config4_grid", "covariance", "seconds_per_grid")
        if c4_s:
            out["strong_scaling"] = strong_scaling_object(c4_s, c4_cov_s, world)
        if world > 1 and pick("rowshard", "rccl_ranks") is not None:
            # a run on several devices whose row-sharded leg did not join them all is a failed run, said so in the line
            out["rowshard_joined_all_ranks"] = bool(pick("rowshard", "rccl_ranks") == world)
        # `value` is quoted on ONE draw of the law (the bench's dataset: the contract); how many passes a path takes depends on
        # the draw, so the mean over the eight OTHER draws the headline_draws leg measures stands beside it at the top level
        if pick("headline_draws", "mean_fits_per_s") is not None:
            out["value_mean_over_draws"] = pick("headline_draws", "mean_fits_per_s")
            out["ms_per_step_mean_over_draws"] = pick("headline_draws", "mean_ms")
            out["value_mean_over_draws_what"] = ("fits/s of the same path on eight other draws of the same law (extra_legs.headline_draws: "
                                                 "3 timed paths each, one GPU, rank 0); `value` is the bench's own dataset")
        out["summary"] = {k: v for k, v in {
            "config4_strong_scaling_fits_per_s": (2500.0 / c4_s) if c4_s else None,
            "config4_strong_scaling_seconds_per_grid": c4_s,
            "config4_strong_scaling_n_gpus": world if c4_s else None,
            "config4_one_gpu_over_x_s": pick("config4_grid", "seconds_per_grid") if world == 1 else None,
            "config4_one_gpu_from_grams_s": pick("config4_grid", "covariance", "seconds_per_grid"),
            "config4_grams_build_s": pick("config4_grid", "covariance", "build_s"),
            "config4_one_gpu_from_grams_three_streams_s": pick("config4_grid", "covariance", "seconds_per_grid_streams"),
            "config4_world8_emulated_over_x_speedup": pick("config4_grid_emulated_world8", "speedup_full_over_max_share"),
            "config4_world8_emulated_shared_grams_speedup_vs_one_gpu_over_x": pick(*e8, "speedup_vs_one_gpu_over_x"),
            "config4_world8_emulated_shared_grams_speedup_vs_best_one_gpu": pick(*e8, "speedup_vs_best_one_gpu"),
            "config4_world8_emulated_shared_grams_speedup_vs_x_with_xgmi_ring_model": pick(*e8, "speedup_vs_one_gpu_over_x_xgmi_ring"),
            "config4_world8_emulated_shared_grams_max_share_s": pick(*e8, "max_share_s"),
            "config4_dense_regime_over_x_s": pick("config4_grid_dense_regime", "seconds_per_grid"),
            "config4_dense_regime_from_grams_s": pick("config4_grid_dense_regime", "covariance", "seconds_per_grid"),
            "config3_referee_rel_inf_err": pick("config3_path", "referee", "beta_rel_inf_err_gpu_vs_oracle"),
            "headline_law_other_draws_mean_fits_per_s": pick("headline_draws", "mean_fits_per_s"),
            "headline_law_other_draws_mean_ms": pick("headline_draws", "mean_ms"),
            "headline_law_other_draws_worst_ms": pick("headline_draws", "worst_ms"),
            "headline_law_other_draws_mean_ms_16_lanes": pick("headline_draws", "mean_ms_16_lanes"),
            "soak_median_fits_per_s": pick("soak", "median_fits_per_s"),
            "soak_worst_fits_per_s": pick("soak", "worst_fits_per_s"),
            "soak_worst_first_solve_fits_per_s": pick("soak", "worst_first_solve_fits_per_s"),
            "soak_worst_fits_per_s_without_model_gram": pick("soak", "worst_fits_per_s_without_model_gram"),
            "config4_dense_regime_over_x_without_model_gram_s": pick("config4_grid_dense_regime", "seconds_per_grid_without_model_gram"),
            "plain_fista_fits_per_s": pick("plain_fista", "fits_per_s"),
            "plain_fista_passes": pick("plain_fista", "passes"),
            "plain_fista_roofline_frac_per_unit": pick("plain_fista", "roofline_frac_per_unit_one_W"),
            "plain_spectral_16_fits_per_s": pick("plain_spectral_16", "fits_per_s"),
            "plain_spectral_16_roofline_frac_per_unit": pick("plain_spectral_16", "roofline_frac_per_unit_one_W"),
            "readme_grid_ms": pick("config1_small", "readme_grid_ms"),
            "lasso_fit_25x30_ms": pick("config1_small", "lasso_fit_25x30_ms"),
            "rowshard_seconds_per_fit": pick("rowshard", "seconds_per_fit"),
            "rowshard_rccl_ranks": pick("rowshard", "rccl_ranks"),
            "rowshard_status": pick("rowshard", "status"),
            "rowshard_collective_us_gradient": pick("rowshard", "collective_us", "gradient_1_lane"),
            "rowshard_collective_us_gram_parts": pick("rowshard", "collective_us", "gram_parts_and_stop_words"),
            "cpu_stock_sklearn_lasso_path_fits_per_s": (out.get("cpu_baseline") or {}).get("value") if (out.get("cpu_baseline") or {}).get("kind") == "stock" else None,
            "cpu_port_c_twin_fits_per_s": ((out.get("cpu_baseline") or {}).get("port") or {}).get("value") if (out.get("cpu_baseline") or {}).get("kind") == "stock" else (out.get("cpu_baseline") or {}).get("value"),
            "gpu_vs_sklearn_rel_inf_err": (out.get("cpu_baseline") or {}).get("beta_rel_inf_err_gpu_vs_sklearn"),
            "wide_rows_50000x20000_roofline_frac_per_pass": pick("wide_rows", "roofline_frac_per_pass"),
            "wide_rows_50000x20000_ms_per_path": pick("wide_rows", "ms_per_path"),
            "literal_config2_ms_per_path": pick("literal_config2", "ms_per_path"),
            "literal_config2_passes": pick("literal_config2", "passes"),
            "literal_config2_worst_rel_inf_vs_c_twin": pick("literal_config2", "worst_rel_inf_vs_c_twin"),
            "path_level_roofline_frac": out["roofline"]["path_level"]["frac"],
        }.items() if v is not None}
        # (the summary last but for the legs it condenses: a reader of the line's tail sees it whole)
        ordered = {k: v for k, v in out.items() if k not in ("extra_legs", "summary")}
        if world > 1 and out.get("rowshard_joined_all_ranks") is False:
            os.write(2, b"[bench] the row-sharded leg's RCCL communicator did not join every rank\n")
        if "extra_legs" in out:
            ordered["extra_legs"] = out["extra_legs"]
        ordered["summary"] = out["summary"]
        print(json.dumps(ordered), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

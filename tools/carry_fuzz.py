"""Random sequences of solves on ONE dataset -- penalties, warm starts, row masks, targets, lane counts, flags changing
from call to call the way estimators and searches change them -- with carried starts allowed (and, under
FLAG_WORKING_SET, the working set taken over with them), against the same sequence with SLM_NO_CARRY=1 on a second
dataset: the same solutions to the solver's tolerance, never more passes, fewer where a call starts at the solution
before it.  Usage: carry_fuzz.py [seeds ...]."""
import os, sys
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "sparse-lm_amd"))
from sparselm_amd import _engine


def run(seed):
    rng = np.random.default_rng(seed)
    n, p = int(rng.integers(600, 3000)), int(rng.integers(140, 900))
    G = int(rng.integers(5, 40))
    X = rng.standard_normal((n, p))
    coef = np.where(rng.random(p) < 0.05, rng.standard_normal(p) * 3, 0.0)
    y = X @ coef + rng.standard_normal(n)
    amax = float(np.max(np.abs(X.T @ y)) / n)
    groups = rng.integers(0, G, p) if rng.random() < 0.5 else None
    eng = _engine.get_engine(0)
    saved = worse = calls = 0
    with eng.dataset(X, y) as A, eng.dataset(X, y) as B:
        if groups is not None:
            for ds in (A, B):
                ds.set_groups(groups, G)
        Gn = A.n_groups
        last = None  # solutions of the previous call, per lane
        masks = [(rng.random(n) < 0.8).astype(float) for _ in range(3)]
        prev = None  # (lanes, mask index or None) of the call before
        for step in range(14):
            again = prev is not None and last is not None and rng.random() < 0.5  # the next round of the same cells
            lanes = prev[0] if again else int(rng.integers(1, 5))
            kind = rng.integers(0, 8)
            flags = 0
            if kind == 4:
                flags = _engine.FLAG_NO_WORKING_SET
            if kind == 5 and rng.random() < 0.5:
                flags = _engine.FLAG_FISTA_ONLY
            if kind >= 6:  # the working set from the first pass: a carried start then takes over the set as well
                flags = _engine.FLAG_WORKING_SET
            if again and prev[2] == _engine.FLAG_WORKING_SET and rng.random() < 0.8:
                flags = prev[2]
            mask_of = prev[1] if again else ([int(rng.integers(0, 3)) for _ in range(lanes)] if rng.random() < 0.4 else None)
            if rng.random() < 0.15:
                ynew = y + rng.standard_normal(n) * 0.1
                for ds in (A, B):
                    ds.set_targets(ynew)
                last = None
            specs = []
            for l in range(lanes):
                al = amax * float(rng.uniform(0.02, 0.5))
                a = al * 1.0 / (np.abs(rng.standard_normal(p)) + 0.3) if rng.random() < 0.7 else al * np.ones(p)
                b = al * rng.uniform(0.2, 2.0, Gn) if groups is not None and rng.random() < 0.6 else np.zeros(Gn)
                spec = dict(points=[(1.0, 1.0, 0.0)] * int(rng.integers(1, 3)), a=a, b=b)
                if last is not None and l < len(last) and (again or rng.random() < 0.8):
                    spec["beta0"] = last[l] if (again or rng.random() < 0.85) else last[l] * (1.0 + 1e-9)
                if mask_of is not None:
                    m = masks[mask_of[l]]
                    spec["row_weight"], spec["n_eff"] = (m.copy() if rng.random() < 0.5 else m), int(m.sum())
                specs.append(spec)
            prev = (lanes, mask_of, flags)
            tol = 1e-9
            ra = A.solve_lanes(specs, tol=tol, flags=flags, max_iter=20000)
            os.environ["SLM_NO_CARRY"] = "1"
            rb = B.solve_lanes(specs, tol=tol, flags=flags, max_iter=20000)
            del os.environ["SLM_NO_CARRY"]
            calls += 1
            ka, kb = ra[0].grad_launches, rb[0].grad_launches
            saved += kb - ka
            worse += ka > kb + 1
            if os.environ.get("CARRY_FUZZ_VERBOSE"):
                print(f"  seed {seed} call {step}: lanes={lanes} flags={flags} again={again} masks={mask_of} warm={[('beta0' in sp) for sp in specs]} "
                      f"points={[len(sp['points']) for sp in specs]} passes carried={ka} plain={kb}", flush=True)
            for l, (u, v) in enumerate(zip(ra, rb)):
                assert u.converged == v.converged, (seed, step, l)
                scale = max(np.max(np.abs(v.betas[-1])), 1e-12)
                err = np.max(np.abs(u.betas[-1] - v.betas[-1])) / scale
                assert err < 2e-6, (seed, step, l, err, ka, kb)
            last = [r.betas[-1].copy() for r in ra]
    return calls, saved, worse


if __name__ == "__main__":
    seeds = [int(s) for s in sys.argv[1:]] or list(range(12))
    tot = [0, 0, 0]
    for s in seeds:
        c = run(s)
        tot = [a + b for a, b in zip(tot, c)]
        print(f"seed {s}: {c[0]} calls, {c[1]} passes saved, {c[2]} calls with more than one extra pass", flush=True)
    print(f"total: {tot[0]} calls, {tot[1]} passes saved, {tot[2]} calls more than a pass worse")
    assert tot[1] > 0

#!/usr/bin/env python3
"""Duration of the two halves of the split pass through slm_gradient_ex (route 1): residuals from X for PROBE_LANES lanes
(rowdot_mfma_kernel, or rowdot_ring_kernel with SLM_ROWDOT_RING=1) + X^T R, and X^T R alone."""
import os, sys, subprocess
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    sys.path.insert(0, os.path.join(ROOT, "sparse-lm_amd"))
    from sparselm_amd import _engine
    eng = _engine.get_engine(0)
    n, p = 100000, 5000
    ds = eng.synthetic_dataset(n, p, seed=7, coef=np.zeros(p), noise_sd=1.0)
    ms = min(ds.gradient(np.ones(p), reps=20, split=True, probe_lanes=int(os.environ.get("PROBE_LANES", "1")),
                         xtr_only="XTR_ONLY" in os.environ)[2] for _ in range(3))
    print(f"{ms:.4f}")
    sys.exit(0)
def run(**env):
    e = dict(os.environ, **env)
    out = subprocess.run([sys.executable, __file__, "child"], env=e, capture_output=True, text=True).stdout.strip().splitlines()
    return float(out[-1])
for lanes in ("1", "5", "16"):
    x = run(PROBE_LANES=lanes, XTR_ONLY="1")
    m = run(PROBE_LANES=lanes)
    r = run(PROBE_LANES=lanes, SLM_ROWDOT_RING="1")
    print(f"lanes={lanes:>2}: xtr alone {x:.3f} ms; rowdot_mfma + xtr {m:.3f} ms (rowdot {m - x:.3f}); rowdot_ring + xtr {r:.3f} ms (rowdot {r - x:.3f})", flush=True)

"""BASELINE config 4 from the folds' Grams (SLM_FLAG_COVARIANCE), to be run under `rocprofv3 --kernel-trace`
(tools/share_timeline.py condenses the trace of the last call).  Usage: python tools/config4_cov_trace.py [noise_sd]"""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "sparse-lm_amd")]
import bench  # noqa: E402
from sparselm_amd import _engine  # noqa: E402

noise = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
c4 = bench.Config4(_engine.get_engine(0), 100_000, 5_000, noise_sd=noise)
calls = c4.calls_of(1, 0)
c4.build_covariance()
for _ in range(3):
    sec, passes = c4.run(calls)
    print(f"{len(calls)} calls, {passes} passes, {1e3 * sec:.2f} ms", file=sys.stderr)
c4.run(calls[:1])  # the trace's last solve: the first call (sixteen lanes x 50 points)
c4.close()

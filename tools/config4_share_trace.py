"""One share of BASELINE config 4 as dealt to `world` ranks, solved `reps` times -- to be run under
`rocprofv3 --kernel-trace` (tools/share_timeline.py condenses the trace) or with SLM_TRACE=2 for the host-side marks.
Usage: python tools/config4_share_trace.py [rank] [world] [reps] [only this call of the share]"""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "sparse-lm_amd")]
import bench  # noqa: E402
from sparselm_amd import _engine  # noqa: E402

rank = int(sys.argv[1]) if len(sys.argv) > 1 else 0
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
c4 = bench.Config4(_engine.get_engine(0), 100_000, 5_000)
if os.environ.get("C4_CONTIGUOUS"):  # the folds as KFold(5) without shuffling makes them: row ranges
    import numpy as np
    fold = np.arange(100_000) * 5 // 100_000
    c4.masks = [(fold != f).astype(float) for f in range(5)]
calls = c4.calls_of(world, rank)
if len(sys.argv) > 4:
    calls = calls[int(sys.argv[4]) :][:1]
for _ in range(reps):
    sec, passes = c4.run(calls)
    print(f"rank {rank} of {world}: {len(calls)} call(s), {passes} passes, {1e3 * sec:.2f} ms", file=sys.stderr)
c4.close()

import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/sparse-lm_amd")
import bench
from sparselm_amd import _engine
eng = _engine.get_engine(0)
for s in (2, 3, 4):
    r = bench.leg_concurrent_paths(eng, 0, 0, 100000, 5000, 50, 1e-8, 16, streams=s, steps=10)
    print(s, round(r["fits_per_s_one_stream"]), round(r["fits_per_s_all_streams"]), r["converged"], flush=True)

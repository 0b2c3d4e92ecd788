import os, sys
import numpy as np
sys.path[:0] = ["/root/repo", "/root/repo/sparse-lm_amd"]
from sparselm_amd import _engine
eng = _engine.get_engine(0)
os.environ["SLM_TRACE"] = "1"
with eng.synthetic_dataset(100000, 5000, seed=1, coef=np.zeros(5000), noise_sd=1.0) as ds:
    print(ds.read_ceiling(reps=10))
    g, _, ms = ds.gradient(None, reps=20)
    print("fused B=1 ms", ms, 8*(100000*5008)/ms/1e6, "GB/s")

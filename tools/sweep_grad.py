#!/usr/bin/env python3
"""Tuning sweep of the fused gradient kernel: (W, C, R) x blocks-per-CU at a given (n, p).

Usage: python tools/sweep_grad.py [--n 100000 --p 5000 --reps 20]
Prints one line per configuration: mean kernel ms and algorithmic GB/s = 8(np+2n+2p)/t.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "sparse-lm_amd"))
from sparselm_amd import _engine  # noqa: E402

CONFIGS = {
    5000: ["8,5,2", "8,5,1", "8,5,3", "8,5,4", "8,6,2"],
    10000: ["8,10,1"],
    2048: ["8,2,4", "8,3,4", "8,3,2", "4,4,4", "8,4,2", "8,4,4", "4,5,2", "4,5,4"],
    512: ["4,1,4", "8,1,4", "2,2,4", "4,2,4", "1,2,4", "2,1,4"],
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=100000)
    ap.add_argument("--p", type=int, default=5000)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--blocks", type=str, default="0,1,2")
    ap.add_argument("--lanes", type=str, default="1")
    ap.add_argument("--ring", type=str, default="0", help="comma list of SLM_GRAD_RING values to sweep (0 = register kernel, 1 = LDS ring)")
    args = ap.parse_args()
    eng = _engine.get_engine(0)
    print(eng.device_info(), flush=True)
    n, p = args.n, args.p
    coef = np.zeros(p)
    coef[:10] = 1.0
    z = np.random.default_rng(0).standard_normal(p)
    nbytes = 8.0 * (n * p + 2 * n + 2 * p)
    cfgs = CONFIGS.get(p)
    if cfgs is None:
        cfgs = [None]
    for ring in args.ring.split(","):
        os.environ["SLM_GRAD_RING"] = ring
        for cfg in (cfgs if ring == "0" else ["ring"]):
            for blocks in args.blocks.split(","):
                if cfg is None or cfg == "ring":
                    os.environ.pop("SLM_GRAD_CONFIG", None)
                else:
                    os.environ["SLM_GRAD_CONFIG"] = cfg
                if blocks == "0":
                    os.environ.pop("SLM_GRAD_BLOCKS_PER_CU", None)
                else:
                    os.environ["SLM_GRAD_BLOCKS_PER_CU"] = blocks
                try:
                    ds = eng.synthetic_dataset(n, p, seed=1, coef=coef, noise_sd=1.0)
                except Exception as exc:  # config does not cover p
                    print(f"cfg={cfg} blocks/CU={blocks}: skipped ({exc})", flush=True)
                    continue
                for lanes in args.lanes.split(","):
                    os.environ["SLM_PROBE_LANES"] = lanes
                    try:
                        g, loss, ms = ds.gradient(z, reps=args.reps)
                    except NotImplementedError as exc:
                        print(f"cfg={cfg} lanes={lanes}: unsupported ({exc})", flush=True)
                        continue
                    print(f"cfg={cfg} lanes={lanes} blocks/CU={blocks or 'occ'}: {ms:8.4f} ms  {nbytes / ms / 1e6:8.1f} GB/s  "
                          f"frac_of_8TB/s={nbytes / ms / 1e6 / 8000:.3f}  |g|={np.linalg.norm(g):.6e}", flush=True)
                ds.close()


if __name__ == "__main__":
    main()

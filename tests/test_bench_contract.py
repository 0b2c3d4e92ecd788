"""Host-side pieces of bench.py that need no GPU: the CPU share the CPU legs are sized to, the argument defaults of the
contract (`python bench.py` alone = one GPU, a few steps), the synthetic coefficient law."""

import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _bench():
    spec = importlib.util.spec_from_file_location("_bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_cpu_share_is_the_affinity_mask_cut_to_the_cgroup_quota(tmp_path, monkeypatch):
    bench = _bench()
    share = bench._host_cpu_share()
    assert 1 <= share <= len(os.sched_getaffinity(0))
    assert bench.HOST_CPUS == share
    # a quota of 2.5 CPUs in a cgroup-v2 file: three threads at most
    real_open = open

    def fake_open(path, *a, **k):
        if path == "/sys/fs/cgroup/cpu.max":
            p = tmp_path / "cpu.max"
            p.write_text("250000 100000\n")
            return real_open(p, *a, **k)
        return real_open(path, *a, **k)

    monkeypatch.setattr("builtins.open", fake_open)
    assert bench._host_cpu_share() == min(3, len(os.sched_getaffinity(0)))


def test_defaults_of_the_contract():
    bench = _bench()
    assert bench.PRIMING_PATHS >= 1 and bench.HBM_PEAK_GBS == 8000.0
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert '"--gpus", type=int, default=1' in src and '"--steps", type=int, default=5' in src
    coef = bench.make_coef(5000, 50, seed=0)
    assert coef.shape == (5000,) and np.count_nonzero(coef) == 50
    np.testing.assert_array_equal(coef, bench.make_coef(5000, 50, seed=0))


def test_strong_scaling_object_and_cpu_baseline_layout():
    """The objects the contract line carries beside `value` (round-5 verdict, items 2 and 8): keys a reader may rely on."""
    bench = _bench()
    ss = bench.strong_scaling_object(0.09, 0.027, 8)
    assert {"metric", "scaling", "n_gpus", "unit", "higher_is_better", "value", "seconds_per_grid", "route",
            "value_from_fold_grams", "seconds_per_grid_from_fold_grams", "note"} <= set(ss)
    assert ss["scaling"] == "strong" and ss["n_gpus"] == 8 and abs(ss["value"] - 2500.0 / 0.09) < 1e-9
    assert bench.strong_scaling_object(0.09, None, 1)["value_from_fold_grams"] is None
    port = {"value": 2.4, "unit": "fits/s", "cores": 16, "kind": "port", "sample": "first 49 ...", "beta_rel_inf_err_gpu_vs_oracle": 1e-8,
            "cvxpy": {"status": "cvxpy unavailable on box"}}
    assert bench.stock_first(dict(port), 50, 100000, 5000) == port  # (no scikit-learn figure: the port is the value)
    both = dict(port, sklearn_lasso_path={"value": 18.5, "unit": "fits/s", "cores": 16, "seconds_per_path": 2.7, "what": "lasso_path(...)",
                                          "beta_rel_inf_err_gpu_vs_sklearn": 2e-11})
    cb = bench.stock_first(both, 50, 100000, 5000)
    assert cb["kind"] == "stock" and cb["value"] == 18.5 and cb["port"]["kind"] == "port" and cb["port"]["value"] == 2.4
    assert {"value", "unit", "cores", "kind", "sample"} <= set(cb) and {"value", "unit", "cores", "kind", "sample"} <= set(cb["port"])

"""Host-side pieces of the ctypes binding that need no device."""

import numpy as np

from sparselm_amd import _engine


def _secant_factors_by_loop(pts):
    """The definition, point by point: gamma_k = (s_k - s_{k-1}) / (s_{k-1} - s_{k-2}) for points that are all
    multiples s_k of one penalty direction; zeros otherwise, and where the quotient is not usable."""
    pts = np.asarray(pts, dtype=np.float64).reshape(-1, 3)
    K = pts.shape[0]
    gam = np.zeros(K)
    if K < 3:
        return gam
    ref = pts[np.argmax(np.abs(pts).sum(axis=1))]
    nrm = float(ref @ ref)
    if nrm <= 0.0:
        return gam
    s = pts @ ref / nrm
    if not np.allclose(np.outer(s, ref), pts, rtol=1e-12, atol=1e-300):
        return gam
    for k in range(2, K):
        den = s[k - 1] - s[k - 2]
        if den != 0.0:
            g = (s[k] - s[k - 1]) / den
            if np.isfinite(g) and abs(g) <= 10.0:
                gam[k] = g
    return gam


def test_secant_factors_match_their_definition():
    rng = np.random.default_rng(0)
    al = np.geomspace(1.0, 1e-3, 50)
    cases = [
        np.c_[al, 0 * al, 0 * al],                          # a Lasso path
        np.c_[0.3 * al, 0.7 * al, 0 * al],                  # a sparse-group path: one direction
        np.c_[0 * al, al, 0.5 * al][:7],
        rng.uniform(size=(10, 3)),                          # no common direction: no prediction
        np.array([[1.0, 0, 0], [1.0, 0, 0], [0.5, 0, 0], [0.5, 0, 0], [0.1, 0, 0]]),  # repeated points
        np.array([[1.0, 0, 0], [0.5, 0, 0]]),               # too short
        np.zeros((5, 3)),
        np.array([[1.0, 0, 0], [0.999999, 0, 0], [1e-9, 0, 0], [0, 0, 0]]),          # quotient beyond the cap
    ]
    for pts in cases:
        want = _secant_factors_by_loop(pts)
        np.testing.assert_array_equal(_engine.path_extrapolation(pts), want)
        np.testing.assert_array_equal(_engine.path_extrapolation(pts), want)  # (second call: from the cache)
    # the cached array is not handed out itself
    a = _engine.path_extrapolation(cases[0])
    a[:] = 7.0
    np.testing.assert_array_equal(_engine.path_extrapolation(cases[0]), _secant_factors_by_loop(cases[0]))


def test_points_block_layout():
    pts = np.arange(12, dtype=np.float64).reshape(4, 3)
    blk = _engine._points_block(pts, np.array([0.0, 0.0, 0.5, 0.25]))
    assert blk.shape == (4, 4) and blk.flags.c_contiguous
    np.testing.assert_array_equal(blk[:, :3], pts)
    np.testing.assert_array_equal(blk[:, 3], [0.0, 0.0, 0.5, 0.25])


def test_host_pool_falls_back_to_numpy_memory():
    """No page-locked memory to be had (no device here), a small request, or a forked child: plain numpy arrays."""
    from sparselm_amd import _engine

    pool = _engine._HostPool()
    small = pool.empty(16)
    assert small.shape == (16,) and small.flags.owndata
    big = pool.empty(1 << 16)  # 512 KiB: asks the library, which has no device to lock pages for
    assert big.shape == (1 << 16,) and big.dtype == np.float64
    big[:] = 1.0
    pool.pid = -1  # "another process"
    assert pool.empty(1 << 16).flags.owndata


def test_roofline_traffic_is_quoted_only_for_the_source_it_was_taken_on(tmp_path, monkeypatch):
    """bench.measured_traffic: the committed PMC figure goes into the bench line only for the workload and kernel it was
    taken on and only while the kernel's source file still has the recorded SHA-256 (the GPU box has no .git to ask)."""
    import hashlib
    import json
    import os
    import sys

    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    sys.path.insert(0, root)
    import bench

    recorded = json.load(open(os.path.join(root, "profiles", "roofline_traffic.json")))
    lanes = recorded["workload"]["lanes"]  # (round 5: eighteen -- the engine's choice for the headline's fifty points)
    kernel = recorded["kernel"].split("::")[1].split("(")[0]  # "slm::xtr18_mfma_kernel(slm::SplitArgs)" -> "xtr18_mfma_kernel"
    value, note = bench.measured_traffic(100_000, 5_000, lanes, kernel)
    src = os.path.join(root, recorded["taken_on"]["kernel_source"])
    if hashlib.sha256(open(src, "rb").read()).hexdigest() == recorded["taken_on"]["kernel_source_sha256"]:
        assert value == recorded["hbm_bytes_per_launch"] and "PMC passes of commit" in note
    else:
        assert value is None and "has changed" in note
    assert bench.measured_traffic(100_000, 5_000, 4, kernel)[0] is None  # another workload
    assert bench.measured_traffic(100_000, 5_000, lanes, "grad_fused_kernel")[0] is None  # another kernel
    # a changed source file: the figure is withheld, with the reason
    fake = tmp_path / "repo"
    (fake / "profiles").mkdir(parents=True)
    (fake / "k.hpp").write_text("// edited\n")
    rec = dict(recorded, taken_on=dict(recorded["taken_on"], kernel_source="k.hpp"))
    (fake / "profiles" / "roofline_traffic.json").write_text(json.dumps(rec))
    monkeypatch.setattr(bench, "ROOT", str(fake))
    value, note = bench.measured_traffic(100_000, 5_000, lanes, kernel)
    assert value is None and "has changed" in note


def test_compiled_binding_loads_and_agrees_with_python_on_the_secant_factors():
    """The pybind11 module loads without a GPU (it only links the engine), reports the ABI it was built against, and its
    secant factors of a path -- computed in C++ for the hot calls -- are those of `_engine.path_extrapolation`."""
    import numpy as np

    from sparselm_amd import _engine

    b = _engine.load_binding()
    assert b is not None, "build it: python sparse-lm_amd/build.py"
    assert b.abi_version() == _engine.ABI_VERSION and b.info_record_bytes() == _engine._INFO_DTYPE.itemsize
    if __import__("os").environ.get("SLM_EXPECT_SANITIZED_BINDING"):  # (tools/sanitize.sh: the ASan + UBSan build is the one under test)
        assert b.__file__.endswith("san/_slm_binding.so"), b.__file__
    rng = np.random.default_rng(0)
    for K in (1, 2, 3, 10, 50):
        al = np.geomspace(1, 1e-3, K)
        for pts in (np.c_[al, 0 * al, 0 * al], np.c_[0.3 * al, 0.7 * al, 0 * al], np.c_[al, al[::-1], 0 * al], rng.uniform(size=(K, 3)),
                    np.c_[al, al, al] * np.r_[np.ones(K // 2), np.ones(K - K // 2)][:, None]):
            pts = np.ascontiguousarray(pts)
            np.testing.assert_allclose(b.path_extrapolation(pts), _engine._path_extrapolation(pts), rtol=1e-13, atol=0)
    with __import__("pytest").raises(ValueError):
        b.path_extrapolation(np.zeros(4))


def test_reweight_rules_are_the_estimators_updates_in_the_engines_terms():
    """``_reweight_rule`` hands ``slm_solve_lanes_reweighted`` the default weight update of each Adaptive* estimator: the
    kernel's expression, scale * (numerator / (|x| + eps)), evaluated here in numpy, gives ``_updated_weights`` bit for bit."""
    from sparselm_amd import model

    rng = np.random.default_rng(3)
    n, p, G = 30, 24, 6
    X = rng.standard_normal((n, p))
    groups = np.arange(p) % G
    gw = rng.uniform(0.5, 2.0, G)
    beta = rng.standard_normal(p) * (rng.random(p) < 0.6)
    ests = [
        model.AdaptiveLasso(alpha=0.37, eps=1e-5),
        model.AdaptiveGroupLasso(groups=groups, alpha=0.21, group_weights=gw),
        model.AdaptiveSparseGroupLasso(groups=groups, alpha=0.8, l1_ratio=0.3, group_weights=gw, eps=1e-4),
        model.AdaptiveRidgedGroupLasso(groups=groups, alpha=1.3, delta=(0.5,), group_weights=gw),
    ]
    for est in ests:
        gidx, GG, w0 = est._adaptive_setup(X)
        gn = np.abs(beta) if gidx is None else np.sqrt(np.bincount(gidx, weights=beta**2, minlength=GG))
        coef_scale, group_scale, numer, eps, tol, n_coef, n_group = est._reweight_rule(p, GG)
        assert eps == est.eps and tol == est.tol
        a0, b0, _ = est._weights_to_penalty(w0, p, GG)
        a = np.array(a0, dtype=float)
        b = np.array(b0, dtype=float)
        if coef_scale != 0.0:
            a[:n_coef] = coef_scale * (numer / (np.abs(beta[:n_coef]) + eps))
        if group_scale is not None:
            b[:n_group] = np.asarray(group_scale) * (numer / (gn[:n_group] + eps))
        want_a, want_b, _ = est._weights_to_penalty(est._updated_weights(beta, gn), p, GG)
        np.testing.assert_array_equal(a, want_a)
        np.testing.assert_array_equal(b, want_b)
    custom = model.AdaptiveLasso(update_function=lambda b, eps: 1.0 / (np.abs(b) + eps))
    custom._adaptive_setup(X)
    assert custom._reweight_rule(p, p) is None

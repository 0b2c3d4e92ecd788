"""RCCL across processes, wherever two devices are visible: a child `torch.distributed.run --nproc-per-node 2` (a fresh
process tree: no exec from a process that has touched the GPU) runs tests/_rccl_two_rank_child.py over `slm_comm_init`.
This pool hands out one-GPU boxes, so the test is skipped here; on a node it is the first multi-rank execution of the
`n_ranks > 1` branch of the engine's RCCL path (the in-process communicator covers the state machine everywhere else,
tests/test_row_sharded_gpu.py).  A hung rank ends the test through the child's timeout and a non-zero exit."""

import os
import signal
import socket
import subprocess
import sys

import numpy as np
import pytest

import oracle
from sparselm_amd import _engine

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.mark.parametrize("world", [1, 2])
def test_ranks_over_rccl(tmp_path, world):
    """world = 2: the real thing, where two devices are visible.  world = 1: the same child on a one-rank communicator --
    a rehearsal of everything but the exchange, which runs on this pool's one-GPU boxes."""
    if _engine.device_count() < world:
        pytest.skip("needs two visible devices (RCCL refuses two ranks on one)")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "_rccl_two_rank_child.py"), str(tmp_path)]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, start_new_session=True)
    try:
        log, _ = proc.communicate(timeout=300)
    except subprocess.TimeoutExpired:
        os.killpg(proc.pid, signal.SIGKILL)  # (the group this test started, nothing else)
        log, _ = proc.communicate()
        pytest.fail("the RCCL child did not finish within 300 s:\n" + log[-3000:])
    assert proc.returncode == 0, log[-3000:]
    ranks = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]
    for r, q in enumerate(ranks):
        assert int(q["rccl_ranks"]) == world and int(q["rccl_rank"]) == r
        assert bool(q["lasso_ok"]) and bool(q["group_ok"])
    # bit-identical coefficients and equal collective counts on the two ranks
    for key in ("lasso", "group", "gram", "gram_c"):
        assert np.array_equal(ranks[0][key], ranks[-1][key]), key
    assert int(ranks[0]["collectives"]) == int(ranks[-1]["collectives"])
    assert int(ranks[0]["collectives_row_sharded"]) == int(ranks[-1]["collectives_row_sharded"])
    assert int(ranks[0]["collectives"]) - int(ranks[0]["collectives_row_sharded"]) == 4  # one sum per fold, nothing else
    # ... which are the single-rank answers: the oracle on the whole matrix, numpy's Gram
    n, p = 4096, 320
    X = np.random.default_rng(0).standard_normal((n, p)) + 0.2
    beta = np.zeros(p)
    beta[::17] = 3.0
    y = X @ beta + 0.5 * np.random.default_rng(1).standard_normal(n)
    amax = float(np.max(np.abs(X.T @ y)) / n)
    gidx, G = oracle.group_index(None, p)
    b = None
    for k, a in enumerate(np.geomspace(amax, 0.02 * amax, 8)):
        b, info = oracle.fista(X, y, a, 0.0, 0.0, gidx, G, beta0=b, tol=1e-13)
        assert np.max(np.abs(ranks[0]["lasso"][k] - b)) <= 1e-8 * max(float(np.max(np.abs(b))), 1e-12)
    gidx, G = oracle.group_index(np.arange(p) // 8, p)
    b = None
    for k, a in enumerate(np.geomspace(amax, 0.05 * amax, 6)):
        b, info = oracle.fista(X, y, 0.0, 4.0 * a, 0.0, gidx, G, beta0=b, tol=1e-13)
        assert np.max(np.abs(ranks[0]["group"][k] - b)) <= 1e-8 * max(float(np.max(np.abs(b))), 1e-12)
    folds = np.random.default_rng(2).permutation(n) % 4
    w = (folds != 2).astype(float)
    Ge = X.T @ (w[:, None] * X) / w.sum()
    assert np.max(np.abs(ranks[0]["gram"] - Ge)) <= 1e-12 * np.max(np.abs(Ge))
    assert np.max(np.abs(ranks[0]["gram_c"] - X.T @ (w * y) / w.sum())) <= 1e-12 * np.max(np.abs(X.T @ y)) / n * 10

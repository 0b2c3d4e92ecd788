"""Synthetic grouped regression problems (counterpart of reference src/sparselm/dataset.py:14-139).

Same call signature and return convention as ``sparselm.dataset.make_group_regression``: a
``make_regression`` design whose informative features are concentrated in ``n_informative_groups``
groups, with group labels returned alongside.  Written against the documented behaviour, not the
reference's random stream, so labels for a given seed differ from the reference's.
"""

from __future__ import annotations

import warnings
from collections.abc import Sequence

import numpy as np
from sklearn.utils import check_random_state


def make_group_regression(
    n_samples=100,
    n_groups=20,
    n_features_per_group=10,
    n_informative_groups=5,
    frac_informative_in_group=1.0,
    bias=0.0,
    effective_rank=None,
    tail_strength=0.5,
    noise=0.0,
    shuffle=True,
    coef=False,
    random_state=None,
):
    """Returns ``(X, y, groups[, coef])``: ``groups[j]`` is the group label of feature j."""
    rng = check_random_state(random_state)
    sizes = (
        [int(n_features_per_group)] * n_groups
        if not isinstance(n_features_per_group, Sequence)
        else [int(s) for s in n_features_per_group]
    )
    if len(sizes) != n_groups:
        raise ValueError("n_features_per_group must have one entry per group")
    if not 0 < n_informative_groups <= n_groups:
        raise ValueError("n_informative_groups must be in (0, n_groups]")
    p = int(sum(sizes))
    if effective_rank is None:
        X = rng.standard_normal((n_samples, p))
    else:
        from sklearn.datasets import make_low_rank_matrix

        X = make_low_rank_matrix(n_samples, p, effective_rank=effective_rank, tail_strength=tail_strength,
                                 random_state=rng)
    groups = np.repeat(np.arange(n_groups), sizes)
    beta = np.zeros(p)
    start = np.concatenate(([0], np.cumsum(sizes)))
    # as in the reference (dataset.py:66-95) the informative groups are the first n_informative_groups
    # (labels are shuffled over the columns below), each with round(frac * size) informative features
    counts = [round(frac_informative_in_group * sizes[g]) for g in range(n_informative_groups)]
    if any(k < 1 for k in counts):
        warnings.warn(
            "The number of features and fraction of informative features per group resulted in "
            "informative groups having no informative features.",
            UserWarning,
        )
    for g, k in enumerate(counts):
        if k < 1:
            continue
        idx = start[g] + rng.choice(sizes[g], size=min(k, sizes[g]), replace=False)
        beta[idx] = 100.0 * rng.uniform(size=len(idx))
    y = X @ beta + bias
    if noise > 0.0:
        y = y + rng.normal(scale=noise, size=n_samples)
    if shuffle:
        rows = rng.permutation(n_samples)
        cols = rng.permutation(p)
        X, y = X[rows][:, cols], y[rows]
        groups, beta = groups[cols], beta[cols]
    return (X, y, groups, beta) if coef else (X, y, groups)

"""Input contract of ``groups=`` / ``group_weights=`` for the group estimators.

What must hold, and which exception class reports a violation, is pinned by the reference's tests
(/root/reference/tests/test_lasso.py:203-260; implementation src/sparselm/_utils/validation.py):
a non-sequence is a ``TypeError``; a wrong shape or length is a ``ValueError``; ``None`` is accepted
everywhere (each feature its own group, unit weights).
"""

from __future__ import annotations

import numpy as np

_SEQUENCE_TYPES = (list, np.ndarray)


def _require_sequence(value, what: str) -> np.ndarray:
    if not isinstance(value, _SEQUENCE_TYPES):
        raise TypeError(f"{what} must be a list or ndarray")
    return np.asarray(value)


def check_groups(groups, n_features: int) -> None:
    """One integer-like label per feature, as a flat list/array."""
    if groups is None:
        return
    labels = _require_sequence(groups, "groups").astype(int)
    if labels.ndim != 1:
        raise ValueError("groups must be a 1D array")
    if labels.shape[0] != n_features:
        raise ValueError(f"groups must be the same length as the number of features {n_features}")


def check_group_weights(group_weights, n_groups: int) -> None:
    """One weight per group (in sorted-label order)."""
    if group_weights is None:
        return
    weights = _require_sequence(group_weights, "group_weights")
    if len(weights) != n_groups:
        raise ValueError(
            f"group_weights must be the same length as the number of groups {len(weights)} != {n_groups}"
        )


def dense_group_index(groups, n_features: int):
    """(index per feature in 0..G-1, G): the i-th sorted unique label is group i, the pairing the
    reference uses for ``group_weights`` / ``delta`` (src/sparselm/model/_lasso.py:248)."""
    if groups is None:
        return None, n_features
    uniq, inverse = np.unique(np.asarray(groups), return_inverse=True)
    return inverse.astype(np.int32).reshape(-1), len(uniq)

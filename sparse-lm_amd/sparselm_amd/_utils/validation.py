"""Input contract of ``groups=`` / ``group_weights=``.

Same checks, in the same order, raising the same exception classes as the reference
(src/sparselm/_utils/validation.py:9-35 and :38-59; pinned by tests/test_lasso.py:203-260).
"""

from __future__ import annotations

import numpy as np


def check_groups(groups, n_features: int) -> None:
    """``groups`` must be a list/ndarray (TypeError), 1-D and of length n_features (ValueError)."""
    if groups is None:
        return
    if not isinstance(groups, (list, np.ndarray)):
        raise TypeError("groups must be a list or ndarray")
    arr = np.asarray(groups).astype(int)
    if arr.ndim != 1:
        raise ValueError("groups must be a 1D array")
    if len(arr) != n_features:
        raise ValueError(f"groups must be the same length as the number of features {n_features}")


def check_group_weights(group_weights, n_groups: int) -> None:
    """``group_weights`` must be a list/ndarray (TypeError) with one entry per group (ValueError)."""
    if group_weights is None:
        return
    if not isinstance(group_weights, (list, np.ndarray)):
        raise TypeError("group_weights must be a list or ndarray")
    arr = np.asarray(group_weights)
    if len(arr) != n_groups:
        raise ValueError(
            f"group_weights must be the same length as the number of groups {len(arr)} != {n_groups}"
        )


def dense_group_index(groups, n_features: int):
    """Dense index 0..G-1 per feature in sorted-unique label order (model/_lasso.py:248)."""
    if groups is None:
        return None, n_features
    uniq, inv = np.unique(np.asarray(groups), return_inverse=True)
    return inv.astype(np.int32).reshape(-1), len(uniq)

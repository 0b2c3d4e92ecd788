"""Estimators with the reference's public names (src/sparselm/model/__init__.py:1-43), Lasso family only.

The mixed-integer estimators (BestSubsetSelection, RegularizedL0, L1L0, L2L0, ...) need a
branch-and-bound MIQP solver and are out of scope for a proximal-gradient engine.
"""

from ._adaptive_lasso import (
    AdaptiveGroupLasso,
    AdaptiveLasso,
    AdaptiveOverlapGroupLasso,
    AdaptiveRidgedGroupLasso,
    AdaptiveSparseGroupLasso,
)
from ._lasso import (
    GroupLasso,
    Lasso,
    OrdinaryLeastSquares,
    OverlapGroupLasso,
    RidgedGroupLasso,
    SparseGroupLasso,
)

__all__ = [
    "OrdinaryLeastSquares",
    "Lasso",
    "GroupLasso",
    "OverlapGroupLasso",
    "SparseGroupLasso",
    "RidgedGroupLasso",
    "AdaptiveLasso",
    "AdaptiveGroupLasso",
    "AdaptiveOverlapGroupLasso",
    "AdaptiveSparseGroupLasso",
    "AdaptiveRidgedGroupLasso",
]

"""Lasso-family estimators under the reference's public names.

The reference exports 16 estimators from ``sparselm.model``; the ten below (plus OLS) are the ones a
proximal-gradient engine can serve.  The mixed-integer ones (BestSubsetSelection, RegularizedL0,
L1L0, L2L0, ...) need a branch-and-bound MIQP solver and are out of scope.
"""

from . import _adaptive_lasso as _adaptive
from . import _lasso as _plain

__all__ = list(_plain.__all__) + list(_adaptive.__all__)
globals().update({name: getattr(_plain, name) for name in _plain.__all__})
globals().update({name: getattr(_adaptive, name) for name in _adaptive.__all__})

"""sparselm_amd -- MI355X-native fit engine behind the sparse-lm Lasso-family estimator API.

Drop-in for ``sparselm.model``'s Lasso / GroupLasso / SparseGroupLasso / RidgedGroupLasso and their
Adaptive* variants: same constructor signatures, same ``fit`` / ``predict`` / ``coef_`` surface, same
warnings and error classes; only the solve path differs (a hand-written HIP FISTA engine instead of
cvxpy).  There is no CPU fallback.
"""

__version__ = "0.1.0"

from . import model, tools  # noqa: F401,E402

"""CPU oracle for the sparse-lm Lasso-family fit path.  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (numpy, plus a plain-C twin in ``fista_ref.c``) of what the
reference's cvxpy ``_solve()`` computes for the Lasso family: the minimiser of

    1/(2n) * ||X b - y||^2 + sum_j a_j |b_j| + sum_g b_g ||b_g||_2 + 1/2 sum_g d_g ||b_g||_2^2

together with the host-side semantics wrapped around it (preprocessing, group ordering, the
Adaptive* re-weighting loops).  Every function cites the reference ``file:line`` it follows
(paths relative to ``/root/reference/src/sparselm``).

Who may use it: ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` -- as the checker / reported CPU baseline, never as the product.  Nothing under
``sparse-lm_amd/`` imports this package; the product path fails loudly without its HIP library.

Pinning status (see DESIGN.md "Oracle"):
  * ``cvxpy`` (the third-party dependency that does the reference's arithmetic, unpinned
    ``cvxpy>=1.2`` in the reference's pyproject.toml:15) is NOT installable here, and the
    reference does not import under the installed scikit-learn 1.7.2.  The oracle is therefore
    pinned by (i) the reference's own known-answer tests (``tests/test_lasso.py:29-61``,
    ``tests/test_ols.py:11-66``), (ii) scikit-learn's coordinate-descent ``Lasso`` which
    minimises the identical l1 objective, (iii) KKT optimality certificates for the group /
    sparse-group / ridged penalties, (iv) group labels captured from the reference's importable
    ``sparselm.dataset.make_group_regression``.
  * Numeric coefficients of Group/SparseGroup/Adaptive fits are "parity unpinned" against a
    live cvxpy run: no reference test holds such numbers and cvxpy cannot run here.
"""

from .penalty import (  # noqa: F401
    group_index,
    kkt_residual,
    objective,
    penalty_value,
    prox,
    prox_fixed_point_residual,
)
from .fista import fista, lipschitz  # noqa: F401
from .primal_dual import kkt_standardized, standardized_sparse_group  # noqa: F401
from .estimators import (  # noqa: F401
    fit_adaptive_group_lasso,
    fit_adaptive_lasso,
    fit_adaptive_overlap_group_lasso,
    fit_adaptive_ridged_group_lasso,
    fit_adaptive_sparse_group_lasso,
    fit_group_lasso,
    fit_lasso,
    fit_overlap_group_lasso,
    overlap_extension,
    fit_ridged_group_lasso,
    fit_sparse_group_lasso,
    preprocess,
)

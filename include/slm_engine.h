/*
 * slm_engine.h -- C ABI of the MI355X (gfx950) proximal-gradient fit engine that replaces the
 * cvxpy solve of CederGroupHub/sparse-lm's Lasso-family estimators.
 *
 * Drop-in boundary.  The one reference interface this library stands in for is
 *
 *     CVXRegressor._solve(self, X, y, solver_options) -> beta
 *         src/sparselm/model/_base.py:512-519        (problem.solve(...); return beta.value)
 *     overridden by AdaptiveLasso._solve
 *         src/sparselm/model/_adaptive_lasso.py:206-232  (re-weighting loop around the same solve)
 *
 * called once per fit at src/sparselm/model/_base.py:201 with already validated, centred and
 * re-weighted float64 arrays.  The problem solved is the one the reference assembles in cvxpy:
 *
 *     minimise_beta  1/(2n) ||X beta - y||^2                      (model/_lasso.py:109-121)
 *                  + sum_j a_j |beta_j|                           (Lasso :99-107, SGL :627-639,
 *                                                                  AdaptiveLasso _adaptive_lasso.py:167-175)
 *                  + sum_g b_g ||beta_g||_2                       (GroupLasso :267-275,
 *                                                                  AdaptiveGroupLasso _adaptive_lasso.py:354-362)
 *                  + 1/2 sum_g d_g ||beta_g||_2^2                 (RidgedGroupLasso :795-811)
 *
 * Conventions: plain C types only; every function returns an slm_status (0 = ok); no C++
 * exception crosses the ABI; slm_last_error() returns a thread-local message for the last failure
 * on the calling thread.  All floating-point buffers are IEEE fp64.  A handle is not thread-safe;
 * different handles are.  Host pointers are borrowed only for the duration of a call.
 * There is NO CPU fallback: every entry point needs a visible gfx950 device and fails with
 * SLM_ERR_NO_DEVICE / SLM_ERR_HIP otherwise.
 */
#ifndef SLM_ENGINE_H
#define SLM_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLM_ABI_VERSION 19

typedef enum slm_status {
  SLM_OK = 0,
  SLM_ERR_BAD_ARG = 1,      /* -> ValueError  */
  SLM_ERR_OOM = 2,          /* -> MemoryError */
  SLM_ERR_HIP = 3,          /* -> RuntimeError (HIP runtime / kernel failure) */
  SLM_ERR_NO_DEVICE = 4,    /* -> RuntimeError (no gfx950 device visible) */
  SLM_ERR_COMM = 5,         /* -> RuntimeError (RCCL failure) */
  SLM_ERR_NOT_CONVERGED = 6,/* informational: reported per path point, never returned by solve */
  SLM_ERR_NON_FINITE = 7,   /* -> RuntimeError: non-finite value in the iterate (the reference's
                               counterpart is cvxpy's SolverError / "infeasible" RuntimeError,
                               model/_adaptive_lasso.py:216-220) */
  SLM_ERR_UNSUPPORTED = 8
} slm_status;

typedef struct slm_engine slm_engine;   /* one per (process, device): stream, workspace          */
typedef struct slm_dataset slm_dataset; /* device-resident (X, y[, row weights][, groups]) + state */

/* ---- library ------------------------------------------------------------------------------- */
int slm_abi_version(void);
const char* slm_last_error(void);
int slm_device_count(int* count_out);

/*
 * Page-locked host memory for result buffers (betas_out / group_norms_out of the solve calls).  Any host pointer is
 * accepted there; a buffer from here is written by the copy engine directly, while into ordinary memory the HIP
 * runtime first locks the caller's pages and keeps the registration -- freeing such memory afterwards (numpy
 * releasing a 2 MB result array, what the reference's `coef_` assignment of model/_base.py:201-202 amounts to) then
 * stalls the next submissions for milliseconds.  The binding recycles these blocks (sparselm_amd._engine._HostPool).
 */
int slm_host_alloc(size_t bytes, void** out);
int slm_host_free(void* ptr);

/* ---- engine lifecycle ------------------------------------------------------------------------ */
int slm_engine_create(int device_id, slm_engine** out);
int slm_engine_destroy(slm_engine* eng);
/* Blocks until all work queued on the engine's stream has finished. */
int slm_engine_synchronize(slm_engine* eng);
/* The SLM_* environment variables are read once per process, when the library first needs one.
   This reads them again: for tests and A/B tools that change a variable after the library was loaded. */
int slm_reload_knobs(void);
/* Device facts used by bench.py / DESIGN.md: out[0]=compute units, out[1]=LDS bytes per CU,
   out[2]=total HBM bytes, out[3]=free HBM bytes, out[4]=wavefront size, out[5]=clock kHz. */
int slm_engine_device_info(slm_engine* eng, int64_t out[6], char* name_out, int name_len);

/* ---- dataset --------------------------------------------------------------------------------- */
/*
 * Upload the (already preprocessed) design matrix and target the reference hands to _solve
 * (model/_base.py:201).  X is n x p with element (i, j) at X[i*row_stride + j*col_stride]
 * (strides in elements): C-order (col_stride == 1) and F-order (row_stride == 1) are accepted.
 * The engine keeps its own padded row-major copy in HBM; `row_weight` (nullable, length n, >= 0)
 * multiplies each row's squared residual: loss = 1/(2n) sum_i w_i (x_i beta - y_i)^2  -- used
 * for CV-fold masks and sample weights without re-uploading X.
 */
int slm_dataset_create(slm_engine* eng, const double* X, int64_t n, int64_t p, int64_t row_stride,
                       int64_t col_stride, const double* y, const double* row_weight,
                       slm_dataset** out);
/*
 * Same, from buffers that already live in this device's HBM (row-major, leading dimension ld >= p
 * elements).  The data are copied into the engine's padded layout; the caller keeps ownership.
 */
int slm_dataset_create_device(slm_engine* eng, const double* dX, int64_t n, int64_t p, int64_t ld,
                              const double* dy, const double* d_row_weight, slm_dataset** out);
/*
 * Does the uploaded X hold a value that is not finite?  *kind_out: 0 = all finite, bit 0 = a NaN, bit 1 = an infinity.
 * One read of X on the device (tens of microseconds per gigabyte) -- what lets the estimators' `fit` skip the host-side
 * scan of `check_array` on large arrays (the reference's `_validate_data`, model/_base.py:173: 0.145 s for the 4 GB of a
 * 100 000 x 5 000 design, two thirds of a whole fit here) and still raise scikit-learn's ValueError.
 */
int slm_dataset_nonfinite(slm_dataset* ds, int32_t* kind_out);
/*
 * Synthetic regression problem generated on the device (no host array): X_ij ~ N(0,1) iid from a
 * counter-based generator keyed by (seed, row_offset + i, j); y = X coef + noise_sd * N(0,1).
 * `coef` is a host vector of length p.  This is the law of sklearn.datasets.make_regression used by
 * BASELINE.json's configs; `row_offset` lets rank r of a row-sharded job own rows
 * [row_offset, row_offset + n) of one global matrix.
 */
int slm_dataset_create_synthetic(slm_engine* eng, int64_t n, int64_t p, uint64_t seed,
                                 int64_t row_offset, const double* coef, double noise_sd,
                                 slm_dataset** out);
/*
 * A copy of (X, y, row weights) on another engine of the SAME device, device to device: solves on different
 * engines run side by side (the launches between the passes of one beside the passes of the other), which a grid
 * search with more (fold, parameter) units than GPUs uses (sparselm_amd.model_selection.GridSearchCV(streams=...);
 * the reference's counterpart is n_jobs > 1 of model_selection.py:273).  Group structure is set again by the caller.
 */
int slm_dataset_clone(slm_dataset* src, slm_engine* eng, slm_dataset** out);
int slm_dataset_destroy(slm_dataset* ds);
int slm_dataset_shape(slm_dataset* ds, int64_t* n, int64_t* p, int64_t* ld);
/* Copy the engine's X (dense n x p, C-order) and/or y back to the host (either may be NULL). */
int slm_dataset_download(slm_dataset* ds, double* X_out, double* y_out);
/*
 * Centre the engine's copy of (X, y) in place by their row-weighted means -- what sklearn's
 * _preprocess_data does on the host for fit_intercept=True (reference model/_base.py:216-222) --
 * without a centred host copy of X: one pass for the means, one to subtract (about 2 ms for 4 GB).
 * x_mean_out (length p) and y_mean_out receive the means (either may be NULL); the caller forms
 * intercept_ = y_mean - x_mean . coef_ (sklearn LinearModel._set_intercept).
 */
int slm_dataset_center(slm_dataset* ds, double* x_mean_out, double* y_mean_out);
/* Replace the row weights (NULL => all ones).  Invalidates the cached Lipschitz constant. */
int slm_dataset_set_row_weights(slm_dataset* ds, const double* row_weight);
/*
 * Replace the targets (host vector of length n, finite) and keep X and everything derived from it.  Several
 * problems on one design -- the sub-problems of the splitting behind SparseGroupLasso(standardize=True)
 * (reference model/_lasso.py:616-639 with :249-252), whose targets change between solves -- then share one
 * upload.  A centred dataset expects targets that are already centred.
 */
int slm_dataset_set_targets(slm_dataset* ds, const double* y);
/*
 * Group structure: gid[j] in [0, n_groups) is the dense group index of feature j in the
 * reference's order (i-th sorted unique label <-> group i, model/_lasso.py:248).  Groups need not
 * be contiguous.  NULL / never called => every feature is its own group (model/_lasso.py:211-217).
 */
int slm_dataset_set_groups(slm_dataset* ds, const int32_t* gid, int32_t n_groups);
/* lambda_max(X^T W X)/n estimate (device power iteration, cached); safe upper-side margin applied. */
int slm_dataset_lipschitz(slm_dataset* ds, double* L_out);
/*
 * One evaluation of the hot kernel: g = X^T W (X z - y) / n  (g_out: length p, host) and
 * loss = 1/(2n) sum_i w_i (x_i z - y_i)^2 (loss_out nullable).  z == NULL means z = 0, which gives
 * -X^T W y / n (used for alpha_max).  `reps` > 1 repeats the launch and reports the mean kernel
 * time in ms through ms_out (nullable) -- the roofline probe.
 */
int slm_gradient(slm_dataset* ds, const double* z, double* g_out, double* loss_out, int32_t reps,
                 double* ms_out);
/*
 * The same evaluation by a NAMED route of the engine -- how the parity tests reach the kernels of the split pass through
 * the boundary (until ABI 17 they were switched by the environment variables SLM_GRAD_SPLIT / SLM_GRAD_LANES /
 * SLM_GRAD_LANE / SLM_PROBE_LANES, read inside slm_gradient).  opts == NULL: slm_gradient.
 *   route 0: the fused one-read kernel (two passes beyond 10 240 columns);
 *   route 1: the split pass -- residuals of the lane slots from X (rowdot*_mfma_kernel, or the ring kernel), then X^T R for
 *            all lane slots on one read of X (xtr*_mfma_kernel).  n_lanes (1..SLM_MAX_LANES) lanes all stand at z and the
 *            gradient of lane `lane_out` is returned: a lane of the second half is xtr32's second plane, or the vector
 *            units' share of xtr18 / xtr20.  More than one lane needs the column-major copy (built on first use).
 *            Where the dataset has no split pass the call falls back to route 0.
 *   probe_lanes: lanes of the timed launches behind `reps` / ms_out (<= 0: one); xtr_only: time X^T R alone (route 1).
 * Reference counterpart: the gradient of the smooth term of /root/reference/src/sparselm/model/_lasso.py:109-121.
 */
typedef struct slm_gradient_opts {
  int32_t route;
  int32_t n_lanes;
  int32_t lane_out;
  int32_t probe_lanes;
  int32_t xtr_only;
} slm_gradient_opts;
int slm_gradient_ex(slm_dataset* ds, const double* z, const slm_gradient_opts* opts, double* g_out, double* loss_out,
                    int32_t reps, double* ms_out);

/*
 * Weighted squared error of m coefficient vectors, as many per pass over X as the fused kernel
 * table has lanes for this p:
 *   sse_out[k] = sum_i w_i (x_i . Z[k] - y_i)^2        (Z: m x p, C-order, host)
 * row_weight (length n, host, nullable => the dataset's) is typically the TEST mask of a CV fold, so
 * that hold-out scores come from the resident X instead of a host GEMM (the reference scores with
 * estimator.predict on X[test], model_selection.py:305-315).
 */
int slm_eval_sse(slm_dataset* ds, const double* Z, int32_t m, const double* row_weight, double* sse_out);

/*
 * The same for coefficient vectors that are zero outside n_cols <= 512 columns (the solutions of a
 * regularisation path): cols[n_cols] are those columns, Zs is m x n_cols (C-order, host).  The columns
 * are gathered once on the device and every vector then costs n x n_cols doubles instead of a share of
 * a pass over X.  SLM_ERR_UNSUPPORTED if n_cols > 512 (callers then use slm_eval_sse).
 */
int slm_eval_sse_sparse(slm_dataset* ds, const int32_t* cols, int32_t n_cols, const double* Zs, int32_t m,
                        const double* row_weight, double* sse_out);

/*
 * Diagnostic: x = H^-1 rhs for one dense symmetric positive definite m x m matrix (row-major, host,
 * m <= 512) with the one-workgroup blocked Cholesky the working-set model solver uses for its direct
 * steps (csrc/newton_kernels.hpp); mu_out (nullable) receives its estimate of lambda_min(H).
 * SLM_ERR_BAD_ARG when H is not numerically positive definite.  No reference counterpart (the
 * factorisations live inside the solvers cvxpy calls, model/_base.py:516-518); used by the tests.
 */
int slm_dense_spd_solve(slm_engine* eng, const double* H, int32_t m, const double* rhs, double* x_out,
                        double* mu_out);

/* ---- solve ----------------------------------------------------------------------------------- */
typedef struct slm_penalty {
  const double* a; /* length p, per-coefficient l1 weight; NULL => all ones            */
  const double* b; /* length G, per-group l2 weight;       NULL => all ones            */
  const double* d; /* length G, per-group ridge weight;    NULL => all ones            */
} slm_penalty;

/* Penalty at path point k is (sa*a, sb*b, sd*d).  `extrap` (k >= 2 only; 0 = plain warm start)
   starts point k from beta_{k-1} + extrap * (beta_{k-1} - beta_{k-2}): a secant prediction along
   the path (exact for the piecewise-linear Lasso path between kinks when
   extrap = (alpha_k - alpha_{k-1}) / (alpha_{k-1} - alpha_{k-2})).  It only moves the starting
   point; the solution of point k is unaffected. */
typedef struct slm_path_point {
  double sa, sb, sd, extrap;
} slm_path_point;

#define SLM_FLAG_NO_RESTART 1u   /* disable the gradient-scheme momentum restart        */
#define SLM_FLAG_PROFILE 2u      /* bracket every 2nd gradient launch with HIP events     */
#define SLM_FLAG_COLD_START 4u   /* do not warm-start point k+1 from point k             */
#define SLM_FLAG_FRESH_L 8u      /* re-estimate the Lipschitz constant even if cached     */
#define SLM_FLAG_FISTA_ONLY 16u  /* never use spectral steps (plain FISTA with restart)  */
#define SLM_FLAG_WORKING_SET 32u     /* Gram-assisted working-set refinement between passes even for
                                        small X (default: only when one pass over X costs more than
                                        the refinement, n*ld >= 2^26)                              */
#define SLM_FLAG_NO_WORKING_SET 64u  /* never refine: every iterate comes from a pass over X        */
#define SLM_FLAG_ON_CHIP 128u        /* the caller accepts the on-chip solver for problems whose Gram matrix fits a
                                        workgroup's LDS (p <= 128, n * ld <= 2^17 -- the sizes of the reference's own
                                        tests and README, tests/conftest.py:17-19, README.md:42-55): ONE launch per call,
                                        a workgroup per lane, accelerated proximal steps + conjugate gradients / Anderson
                                        acceleration on X^T W X / n held in LDS (small_kernels.hpp).  Same minimiser and the
                                        same meaning of `tol`; slm_point_info.mode is 2, n_iter counts matrix-vector products,
                                        slm_solve_stats.grad_launches is 1.  Ignored together with the flags that ask for
                                        a particular iteration (1, 2, 16, 32, 64); a point it does not settle hands the
                                        call to the general path.  Up to SLM_MAX_CELLS lanes whatever p.              */

#define SLM_FLAG_COVARIANCE 256u      /* passes take their gradients from the Grams of the call's row sets where
                                        slm_dataset_covariance() has built every one of them: G z - c, 8 p^2 bytes per row
                                        set and pass instead of a read of X (cov_kernels.hpp).  Same iteration, same
                                        results to rounding; ignored where a Gram is missing, on row-sharded datasets and
                                        for calls the on-chip solver takes                                            */

#define SLM_FLAG_NO_MODEL_GRAM 512u   /* lanes whose solutions outgrow the working set's 512 columns finish with plain
                                        steps (two reads of X each) instead of rounds on the model Gram -- for
                                        measurements: the results agree to the tolerance either way              */

#define SLM_FLAG_PROFILE_UNIT 1024u   /* with SLM_FLAG_PROFILE: the bracket of HIP events opens before the residual kernels of
                                        the split pass -- the whole gradient unit (residuals + X^T R), not the launch that
                                        streams X alone (bench.py: roofline.frac)                                   */

typedef struct slm_solve_opts {
  double tol;          /* relative distance to the minimiser a point is accepted at: stop when the KKT
                          residual ||G(z)||_2 (prox-gradient mapping) <= tol * mu * ||beta||_2, mu the
                          strong-convexity estimate of the face (slm_point_info.mu); <= 0 => 1e-8 */
  int32_t max_iter;    /* per path point; <= 0 => 10000                                  */
  int32_t check_every; /* iterations queued between host polls; <= 0 => automatic        */
  double L;            /* Lipschitz constant to use; <= 0 => slm_dataset_lipschitz()     */
  uint32_t flags;
} slm_solve_opts;

typedef struct slm_point_info {
  int32_t n_iter;    /* gradient evaluations spent on this point                         */
  int32_t status;    /* SLM_OK or SLM_ERR_NOT_CONVERGED                                  */
  double resid;      /* ||beta+ - z||_2 at exit                                          */
  double beta_norm;  /* ||beta||_2                                                       */
  double loss;       /* 1/(2n)||X z - y||_W^2 at the last gradient point                 */
  double L;          /* inverse step in use at exit (Lipschitz constant in FISTA mode)   */
  int32_t mode;      /* 1 = spectral (Barzilai-Borwein) steps, 0 = FISTA (after fallback), 2 = the on-chip
                        solver (SLM_FLAG_ON_CHIP)                            */
  int32_t rejects;   /* spectral candidates rejected so far in this solve                */
  double kkt;        /* KKT residual at exit: ||(z - prox_s(z - s grad f(z))) / s||_2 -- zero exactly at
                        the minimiser (the certificate SURVEY 8c-3 names; the reference's solver reports
                        its own through cvxpy's solver_stats, model/_base.py:516-518)               */
  double mu;         /* strong-convexity estimate the point was accepted with: kkt / mu bounds the
                        distance to the minimiser (smallest eigenvalue of the face Hessian from the model
                        solver's Cholesky factor, or the smallest curvature measured along the steps)  */
} slm_point_info;

typedef struct slm_solve_stats {
  int64_t grad_launches;  /* gradient kernels that did work                              */
  int64_t grad_timed;     /* how many of them were bracketed by events (SLM_FLAG_PROFILE) */
  double grad_ms_total;   /* sum of the device durations of the timed ones               */
  double wall_ms;         /* host wall clock of the call                                 */
  double lipschitz_ms;    /* part of wall_ms spent estimating L (0 if cached / given)    */
  int64_t ws_builds;      /* working sets selected and their Grams built from scratch       */
  int64_t ws_appends;     /* times columns were appended to the working set instead        */
  int64_t ws_refined;     /* iterates moved by the working-set refinement, all lanes       */
  int64_t ws_misses;      /* times an iterate left the working set (columns get appended)  */
  int64_t ws_columns;     /* columns in the working set at the end of the solve            */
  int64_t ws_inner_iters; /* proximal-gradient iterations of the model solver, all refinements */
  int64_t ws_direct_steps;/* direct (Cholesky) steps of the model solver: exact minimisation over the
                             face of the iterate, taken when the iteration is slow (ill-conditioned faces) */
  int64_t mg_rounds;      /* (lane, pass) pairs in which a lane beyond the working set took its next point from
                             the model Gram (an fp16 product of the whole of X^T W X / n: mg_kernels.hpp) instead
                             of a plain step; 0: the model Gram was not used                              */
  int64_t mg_inner_iters; /* proximal-gradient iterations on the model Gram, all lanes (one read of the
                             8 p^2-byte matrix each, for sixteen lanes)                                    */
  int64_t mg_rejected;    /* proposals from the model Gram that the true objective rejected              */
  double mg_build_ms;     /* host wall clock spent building the model Gram inside this call (0: it was
                             there already, or not used)                                                  */
  int64_t light_passes;   /* (ABI 19) re-verifications after a miss that were CERTIFIED PARTIAL passes instead of passes over
                             X (csrc/light_kernels.hpp): exact gradient on the working set and on the borderline columns,
                             a Cauchy-Schwarz certificate for the rest; not counted in grad_launches            */
  int64_t light_columns;  /* borderline columns those passes read from the column-major copy, in all */
} slm_solve_stats;

/*
 * Warm-started regularisation path: for k = 0..n_points-1 minimise the objective with penalty
 * (sa_k a, sb_k b, sd_k d), each point started from the previous solution (the first from
 * beta0, NULL => 0).  The whole path runs as one device-resident state machine; the host only
 * queues iterations and polls a flag.  betas_out: n_points x p (C-order, host);
 * group_norms_out: n_points x G (nullable); infos: n_points (nullable); stats nullable.
 * n_points == 1 is the plain _solve().
 *
 * Two things the solves of a dataset do with what the data already told them (working-set solves on a large X; every
 * reported point is verified by a gradient over all rows under the unchanged rule either way):
 *  - carried start: when beta0 of every lane is, bit for bit, the solution the dataset's LAST solve reported for that
 *    lane -- over the same rows and row weights; the penalty may differ: the rounds of an Adaptive* estimator
 *    (model/_adaptive_lasso.py:206-232), a refit -- the solve starts at the point whose gradient the engine still holds
 *    (within the tolerance of beta0) and does not run its first pass over the data.  SLM_NO_CARRY=1 turns it off.
 *  - sample start: a path on several lanes without a warm start opens on the first quarter of the rows -- enough to
 *    rank the features for the first working set, on which the model's linear term is then formed exactly from the
 *    gathered columns; nothing is accepted on the estimate.  SLM_NO_SAMPLE_START=1 opens on all rows.
 */
int slm_solve_path(slm_dataset* ds, const slm_penalty* pen, const slm_path_point* points,
                   int32_t n_points, const slm_solve_opts* opts, const double* beta0,
                   double* betas_out, double* group_norms_out, slm_point_info* infos,
                   slm_solve_stats* stats);

/*
 * Several independent problems ("lanes") on ONE pass over X per iteration: lane l has its own
 * penalty, path, warm start and -- optionally -- its own row weights (a CV-fold mask) and 1/n
 * scaling.  The fused gradient kernel streams X once and produces all lanes' gradients, so B lanes
 * cost the HBM traffic of one.  Used for the alpha sub-paths of one long path, for the folds of a
 * cross-validation, or both.  n_lanes <= SLM_MAX_LANES; SLM_ERR_UNSUPPORTED if no kernel variant
 * covers (p, n_lanes) -- callers then fall back to fewer lanes.  How many lanes a pass can serve:
 * the fused kernels go up to 4 at p = 5000 (6 up to 3072 columns: z and the accumulators of every
 * lane live in registers); on rows of up to 5120 columns working-set solves, and any solve on a large X
 * (n * ld >= 2^26 doubles), use the split pass, whose two halves -- residuals, then X^T R -- run on the
 * matrix cores for sixteen lanes per read of X.  slm_dataset_max_lanes() tells.  Working-set solves on such rows take up
 * to SLM_MAX_LANES = 32 lanes: two halves of sixteen, whose X^T R is ONE read of X for all thirty-two (a row of X in
 * registers is multiplied by both halves' residuals: 0.60-0.71 ms against 0.57 at 100k x 5k); calls of 17 to 20 lanes cost
 * what sixteen cost (lanes 17..20 ride on the vector units beside the sixteen on the matrix cores); covariance passes
 * take thirty-two as well (both halves' points against one read of every Gram); row-sharded solves stay at sixteen.
 */
#define SLM_MAX_LANES 32
/* ... and of a call the on-chip solver takes (SLM_FLAG_ON_CHIP on a dataset of p <= 128, n * ld <= 2^17): a workgroup per
 * lane, so the cells of a small grid search -- (candidate, fold) pairs, 50 in the reference's README example -- go in ONE
 * call.  slm_dataset_max_lanes() tells which of the two limits applies; a point the kernel does not settle sends the call
 * to the general path in chunks of SLM_MAX_LANES. */
#define SLM_MAX_CELLS 64
typedef struct slm_lane {
  const slm_penalty* pen;         /* NULL => all-ones base vectors                              */
  const slm_path_point* points;   /* this lane's warm-started path                              */
  int32_t n_points;
  const double* beta0;            /* length p warm start, NULL => 0                             */
  const double* row_weight;       /* length n, NULL => the dataset's row weights                */
  int64_t n_eff;                  /* 1/n_eff scaling of loss and gradient; <= 0 => dataset's    */
  double* betas_out;              /* n_points x p                                               */
  double* group_norms_out;        /* n_points x G, nullable                                     */
  slm_point_info* infos;          /* n_points, nullable                                         */
} slm_lane;

int slm_solve_lanes(slm_dataset* ds, const slm_lane* lanes, int32_t n_lanes,
                    const slm_solve_opts* opts, slm_solve_stats* stats);
/*
 * The re-weighting loop of the Adaptive* estimators inside ONE call (reference: AdaptiveLasso._solve,
 * src/sparselm/model/_adaptive_lasso.py:206-232 -- solve, renew the penalty weights from the solution, :364-374 for the
 * group norms, stop when the weights no longer move or after max_iter solves; the reference re-enters cvxpy once per
 * round).  Lane l's points are its rounds (n_points = max_iter, normally all (1, 1, 1)): after round k the weights become
 *     a_j <- coef_scale * (numerator / (|beta_j| + eps))            j < n_coef     (coef_scale != 0)
 *     b_g <- group_scale[g] * (numerator / (||beta_g||_2 + eps))    g < n_group    (group_scale != NULL)
 * -- the default update functions of the reference, operation for operation -- and the rounds end once the 2-norm of the
 * change of the weights is <= tol.  betas_out / group_norms_out / infos hold one row per round; rounds that were not run
 * have infos[k].n_iter == 0 and mode == 3, and rounds_out[l] says how many were.  Only for problems the on-chip solver
 * takes (the reference-sized ones): SLM_ERR_UNSUPPORTED otherwise, or when a round was not settled on chip -- the
 * caller then runs the loop itself over slm_solve_lanes, which is what it did before this call existed.
 */
typedef struct slm_reweight {
  double coef_scale;
  const double* group_scale;   /* length n_group, nullable */
  double numerator;
  double eps;
  double tol;
  int32_t n_coef;              /* leading coefficients that carry adaptive weights (an intercept column stays out) */
  int32_t n_group;
} slm_reweight;
int slm_solve_lanes_reweighted(slm_dataset* ds, const slm_lane* lanes, const slm_reweight* rules, int32_t n_lanes,
                               const slm_solve_opts* opts, slm_solve_stats* stats, int32_t* rounds_out);
/* How many lanes one slm_solve_lanes call on this dataset can take with these solve flags (callers
 * size their batches of CV folds / grid rows with it instead of probing for SLM_ERR_UNSUPPORTED). */
int slm_dataset_max_lanes(slm_dataset* ds, uint32_t flags, int32_t* max_lanes_out);
/* The lane count slm_solve_path_lanes takes for a path of n_points when the caller passes n_lanes = 0 (ABI 17: until then
   n_lanes = 0 was clamped to one lane, the strictly sequential warm-started path; callers that want that pass 1). */
int slm_dataset_path_lanes(slm_dataset* ds, int32_t n_points, uint32_t flags, int32_t* lanes_out);

/*
 * ONE warm-started path walked by n_lanes lanes that share each pass over X: the path is cut into
 * n_lanes contiguous ranges (the first warm-started from beta0, the others cold); a lane that runs
 * out of points takes over the upper half of what the busiest lane has left, so the lanes finish
 * together whatever the per-point cost profile is.  Same outputs as slm_solve_path.
 * n_lanes = 0 leaves the count to the engine (ABI 17), which weighs passes against their price: sixteen; eighteen or
 * twenty where that saves a pass of a path over a large X (a 50-point path: three passes instead of four when no point
 * meets a feature outside the working set); up to thirty-two for paths in contiguous ranges (group penalties: 50 points
 * on twenty-five lanes, two passes).  slm_dataset_path_lanes() tells.
 */
int slm_solve_path_lanes(slm_dataset* ds, const slm_penalty* pen, const slm_path_point* points,
                         int32_t n_points, int32_t n_lanes, const slm_solve_opts* opts,
                         const double* beta0, double* betas_out, double* group_norms_out,
                         slm_point_info* infos, slm_solve_stats* stats);

/*
 * SparseGroupLasso(standardize=True) -- the reference's lambda1 ||b||_1 + lambda2 sum_g w_g ||X_g b_g||_2
 * (src/sparselm/model/_lasso.py:616-639 with the standardised group norms of :249-252), which is no proximal penalty
 * on b -- by operator splitting with ALL sweeps in one launch (small_split_kernels.hpp), for problems the on-chip solver
 * takes (p <= 128, n * ld <= 131072, no row weights, not sharded, the groups' factors fitting LDS; otherwise
 * SLM_ERR_UNSUPPORTED and the caller runs the sweeps itself over slm_solve_lanes, sparselm_amd/model/_split.py):
 *   minimise 1/(2n)||X b - y||^2 + sum_j a[j] |b_j| + sum_g b[g] ||X_g b_g||_2 .
 * Stops when the primal residual ||M b - gamma|| and the dual residual are below opts->tol relative to their scales
 * (the rule of _split.py); the b-steps are solved to tol_inner (<= 0: min(tol, 1e-10)).  max_sweeps <= 0: 500.
 * warm != 0 continues from the splitting variables of the previous call on this dataset (the re-weighting loop of the
 * adaptive estimator) instead of rebuilding them from beta0.  info->n_iter = sweeps, info->rejects = matrix-vector
 * products of the b-steps, info->kkt / info->mu = primal / dual residual, info->L = rho at exit, info->status =
 * SLM_OK or SLM_ERR_NOT_CONVERGED (outputs are the last iterate).
 */
int slm_solve_standardized_sgl(slm_dataset* ds, const double* a, const double* b, const slm_solve_opts* opts,
                               double tol_inner, int32_t max_sweeps, const double* beta0, int32_t warm,
                               double* beta_out, double* group_norms_out, slm_point_info* info);

/*
 * The Gram of a row set, for covariance passes (SLM_FLAG_COVARIANCE): G = X^T W X / n_eff, c = X^T W y / n_eff and
 * y^T W y / n_eff, kept with the dataset (8 ld^2 bytes each) and found again by a fingerprint of the row weights and
 * n_eff -- what a lane of slm_solve_lanes brings as (row_weight, n_eff).  row_weight: length n on the host, NULL = the
 * dataset's own; n_eff <= 0 = the dataset's row count (the rule of slm_lane).  A 0/1 mask that leaves out at most half
 * of the rows costs the Gram of the rows left out (the Gram of all rows is built once per dataset and kept); other
 * weights cost a scaled copy of X and a full product.  The product is cov_syrk_kernel (fp64 matrix cores, the lower
 * triangle: 50 ms for all rows at 100 000 x 5 000, 11 ms for a fold's test rows).  Worth it when many solves share the row set: the ten
 * l1_ratio rows x fifty alphas of a CV fold, the rounds of an Adaptive* grid.  Replaces nothing in the
 * reference (cvxpy has no Gram form; scikit-learn's lasso_path(precompute=True) is the same idea on the host).
 * SLM_ERR_UNSUPPORTED on row-sharded datasets and rows beyond the split pass's 10 240 columns.  New targets
 * (slm_dataset_set_targets) and centring (slm_dataset_center) drop the Grams built so far.  At most sixteen Grams are
 * kept per dataset; the oldest goes first.
 */
int slm_dataset_covariance(slm_dataset* ds, const double* row_weight, int64_t n_eff);
/* The folds of a K-fold split at once: count (row_weight, n_eff) pairs.  When the masks' zeros are a partition of the rows
 * -- the test rows of K folds -- the Gram of all rows is the sum of the test rows' Grams and is never formed from X: K
 * products over n / K rows each.  Anything else is built mask by mask, as by slm_dataset_covariance. */
int slm_dataset_covariance_folds(slm_dataset* ds, const double* const* row_weights, const int64_t* n_effs, int32_t count);
/* The same in two steps.  _begin queues the products on the engine's stream and returns (*started_out = 0 and nothing queued
 * when the masks are no such partition: use slm_dataset_covariance then); _finish forms the Grams and files them.  Between the
 * two the caller may queue other work -- and a REPLICA (slm_dataset_set_replicated) on an engine with a communicator enters
 * its collective in _finish only: _begin builds the parts of this rank's n_ranks-th of the rows ([X_f^T X_f | X_f^T y_f |
 * y_f . y_f] of every fold's test rows among them: 1 / n_ranks of the products), _finish sums each part over the ranks (one
 * all-reduce of ld^2 + ld + 16 doubles per fold on a second stream, part f under way while part f + 1 is still being
 * multiplied) -- the grid mode's only collective (SURVEY 8e-1 has none; the reference dispatches whole fits,
 * src/sparselm/model_selection.py:273,304-323).  Every rank must call both with the same masks. */
int slm_dataset_covariance_folds_begin(slm_dataset* ds, const double* const* row_weights, const int64_t* n_effs, int32_t count,
                                       int32_t* started_out);
int slm_dataset_covariance_folds_finish(slm_dataset* ds);
int slm_dataset_covariance_count(slm_dataset* ds, int32_t* count_out);
/* Drops every Gram built so far (and a build under way): the memory goes back, later solves run over X. */
int slm_dataset_covariance_clear(slm_dataset* ds);
/* Diagnostic (tests): Gram `index` (oldest first) to the host -- G_out p x p (C-order), c_out length p (either may be NULL),
 * scalars_out = {y^T W y / n_eff, n_eff, the two fingerprint sums of the row weights}. */
int slm_dataset_covariance_download(slm_dataset* ds, int32_t index, double* G_out, double* c_out, double scalars_out[4]);

/*
 * Measurement: the rate at which THIS device streams the dataset's copy of X through plain 16-byte loads that are only
 * summed up (`reps` sweeps after one warm-up, HIP events) -- the read-only ceiling the passes over X are read against
 * (bench.py `roofline.read_stream_ceiling_gbs`; SURVEY Appendix D asks for it, the reference has no counterpart).
 */
int slm_dataset_read_ceiling(slm_dataset* ds, int32_t reps, double* gbs_out, double* ms_out);

/*
 * The model Gram (csrc/mg_kernels.hpp): G~ ~ X^T W X / n of the dataset's own rows and row weights from ONE product on the
 * fp16 matrix cores (columns scaled by powers of two, fp32 accumulation over chunks of rows, chunks summed in fp64):
 * relative error ~1e-4, built in a few milliseconds where the fp64 Gram of slm_dataset_covariance takes ten times as
 * long.  It only ever PROPOSES points: lanes of a working-set solve whose solutions outgrow the working set's 512 columns
 * take their next point from proximal-gradient steps on it (one read of its 8 p^2 bytes per step for sixteen lanes) and
 * every proposal is verified by a pass over X in fp64 under the unchanged acceptance test and stopping rule -- the
 * reference's solve has no density regime either (src/sparselm/model/_base.py:512-519).  Solves build it themselves when
 * they need it (slm_solve_stats.mg_*; SLM_FLAG_NO_MODEL_GRAM keeps them from it) and it stays with the dataset; this
 * entry point builds it now -- e.g. before the paths of a search -- and, for tests, hands it to the host: G_out ld x ld
 * (C-order, ld = p rounded up to a multiple of 16) or NULL.  SLM_ERR_UNSUPPORTED on row-sharded datasets and for
 * p > 16 384.
 */
int slm_dataset_model_gram(slm_dataset* ds, double* G_out);
/* On an engine with a communicator a dataset is a row block of one tall matrix (the row-sharded mode below) unless it is
 * marked as a replica: every rank then holds ALL rows, solves its own lanes without any per-pass collective (grid mode), and
 * the communicator only carries the folds' Grams (slm_dataset_covariance_folds). */
int slm_dataset_set_replicated(slm_dataset* ds, int32_t replicated);

/* ---- row-sharded mode (very tall X split by rows over ranks) ----------------------------------------
 * Every rank holds a block of rows and runs the whole state machine; per pass the ranks enter TWO all-reduces: the
 * lanes' gradients (ld + 16 doubles each), and a 16-double stop vector -- in working-set solves riding behind the
 * staged Gram parts in one buffer.  slm_dataset_center all-reduces the weighted sums and X^T w, so the global means are
 * subtracted.  HBM per rank: X plus, in working-set solves, its column-major copy (2 x 8 n p bytes). */
#define SLM_COMM_ID_BYTES 128
/* Rank 0 creates the id and distributes the 128 bytes to the other ranks out of band. */
int slm_comm_unique_id(uint8_t id_out[SLM_COMM_ID_BYTES]);
/* Collective over all ranks: afterwards slm_gradient / slm_solve_path on datasets of this engine
   sum X^T r (and the loss / n) over ranks with RCCL; n_global replaces n in the 1/n scaling. */
int slm_comm_init(slm_engine* eng, int32_t rank, int32_t n_ranks,
                  const uint8_t id[SLM_COMM_ID_BYTES]);
/* Rank and size of the engine's communicator as RCCL reports them (1 rank, rank 0 without one). */
int slm_comm_info(slm_engine* eng, int32_t* rank_out, int32_t* n_ranks_out);
/* All-reduces this engine has entered since its communicator was created: ranks of one job must agree
   on it at every quiescent point (the row-sharded state machine takes its stop decision from a word the
   ranks all-reduce after every pass, so that they do whatever their own states say). */
int slm_comm_collectives(slm_engine* eng, int64_t* count_out);
/* Measurement: microseconds per all-reduce of `count` doubles on the engine's communicator (`reps` of them on its stream,
 * after one warm-up, HIP events) -- the per-collective floor of a row-sharded pass (bench.py `rowshard.collective_us`).
 * Every rank of the communicator calls it with the same arguments. */
int slm_comm_all_reduce_probe(slm_engine* eng, int64_t count, int32_t reps, double* us_out);
/*
 * In-process communicator: the n_ranks engines (one process, ONE device, each driven by its own host
 * thread) become the ranks of a row-sharded job; the all-reduce is a rendezvous of their streams with the
 * staged buffers added in rank order.  For single-GPU boxes, where RCCL cannot form a group of several
 * ranks, and for the tests of the multi-rank state machine; a rank that fails to enter a collective within
 * timeout_s (<= 0: 30 s) fails it with SLM_ERR_COMM on the others instead of hanging them.
 */
int slm_comm_init_local(slm_engine** engines, int32_t n_ranks, double timeout_s);
/* n_global replaces n in the 1/n (gradient) and 1/(2n) (loss) scaling: the global row count of a
   row-sharded matrix, or the number of unmasked rows when row weights act as a CV-fold mask. */
int slm_dataset_set_global_rows(slm_dataset* ds, int64_t n_global);
int slm_comm_destroy(slm_engine* eng);

#ifdef __cplusplus
}
#endif
#endif /* SLM_ENGINE_H */

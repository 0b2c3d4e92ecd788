// Host side of the MI355X fit engine, third unit: the device-resident solve loop of a call's lanes -- what replaces the cvxpy
// `problem.solve` call of /root/reference/src/sparselm/model/_base.py:512-519 (and the inner solve of
// model/_adaptive_lasso.py:213-215) -- and the entry points that reach it (slm_solve_path, slm_solve_lanes,
// slm_solve_lanes_reweighted, slm_solve_path_lanes).  Kernel tables, the launches of a pass and the step-size seeds are
// engine_solve.hip's.
#include "engine_internal.hpp"

// ------------------------------------------------------------------------------------------------
// path solves
// ------------------------------------------------------------------------------------------------
// shared_path: the lanes are contiguous, ordered ranges of ONE path (slm_solve_path_lanes): global
// point indices on the device and work stealing between lanes.
// Working-set refinement policy (see solve_core): 0 = never, 1 = when a path point turns out hard
// (small problems), 2 = from the first pass.
static int ws_policy(const slm_dataset* ds, uint32_t flags) {
  // (a group has to fit the working set with room for others: up to half its capacity.  Until round 6 the bound was 64 -- the
  //  width of a wavefront, not a property of the kernels: the model solver's group sums walk a group's members through LDS
  //  whatever their number, the scores and the selection move groups as blocks -- and ONE larger group switched the working
  //  set off for the whole dataset.)
  if (ds->max_group > WS_KCAP / 2 || ds->n < 4) return 0;
  if (knobs().ws == 0 || (flags & SLM_FLAG_NO_WORKING_SET)) return 0;
  const bool big = (double)ds->n * (double)ds->ld >= 67108864.0;  // 2^26 doubles = 512 MiB
  return (big || (flags & SLM_FLAG_WORKING_SET) || knobs().ws == 1) ? 2 : 1;
}

// The on-chip solver (small_kernels.hpp) takes a call when the caller allows it (SLM_FLAG_ON_CHIP), the Gram matrix
// fits the LDS and nothing asks for a particular iteration of the general path.
static bool small_ok(const slm_dataset* ds, uint32_t flags) {
  if (!(flags & SLM_FLAG_ON_CHIP)) return false;
  if (flags & (SLM_FLAG_NO_RESTART | SLM_FLAG_PROFILE | SLM_FLAG_FISTA_ONLY | SLM_FLAG_WORKING_SET | SLM_FLAG_NO_WORKING_SET))
    return false;
  if (!knobs().on_chip) return false;
  return ds->p <= SM_PMAX && (double)ds->n * (double)ds->ld <= 131072.0 && !row_sharded(ds);
}
// most lanes one solve can run: the fused kernels' table, or the split pass's sixteen when the working
// set is on from the start.  Host logic only -- no device call, no allocation: the answer to "how many lanes" must not
// depend on which device is current, and must not queue work (the column-major copy of X that more than sixteen lanes
// need is built by the solve that uses them: lanes_with_copy below).
static bool split_possible(const slm_dataset* ds) {  // (split_usable without building anything)
  return ds->sk != nullptr && (ds->sk->rowdot != nullptr || !ds->XT_failed);
}
static int max_lanes_for(const slm_dataset* ds, uint32_t flags) {
  if (small_ok(ds, flags)) return ds->lane_cap;  // a workgroup per lane
  if ((ws_policy(ds, flags) == 2 || (double)ds->n * (double)ds->ld >= 67108864.0) && split_possible(ds)) {
    // Two halves of sixteen on ONE read of X (xtr32_mfma_kernel, 0.71 ms against 0.57 at 100k x 5k) where the solve is a
    // working-set solve over X on this device: lanes that advance a point per pass -- the units of a grid, the folds of a
    // search -- then cost 0.6 of what they cost on sixteen (config 4 over X: 159 passes / 0.147 s -> 85 / 0.092 s).
    // Covariance passes take thirty-two as well (a launch of the Gram product per half: 13-37 us each against the chain
    // of a whole pass saved).  Row-sharded solves stay at sixteen; so do rows beyond 5120 columns (no ring variant:
    // every residual from X is a read of the column-major copy per half).
    if (ws_policy(ds, flags) == 2 && !row_sharded(ds) && ds->sk->rowdot != nullptr && knobs().wide_lanes &&
        !ds->XT_failed)
      return kMaxLanes;
    return SPLIT_LANES;
  }
  int B = kMaxLanes;
  while (B > 1 && !ds->gk[B - 1]) --B;
  return B;
}
// A lane count beyond sixteen as a solve can really take it: the column-major copy is built here, on the dataset's own
// device, and a dataset that has no memory for it stays at sixteen (XT_failed: max_lanes_for then says so as well).
static int lanes_with_copy(slm_dataset* ds, uint32_t flags, int want, int* lanes_out) {
  *lanes_out = want;
  if (want <= SPLIT_LANES || small_ok(ds, flags)) return SLM_OK;
  HIP_TRY(hipSetDevice(ds->eng->device));
  SLM_TRY(ensure_xt(ds));
  if (ds->XT == nullptr) *lanes_out = SPLIT_LANES;
  return SLM_OK;
}

static int solve_core(slm_dataset* ds, const slm_lane* lanes, int32_t n_lanes, const slm_solve_opts* opts,
                      slm_solve_stats* stats, bool shared_path, const slm_reweight* rules = nullptr, int32_t* rounds_out = nullptr);

// A call the on-chip solver was offered, on the general path: in as many calls as that path needs for the lane count
// (sixteen workgroups take sixteen lanes whatever p; the fused kernels' table stops earlier).
static int solve_without_chip(slm_dataset* ds, const slm_lane* lanes, int32_t B, const slm_solve_opts& o, slm_solve_stats* stats,
                              bool shared_path) {
  slm_solve_opts again = o;
  again.flags &= ~SLM_FLAG_ON_CHIP;
  const int per_call = shared_path ? B : std::min<int>(B, max_lanes_for(ds, again.flags));
  if (per_call >= B) return solve_core(ds, lanes, B, &again, stats, shared_path);
  slm_solve_stats sum, part;
  memset(&sum, 0, sizeof(sum));
  for (int l0 = 0; l0 < B; l0 += per_call) {
    SLM_TRY(solve_core(ds, lanes + l0, std::min(per_call, B - l0), &again, &part, false));
    sum.grad_launches += part.grad_launches;
    sum.wall_ms += part.wall_ms;
    sum.lipschitz_ms += part.lipschitz_ms;
    sum.ws_builds += part.ws_builds; sum.ws_appends += part.ws_appends; sum.ws_refined += part.ws_refined;
    sum.ws_misses += part.ws_misses; sum.ws_columns = std::max(sum.ws_columns, part.ws_columns);
    sum.ws_inner_iters += part.ws_inner_iters; sum.ws_direct_steps += part.ws_direct_steps;
  }
  if (stats) *stats = sum;
  return SLM_OK;
}

// ------------------------------------------------------------------------------------------------
// One call of the solve loop: what replaces `problem.solve()` (/root/reference/src/sparselm/model/_base.py:512-519) for the
// lanes of a call.  The phases run in the order of run(); everything they share lives here.
//   shape()            arguments, route (fused / split pass / covariance passes / on chip), lane count
//   stage_row_weights  per-lane row weights and 1/n scalings, their checksums
//   find_covariance    the Grams of the call's row sets (SLM_FLAG_COVARIANCE)
//   seed_lipschitz     step-size seeds (kept sketch, per-lane bounds, full power iteration)
//   stage_lanes        carried start, control blocks, penalties and warm starts on the device, the tail kernel's arguments
//   run_on_chip        problems that fit a workgroup: one launch for the whole call
//   prepare_working_set, plan_queue   working-set buffers; chunks, expected end, sample start, model-Gram eligibility
//   pass_loop          queue passes (gradient, tail, refinement), poll snapshots, model-Gram rounds
//   finish             results, statistics, trace, the state a later carried start finds
// ------------------------------------------------------------------------------------------------
struct PathCall {
  // ---- arguments
  slm_dataset* ds = nullptr;
  const slm_lane* lanes = nullptr;
  int B = 0;
  slm_solve_opts o;
  slm_solve_stats* stats = nullptr;
  bool shared_path = false;
  const slm_reweight* rules = nullptr;
  int32_t* rounds_out = nullptr;
  // ---- shape and route
  slm_engine* eng = nullptr;
  hipStream_t s = nullptr;
  int64_t n = 0, p = 0, ld = 0;
  int G = 0;
  bool big_x = false, split = false, want_cov = false, interleave = false, sharded = false, small = false, profile = false;
  int64_t total_points = 0;
  bool any_rw = false, any_gn = false, custom_scale = false;
  // ---- clocks (SLM_TRACE)
  std::chrono::steady_clock::time_point t_begin;
  double tr[6] = {0, 0, 0, 0, 0, 0}, tr_rw = 0.0;
  double t_mark() const { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); }
  // ---- row weights, Grams, step sizes
  LaneSetup ls;
  double wmax[SLM_MAX_CELLS];      // largest row weight of each lane (< 0: unknown)
  double rw_fp[SLM_MAX_CELLS][2];  // two checksums of each lane's row weights (carried starts)
  int cov_entry[SLM_MAX_CELLS];
  bool cov_on = false;
  double L[SLM_MAX_CELLS], L_factor[SLM_MAX_CELLS];  // (lane l uses L_factor[l] times the device-resident estimate)
  double lipschitz_ms = 0.0;
  bool L_on_device = false;  // the estimate stays on the device (no host round trip before the first pass)
  bool L_kept = false;       // ... in the dataset's kept slot (lambda[lane_cap]) rather than in lane 0's
  // ---- carried start, control blocks
  bool carry = false, ws_carry = false;
  bool ws_begun = false;  // solve_begin_kernel has started the working set's state with the solve
  int ws_set_of[SLM_MAX_LANES], ws_set_lane[SLM_MAX_LANES], ws_n_sets = 0;
  PathCtl* h = nullptr;  // host staging of the control blocks
  bool infos_in_snap = false;
  slm_point_info* d_infos = nullptr;
  TailArgs ta;
  // ---- working set
  bool use_ws = false, ws_late = false;
  WsArgs wa;
  const int* done_flag = nullptr;
  int ws_comm_rc = 0;        // first RCCL error of the per-pass Gram all-reduce (checked after each chunk)
  bool mg_handover = false;  // the rounds on the model Gram are on: tail points change hands (tail_handover_kernel)
  bool fix_start = false;    // the refinement being queued follows the pass on the row sample (sample start)
  // ---- queue
  int chunk = 0;
  int64_t max_total = 0, enq = 0, expected = 0, n_sample = 0, prof_off = 0;
  int64_t planned_end = 0;  // `expected` as the lanes' walks give it (tracing and tests move `expected` itself: how the host polls)
  int slot = 0, final_slot = 0;
  bool pending[2] = {false, false};
  bool done = false, results_queued = false, results_final = false, deferred = false, trace3 = false;
  // ---- certified partial passes (light_kernels.hpp)
  bool light_begun = false;         // the stamps of this solve have been cleared (first attempt)
  const int* light_skip = nullptr;  // the flag the kernels of the pass being queued return on (LightCtl::ok), or nullptr
  std::vector<char> prof_rec;  // profile slots recorded in THIS solve (a pass that may be skipped on the device is not timed)
  // ---- model Gram
  bool mg_forced = false, mg_ok = false, mg_on = false;
  int mg_cap = 0, mg_inner = 20, mg_built = 0;
  double mg_build_ms = 0.0;
  int mg_entry_of_set[SLM_MAX_LANES];

  int run();
  int shape(const slm_solve_opts* opts);
  int stage_row_weights();
  int find_covariance();
  int kept_sketch();
  int seed_lipschitz();
  int stage_lanes();
  int enqueue_result_copies();
  int run_on_chip();
  int ws_setup(bool late);
  void ws_release();
  int prepare_working_set();
  int enqueue_pass_gradient(hipEvent_t e0, hipEvent_t e1);
  void enqueue_tail();
  void enqueue_refinement();
  void plan_queue();
  int mg_sets();
  bool mg_wanted(const DevCtl& c) const;
  int mg_consider(const DevCtl& c);
  void trace_pass(const DevCtl& now) const;
  bool light_eligible();
  int enqueue_light_attempt();
  int queue_chunk();
  int pass_loop();
  int finish();
  void report(const DevCtl& snap, int64_t passes);
};

static int solve_core(slm_dataset* ds, const slm_lane* lanes, int32_t n_lanes, const slm_solve_opts* opts,
                      slm_solve_stats* stats, bool shared_path, const slm_reweight* rules, int32_t* rounds_out) {
  if (!ds || !lanes) return fail(SLM_ERR_BAD_ARG, "NULL argument");
  HIP_TRY(hipSetDevice(ds->eng->device));  // (before anything that may allocate or launch: split_usable / ensure_xt below)
  // (heap, not stack: the call's state carries two kernels' argument blocks and the lanes' tables)
  std::unique_ptr<PathCall> c(new PathCall());
  c->ds = ds; c->lanes = lanes; c->B = n_lanes; c->stats = stats; c->shared_path = shared_path; c->rules = rules; c->rounds_out = rounds_out;
  memset(&c->o, 0, sizeof(c->o));
  if (opts) c->o = *opts;
  memset(c->cov_entry, 0, sizeof(c->cov_entry));
  memset(c->rw_fp, 0, sizeof(c->rw_fp));
  memset(c->ws_set_of, 0, sizeof(c->ws_set_of));
  memset(c->ws_set_lane, 0, sizeof(c->ws_set_lane));
  memset(c->mg_entry_of_set, 0, sizeof(c->mg_entry_of_set));
  memset(&c->wa, 0, sizeof(c->wa));
  SLM_TRY(c->shape(opts));
  // Uploads from the caller's buffers and from the dataset's staging area are asynchronous: whichever way this
  // function is left, the stream is drained first (on the normal path it already is: a no-op then).
  struct DrainOnExit {
    hipStream_t s;
    ~DrainOnExit() { (void)hipStreamSynchronize(s); }
  } drain_on_exit{c->s};
  return c->run();
}

int PathCall::run() {
  SLM_TRY(stage_row_weights());
  SLM_TRY(find_covariance());
  SLM_TRY(seed_lipschitz());
  SLM_TRY(stage_lanes());
  if (small) return run_on_chip();
  SLM_TRY(prepare_working_set());
  plan_queue();
  SLM_TRY(pass_loop());
  return finish();
}

int PathCall::shape(const slm_solve_opts* opts) {
  if (rules) {
    // re-weighted rounds run inside the on-chip kernel, nowhere else: other problems keep their loop on the caller's side
    if (!rounds_out) return fail(SLM_ERR_BAD_ARG, "rounds_out is NULL");
    if (shared_path || !small_ok(ds, opts ? opts->flags : 0u))
      return fail(SLM_ERR_UNSUPPORTED, "re-weighted rounds need a problem the on-chip solver takes (p <= %d, n * ld <= 131072)", SM_PMAX);
    for (int l = 0; l < B && l < SLM_MAX_CELLS; ++l) {
      const slm_reweight& r = rules[l];
      if (!(r.eps >= 0.0) || !(r.tol >= 0.0) || !std::isfinite(r.coef_scale) || !std::isfinite(r.numerator) || !std::isfinite(r.eps) ||
          r.n_coef < 0 || r.n_coef > ds->p || r.n_group < 0 || r.n_group > ds->G)
        return fail(SLM_ERR_BAD_ARG, "lane %d: bad re-weighting rule", l);
      if (r.group_scale)
        for (int g = 0; g < r.n_group; ++g)
          if (!(r.group_scale[g] >= 0.0) || !std::isfinite(r.group_scale[g]))
            return fail(SLM_ERR_BAD_ARG, "lane %d: group_scale[%d] is negative or not finite", l, g);
    }
  }
  {
    // sixteen lanes; the on-chip solver, a workgroup per lane, takes SLM_MAX_CELLS (what it does not settle comes back here
    // through solve_without_chip in chunks of sixteen)
    const int cap = small_ok(ds, opts ? opts->flags : 0u) ? ds->lane_cap : kMaxLanes;
    if (B < 1 || B > cap) return fail(SLM_ERR_BAD_ARG, "n_lanes must be in [1, %d] (got %d)", cap, B);
  }
  // the split pass costs four launches where the fused kernel costs one: take it when X is large (the
  // accumulate-only stream is then all that matters) or when only it has enough lanes
  // -- the latter also without the working set when X is large: sixteen lanes on the two matrix-core halves
  // (two reads of X per pass) move more problems per byte than four on the fused kernel (one read)
  big_x = (double)ds->n * (double)ds->ld >= 67108864.0;
  // (rows beyond 5120 columns have no ring variant for the residuals that need X: every such residual is a
  //  second full read, of the column-major copy -- the split pass is worth it there only for more lanes than
  //  the fused kernel serves: measured on config 5's shape, one lane, 2.9 ms per pass against 2.4 ms fused)
  const bool wide = ds->sk != nullptr && ds->sk->rowdot == nullptr;
  // (the fused kernels' table stops at SLM_MAX_LANES; calls of more lanes exist on the on-chip route only)
  const GradKernel* gk_B = B <= kMaxLanes ? ds->gk[B - 1] : nullptr;
  // (rows beyond 10 240 columns: the fused table has only the two-pass kernels, two reads of X per gradient -- a working-set
  //  solve takes the split pass there whatever the lane count: one read, and the residuals of points on W from the gathered columns)
  const bool two_pass_only = gk_B != nullptr && gk_B->D < 0;
  const bool want_split = (ws_policy(ds, opts ? opts->flags : 0u) == 2 && (!wide || two_pass_only)) ? (big_x || !gk_B) : (big_x && !gk_B);
  // (covariance passes are a form of the split pass: the flag asks for it whatever the size, where Grams exist)
  want_cov = opts && (opts->flags & SLM_FLAG_COVARIANCE) && !ds->cov.empty() && !row_sharded(ds) && B <= kMaxLanes;
  split = (want_split || want_cov) && split_usable(ds);
  // Shared path with the working set on from the start: the lanes take the points of the path in turn
  // (lane l: l, l + B, ...) instead of contiguous ranges.  Every lane then starts near alpha_max, where
  // the first working set (chosen from the gradient at zero) is enough, and all lanes move down the
  // path together, so W only ever has to cover one band of alphas; a contiguous split starts some lanes
  // cold at small alpha, whose first refinement misses features W could not know about (one extra pass).
  // (Only for per-feature penalties.  With group penalties the cold starts do not miss -- config 3: no
  // miss either way -- while looking a whole stride ahead pulls noise groups into W: 380 columns and
  // 10.9 ms per path against 250 columns and 10.3 ms with contiguous ranges.)
  interleave = shared_path && ds->singleton && ws_policy(ds, opts ? opts->flags : 0u) == 2 &&
                          knobs().interleave;
  if (!split && !gk_B && !small_ok(ds, opts ? opts->flags : 0u))
    return fail(SLM_ERR_UNSUPPORTED, "no %d-lane gradient kernel covers p = %lld", B, (long long)ds->p);
  // more than sixteen lanes: two halves on one read of X (xtr32_mfma_kernel) -- working-set solves on the split pass, all rows here
  if (B > SPLIT_LANES && !small_ok(ds, opts ? opts->flags : 0u) && (!split || row_sharded(ds) || ws_policy(ds, opts ? opts->flags : 0u) != 2))
    return fail(SLM_ERR_UNSUPPORTED, "%d lanes: more than %d need a working-set solve on the split pass of an unsharded dataset", B, SPLIT_LANES);
  if (split && B > ROWDOT_LANES) SLM_TRY(ensure_xt(ds));  // rowdot_mfma_kernel reads the column-major copy (optional)
  for (int l = 0; l < B; ++l) {
    const slm_lane& ln = lanes[l];
    if (!ln.points || !ln.betas_out) return fail(SLM_ERR_BAD_ARG, "lane %d: NULL points or betas_out", l);
    if (ln.n_points <= 0) return fail(SLM_ERR_BAD_ARG, "lane %d: n_points must be positive", l);
    for (int k = 0; k < ln.n_points; ++k) {
      const slm_path_point& q = ln.points[k];
      if (!(q.sa >= 0.0) || !(q.sb >= 0.0) || !(q.sd >= 0.0) || !std::isfinite(q.sa + q.sb + q.sd))
        return fail(SLM_ERR_BAD_ARG, "path point %d has a negative or non-finite scale", k);
      if (!std::isfinite(q.extrap) || std::fabs(q.extrap) > 1e3)
        return fail(SLM_ERR_BAD_ARG, "path point %d has an unreasonable extrapolation factor", k);
    }
    total_points += ln.n_points;
    any_rw = any_rw || ln.row_weight != nullptr;
    any_gn = any_gn || ln.group_norms_out != nullptr;
  }
  eng = ds->eng;
  sharded = row_sharded(ds);  // (a replica on an engine with a communicator -- grid mode -- is not)
  HIP_TRY(hipSetDevice(eng->device));
  s = eng->stream;
  t_begin = std::chrono::steady_clock::now();
  p = ds->p; ld = ds->ld; n = ds->n;
  G = ds->G;
  if (!(o.tol > 0.0)) o.tol = 1e-8;
  if (o.max_iter <= 0) o.max_iter = 10000;
  profile = (o.flags & SLM_FLAG_PROFILE) != 0;

  return SLM_OK;
}

int PathCall::stage_row_weights() {
  // ---- per-lane row weights / scaling -----------------------------------------------------------
  ls = default_lanes(ds, B);
  for (int l = 0; l < kMaxCells; ++l) wmax[l] = ds->rw ? ds->rw_max : 1.0;
  if (any_rw) {
    if (!ds->rw_lanes) SLM_TRY(dalloc(&ds->rw_lanes, (size_t)ds->lane_cap * n));
    for (int l = 0; l < B; ++l) {
      double* dst = ds->rw_lanes + (size_t)l * n;
      if (lanes[l].row_weight) {
        const double* w = lanes[l].row_weight;
        // lanes that bring the same host array (the grid rows of one CV fold) share one check and one upload
        int same = -1;
        for (int m = 0; m < l && same < 0; ++m)
          if (lanes[m].row_weight == w) same = m;
        if (same >= 0) {
          wmax[l] = wmax[same];
          rw_fp[l][0] = rw_fp[same][0];
          rw_fp[l][1] = rw_fp[same][1];
          HIP_TRY(hipMemcpyAsync(dst, ds->rw_lanes + (size_t)same * n, sizeof(double) * n, hipMemcpyDeviceToDevice, s));
          continue;
        }
        double top = 0.0, f0 = 0.0, f1 = 0.0;
        for (int64_t i = 0; i < n; ++i) {
          if (!(w[i] >= 0.0) || !std::isfinite(w[i]))
            return fail(SLM_ERR_BAD_ARG, "lane %d: row_weight[%lld] is negative or not finite", l, (long long)i);
          top = std::max(top, w[i]);
          f0 += w[i];
          f1 += w[i] * (double)(((uint32_t)i * 2654435761u) >> 8);  // (position-dependent; no chain beside the sums')
        }
        wmax[l] = top;
        rw_fp[l][0] = f0;
        rw_fp[l][1] = f1;
        HIP_TRY(hipMemcpyAsync(dst, w, sizeof(double) * n, hipMemcpyHostToDevice, s));
      } else if (ds->rw) {
        HIP_TRY(hipMemcpyAsync(dst, ds->rw, sizeof(double) * n, hipMemcpyDeviceToDevice, s));
      } else {
        hipLaunchKernelGGL(fill_kernel, dim3(256), dim3(256), 0, s, dst, n, 1.0);
      }
    }
    ls.rw = ds->rw_lanes;
    ls.rw_stride = n;
  }
  tr_rw = t_mark();
  for (int l = 0; l < B; ++l)
    if (lanes[l].n_eff > 0) {
      ls.n_eff[l] = (double)lanes[l].n_eff;
      custom_scale = true;
    }

  small = small_ok(ds, o.flags);
  return SLM_OK;
}

int PathCall::find_covariance() {
  // ---- covariance passes: every row set of the call has its Gram (slm_dataset_covariance) --------------------------
  if (want_cov && split && !small) {
    const double* wdev[SLM_MAX_CELLS];
    int uniq_of[SLM_MAX_CELLS], first_lane[SLM_MAX_CELLS], nu = 0;
    for (int l = 0; l < B; ++l) {
      int u = -1;
      for (int m = 0; m < l && u < 0; ++m)
        if (lanes[m].row_weight == lanes[l].row_weight && lanes[m].n_eff == lanes[l].n_eff) u = uniq_of[m];
      if (u < 0) {
        u = nu++;
        first_lane[u] = l;
        wdev[u] = ls.rw ? ls.rw + (int64_t)l * ls.rw_stride : nullptr;
      }
      uniq_of[l] = u;
    }
    double fp[2 * SLM_MAX_LANES];
    SLM_TRY(cov_fingerprints(ds, wdev, nu, fp));
    cov_on = true;
    int entry_of_set[SLM_MAX_LANES];
    for (int u = 0; u < nu && cov_on; ++u) {
      const int l = first_lane[u];
      const double ne = ls.n_eff[l] > 0 ? ls.n_eff[l] : (double)ds->n_global;
      entry_of_set[u] = cov_find(ds, fp[2 * u], fp[2 * u + 1], ne);
      cov_on = entry_of_set[u] >= 0;
    }
    if (cov_on) {
      for (int l = 0; l < B; ++l) cov_entry[l] = entry_of_set[uniq_of[l]];
      if (!ds->cov_Z) SLM_TRY(dalloc(&ds->cov_Z, (size_t)ld * SPLIT_RSTRIDE * SPLIT_HALVES));
    }
  }
  return SLM_OK;
}

// the sketch's estimate for the dataset's own rows and weights, computed once and kept on the device
int PathCall::kept_sketch() {
  if (ds->sketch_valid && !(o.flags & SLM_FLAG_FRESH_L) && knobs().sketch_cache) return SLM_OK;
  SLM_TRY(power_iteration(ds, default_lanes(ds, 1), nullptr, sketch_iters(), sketch_rows(ds->n)));
  HIP_TRY(hipMemcpyAsync(ds->lambda + ds->lane_cap, ds->lambda, sizeof(double), hipMemcpyDeviceToDevice, eng->stream));
  ds->sketch_valid = true;
  return SLM_OK;
}

int PathCall::seed_lipschitz() {
  // ---- Lipschitz constants -----------------------------------------------------------------------
  for (int l = 0; l < kMaxCells; ++l) L_factor[l] = 1.0;
  if (o.L > 0.0 || small) {  // (the on-chip solver bounds its own steps from the Gram matrix)
    for (int l = 0; l < B; ++l) L[l] = o.L > 0.0 ? o.L : 1.0;
  } else {
    const auto t0 = std::chrono::steady_clock::now();
    bool ran = false;
    // working-set solves barely use L (first candidate, fallback steps): a bound from the first thirty-second
    // of the rows, three power steps, costs a sixth of the two full passes
    const bool sketch = ws_policy(ds, o.flags) == 2 && n >= 65536 && knobs().l_sketch;
    if (sketch && !(ds->L_valid && !(o.flags & SLM_FLAG_FRESH_L) && !any_rw && !custom_scale)) {
      const bool per_lane = any_rw || custom_scale;
      bool bounded = per_lane && !sharded;
      for (int l = 0; l < B && bounded; ++l) bounded = wmax[l] > 0.0;
      if (bounded) {
        // Lanes with their own row weights / scaling (CV folds: 0/1 masks with 1/n_train): ONE estimate, of the
        // unweighted operator X_S^T X_S / |S|, and per lane the bound lambda_max(X^T W_l X) / n_l <= max(w_l) n / n_l
        // times it -- 1.25 for the folds of a 5-fold split, whose own lambda_max is that of the whole matrix to a few
        // per cent.  A step-size seed may be long by that much (it only shortens the first candidate step, and the
        // sketch is already long by 2-3 x); what it must not cost is what the per-lane power iteration did: three split
        // passes over the sketch for sixteen lanes, 0.9 ms of stream and a host round trip before every call of a grid.
        // (row-sharded: the largest weight of THIS rank's rows would give every rank its own L -- the lanes' own
        //  estimates, all-reduced like every gradient, stay in use there)
        LaneSetup plain = default_lanes(ds, 1);
        plain.rw = nullptr;
        if (ds->rw) {  // (not the dataset's own operator: not kept)
          SLM_TRY(power_iteration(ds, plain, nullptr, sketch_iters(), sketch_rows(n)));
        } else {
          SLM_TRY(kept_sketch());
          L_kept = true;
        }
        for (int l = 0; l < B; ++l) {
          L[l] = 0.0;
          L_factor[l] = wmax[l] * (double)ds->n_global / (ls.n_eff[l] > 0 ? ls.n_eff[l] : (double)ds->n_global);
        }
        L_on_device = true;
      } else if (!per_lane && !ds->rw) {
        // one operator for all lanes and no row weights that could blank the window: nothing on the host needs
        // the number -- the power steps are queued, seed_step_kernel writes L, the first inverse step and the
        // curvature floor into the control blocks, and the host goes on preparing the solve meanwhile
        // (it used to wait for them: 0.2 ms of idle stream per path)
        // (on a side stream beside the first pass, on vectors of its own, the seed saved nothing: the pass is bound by
        //  the memory system, and the 0.4 GB the three power steps read through it come out of the same budget -- 4.26 ms
        //  per path either way, profiles/r03a_seed_beside_ab.txt)
        // (the estimate belongs to the dataset -- its rows, its weights, nothing of the call: kept on the device beside the
        //  lanes' values, like the bound of the full power iteration is kept on the host (estimate_lipschitz); 45 us of
        //  four launches per solve otherwise)
        SLM_TRY(kept_sketch());
        L_kept = true;
        for (int l = 0; l < B; ++l) L[l] = 0.0;
        L_on_device = true;
      } else {
      SLM_TRY(power_iteration(ds, per_lane ? ls : default_lanes(ds, 1), L, sketch_iters(), sketch_rows(n)));
      // A lane whose row weights vanish on the window (scikit-learn's default cv = unshuffled KFold: the first
      // fold's training mask is zero on the first n / k rows) measured nothing there: all rows, then.
      bool blank = false;
      for (int l = 0; l < (per_lane ? B : 1); ++l) blank = blank || !(L[l] > 0.0);
      if (blank) SLM_TRY(power_iteration(ds, per_lane ? ls : default_lanes(ds, 1), L, kPowerItersSolve));
      if (!per_lane)
        for (int l = 1; l < B; ++l) L[l] = L[0];
      }
      ran = true;
    } else if (any_rw || custom_scale) {
      SLM_TRY(power_iteration(ds, ls, L, kPowerItersSolve));  // lane-specific operators: not cached
      ran = true;
    } else {
      if (o.flags & SLM_FLAG_FRESH_L) ds->L_valid = false;
      ran = !ds->L_valid;
      double L1 = 0.0;
      SLM_TRY(estimate_lipschitz(ds, &L1, kPowerItersSolve));
      for (int l = 0; l < B; ++l) L[l] = L1;
    }
    if (ran)
      lipschitz_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  }

  tr[5] = t_mark();
  return SLM_OK;
}

int PathCall::stage_lanes() {
  // ---- buffers -----------------------------------------------------------------------------------
  if (total_points > ds->cap_points) {
    dfree(ds->pts); dfree(ds->betas_out); dfree(ds->infos);
    ds->cap_points = 0;
    SLM_TRY(dalloc(&ds->pts, total_points));
    SLM_TRY(dalloc(&ds->betas_out, (size_t)total_points * p));
    SLM_TRY(dalloc(&ds->infos, total_points));
    ds->cap_points = total_points;
  }
  if (any_gn && total_points * G > ds->cap_gn) {
    dfree(ds->gn_out);
    ds->cap_gn = 0;
    SLM_TRY(dalloc(&ds->gn_out, (size_t)total_points * G));
    ds->cap_gn = total_points * G;
  }
  // ---- carried start ---------------------------------------------------------------------------------------------
  // A solve that starts where the dataset's last solve ended -- every lane's warm start IS the solution that solve
  // reported for the lane, over the same rows with the same weights -- already has what its first step needs on the
  // device: the tail kernels keep the last point whose gradient they saw (zprev), that gradient (gprev) and its loss.
  // The penalty may differ (the gradient is that of the smooth part): the rounds of an Adaptive* estimator
  // (reference: model/_adaptive_lasso.py:142-178 re-solves the same problem with new weights), the refit of a search,
  // repeated fits with warm_start.  The solve then starts at zprev -- the point the reported solution is one proximal
  // step of converged length away from -- and its first pass over the data is not run: 9 -> 7 passes for BASELINE
  // config 5.  Decided on the host from the caller's arrays alone, so the ranks of a row-sharded solve agree.
  if (ds->carry_valid && !small && !shared_path && B <= ds->carry_lanes && !(o.flags & SLM_FLAG_COLD_START) &&
      knobs().carry) {
    carry = true;
    for (int l = 0; l < B && carry; ++l) {
      const slm_dataset::CarryLane& c = ds->carry_lane[l];
      carry = lanes[l].beta0 != nullptr && lanes[l].n_eff == c.n_eff && (lanes[l].row_weight != nullptr) == c.has_rw &&
              (!c.has_rw || (rw_fp[l][0] == c.fp[0] && rw_fp[l][1] == c.fp[1])) &&
              memcmp(lanes[l].beta0, ds->carry_out.data() + (size_t)l * p, sizeof(double) * (size_t)p) == 0;
    }
  }
  ds->carry_valid = false;  // (this solve rewrites the state; it describes its own end below)
  // the row sets of the lanes' Grams (lanes with the same row weights -- same host pointer: the folds of a CV grid -- and
  // the same 1/n scaling share one), as ws_setup forms them
  {
    const void* rwp[SLM_MAX_LANES];
    int64_t nef[SLM_MAX_LANES];
    const int nl = std::min<int>(B, kMaxLanes);
    for (int l = 0; l < nl; ++l) {
      rwp[l] = lanes[l].row_weight;
      nef[l] = lanes[l].n_eff;
    }
    ws_n_sets = slm_host::row_sets(nl, rwp, nef, ws_set_of, ws_set_lane);
  }
  // a carried start on the same row sets takes over the working set too (ws_ctl_carry_kernel)
  ws_carry = carry && ds->ws_carry_valid && ws_policy(ds, o.flags) == 2 && ws_n_sets == ds->ws_carry_sets &&
                  ds->ws_sets >= ws_n_sets && ds->ws_carry_cov == cov_on && knobs().ws_carry;
  for (int l = 0; l < B && ws_carry; ++l) ws_carry = ws_set_of[l] == ds->ws_carry_set_of[l];
  ds->ws_carry_valid = false;
  if (!ds->h_stage) {  // (page-locked: the set-up kernel reads the control blocks from it)
    hipError_t eh = hipHostMalloc((void**)&ds->h_stage, sizeof(PathCtl) * SLM_MAX_CELLS, hipHostMallocDefault);
    if (eh != hipSuccess) return fail(SLM_ERR_OOM, "hipHostMalloc: %s", hipGetErrorString(eh));
  }
  h = ds->h_stage;  // (lives as long as the dataset: the kernel below reads it asynchronously)
  memset(h, 0, sizeof(PathCtl) * SLM_MAX_CELLS);
  SetupArgs su;
  memset(&su, 0, sizeof(su));
  su.beta = ds->beta; su.z = ds->z; su.zprev = ds->zprev; su.gprev = ds->gprev;
  su.g = ds->g;
  su.carry = carry ? 1 : 0;
  if (carry)
    for (int l = 0; l < B; ++l) su.carry_loss[l] = ds->carry_lane[l].loss;
  su.a0 = ds->a0; su.b0 = ds->b0; su.d0 = ds->d0;
  // (small solves keep their per-point records inside the control block: one blocking copy less at the end)
  infos_in_snap = total_points <= kSnapInfos;
  d_infos = infos_in_snap ? ds->dctl->infos : ds->infos;
  su.infos = reinterpret_cast<unsigned char*>(d_infos);
  su.infos_bytes = (int64_t)(sizeof(slm_point_info) * total_points);
  static_assert(sizeof(slm_point_info) % 8 == 0, "infos are zeroed in 8-byte words");
  su.ld = ld; su.p = p; su.G = G; su.n_lanes = B; su.max_lanes = ds->lane_cap;
  if (!ds->h_vec) {
    hipError_t eh = hipHostMalloc((void**)&ds->h_vec, sizeof(double) * 4 * (size_t)ds->lane_cap * (size_t)ld, hipHostMallocDefault);
    if (eh != hipSuccess) return fail(SLM_ERR_OOM, "hipHostMalloc: %s", hipGetErrorString(eh));
    memset(ds->h_vec, 0, sizeof(double) * 4 * (size_t)ds->lane_cap * (size_t)ld);
  }
  if (total_points > ds->h_pts_cap) {
    if (ds->h_pts) (void)hipHostFree(ds->h_pts);
    ds->h_pts = nullptr;
    ds->h_pts_cap = 0;
    hipError_t eh = hipHostMalloc((void**)&ds->h_pts, sizeof(slm_path_point) * (size_t)total_points, hipHostMallocDefault);
    if (eh != hipSuccess) return fail(SLM_ERR_OOM, "hipHostMalloc: %s", hipGetErrorString(eh));
    ds->h_pts_cap = total_points;
  }
  const size_t cap = (size_t)ds->lane_cap;
  int up_lo[4] = {ds->lane_cap, ds->lane_cap, ds->lane_cap, ds->lane_cap}, up_hi[4] = {-1, -1, -1, -1};  // lanes that bring a, b, d, beta0
  int64_t off = 0;
  bool same_pen = B > 1;
  for (int l = 1; l < B; ++l) same_pen = same_pen && lanes[l].pen == lanes[0].pen;
  for (int l = 0; l < B; ++l) {
    const slm_lane& ln = lanes[l];
    const slm_penalty* pen = ln.pen;
    // what the caller gave is uploaded (lanes that share one penalty -- the ranges of a shared path -- copy lane
    // 0's on the device); everything else is filled by solve_setup_kernel below, in one launch
    const double* src[3] = {pen ? pen->a : nullptr, pen ? pen->b : nullptr, pen ? pen->d : nullptr};
    double* dst[3] = {ds->a0 + (size_t)l * ld, ds->b0 + (size_t)l * ld, ds->d0 + (size_t)l * ld};
    unsigned char* mode[3] = {&su.a_mode[l], &su.b_mode[l], &su.d_mode[l]};
    const int64_t cnt[3] = {p, (int64_t)G, (int64_t)G};
    for (int v = 0; v < 3; ++v) {
      if (!src[v]) *mode[v] = 1;
      else if (same_pen && l > 0) *mode[v] = 2;
      else {
        *mode[v] = 0;
        for (int64_t i = 0; i < cnt[v]; ++i)
          if (!(src[v][i] >= 0.0) || !std::isfinite(src[v][i]))
            return fail(SLM_ERR_BAD_ARG, "penalty weights must be finite and >= 0 (index %lld)", (long long)i);
        memcpy(ds->h_vec + ((size_t)v * cap + l) * ld, src[v], sizeof(double) * cnt[v]);
        up_lo[v] = std::min(up_lo[v], l);
        up_hi[v] = std::max(up_hi[v], l);
      }
    }
    memcpy(ds->h_pts + off, ln.points, sizeof(slm_path_point) * (size_t)ln.n_points);
    if (ln.beta0) {
      for (int64_t j = 0; j < p; ++j)
        if (!std::isfinite(ln.beta0[j])) return fail(SLM_ERR_BAD_ARG, "beta0[%lld] is not finite", (long long)j);
      if (!carry) {  // (a carried start takes the point from the device)
        memcpy(ds->h_vec + ((size_t)3 * cap + l) * ld, ln.beta0, sizeof(double) * p);
        up_lo[3] = std::min(up_lo[3], l);
        up_hi[3] = std::max(up_hi[3], l);
      }
      su.beta_mode[l] = 1;
    }
    h[l].n_points = ln.n_points;
    h[l].max_iter = o.max_iter;
    h[l].t = 1.0;
    h[l].L = L[l];
    h[l].tol = o.tol;
    h[l].flags = o.flags;
    h[l].pt_off = shared_path ? 0 : (int32_t)off;
    h[l].stride = 1;
    h[l].tail_pt = -1;
    if (shared_path && interleave) {  // lane l takes points l, l + B, l + 2B, ... of the whole path
      // The points beyond the last full band (two of a 50-point path on sixteen lanes) go to the LAST lanes -- the
      // ones that have just solved their neighbours -- not to the first, which would reach them from sixteen points
      // up the path: there the features of the last decade of alpha cannot be told yet, the first verification
      // misses and a large append follows (0.33 ms on the headline path).  (host_logic.hpp: interleaved_walk)
      const slm_host::LaneWalk w = slm_host::interleaved_walk(l, B, total_points, knobs().tail_band, knobs().slack_deep);
      h[l].point = w.first;
      h[l].pt_lo = w.first;
      h[l].n_points = w.n_points;
      h[l].stride = w.stride;
      h[l].tail_pt = w.tail_pt;
    } else if (shared_path) {  // global indices: [off, off + n_points)
      h[l].point = (int32_t)off;
      h[l].pt_lo = (int32_t)off;
      h[l].n_points = (int32_t)(off + ln.n_points);
    }
    h[l].zzero = ln.beta0 ? 0 : 1;
    if (rules) {
      const slm_reweight& r = rules[l];
      h[l].rw_coef = r.coef_scale; h[l].rw_numer = r.numerator; h[l].rw_eps = r.eps; h[l].rw_tol = r.tol;
      h[l].rw_ncoef = r.n_coef; h[l].rw_ngroup = r.n_group;
      // (by what the rule covers, not by the value of its scale: a zero scale -- AdaptiveLasso(alpha=0) -- renews the weights
      //  to what they were, the round is counted and the rounds end on `moved <= tol` as the loop of calls does after one)
      h[l].rw_on = (r.n_coef > 0 ? 1 : 0) | ((r.group_scale && r.n_group > 0) ? 2 : 0);
      if (r.group_scale)  // (gscale: the general path's scratch for group factors, free on chip; pageable source: staged by the runtime)
        HIP_TRY(hipMemcpyAsync(ds->gscale + (size_t)l * G, r.group_scale, sizeof(double) * (size_t)r.n_group, hipMemcpyHostToDevice, s));
    }
    h[l].mode = (o.flags & SLM_FLAG_FISTA_ONLY) ? 0 : 1;
    h[l].ak = 1.25 * L[l];  // a slightly short first step; the scheme measures its own curvature after it
    h[l].Lhat = 0.5 * L[l];  // a sure lower bound of lambda_max for the residual scaling
    off += ln.n_points;
  }
  {  // the staged rows, first to last lane that brings any (rows in between are filled by solve_setup_kernel afterwards)
    double* dev[4] = {ds->a0, ds->b0, ds->d0, ds->beta};
    for (int v = 0; v < 4; ++v)
      if (up_hi[v] >= 0)
        HIP_TRY(hipMemcpyAsync(dev[v] + (size_t)up_lo[v] * ld, ds->h_vec + ((size_t)v * cap + up_lo[v]) * ld,
                               sizeof(double) * (size_t)(up_hi[v] - up_lo[v] + 1) * ld, hipMemcpyHostToDevice, s));
  }
  // the rest of the set-up in ONE launch: vectors, path points and control blocks (fetched from the page-locked staging by the
  // kernel itself), the head of the control block cleared -- stop words and working-set counters; a working set taken over
  // from the solve before keeps its block: ws_setup --, the step-size seed (the power steps are still in flight: their result
  // goes into the control blocks on the device) and the working set's first state
  static_assert(offsetof(DevCtl, lane) >= offsetof(DevCtl, ws) + sizeof(WsCtl) && offsetof(DevCtl, g) == 0, "g, ws, lane");
  {
    BeginArgs bg;
    memset(&bg, 0, sizeof(bg));
    bg.h_ctl = h; bg.ctl = ds->ctl; bg.n_lanes = B;
    bg.h_pts = ds->h_pts; bg.pts = ds->pts; bg.n_pts = total_points;
    bg.head = reinterpret_cast<int32_t*>(ds->dctl);
    bg.head_words = (int32_t)((ws_carry ? offsetof(DevCtl, ws) : offsetof(DevCtl, lane)) / sizeof(int32_t));
    if (L_on_device) {
      bg.lambda = L_kept ? ds->lambda + ds->lane_cap : ds->lambda;
      bg.margin = 1.08;
      for (int l = 0; l < kMaxLanes; ++l) bg.factor[l] = L_factor[l];
    }
    // (the working set's state where it starts with the solve: ws_setup then has nothing left to launch)
    ws_begun = !small && !ws_carry && ws_policy(ds, o.flags) == 2;
    if (ws_begun) {
      bg.ws = &ds->dctl->ws;
      bg.max_builds = kWsMaxBuilds;
    }
    hipLaunchKernelGGL(solve_begin_kernel, dim3(128), dim3(256), 0, s, su, bg);
  }
  tr[0] = t_mark();
  // (no wait here: the caller's buffers outlive the call, the control blocks are staged in the dataset, and
  //  everything the host still has to prepare overlaps with the step-size seed running on the device)
  tr[1] = t_mark();

  ta.ctl = ds->ctl;
  ta.gdone = reinterpret_cast<int*>(ds->gctl);
  ta.n_lanes = B;
  ta.done_slot = sharded ? 3 : 0;
  ta.provisional = 0;
  ta.steal = (shared_path && !interleave) ? 1 : 0;  // interleaved lanes are balanced by construction
  ta.pts = ds->pts;
  ta.p = (int)p;
  ta.G = G;
  ta.singleton = ds->singleton;
  ta.team = ds->team;
  ta.beta = ds->beta;
  ta.z = ds->z;
  ta.g = ds->g;
  ta.ld = ld;
  ta.zprev = ds->zprev;
  ta.gprev = ds->gprev;
  ta.gscale = ds->gscale;
  ta.uscratch = ds->u;
  ta.a0 = ds->a0;
  ta.b0 = ds->b0;
  ta.d0 = ds->d0;
  ta.order = ds->order;
  ta.gid = ds->gid;
  ta.gstart = ds->gstart;
  ta.betas_out = ds->betas_out;
  ta.gn_out = any_gn ? ds->gn_out : nullptr;
  ta.infos = d_infos;

  return SLM_OK;
}

// results: lanes whose host buffers follow each other (the ranges of one shared path do) travel in one
// copy -- a device-to-host copy into pageable memory costs ~40 us before the first byte moves.  Queued on the
// solve's stream; the caller waits for it.
int PathCall::enqueue_result_copies() {
  int64_t at = 0;
  for (int l = 0; l < B;) {
    int l1 = l + 1;
    int64_t pts = lanes[l].n_points;
    const bool gn = lanes[l].group_norms_out != nullptr, inf = lanes[l].infos != nullptr;
    while (l1 < B && lanes[l1].betas_out == lanes[l].betas_out + (size_t)pts * p &&
           (lanes[l1].group_norms_out != nullptr) == gn && (lanes[l1].infos != nullptr) == inf &&
           (!gn || lanes[l1].group_norms_out == lanes[l].group_norms_out + (size_t)pts * G) &&
           (!inf || lanes[l1].infos == lanes[l].infos + pts)) {
      pts += lanes[l1].n_points;
      ++l1;
    }
    HIP_TRY(hipMemcpyAsync(lanes[l].betas_out, ds->betas_out + (size_t)at * p, sizeof(double) * (size_t)pts * p,
                           hipMemcpyDeviceToHost, s));
    if (gn)
      HIP_TRY(hipMemcpyAsync(lanes[l].group_norms_out, ds->gn_out + (size_t)at * G, sizeof(double) * (size_t)pts * G,
                             hipMemcpyDeviceToHost, s));
    if (inf && !infos_in_snap)
      HIP_TRY(hipMemcpyAsync(lanes[l].infos, ds->infos + at, sizeof(slm_point_info) * (size_t)pts,
                             hipMemcpyDeviceToHost, s));
    at += pts;
    l = l1;
  }
  return SLM_OK;
}

// ---- problems that fit a workgroup: one launch for the whole call (small_kernels.hpp) -------------------------
int PathCall::run_on_chip() {
  SmallArgs sm;
  memset(&sm, 0, sizeof(sm));
  sm.t = ta;
  sm.X = ds->X; sm.y = ds->y; sm.rw = ls.rw; sm.rw_stride = ls.rw_stride; sm.n = n;
  for (int l = 0; l < kMaxCells; ++l) sm.inv_n[l] = 1.0 / (ls.n_eff[l] > 0 ? ls.n_eff[l] : (double)ds->n_global);
  sm.max_iters = (int)std::min<int64_t>((int64_t)o.max_iter, 1500);  // (products per point; then the general path takes over)
  sm.cold = (o.flags & SLM_FLAG_COLD_START) ? 1 : 0;
  // LDS: the Gram matrix, three vectors, and the rest as the stage of the rows while the matrix is built
  const size_t fixed = sizeof(double) * ((size_t)p * p + 3 * (size_t)p);
  const size_t lds = (size_t)SM_LDS_BYTES;
  sm.stage_doubles = (int)((lds - fixed - 64) / sizeof(double));
  SLM_TRY(allow_big_lds((const void*)small_solve_kernel, eng->device));
  // The coefficients of a call that fits the dataset's pinned stage are stored there by the kernel itself and moved to
  // the caller's arrays after the wait: a copy command into pageable memory is 20-40 us behind a 0.25 ms kernel
  // (SLM_NO_SMALL_STAGE: the copy commands, for comparison).
  bool staged_out = (size_t)total_points * (size_t)p <= kSmallOutDoubles && !any_gn && knobs().small_stage;
  if (staged_out && !ds->h_small_out &&
      hipHostMalloc((void**)&ds->h_small_out, sizeof(double) * kSmallOutDoubles, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    ds->h_small_out = nullptr;
    staged_out = false;
  }
  if (staged_out) sm.t.betas_out = ds->h_small_out;
  hipLaunchKernelGGL(small_solve_kernel, dim3(B), dim3(SM_THREADS), lds, s, sm);
  SLM_TRY(check_launch());
  if (infos_in_snap) HIP_TRY(hipMemcpyAsync(&ds->hctl[0].c, ds->dctl, sizeof(DevCtl), hipMemcpyDeviceToHost, s));
  else HIP_TRY(hipMemcpyAsync(&ds->hctl[0].c, ds->dctl, offsetof(DevCtl, infos), hipMemcpyDeviceToHost, s));
  if (!staged_out) SLM_TRY(enqueue_result_copies());
  else if (!infos_in_snap) {  // (records too many for the snapshot: they travel as before)
    int64_t at_i = 0;
    for (int l = 0; l < B; ++l) {
      if (lanes[l].infos)
        HIP_TRY(hipMemcpyAsync(lanes[l].infos, ds->infos + at_i, sizeof(slm_point_info) * (size_t)lanes[l].n_points, hipMemcpyDeviceToHost, s));
      at_i += lanes[l].n_points;
    }
  }
  HIP_TRY(hipStreamSynchronize(s));
  if (staged_out) {
    int64_t at_o = 0;
    for (int l = 0; l < B; ++l) {
      memcpy(lanes[l].betas_out, ds->h_small_out + (size_t)at_o * p, sizeof(double) * (size_t)lanes[l].n_points * p);
      at_o += lanes[l].n_points;
    }
  }
  const double t_small = t_mark();
  const DevCtl& snap = ds->hctl[0].c;
  bool nonfinite = false, unconverged = false;
  int64_t at = 0, sweeps = 0;
  std::vector<slm_point_info> far_infos;
  if (!infos_in_snap) {  // (large calls: the records were fetched into the lanes' own arrays, or not asked for)
    far_infos.resize((size_t)total_points);
    HIP_TRY(hipMemcpy(far_infos.data(), ds->infos, sizeof(slm_point_info) * (size_t)total_points, hipMemcpyDeviceToHost));
  }
  const slm_point_info* all = infos_in_snap ? snap.infos : far_infos.data();
  for (int l = 0; l < B; ++l) {
    nonfinite = nonfinite || snap.lane[l].nonfinite;
    sweeps += snap.lane[l].iter;
    for (int k = 0; k < lanes[l].n_points; ++k) unconverged = unconverged || all[at + k].status == SLM_ERR_NOT_CONVERGED;
    if (infos_in_snap && lanes[l].infos) memcpy(lanes[l].infos, snap.infos + at, sizeof(slm_point_info) * (size_t)lanes[l].n_points);
    at += lanes[l].n_points;
  }
  if (nonfinite) return fail(SLM_ERR_NON_FINITE, "non-finite iterate (diverged or non-finite data)");
  if (rules) {
    // a round the kernel did not settle ends the lane's rounds there: the caller runs its own loop (over slm_solve_lanes,
    // whose general path takes what the chip gives up) -- nothing half-done is handed back
    if (unconverged) return fail(SLM_ERR_UNSUPPORTED, "a re-weighted round was not settled on chip");
    for (int l = 0; l < B; ++l) {
      // (a rule that covers nothing runs no round: the caller's loop takes the call rather than an index of -1)
      if (snap.lane[l].rounds < 1) return fail(SLM_ERR_UNSUPPORTED, "lane %d: the re-weighting rule covers no weight", l);
      rounds_out[l] = snap.lane[l].rounds;
    }
  }
  if (unconverged && knobs().on_chip_fallback) {  // (SLM_ON_CHIP_NO_FALLBACK: diagnostics -- the on-chip records as they are)
    if (knobs().trace == 2) fprintf(stderr, "[slm] on-chip solve gave a point up after %.3f ms (%lld products): the general path takes the call\n", t_small, (long long)sweeps);
    // the on-chip iteration did not settle some point within its products (an ill-conditioned face): the general
    // path, with its Newton steps, takes the call over from the start
    return solve_without_chip(ds, lanes, B, o, stats, shared_path);
  }
  if (stats) {
    memset(stats, 0, sizeof(*stats));
    stats->grad_launches = 1;  // X is read once, for the Gram matrices
    stats->ws_inner_iters = sweeps;
    stats->wall_ms = t_mark();
  }
  if (knobs().trace == 2) {
      int sl = 0;  // the lane that took longest
      double worst = -1.0;
      for (int l = 0; l < B; ++l) {
        const double t = snap.lane[l].hist[0] + snap.lane[l].hist[1] + snap.lane[l].hist[2] + snap.lane[l].hist[3] + snap.lane[l].hist[4];
        if (t > worst) { worst = t; sl = l; }
      }
      fprintf(stderr, "[slm] on-chip solve: row weights %.3f setup %.3f launched+synced %.3f end %.3f ms, %lld products; slowest lane (%d of %d, %d points, "
              "%d products, %d face solves) in the kernel: Gram %.3f, lambda_max %.3f, proximal steps %.3f, faces %.3f, records %.3f ms\n", tr_rw, tr[0],
              t_small, t_mark(), (long long)sweeps, sl, B, lanes[sl].n_points, snap.lane[sl].iter, (int)snap.lane[sl].n_hist, snap.lane[sl].hist[0],
              snap.lane[sl].hist[1], snap.lane[sl].hist[2], snap.lane[sl].hist[3], snap.lane[sl].hist[4]);
    }
  return SLM_OK;
}

// ---- working-set refinement (ws_kernels.hpp) -----------------------------------------------------
// Worth it when a pass over X costs more than the one-workgroup model solve that replaces several
// of them; row-sharded datasets would need the Gram all-reduced (not built).
// Small problems start with plain steps (their passes cost less than a model solve) and switch the
// refinement on when a path point turns out to be hard (ws_late: more than kWsLateIters passes on
// one point -- ill-conditioned designs, where FISTA needs thousands).
int PathCall::ws_setup(bool late) {
  // lanes with the same row weights (same host pointer: the folds of a CV grid) and the same 1/n
  // scaling share one Gram
  const int* set_of = ws_set_of;
  const int* set_lane = ws_set_lane;
  const int n_sets = ws_n_sets;
  const int ws_nblk_most = (int)std::max<int64_t>(1, std::min<int64_t>(eng->cus, n / 64));  // (2 MiB of partials each)
  // Row sets that share partial Grams (ws_block_owner_kernel): the row blocks that are worked on are one per block plus the few
  // where a set's weights change (two per contiguous fold) -- a handful more than there are CUs is a second round of workgroups
  // for the stragglers, i.e. the time of two Grams for the work of one.  A few blocks fewer, and everything is one round.
  // (whether or not SLM_NO_GRAM_OWNER switches the sharing off: the same blocks, the same sums, bit for bit)
  const bool share_blocks = ls.rw != nullptr && !sharded && n_sets > 1;
  const int ws_nblk = share_blocks ? std::max(1, std::min(ws_nblk_most, eng->cus - 2 * n_sets)) : ws_nblk_most;
  // (each on its own: slm_eval_sse_sparse may already have brought idx and XW in)
  if (!ds->ws_idx) SLM_TRY(dalloc(&ds->ws_idx, WS_KCAP));
  if (!ds->ws_gs) SLM_TRY(dalloc(&ds->ws_gs, WS_KCAP));
  if (!ds->ws_gl) SLM_TRY(dalloc(&ds->ws_gl, WS_KCAP));
  if (!ds->ws_pos) SLM_TRY(dalloc(&ds->ws_pos, (size_t)ld));
  if (!ds->ws_score) SLM_TRY(dalloc(&ds->ws_score, (size_t)ld));
  if (!ds->ws_XW) SLM_TRY(dalloc(&ds->ws_XW, (size_t)n * WS_KCAP));
  if (!ds->ws_nt && knobs().direct) SLM_TRY(dalloc(&ds->ws_nt, (size_t)kMaxLanes * NT_SCRATCH));
  if (ds->ws_sets < n_sets) {
    dfree(ds->ws_part); dfree(ds->ws_G); dfree(ds->ws_Gx);
    ds->ws_sets = 0;
    if (sharded) SLM_TRY(dalloc(&ds->ws_Gx, (size_t)n_sets * WS_KCAP * WS_KCAP + STOP_WORDS));
    SLM_TRY(dalloc(&ds->ws_part, (size_t)ws_nblk_most * n_sets * WS_KCAP * WS_KCAP));  // (ws_nblk_most depends on n only)
    SLM_TRY(dalloc(&ds->ws_G, (size_t)n_sets * WS_KCAP * WS_KCAP));
    ds->ws_sets = n_sets;
  }
  // column-major copy of X (a layout of the data like the padded row-major one: depends on nothing
  // but X, kept for the life of the dataset; 2 ms for 4 GB).  Optional: without the memory for it
  // the gathers read the row-major X, one 64-byte sector per element.
  SLM_TRY(ensure_xt(ds));
  // (initialised on the device: a host-side copy would need the stream drained before its buffer goes away)
  if (late) HIP_TRY(hipMemsetAsync(ds->ws_ctl, 0, sizeof(WsCtl), s));  // (a fresh solve has cleared it already)
  if (ws_carry && !late) hipLaunchKernelGGL(ws_ctl_carry_kernel, dim3(1), dim3(256), 0, s, ds->ws_ctl, kWsMaxBuilds);
  else if (late || !ws_begun) hipLaunchKernelGGL(ws_ctl_init_kernel, dim3(1), dim3(64), 0, s, ds->ws_ctl, kWsMaxBuilds);  // (else: solve_begin_kernel has)
  wa.ws = ds->ws_ctl;
  wa.idx = ds->ws_idx; wa.pos = ds->ws_pos; wa.gs = ds->ws_gs; wa.gl = ds->ws_gl;
  wa.score = ds->ws_score; wa.XW = ds->ws_XW; wa.part = ds->ws_part; wa.Gm = ds->ws_G;
  wa.nt = knobs().direct ? ds->ws_nt : nullptr;
  if (sharded && !ds->ws_Gx) SLM_TRY(dalloc(&ds->ws_Gx, (size_t)ds->ws_sets * WS_KCAP * WS_KCAP + STOP_WORDS));
  wa.Gx = sharded ? ds->ws_Gx : nullptr;  // row-sharded: Gram parts are summed over ranks before use
  wa.X = ds->X; wa.XT = ds->XT; wa.n = n; wa.ld = ld;
  wa.rw = ls.rw; wa.rw_stride = ls.rw_stride;
  for (int l = 0; l < kMaxLanes; ++l) {
    wa.set_of[l] = l < B ? set_of[l] : 0;
    wa.set_lane[l] = l < n_sets ? set_lane[l] : 0;
    const int rep = wa.set_lane[l];
    wa.inv_n[l] = 1.0 / (ls.n_eff[rep] > 0 ? ls.n_eff[rep] : (double)ds->n_global);
  }
  wa.n_sets = n_sets;
  wa.nblk = ws_nblk;
  // row blocks on which a set's row weights are all zeros, or all ones like another set's: ws_block_owner_kernel
  wa.owner = nullptr;
  if (ls.rw != nullptr && !sharded && knobs().gram_owner && ws_nblk <= 512) {
    if (!ds->ws_owner) SLM_TRY(dalloc(&ds->ws_owner, (size_t)kMaxLanes * 512));
    hipLaunchKernelGGL(ws_block_owner_kernel, dim3((unsigned)ws_nblk), dim3(256), 0, s, ls.rw, ls.rw_stride, wa, ds->ws_owner);
    wa.owner = ds->ws_owner;
  }
  // measured on the headline path (tools/ws_sweep.py, 24 combinations within 8 % of each other):
  // theta 0.85 / look-ahead 2 / 16 newcomers per pass / 112 initial columns was the fastest
  // (append 48: interleaved lanes need the next band of the path at once; elsewhere 16 cost a pass now and then)
  const slm_host::Knobs& kn = knobs();
  wa.theta = kn.ws_theta;
  wa.lookahead = kn.ws_lookahead;
  wa.append_max = kn.ws_append;
  // The first selection.  A path that walks down from alpha_max on interleaved lanes starts small: 112 columns (160: the
  // same; 208: 4.0 ms per headline path against 3.65).  Lanes that start cold at an alpha of their own (single fits, the
  // pieces of a grid's paths) have nothing that limits what is active at their first point: up to 256 columns -- a
  // selection cut short is repaired at 48 columns per pass, and every repair is a pass over X (one cold Lasso point with
  // 300 informative features: 3-5 passes from 112 columns, 2-4 from 256; 384 costs a sparse fit at a noise-level alpha
  // a millisecond of Gram and model solves on 330 noise columns, tools/single_fit_big.py).  Groups bring their features
  // in blocks: up to 384 (config 5's cold solve: 5 passes from 256 columns, 2 from 384 -- 13.2 -> 8.3 ms per fit of three
  // solves; configs 3 and 4 the same either way).
  wa.k_init = kn.ws_kinit > 0 ? kn.ws_kinit : (ds->singleton ? (shared_path ? 112 : 256) : 384);
  wa.bb_steps = kn.ws_bb;
  wa.one_solver = kn.ws_one_solver;
  wa.hard_call = kn.hard_callwide;
  wa.power_iters = kn.ws_power_iters;
  wa.miss_factor = kn.ws_miss_factor;
  wa.miss_div = kn.ws_miss_div;
  // (0.5 left a first working set of 56-112 columns to the luck of the bisection: 65 on one draw of the headline's law, 102 on
  //  another -- and the features of the first band's deepest points outside the small ones; 0.75: the soak law's twelve paths 92.6 -> 85.0 ms)
  // (per-feature penalties only: groups come in blocks and start from 384 columns -- config 3's first set 250 -> 300 columns at
  //  0.75, 2.79 -> 3.44 ms per path)
  // (and shared paths only: a single cold fit pays for the larger first set without a band of points to serve with it --
  //  1.45 -> 1.51 ms at 0.3 alpha_max, 2.46 -> 2.78 ms at 0.005, tools/single_fit_big.py)
  wa.fill = kn.ws_fill > 0.0 ? kn.ws_fill : ((ds->singleton && shared_path) ? 0.75 : 0.5);
  return SLM_OK;
}

// no memory for the working-set buffers: the plain iteration still works (unless this solve runs
// more lanes than the fused kernels serve, which only the split pass can do)
void PathCall::ws_release() {
  ds->ws_carry_valid = false;
  dfree(ds->ws_idx); dfree(ds->ws_pos); dfree(ds->ws_gs); dfree(ds->ws_gl);
  dfree(ds->ws_score); dfree(ds->ws_XW); dfree(ds->ws_part); dfree(ds->ws_G); dfree(ds->ws_Gx); dfree(ds->ws_nt);
  ds->ws_sets = 0;
  (void)hipGetLastError();
}

int PathCall::prepare_working_set() {
  {
    const int pol = ws_policy(ds, o.flags);
    use_ws = pol == 2;
    // (row-sharded: the switch would change the collectives of a pass on the strength of one rank's state)
    ws_late = pol == 1 && !sharded;
  }
  if (sharded && !ds->stop_words) SLM_TRY(dalloc(&ds->stop_words, STOP_WORDS));
  if (use_ws) {
    const int rc = ws_setup(false);
    // (row-sharded: a rank that fell back on its own would stop entering the per-pass Gram all-reduce while its peers
    //  still do -- mismatched collectives, which RCCL answers with a hang: no memory for the working set is an error
    //  there, reported by the rank that ran out, and the caller frees memory or passes SLM_FLAG_NO_WORKING_SET on all)
    if (rc == SLM_ERR_OOM && !split && !sharded) {
      ws_release();
      use_ws = false;
    } else if (rc != SLM_OK) {
      return rc;
    }
  }
  done_flag = &ds->gctl->done;
  return SLM_OK;
}

// the gradient of one pass: split pass (sixteen lane slots, residuals from the gathered columns where
// possible) when the working set runs from the start, the fused kernel otherwise
int PathCall::enqueue_pass_gradient(hipEvent_t e0, hipEvent_t e1) {
  if (cov_on) return enqueue_gradient_cov(ds, B, cov_entry, done_flag, e0, e1, ds->ctl, use_ws ? &wa : nullptr);
  if (split) return enqueue_gradient_split(ds, ls, ds->y, done_flag, ds->ctl, &wa, e0, e1, 0, (o.flags & SLM_FLAG_PROFILE_UNIT) != 0, light_skip);
  return enqueue_gradient(ds, ls, ds->y, done_flag, e0, e1, 0, light_skip);
}

// everything that follows the gradient of one pass
// (in two halves: behind the pass a solve is expected to end with, the second half waits for the verdict)
void PathCall::enqueue_tail() {
  launch_tail(ta, s);
  if (shared_path && !interleave) hipLaunchKernelGGL(steal_kernel, dim3(1), dim3(256), 0, s, ta);
  // (the dense end of an interleaved path: finished lanes take over tail points their owners have not started)
  if (shared_path && interleave && mg_handover) hipLaunchKernelGGL(tail_handover_kernel, dim3(1), dim3(64), 0, s, ta);
  // (the sparse end: a lane that has fallen a pass behind gives its tail point to a lane that has finished)
  else if (shared_path && interleave && use_ws && !ws_late && !sharded && planned_end > 0 && knobs().lag_handover)
    hipLaunchKernelGGL(lag_handover_kernel, dim3(1), dim3(64), 0, s, ta, &ds->ws_ctl->builds, &ds->ws_ctl->stale, (int)(planned_end - enq - 1));
  if (sharded) {  // the ranks agree on "finished" before anything acts on it
    if (use_ws && wa.Gx) {
      // working-set solves: the stop words ride behind the staged Gram parts, in the one all-reduce of the refinement
      // (enqueue_refinement) -- two collectives per pass, not three.  Until then this pass's kernels see the flag of
      // the pass before, which is what they would see on a rank that has not finished.
      hipLaunchKernelGGL(stop_pack_kernel, dim3(1), dim3(64), 0, s, ta.gdone, ds->ctl, B,
                         wa.Gx + (size_t)wa.n_sets * WS_KCAP * WS_KCAP);
    } else {
      hipLaunchKernelGGL(stop_pack_kernel, dim3(1), dim3(64), 0, s, ta.gdone, ds->ctl, B, ds->stop_words);
      if (ws_comm_rc == 0) ws_comm_rc = all_reduce_sum(eng, ds->stop_words, STOP_WORDS);
      hipLaunchKernelGGL(stop_apply_kernel, dim3(1), dim3(64), 0, s, ta.gdone, ds->stop_words);
    }
  }
}

void PathCall::enqueue_refinement() {
  if (use_ws) {
    {
      const int bs = ds->singleton ? 256 : 64;
      const int64_t items = ds->singleton ? p : 16 * (int64_t)G;  // groups: one thread per (group, lane)
      hipLaunchKernelGGL(ws_score_kernel, dim3((unsigned)((items + bs - 1) / bs)), dim3(bs), 0, s, ta, wa);
      hipLaunchKernelGGL(ws_select_kernel, dim3(1), dim3(WS_THREADS), 0, s, ta, wa);
    }
    if (cov_on) {
      // covariance passes: the working set's Gram is a sub-matrix of the row set's (no gathered columns, no product
      // over the rows; nothing reads XW in this mode -- the residuals of a pass are not formed at all)
      CovSets cs;
      for (int st = 0; st < kMaxLanes; ++st) cs.G[st] = st < wa.n_sets ? ds->cov[(size_t)cov_entry[wa.set_lane[st]]].G : nullptr;
      hipLaunchKernelGGL(ws_gram_cov_kernel, dim3(WS_TILES * WS_TILES, (unsigned)wa.n_sets), dim3(256), 0, s, wa, cs);
    } else {
    hipLaunchKernelGGL(ws_gather_kernel, dim3((unsigned)std::min<int64_t>((n + 31) / 32, 1024), WS_KCAP / 32), dim3(256), 0, s, wa);
    if (fix_start) {  // the exact gradient at zero on W, from the gathered columns (ws_kernels.hpp (ii-b))
      XtyArgs xa;
      xa.ws = wa.ws; xa.idx = wa.idx; xa.XW = wa.XW; xa.y = ds->y; xa.part = ds->partial; xa.g = ds->g; xa.gprev = ds->gprev; xa.z = ds->z;
      xa.ctl = ds->ctl; xa.done = done_flag; xa.n = n; xa.ld = ld; xa.inv_n = 1.0 / (double)ds->n_global; xa.n_lanes = B;
      xa.nblk = (int)std::max<int64_t>(1, std::min<int64_t>(2 * eng->cus, n / 64));
      hipLaunchKernelGGL(ws_xty_partial_kernel, dim3((unsigned)xa.nblk), dim3(512), 0, s, xa);
      hipLaunchKernelGGL(ws_xty_apply_kernel, dim3(WS_KCAP / 128), dim3(512), 0, s, xa);
    }
    hipLaunchKernelGGL(ws_gram_kernel, dim3((unsigned)wa.nblk, (unsigned)wa.n_sets, 1), dim3(WS_GRAM_THREADS), 0, s,
                       wa);
    if (wa.Gx)  // (zero where this pass builds nothing, so the unconditional all-reduce below is harmless)
      (void)hipMemsetAsync(wa.Gx, 0, sizeof(double) * (size_t)wa.n_sets * WS_KCAP * WS_KCAP, s);
    hipLaunchKernelGGL(ws_gram_reduce_kernel, dim3(WS_TILES * WS_TILES, (unsigned)wa.n_sets), dim3(256),
                       0, s, wa);
    }
    if (wa.Gx) {
      // one collective per pass on every rank whether or not a build is under way: the ranks run the
      // same state machine on the same all-reduced gradients, so they agree on when that is
      const size_t gram_words = (size_t)wa.n_sets * WS_KCAP * WS_KCAP;
      if (ws_comm_rc == 0) ws_comm_rc = all_reduce_sum(eng, wa.Gx, gram_words + STOP_WORDS);  // (+ the stop words: enqueue_tail)
      hipLaunchKernelGGL(ws_publish_kernel, dim3(WS_PUBLISH_BLOCKS, (unsigned)wa.n_sets), dim3(256), 0, s, wa);
      hipLaunchKernelGGL(stop_apply_kernel, dim3(1), dim3(64), 0, s, ta.gdone, wa.Gx + gram_words);
    }
    // the iteration alone, then -- for the lanes it left -- the solver with direct steps (ws_refine_lane)
    {
      if (wa.one_solver && wa.nt) {
      } else if (ds->singleton) hipLaunchKernelGGL((ws_solve_kernel<false, 0>), dim3(B), dim3(WS_THREADS), 0, s, ta, wa);
      else hipLaunchKernelGGL((ws_solve_kernel<true, 0>), dim3(B), dim3(WS_THREADS), 0, s, ta, wa);
      if (wa.nt) {
        if (ds->singleton) hipLaunchKernelGGL((ws_solve_kernel<false, 1>), dim3(B), dim3(WS_THREADS), 0, s, ta, wa);
        else hipLaunchKernelGGL((ws_solve_kernel<true, 1>), dim3(B), dim3(WS_THREADS), 0, s, ta, wa);
      }
    }
  }
}

void PathCall::plan_queue() {
  // ---- queue iterations; the device decides when each point / lane / the solve is finished ------
  chunk = o.check_every;
  if (chunk <= 0) {
    // passes queued per status poll.  The host learns of the stop one chunk late, so up to two chunks
    // of launches return at once at the end of a solve (4.5 us each): small chunks win even for tiny
    // problems (measured, tools/chunk_probe.py: 19-pass fit 0.63 ms at 32, 0.44 ms at 4).
    const double est_us = std::max(12.0, (double)n * (double)ld * 8.0 / 5.0e6);
    chunk = est_us > 150.0 ? 2 : 4;
  }
  if (use_ws) chunk = std::min(chunk, 8);  // a queued pass is nine launches even when it returns at once
  int max_points = 0;
  for (int l = 0; l < B; ++l) max_points = std::max(max_points, (int)lanes[l].n_points);
  if (shared_path) max_points = (int)total_points;  // a lane may end up walking most of the path
  max_total = (int64_t)max_points * o.max_iter;

  // (hipGraph replay of a chunk of passes was tried in round 1 and removed: the loop is bound by the ~1.5 us
  //  dependent-kernel boundaries on the device, not by host launches -- 18.7 against 16.9 us per three-kernel pass on
  //  small problems -- and instantiation cost 0.6 ms per solve)
  tr[2] = t_mark();
  // Working-set solves from the start verify one point per lane and pass, after the pass at zero: the
  // queue is cut to end exactly there, and polls go pass by pass after it (a miss adds a pass or two).
  // Without this a 5-pass path drags three queued no-op passes behind it (12 launches each).
  if (use_ws && !ws_late && o.check_every <= 0) {
    int64_t most = 0;
    for (int l = 0; l < B; ++l) {
      int64_t mine = lanes[l].n_points;
      if (shared_path && interleave) mine = slm_host::interleaved_points(slm_host::LaneWalk{h[l].pt_lo, h[l].n_points, h[l].stride, h[l].tail_pt});
      most = std::max<int64_t>(most, mine);
    }
    expected = 1 + most;
    planned_end = expected;
  }
  // ---- sample start ------------------------------------------------------------------------------------------------
  // A cold path -- no lane brings a warm start -- used to open with a pass over X for the gradient at zero, of which the solve
  // uses two things: the choice of the first working set, and -- on it -- the exact linear term of the model.  The choice
  // needs the ranking of |X_j^T y|, which a quarter of the rows gives (a feature that enters on the first band of alphas
  // stands above the sampling noise); the linear term on W is X_W^T y, one read of the gathered columns.  So the path
  // opens on the first n / 4 rows (150 us instead of 570), nothing is accepted on that estimate (TailArgs::provisional),
  // the model of the first refinement is exact on W, and the first pass over ALL of X already verifies the first band:
  // 4 passes per 50-alpha path instead of 5.  What the sample ranks wrongly the verification finds (a miss: the columns
  // are appended and the point is verified again, as after any pass) -- rows in an order that makes their head
  // unrepresentative cost a pass, not a digit.  SLM_NO_SAMPLE_START=1 opens on all rows.
  // Paths only: their first band sits at the top of the alpha range, where what enters stands far above the sampling
  // noise.  A single cold point at a small alpha admits features the sample cannot tell from noise -- measured on the
  // headline's data (tools/single_fit_big.py): 1.51 -> 1.01 ms at 0.3 alpha_max, 1.53 -> 1.91 ms at 0.05 (a miss and its
  // append on top of the sample's launches), 2.33 -> 1.92 ms at 0.005; SLM_SAMPLE_START_ALL=1 takes that gamble.
  {
    bool cold = true;
    for (int l = 0; l < B; ++l) cold = cold && lanes[l].beta0 == nullptr;
    if (cold && (shared_path || knobs().sample_start_all) && use_ws && !ws_late && !sharded && !cov_on && split && !any_rw && !ds->rw &&
        !custom_scale && expected > 0 && o.max_iter >= 4 && !(o.flags & SLM_FLAG_FISTA_ONLY) && knobs().sample_start) {
      const int64_t least = knobs().sample_min_rows;  // (65536: below it a pass costs little more than the launches of the sample's; SLM_SAMPLE_START_MIN_ROWS: tests)
      // A quarter of the rows (round 4: an eighth).  The sample has to rank the features of the first band's DEEPEST point
      // above the noise features: a gradient entry of the sample carries noise sd(y) / sqrt(rows) -- on the headline's law
      // 3.7 from an eighth of the rows, 2.6 from a quarter -- and the largest of 5 000 noise entries is 3.7 sd: from an
      // eighth the features entering at point 16-17 of eighteen lanes (|beta| about 9) sit INSIDE the noise features' range
      // (67 of those above 9), from a quarter above it (2).  Measured over eight draws of the headline's law
      // (tools/headline_data_seeds.py): eighteen lanes 34 passes / 4.43 ms per path on an eighth, 30 / 3.75 on a quarter,
      // 27 / 3.39 on a half; sixteen lanes 4.11 / 3.89 / 3.78; the bench's own draw 2.86 either way; the soak law's twelve
      // 91.6 -> 92.8 ms in total (a half: 98.2).  SLM_SAMPLE_DIV sets the divisor.
      if (n >= least) n_sample = n / knobs().sample_div;
    }
  }
  // ---- model Gram (mg_kernels.hpp) -----------------------------------------------------------------------------------
  // Lanes whose solutions outgrow the working set used to finish with plain steps, two reads of X each.  When a snapshot
  // shows that this has begun (WsCtl::outgrown: a selection did not fit the working set's 512 columns), the
  // model Gram of the dataset is built -- once, it stays with the dataset -- and every later pass is followed by a round
  // of proximal-gradient steps on it for the lanes the working set does not serve (mg_enqueue_round).  From then on the
  // host looks at every pass's snapshot before it queues the next: a round is sized by what the last one needed.
  mg_forced = knobs().mg == 2;  // (tests: any size, from the first snapshot on)
  // (the model Grams of a dataset, fp32, are kept within 3 GB: sixteen row sets at p = 5 000, seven at 10 000)
  mg_cap = slm_host::model_gram_cap(ld, 3.0e9, kMgEntries);
  mg_ok = use_ws && split && (big_x || mg_forced) && !sharded && !cov_on && !(o.flags & SLM_FLAG_NO_MODEL_GRAM) &&
                     (size_t)ds->lane_cap >= (size_t)kMaxLanes && ws_n_sets <= mg_cap && mg_possible(ds);
  if (mg_ok && mg_forced) expected = 0;  // (tests: polled from the first chunk on, so that short solves reach the rounds too)
  // A dataset that already holds the model Gram of every row set of this call (an earlier solve outgrew the working set
  // and built them: the same path again, a refit, the next search on the data) will be served by the rounds the moment
  // its selection stops fitting: the working set then stays as it is from the first overflow on, instead of being
  // selected, gathered and multiplied afresh once (2-3 ms at 500 columns) before the host has seen the counter.
  // (lanes on the dataset's own rows only: other row sets are told apart by fingerprints, a kernel and a round trip)
  if (mg_ok && !mg_forced && !ds->mg.empty() && knobs().mg_keep) {
    bool all = ws_n_sets > 0;
    for (int st = 0; st < ws_n_sets && all; ++st) {
      const int l = ws_set_lane[st];
      all = lanes[l].row_weight == nullptr;
      const double ne = ls.n_eff[l] > 0 ? ls.n_eff[l] : (double)ds->n_global;
      bool have = false;
      for (const auto& e : ds->mg) have = have || (e.own && e.n_eff == ne);
      all = all && have;
    }
    if (all) wa.keep_full = 1;
  }
  prof_off = n_sample > 0 ? 1 : 0;  // (the pass on the sample is no launch of the roofline's kernel on X)
  // SLM_TRACE=3: one line per polled snapshot -- where every lane stands, the working set, the model Gram's rounds
  // (with SLM_TRACE_POLL=1 every pass is polled: a diagnostic, the queue then drains between passes)
  trace3 = knobs().trace == 3;
  if (trace3 && knobs().trace_poll && expected > 0) expected = n_sample > 0 ? 2 : 1;  // (the sample pass's refinement is never held back)
}

// the model Gram of every row set of the call (ws_set_of: lanes with the same row weights and scaling share one), found
// by the fingerprint of the row weights as the lanes brought them, or built
int PathCall::mg_sets() {
  const double* wdev[SLM_MAX_LANES];
  double fp[2 * SLM_MAX_LANES] = {};
  int n_fp = 0, fp_at[SLM_MAX_LANES];
  for (int st = 0; st < ws_n_sets; ++st) {
    const int l = ws_set_lane[st];
    fp_at[st] = -1;
    if (lanes[l].row_weight != nullptr) {
      fp_at[st] = n_fp;
      wdev[n_fp++] = ls.rw + (int64_t)l * ls.rw_stride;
    }
  }
  if (n_fp > 0) SLM_TRY(cov_fingerprints(ds, wdev, n_fp, fp));
  int missing = 0;
  for (int pass = 0; pass < 2; ++pass) {
    // (first round: is there room for what is missing?  if not, every entry goes and all of the call's are built)
    if (pass == 1 && (int)ds->mg.size() + missing > mg_cap) mg_invalidate(ds);
    if (pass == 1) mg_built += missing;
    for (int st = 0; st < ws_n_sets; ++st) {
      const int l = ws_set_lane[st];
      const bool own = lanes[l].row_weight == nullptr;
      const double ne = ls.n_eff[l] > 0 ? ls.n_eff[l] : (double)ds->n_global;
      const double f1 = own ? 0.0 : fp[2 * fp_at[st]], f2 = own ? 0.0 : fp[2 * fp_at[st] + 1];
      if (pass == 0) {
        bool have = false;
        for (const auto& e : ds->mg) have = have || (e.n_eff == ne && (own ? e.own : (!e.own && e.fp1 == f1 && e.fp2 == f2)));
        missing += have ? 0 : 1;
      } else {
        SLM_TRY(mg_ensure(ds, own ? nullptr : ls.rw + (int64_t)l * ls.rw_stride, ne, own, f1, f2, &mg_entry_of_set[st]));
      }
    }
  }
  return SLM_OK;
}

bool PathCall::mg_wanted(const DevCtl& c) const {
  if (mg_forced) return true;
  // (capacity, not difficulty: a lane that spends passes on an ill-conditioned face inside the working set is served by
  //  the model solver's direct steps, and the set must stay free to be selected afresh there)
  return c.ws.outgrown > 0 || c.ws.overflows > 0 || c.ws.disabled != 0;
}

int PathCall::mg_consider(const DevCtl& c) {
  if (!mg_ok || mg_on || !mg_wanted(c)) return SLM_OK;
  const auto t0 = std::chrono::steady_clock::now();
  const int rc = mg_sets();
  if (rc == SLM_OK) {
    mg_on = true;
    mg_handover = knobs().handover;
    wa.keep_full = 1;  // (from here on the working set serves what it holds: enqueue_refinement passes wa by value)
    if (mg_built > 0) mg_build_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  } else if (rc != SLM_ERR_OOM) {
    return rc;
  }
  return SLM_OK;  // (no memory for it: the solve goes on with plain steps)
}

// SLM_TRACE=3: one line per polled snapshot -- where every lane stands, the working set, the model Gram's rounds
void PathCall::trace_pass(const DevCtl& now) const {
  if (!trace3) return;
  fprintf(stderr, "[slm] pass %lld at %.3f ms: K %d builds %d appends %d misses %d stale %d refined %d | mg on %d rounds %d inner %d most %d rej %d | light %d of %d (last: %d columns, why %d, D %.3g) | lanes (point.iter/flags):",
          (long long)enq, t_mark(), now.ws.Kreal, now.ws.builds, now.ws.appends, now.ws.misses, now.ws.stale, now.ws.refined, (int)mg_on,
          now.mg.rounds, now.mg.inner_iters, now.mg.most_iters, now.mg.rejected, now.lt.used, now.lt.attempts, now.lt.n_cols, now.lt.why, now.lt.D[0]);
  for (int l = 0; l < B; ++l)
    fprintf(stderr, " %d.%d%s%s%s", now.lane[l].point, now.lane[l].iter, now.lane[l].done ? "d" : "", now.lane[l].zsup ? "w" : "",
            l < SLM_MAX_LANES && now.mg.lane[l].active ? "m" : "");
  fprintf(stderr, "\n");
}

// A pass beyond the expected end of a working-set path re-verifies the few lanes whose last verification missed: such a pass
// may be a CERTIFIED PARTIAL one (light_kernels.hpp) -- the host queues the attempt in front of the pass over X, the device
// decides whether it stands (then the pass's kernels return at once) or stands down (then they run).  Shared paths over the
// dataset's own unweighted rows, one device; not beside model-Gram rounds or covariance passes.
bool PathCall::light_eligible() {
  if (!knobs().light_pass || !shared_path || !use_ws || ws_late || cov_on || sharded || mg_on) return false;
  if (any_rw || ds->rw || custom_scale || B > kMaxLanes || !(expected > 0 && enq >= expected)) return false;
  return ds->XT != nullptr && ds->XT_ready && ds->colnorm != nullptr && ds->colnorm_ready;
}

int PathCall::enqueue_light_attempt() {
  // row blocks of the attempt's own kernels: one per CU, more when a block's image of dR (8 lanes) would not fit 60 KB of LDS
  const int nblk = (int)std::max<int64_t>(std::max<int64_t>(1, std::min<int64_t>(eng->cus, n)), (n + 959) / 960);
  if (!ds->lt_dR) SLM_TRY(dalloc(&ds->lt_dR, (size_t)n * LT_LANES));
  if (!ds->lt_d2) SLM_TRY(dalloc(&ds->lt_d2, (size_t)nblk * (LT_LANES + kMaxLanes)));  // (+ the block sums of the losses)
  if (!ds->lt_part) SLM_TRY(dalloc(&ds->lt_part, (size_t)nblk * LT_CAP * LT_LANES));
  if (!ds->lt_cols) SLM_TRY(dalloc(&ds->lt_cols, (size_t)LT_CAP));
  if (!ds->lt_stamp) SLM_TRY(dalloc(&ds->lt_stamp, (size_t)ld));
  if (!light_begun) {  // (attempt numbers start at 1 with every solve: the stamps of the solve before mean nothing)
    HIP_TRY(hipMemsetAsync(ds->lt_stamp, 0, sizeof(int32_t) * (size_t)ld, s));
    light_begun = true;
  }
  LightArgs la;
  memset(&la, 0, sizeof(la));
  la.lt = &ds->dctl->lt; la.ctl = ds->ctl; la.done = done_flag; la.pts = ds->pts; la.ws = wa.ws; la.idx = wa.idx; la.pos = wa.pos;
  la.XW = wa.XW; la.XT = ds->XT; la.Gm = wa.Gm; la.y = ds->y; la.colnorm = ds->colnorm;
  la.z = ds->z; la.zprev = ds->zprev; la.gprev = ds->gprev; la.a0 = ds->a0; la.b0 = ds->b0; la.g = ds->g;
  la.loss_partial = ds->lt_d2 + (size_t)nblk * LT_LANES; la.dR = ds->lt_dR; la.d2_part = ds->lt_d2; la.cols = ds->lt_cols; la.stamp = ds->lt_stamp; la.part = ds->lt_part;
  la.n = n; la.ld = ld; la.rows_base = n / nblk; la.rows_rem = n % nblk;
  la.p = (int)p; la.n_lanes = B; la.slots = SPLIT_LANES * ((B + SPLIT_LANES - 1) / SPLIT_LANES); la.nblk = nblk;
  la.inv_n = 1.0 / (double)ds->n_global;
  la.order = ds->order; la.gstart = ds->gstart; la.G = G; la.singleton = ds->singleton;
  hipLaunchKernelGGL(light_prepare_kernel, dim3(1), dim3(1024), 0, s, la);
  hipLaunchKernelGGL(light_resid_kernel, dim3((unsigned)nblk), dim3(LT_WAVES * 64), 0, s, la);
  hipLaunchKernelGGL(light_select_kernel, dim3(1), dim3(1024), 0, s, la);
  const size_t lds = sizeof(double) * (size_t)(la.rows_base + 1) * LT_LANES;
  hipLaunchKernelGGL(light_coldot_kernel, dim3((unsigned)nblk), dim3(256), lds, s, la);
  hipLaunchKernelGGL(light_assemble_kernel, dim3(LT_LANES), dim3(1024), 0, s, la);
  return SLM_OK;
}

// passes of one chunk: gradient, tail, (refinement), and the snapshot the host will read
int PathCall::queue_chunk() {
  // (a solve with an expected end queues all of its passes at once: launches behind the device-side stop flag return
  //  at once, and every snapshot in between -- a copy, an event, 6 us of idle stream around them -- told the host
  //  nothing it acts on)
  const int this_chunk = mg_on ? 1 : (expected <= 0 ? chunk : (enq < expected ? (int)std::min<int64_t>(64, expected - enq) : 1));
  for (int i = 0; i < this_chunk; ++i) {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool light = light_eligible();  // (the pass over X of this step may be skipped on the device: not timed)
    if (profile && !light && enq >= prof_off && (enq - prof_off) % kProfStride == 0) {  // sampled: an event pair costs ~8 us of stream time
      const int64_t slot_id = (enq - prof_off) / kProfStride;
      while ((int64_t)ds->prof.size() < 2 * (slot_id + 1)) {
        hipEvent_t ev;
        HIP_TRY(hipEventCreate(&ev));
        ds->prof.push_back(ev);
      }
      e0 = ds->prof[2 * slot_id];
      e1 = ds->prof[2 * slot_id + 1];
      if ((int64_t)prof_rec.size() <= slot_id) prof_rec.resize((size_t)slot_id + 1, 0);
      prof_rec[(size_t)slot_id] = 1;
    }
    if (light) SLM_TRY(enqueue_light_attempt());
    light_skip = light ? &ds->dctl->lt.ok : nullptr;
    const bool sample_pass = n_sample > 0 && enq == 0;
    if (sample_pass) {
      LaneSetup part = ls;
      for (int l = 0; l < kMaxLanes; ++l) part.n_eff[l] = (double)ds->n_global * (double)n_sample / (double)n;
      SLM_TRY(enqueue_gradient_split(ds, part, ds->y, done_flag, ds->ctl, &wa, nullptr, nullptr, n_sample));
      TailArgs first = ta;
      first.provisional = 1;
      launch_tail(first, s);
    } else {
      if (!(carry && enq == 0)) SLM_TRY(enqueue_pass_gradient(e0, e1));  // (a carried start has its first gradient)
      enqueue_tail();
    }
    fix_start = sample_pass;
    ++enq;
    // behind the pass the solve is expected to end with -- and behind the few re-verifications past it, which the host
    // polls one by one (pass_loop: at_end) and which mostly end the solve -- the launches of the refinement would only find
    // out that there is nothing left to refine (30-40 us): they follow once the snapshot says otherwise
    deferred = (expected > 0 && enq >= expected && enq < expected + 4 && !sharded) || mg_on;
    if (!deferred) enqueue_refinement();
    fix_start = false;
  }
  SLM_TRY(check_launch());
  if (ws_comm_rc != 0) return ws_comm_rc;  // (all_reduce_sum has set the message)
  HIP_TRY(hipMemcpyAsync(&ds->hctl[slot].c, ds->dctl, sizeof(DevCtl), hipMemcpyDeviceToHost, s));
  return SLM_OK;
}

int PathCall::pass_loop() {
  while (!done) {
    SLM_TRY(queue_chunk());
    // The pass the solve is expected to end with: the host waits for THIS chunk instead of queueing another pass
    // behind it -- when the solve does end there (the usual case) the snapshot is final and only the coefficients
    // remain to be fetched.  Polling one chunk behind cost a queued pass that returned at once (eighteen launches,
    // 0.09 ms) and four blocking copies (0.2 ms of host round trips) on every 5 ms path.  A solve that overruns
    // gets a few more passes polled this way, then the pipelined polls.
    const bool at_end = mg_on || (expected > 0 && enq >= expected && (enq < expected + 4 || (trace3 && knobs().trace_poll)));
    HIP_TRY(hipEventRecord(ds->ev[slot], s));
    // behind the pass the solve is expected to end with, the results set off at once: when it does end there they are
    // under way while the host still reads the snapshot (37 us of idle stream per path); when it does not, they are
    // fetched again at the real end
    if (expected > 0 && enq == expected && !results_queued) {
      SLM_TRY(enqueue_result_copies());
      results_queued = true;
    }
    pending[slot] = true;
    const int other = slot ^ 1;
    if (at_end) {
      // sleep until the chunk before this one is through, then watch this one's event: a blocking wait wakes up
      // 20-40 us after the event (interrupt + scheduler), a query loop within a microsecond or two -- and it
      // runs for one chunk (a pass or two) at most
      if (pending[other]) HIP_TRY(hipEventSynchronize(ds->ev[other]));
      const auto t_spin = std::chrono::steady_clock::now();
      for (;;) {
        const hipError_t q = hipEventQuery(ds->ev[slot]);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) HIP_TRY(q);
        if (std::chrono::steady_clock::now() - t_spin > std::chrono::milliseconds(20)) {
          HIP_TRY(hipEventSynchronize(ds->ev[slot]));
          break;
        }
      }
      pending[slot] = pending[other] = false;
      if (ds->hctl[slot].c.g.done) {
        done = true;
        final_slot = slot;
        results_final = results_queued && enq == expected;  // (queued behind exactly this pass)
      } else {  // the solve goes on: what was held back, then the next pass
        const DevCtl& now = ds->hctl[slot].c;
        trace_pass(now);
        SLM_TRY(mg_consider(now));
        if (deferred) {
          // (once the working set is frozen and no live lane stands on it any more -- the dense end of a path -- its
          //  seven launches would only find that out again, 70 us a pass: the rounds below serve every lane)
          bool ws_serves = !(mg_on && now.ws.stale != 0 && wa.keep_full != 0);
          for (int l = 0; l < B && !ws_serves; ++l) ws_serves = !now.lane[l].done && !now.lane[l].idle && now.lane[l].zsup != 0;
          if (ws_serves) enqueue_refinement();
          deferred = false;
        }
        if (mg_on) {
          // (a round that left a lane short of its tolerance -- an ill-conditioned face -- is followed by one twice as long:
          //  an inner iteration costs a twentieth of a pass)
          if (now.mg.rounds > 0) mg_inner = now.mg.most_iters >= mg_inner ? std::min(96, 2 * mg_inner) : std::max(6, std::min(96, (int)now.mg.most_iters + 2));
          SLM_TRY(mg_enqueue_round(ds, ta, B, mg_inner, done_flag, ws_n_sets, mg_entry_of_set, ws_set_of));
        }
      }
    } else if (pending[other]) {
      HIP_TRY(hipEventSynchronize(ds->ev[other]));
      pending[other] = false;
      if (ds->hctl[other].c.g.done) {
        done = true;
        final_slot = other;
      }
      if (!done) SLM_TRY(mg_consider(ds->hctl[other].c));  // (rounds follow the passes queued from here on)
      if (!done && ws_late && ds->hctl[other].c.g.hard >= kWsLateIters) {
        const int rc = ws_setup(true);  // (waits for the stream: the queued passes simply finish first)
        ws_late = false;
        if (rc == SLM_OK) {
          use_ws = true;
          chunk = std::min(chunk, 8);
        } else if (rc == SLM_ERR_OOM) {
          ws_release();  // carry on with plain steps
        } else {
          return rc;
        }
      }
    }
    slot = other;
    if (!done && enq >= max_total + 2 * (int64_t)chunk) {
      HIP_TRY(hipStreamSynchronize(s));
      return fail(SLM_ERR_HIP, "internal error: path state machine did not terminate");
    }
  }
  return SLM_OK;
}

int PathCall::finish() {
  if (!results_final) SLM_TRY(enqueue_result_copies());
  HIP_TRY(hipStreamSynchronize(s));
  tr[3] = t_mark();
  const DevCtl& snap = ds->hctl[final_slot].c;  // (nothing in the block changes after `done`)
  if (sharded && snap.g.diverged)
    return fail(SLM_ERR_COMM, "row-sharded solve aborted: the ranks' solver states differ (different arguments on "
                "different ranks, or an all-reduce that is not bit-identical on every rank)");
  const PathCtl* fin = snap.lane;
  if (infos_in_snap) {
    int64_t at = 0;
    for (int l = 0; l < B; ++l) {
      if (lanes[l].infos) memcpy(lanes[l].infos, snap.infos + at, sizeof(slm_point_info) * (size_t)lanes[l].n_points);
      at += lanes[l].n_points;
    }
  }
  int64_t passes = 0;
  bool nonfinite = false;
  for (int l = 0; l < B; ++l) {
    passes = std::max<int64_t>(passes, fin[l].total_iter);
    nonfinite = nonfinite || fin[l].nonfinite;
  }
  if (stats) {
    // launches over the data that did work (every launch serves all lanes): passes the tail kernels saw, less the carried
    // first one, the pass on the row sample, and the certified partial passes (no read of X)
    stats->grad_launches = passes - (carry ? 1 : 0) - prof_off - snap.lt.used;
    stats->light_passes = snap.lt.used;
    stats->light_columns = snap.lt.cols_total;
    stats->grad_ms_total = 0.0;
    stats->grad_timed = 0;
    if (profile) {
      double tot = 0.0;
      int64_t cnt = 0;
      // iterations 0, kProfStride, 2 kProfStride, ... below `passes` did real work and were timed
      for (int64_t k = 0; k * kProfStride < passes - prof_off && 2 * k + 1 < (int64_t)ds->prof.size(); ++k) {
        if (k >= (int64_t)prof_rec.size() || !prof_rec[(size_t)k]) continue;  // (not recorded in this solve)
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ds->prof[2 * k], ds->prof[2 * k + 1]) == hipSuccess) {
          tot += ms;
          ++cnt;
        }
      }
      stats->grad_ms_total = tot;
      stats->grad_timed = cnt;
    }
    stats->lipschitz_ms = lipschitz_ms;
    stats->ws_builds = stats->ws_appends = stats->ws_refined = stats->ws_misses = stats->ws_columns = 0;
    stats->ws_inner_iters = stats->ws_direct_steps = 0;
    stats->mg_rounds = snap.mg.rounds;
    stats->mg_inner_iters = snap.mg.inner_iters;
    stats->mg_rejected = snap.mg.rejected;
    stats->mg_build_ms = mg_build_ms;
    if (use_ws) {
      const WsCtl& wc = snap.ws;
      stats->ws_builds = wc.builds;
      stats->ws_appends = wc.appends;
      stats->ws_refined = wc.refined;
      stats->ws_misses = wc.misses;
      stats->ws_columns = wc.Kreal;
      stats->ws_inner_iters = wc.inner_iters;
      stats->ws_direct_steps = wc.newton_steps;
      if (knobs().trace == 2) {
        fprintf(stderr, "[slm] working set: %d model-solver iterations over %d refinements, %d direct steps (%d refused, %d of them not positive definite), K = %d, lambda_max bound of the first Gram %.4g (seed L %.4g)\n",
                wc.inner_iters, wc.refined, wc.newton_steps, wc.newton_fails, wc.newton_nopd, wc.K, wc.Lw[0], fin[0].L);
        fprintf(stderr, "[slm] model solver, lane 0, ms over the solve: set-up %.3f, lambda_max of a new Gram %.3f, start value %.3f, "
                "iteration %.3f, acceptance + write-back %.3f\n", wc.solve_ticks[0] * 1e-5, wc.solve_ticks[1] * 1e-5,
                wc.solve_ticks[2] * 1e-5, wc.solve_ticks[3] * 1e-5, wc.solve_ticks[4] * 1e-5);
        fprintf(stderr, "[slm] model solves by iterations:");
        for (int i = 0; i < 32; ++i)
          if (wc.iters_hist[i]) fprintf(stderr, " %d:%d", i, wc.iters_hist[i]);
        fprintf(stderr, "\n");
        fprintf(stderr, "[slm] model solver, ms per lane over the solve:");
        for (int l = 0; l < B; ++l) fprintf(stderr, " %.3f", wc.lane_ticks[l] * 1e-5);
        fprintf(stderr, "\n");
        if (wc.newton_factors) {
          fprintf(stderr, "[slm] direct steps: accepted at t = 1: %d, 1/2: %d, 1/4: %d, first sign change: %d; %d factorisations, %.0f unknowns on average\n",
                  wc.newton_trial[0], wc.newton_trial[1], wc.newton_trial[2], wc.newton_trial[3], wc.newton_factors,
                  (double)wc.newton_unknowns / wc.newton_factors);
          fprintf(stderr, "[slm] direct steps without a usable segment: t = 0: %d, slope <= 0: %d, curvature <= 0: %d; no decrease on it: %d\n",
                  wc.newton_ref[0], wc.newton_ref[1], wc.newton_ref[2], wc.newton_ref[3]);
          int worst = 0;
          double worst_ms = -1.0;
          for (int l = 0; l < B; ++l) {
            double t = 0.0;
            for (int k = 0; k < 6; ++k) t += wc.nt_ticks[l][k] * 1e-5;
            if (t > worst_ms) {
              worst_ms = t;
              worst = l;
            }
          }
          const unsigned long long* tk = wc.nt_ticks[worst];
          fprintf(stderr, "[slm] direct steps of the busiest lane (%d: %d factorisations), ms: matvec + free set %.3f, assembly %.3f, "
                  "factorisation %.3f, solve %.3f, trial points %.3f, mu %.3f\n", worst, wc.nt_factors[worst],
                  tk[0] * 1e-5, tk[1] * 1e-5, tk[2] * 1e-5, tk[3] * 1e-5, tk[4] * 1e-5, tk[5] * 1e-5);
        }
      }
    }
    stats->wall_ms =
        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  }
  tr[4] = t_mark();
  if (knobs().trace != 0)  // 1: slow solves only, 2: every solve (cumulative ms since entry)
    if (tr[4] > 15.0 || knobs().trace == 2)
      fprintf(stderr, "[slm] solve: row weights %.3f L %.3f setup %.3f sync %.3f prequeue %.3f loop %.3f end %.3f ms\n", tr_rw, tr[5], tr[0], tr[1], tr[2], tr[3], tr[4]);
  if (nonfinite) return fail(SLM_ERR_NON_FINITE, "non-finite iterate (diverged or non-finite data)");
  if (!shared_path && !(o.flags & SLM_FLAG_COLD_START) && B <= kMaxLanes) {  // where the solve ended (carried starts, above)
    ds->carry_out.resize((size_t)B * (size_t)p);
    for (int l = 0; l < B; ++l) {
      memcpy(ds->carry_out.data() + (size_t)l * p, lanes[l].betas_out + (size_t)(lanes[l].n_points - 1) * p, sizeof(double) * (size_t)p);
      slm_dataset::CarryLane& c = ds->carry_lane[l];
      c.n_eff = lanes[l].n_eff;
      c.has_rw = lanes[l].row_weight != nullptr;
      c.fp[0] = rw_fp[l][0];
      c.fp[1] = rw_fp[l][1];
      c.loss = fin[l].loss_base;
    }
    ds->carry_lanes = B;
    ds->carry_valid = true;
    if (use_ws && snap.ws.valid && !snap.ws.building && !snap.ws.disabled && !snap.ws.stale) {
      ds->ws_carry_valid = true;
      ds->ws_carry_cov = cov_on;
      ds->ws_carry_sets = ws_n_sets;
      for (int l = 0; l < kMaxLanes; ++l) ds->ws_carry_set_of[l] = l < B ? ws_set_of[l] : 0;
    }
  }
  return SLM_OK;
}

extern "C" int slm_dataset_max_lanes(slm_dataset* ds, uint32_t flags, int32_t* max_lanes_out) {
  if (!ds || !max_lanes_out) return fail(SLM_ERR_BAD_ARG, "NULL argument");
  *max_lanes_out = max_lanes_for(ds, flags);
  return SLM_OK;
}

extern "C" int slm_solve_lanes(slm_dataset* ds, const slm_lane* lanes, int32_t n_lanes,
                               const slm_solve_opts* opts, slm_solve_stats* stats) {
  return solve_core(ds, lanes, n_lanes, opts, stats, false);
}

extern "C" int slm_solve_lanes_reweighted(slm_dataset* ds, const slm_lane* lanes, const slm_reweight* rules, int32_t n_lanes,
                                          const slm_solve_opts* opts, slm_solve_stats* stats, int32_t* rounds_out) {
  if (!rules) return fail(SLM_ERR_BAD_ARG, "rules is NULL");
  return solve_core(ds, lanes, n_lanes, opts, stats, false, rules, rounds_out);
}

// the engine's choice of lanes for a shared path (n_lanes = 0): the fewest passes over X at the price of sixteen lanes
static int auto_lanes(const slm_dataset* ds, int32_t n_points, uint32_t fl) {
  const int cap = max_lanes_for(ds, fl);
  const bool big = ws_policy(ds, fl) == 2 && (double)ds->n * (double)ds->ld >= 67108864.0 && !small_ok(ds, fl);
  const bool interleaved = ds->singleton && knobs().interleave;  // (solve_core: per-feature penalties take the points in turn)
  int B = slm_host::auto_path_lanes(n_points, cap, big, !interleaved);
  if (knobs().auto_lanes > 0) B = std::max(1, std::min<int>(std::min(knobs().auto_lanes, cap), n_points));  // (SLM_AUTO_LANES: A/B runs)
  return B;
}
extern "C" int slm_dataset_path_lanes(slm_dataset* ds, int32_t n_points, uint32_t flags, int32_t* lanes_out) {
  if (!ds || !lanes_out) return fail(SLM_ERR_BAD_ARG, "NULL argument");
  if (n_points <= 0) return fail(SLM_ERR_BAD_ARG, "n_points must be positive");
  *lanes_out = auto_lanes(ds, n_points, flags);
  return SLM_OK;
}

extern "C" int slm_solve_path_lanes(slm_dataset* ds, const slm_penalty* pen, const slm_path_point* points,
                                    int32_t n_points, int32_t n_lanes, const slm_solve_opts* opts,
                                    const double* beta0, double* betas_out, double* group_norms_out,
                                    slm_point_info* infos, slm_solve_stats* stats) {
  if (!ds || !points || !betas_out) return fail(SLM_ERR_BAD_ARG, "NULL argument");
  if (n_points <= 0) return fail(SLM_ERR_BAD_ARG, "n_points must be positive");
  const uint32_t fl = opts ? opts->flags : 0u;
  int B;
  if (n_lanes == 0) {  // the engine's choice (host_logic.hpp)
    B = auto_lanes(ds, n_points, fl);
  } else {
    B = std::max(1, std::min<int>(std::min<int>(n_lanes, kMaxLanes), n_points));
    B = std::min(B, max_lanes_for(ds, fl));  // no kernel variant for (p, B): fewer lanes
  }
  SLM_TRY(lanes_with_copy(ds, fl, B, &B));  // (more than sixteen: the column-major copy, or sixteen)
  slm_lane lanes[SLM_MAX_LANES];
  memset(lanes, 0, sizeof(lanes));
  int64_t lo = 0;
  for (int l = 0; l < B; ++l) {
    const int64_t hi = (int64_t)n_points * (l + 1) / B;
    lanes[l].pen = pen;
    lanes[l].points = points + lo;
    lanes[l].n_points = (int32_t)(hi - lo);
    lanes[l].beta0 = (l == 0) ? beta0 : nullptr;
    lanes[l].betas_out = betas_out + lo * ds->p;
    lanes[l].group_norms_out = group_norms_out ? group_norms_out + lo * ds->G : nullptr;
    lanes[l].infos = infos ? infos + lo : nullptr;
    lo = hi;
  }
  return solve_core(ds, lanes, B, opts, stats, B > 1);
}

extern "C" int slm_solve_path(slm_dataset* ds, const slm_penalty* pen, const slm_path_point* points,
                              int32_t n_points, const slm_solve_opts* opts, const double* beta0,
                              double* betas_out, double* group_norms_out, slm_point_info* infos,
                              slm_solve_stats* stats) {
  if (!ds || !points || !betas_out) return fail(SLM_ERR_BAD_ARG, "NULL argument");
  slm_lane lane;
  memset(&lane, 0, sizeof(lane));
  lane.pen = pen;
  lane.points = points;
  lane.n_points = n_points;
  lane.beta0 = beta0;
  lane.betas_out = betas_out;
  lane.group_norms_out = group_norms_out;
  lane.infos = infos;
  return slm_solve_lanes(ds, &lane, 1, opts, stats);
}

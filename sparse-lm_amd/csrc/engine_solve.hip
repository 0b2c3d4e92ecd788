// Host side of the MI355X fit engine, second unit: kernel tables, the launches of a pass, step-size seeds, centring and the
// device-resident solve loop -- what replaces the cvxpy `problem.solve` call of
// /root/reference/src/sparselm/model/_base.py:512-519 (and the inner solve of model/_adaptive_lasso.py:213-215).
#include "engine_internal.hpp"

// ------------------------------------------------------------------------------------------------
// gradient kernel table
// ------------------------------------------------------------------------------------------------
#define SLM_GK(W, C, R, B) {W, C, R, B, 0, grad_fused_kernel<W, C, R, B>}
#define SLM_RK(W, C, B, D) {W, C, 1, B, D, grad_ring_kernel<W, C, B, D>}
// LDS-ring variants, preferred where they exist (512-thread workgroups, rows of up to 5120 columns);
// ordered by capacity within each lane count.
static const GradKernel kGradRing[] = {
    SLM_RK(8, 1, 1, 3), SLM_RK(8, 2, 1, 3), SLM_RK(8, 3, 1, 3), SLM_RK(8, 4, 1, 3), SLM_RK(8, 5, 1, 2),
    SLM_RK(8, 1, 2, 3), SLM_RK(8, 2, 2, 3), SLM_RK(8, 3, 2, 3), SLM_RK(8, 4, 2, 3), SLM_RK(8, 5, 2, 2),
    SLM_RK(8, 1, 3, 3), SLM_RK(8, 2, 3, 3), SLM_RK(8, 3, 3, 3), SLM_RK(8, 4, 3, 3), SLM_RK(8, 5, 3, 2),
    SLM_RK(8, 1, 4, 3), SLM_RK(8, 2, 4, 3), SLM_RK(8, 3, 4, 3), SLM_RK(8, 4, 4, 3), SLM_RK(8, 5, 4, 2),
    // five and six lanes only where the 8C VGPRs per lane leave the kernel spill-free: a variant with
    // 12 spilled registers in the row loop (B = 5 at C = 5) measured 1.62 ms against 0.59 ms
    SLM_RK(8, 1, 5, 3), SLM_RK(8, 2, 5, 3), SLM_RK(8, 3, 5, 3), SLM_RK(8, 4, 5, 3),
    SLM_RK(8, 1, 6, 3), SLM_RK(8, 2, 6, 3), SLM_RK(8, 3, 6, 3),
};
// Default choice per (lanes B, capacity 64*W*C chunks of 16 bytes); every list is ordered by
// capacity.  R (rows held per step) is the largest that keeps the kernel free of (large) spills at
// 256 VGPRs; measured register counts are in DESIGN.md.
static const GradKernel kGradDefault[] = {
    // one lane
    SLM_GK(1, 1, 4, 1), SLM_GK(2, 1, 4, 1), SLM_GK(4, 1, 4, 1), SLM_GK(8, 1, 4, 1), SLM_GK(8, 2, 4, 1),
    SLM_GK(8, 3, 4, 1), SLM_GK(8, 4, 2, 1), SLM_GK(8, 5, 2, 1), SLM_GK(8, 6, 2, 1), SLM_GK(8, 8, 2, 1),
    SLM_GK(8, 10, 1, 1),
    // two lanes
    SLM_GK(1, 1, 4, 2), SLM_GK(2, 1, 4, 2), SLM_GK(4, 1, 4, 2), SLM_GK(8, 1, 4, 2), SLM_GK(8, 2, 4, 2),
    SLM_GK(8, 3, 4, 2), SLM_GK(8, 4, 2, 2), SLM_GK(8, 5, 2, 2), SLM_GK(8, 6, 2, 2), SLM_GK(8, 8, 1, 2),
    // three lanes
    SLM_GK(1, 1, 4, 3), SLM_GK(2, 1, 4, 3), SLM_GK(4, 1, 4, 3), SLM_GK(8, 1, 4, 3), SLM_GK(8, 2, 4, 3),
    SLM_GK(8, 3, 4, 3), SLM_GK(8, 4, 2, 3), SLM_GK(8, 5, 2, 3), SLM_GK(8, 6, 1, 3),
    // four lanes
    SLM_GK(1, 1, 4, 4), SLM_GK(2, 1, 4, 4), SLM_GK(4, 1, 4, 4), SLM_GK(8, 1, 4, 4), SLM_GK(8, 2, 4, 4),
    SLM_GK(8, 3, 2, 4), SLM_GK(8, 4, 2, 4), SLM_GK(8, 5, 1, 4),
};
// Extra instantiations reachable through SLM_GRAD_CONFIG=W,C,R (tuning sweeps).
static const GradKernel kGradExtra[] = {
    SLM_GK(8, 5, 1, 1), SLM_GK(8, 5, 3, 1), SLM_GK(8, 5, 4, 1), SLM_GK(8, 4, 4, 1), SLM_GK(8, 6, 1, 1),
    SLM_GK(8, 8, 1, 1), SLM_GK(8, 5, 1, 2), SLM_GK(8, 5, 1, 3), SLM_GK(8, 4, 4, 2), SLM_GK(8, 6, 1, 2),
};

// Rows longer than the fused kernels cover: two-pass fallback (D = -1), one lane, any p.
static const GradKernel kGradTwoPass = {8, 4, 2, 1, -1, nullptr};
static const int kTwoPassC = 4;  // column tile of xtr_kernel: 512 * 4 chunks = 4096 columns

// Split pass (split_kernels.hpp) for working-set solves: sixteen lanes per read of X.  The table is for
// rowdot_ring_kernel (rows of up to 5120 columns, D rows in flight as for the fused ring kernel); rows of
// 5 121 ... 10 240 columns (BASELINE config 5: p = 10 000) have no ring variant -- their LDS ring would not
// fit -- and take every residual that needs X from rowdot_mfma_kernel, which has no column limit but needs
// the column-major copy of X (`rowdot == nullptr`: the split pass is then only used when that copy exists).
#define SLM_SK(C, D)                                                                                   \
  {8, C, SPLIT_LANES, D, rowdot_ring_kernel<8, C, ROWDOT_LANES, D>, resid_ws_kernel<SPLIT_LANES>}
static const SplitKernel kSplit[] = {SLM_SK(1, 3), SLM_SK(2, 3), SLM_SK(3, 3), SLM_SK(4, 3), SLM_SK(5, 2)};
// Rows beyond 5120 columns, ANY width (round 6; until then the table stopped at 10 240 columns and wider rows had one lane on
// the two-pass kernels, at most half of the roofline by construction): nothing in the split pass depends on the row length --
// X^T R walks column blocks of 512, the residuals from X contract the column-major copy group by group, the working set's
// kernels see its <= 512 columns -- so sixteen lanes and the working set serve p = 20 000 or 40 000 like 5 000.
static const SplitKernel kSplitAnyWidth = {8, 0, SPLIT_LANES, 0, nullptr, resid_ws_kernel<SPLIT_LANES>};
const SplitKernel* pick_split_kernel(int64_t p2) {
  if (!knobs().split) return nullptr;
  for (const auto& k : kSplit)
    if (64LL * k.W * k.C >= p2) return &k;
  return &kSplitAnyWidth;
}

// X^T R of the split pass on the matrix cores (xtr_mfma_kernel): grid = (column blocks of 512, row blocks), ONE
// workgroup (four wavefronts, 64 KB of rows in flight) per CU; rows per block a multiple of 8.  Two workgroups per CU
// -- the first choice: more bytes in flight -- measured 4-6 % slower on every box (0.603 against 0.566 ms, 0.622
// against 0.592 ms at n = 100k, p = 5k; tools/xtr_wgs_probe.py): twice as many row streams open at once, and the
// kernel has the bytes in flight it needs with four wavefronts.  SLM_XTR_WGS_PER_CU=2 brings the old grid back.
int xtr_max_row_blocks(int cus, int64_t ld) {  // (sizes the partial buffer: the larger of the two grids)
  return slm_host::xtr_row_blocks_most(cus, ld, XTR_CB);
}
// sets a.xrows; returns the number of row blocks (= blocks of `partial` to reduce)
int launch_xtr(int cus, SplitArgs& a, hipStream_t s, bool sample) {
  double per_cu = knobs().xtr_wgs_per_cu;  // (SLM_XTR_WGS_PER_CU, A/B runs: workgroups per CU, up to 2)
  const bool wide = a.lane_slots > SPLIT_LANES;  // thirty-two lanes: both planes of R per row of X
  if (wide) per_cu = 1.0;                        // (its partial sums fill the buffer at one workgroup per CU)
  const slm_host::XtrGrid g = slm_host::xtr_grid(a.n, a.ld, XTR_CB, (int64_t)(xtr_max_row_blocks(cus, a.ld) * per_cu / 2.0));
  const int xb = g.xb, yb = g.yb;
  a.xrows = g.rows;
  // seventeen to twenty lanes: the lanes beyond sixteen on the vector units beside the sixteen on the matrix cores
  // (xtr18 / xtr20_mfma_kernel: the price of sixteen; SLM_XTR_EXTRAS=0: both halves on the matrix cores)
  const int extra = wide && knobs().xtr_extras ? a.n_lanes - SPLIT_LANES : 0;
  if (wide && extra >= 1 && extra <= 2) {
    if (sample) hipLaunchKernelGGL(xtr18_sample_kernel, dim3(xb, yb), dim3(XTR_WAVES * 64), 0, s, a);
    else hipLaunchKernelGGL(xtr18_mfma_kernel, dim3(xb, yb), dim3(XTR_WAVES * 64), 0, s, a);
  } else if (wide && extra >= 3 && extra <= 4) {
    if (sample) hipLaunchKernelGGL(xtr20_sample_kernel, dim3(xb, yb), dim3(XTR_WAVES * 64), 0, s, a);
    else hipLaunchKernelGGL(xtr20_mfma_kernel, dim3(xb, yb), dim3(XTR_WAVES * 64), 0, s, a);
  } else if (wide) {
    if (sample) hipLaunchKernelGGL(xtr32_sample_kernel, dim3(xb, yb), dim3(XTR_WAVES * 64), 0, s, a);
    else hipLaunchKernelGGL(xtr32_mfma_kernel, dim3(xb, yb), dim3(XTR_WAVES * 64), 0, s, a);
  } else {
    if (sample) hipLaunchKernelGGL(xtr_sample_kernel, dim3(xb, yb), dim3(XTR_WAVES * 64), 0, s, a);
    else hipLaunchKernelGGL(xtr_mfma_kernel, dim3(xb, yb), dim3(XTR_WAVES * 64), 0, s, a);
  }
  return yb;
}

// the product of a covariance pass (cov_gz_mfma_kernel): xtr_mfma_kernel's grid; when only the working set's rows are read a
// workgroup row takes the next multiple of four of WS_KCAP / row blocks list entries (at most 32: eight steps in registers)
static int launch_cov_gz(int cus, SplitArgs& a, hipStream_t s, const CovBatch& cb, int n_sets) {
  const slm_host::XtrGrid g = slm_host::xtr_grid(a.n, a.ld, XTR_CB, xtr_max_row_blocks(cus, a.ld) / 2);
  const int xb = g.xb, yb = g.yb;
  a.xrows = g.rows;
  const int per = ((WS_KCAP + yb - 1) / yb + 3) / 4 * 4;
  a.xrows_ws = (a.ctl != nullptr && per <= 32) ? per : 0;
  if (a.r_plane != 0) hipLaunchKernelGGL(cov_gz32_mfma_kernel, dim3(xb, yb, (unsigned)n_sets), dim3(XTR_WAVES * 64), 0, s, a, cb);  // (both halves)
  else hipLaunchKernelGGL(cov_gz_mfma_kernel, dim3(xb, yb, (unsigned)n_sets), dim3(XTR_WAVES * 64), 0, s, a, cb);
  return yb;
}

const GradKernel* pick_grad_kernel(int64_t p2, int B) {
  if (p2 > kMaxChunks) return B == 1 ? &kGradTwoPass : nullptr;
  // LDS-ring variants: measured flat in B (0.60-0.61 ms for B = 1..4 at p = 5000) where the register
  // variants grow (0.598 / 0.599 / 0.615 / 0.733 ms on the same box), so they take over from B = 3.
  // SLM_GRAD_RING=0 disables them, =1 forces them for every B.
  const bool ring_off = knobs().grad_ring == 0, ring_all = knobs().grad_ring == 1;
  if (!ring_off && p2 > 256 && (B >= 3 || ring_all)) {
    for (const auto& k : kGradRing)
      if (k.B == B && 64LL * k.W * k.C >= p2) return &k;
  }
  if (knobs().grad_cfg[0] > 0) {  // (SLM_GRAD_CONFIG=W,C,R: tuning sweeps)
    const int W = knobs().grad_cfg[0], C = knobs().grad_cfg[1], R = knobs().grad_cfg[2];
    for (const auto& k : kGradDefault)
      if (k.B == B && k.W == W && k.C == C && k.R == R && 64LL * W * C >= p2) return &k;
    for (const auto& k : kGradExtra)
      if (k.B == B && k.W == W && k.C == C && k.R == R && 64LL * W * C >= p2) return &k;
  }
  for (const auto& k : kGradDefault)
    if (k.B == B && 64LL * k.W * k.C >= p2) return &k;
  return nullptr;
}

// ------------------------------------------------------------------------------------------------
// launches
// ------------------------------------------------------------------------------------------------
LaneSetup default_lanes(slm_dataset* ds, int B) {
  LaneSetup ls;
  ls.B = B;
  ls.rw = ds->rw;
  ls.rw_stride = 0;
  for (int l = 0; l < kMaxCells; ++l) ls.n_eff[l] = (double)ds->n_global;
  return ls;
}

// grad -> reduce (-> all-reduce) for B lanes on ONE pass over X:
// g_l = X^T W_l (X z_l - y) / n_eff_l in ds->g + l*(ld+16), loss_l in g_l[ld].
int enqueue_gradient(slm_dataset* ds, const LaneSetup& ls, const double* y, const int* done,
                     hipEvent_t ev_start, hipEvent_t ev_stop, int64_t n_rows, const int* skip) {
  hipStream_t s = ds->eng->stream;
  const int B = ls.B;
  const GradKernel* gk = ds->gk[B - 1];
  if (!gk) return fail(SLM_ERR_UNSUPPORTED, "no %d-lane gradient kernel covers p = %lld", B, (long long)ds->p);
  const int nblk = ds->nblk[B - 1];
  GradArgs a;
  a.X = ds->X;
  a.y = y;
  a.rw = ls.rw;
  a.z = ds->z;
  a.partial = ds->partial;
  a.loss_partial = ds->loss_partial;
  a.done = done;
  a.skip = skip;
  const int64_t nr = n_rows > 0 ? n_rows : ds->n;  // n_rows: only the first rows (sketched Lipschitz bound)
  a.n = nr;
  a.ld = ds->ld;
  a.rows_base = nr / nblk;
  a.rows_rem = nr % nblk;
  a.rw_stride = ls.rw_stride;
  a.p2 = (int)(ds->ld / 2);
  if (ev_start) HIP_TRY(hipEventRecord(ev_start, s));
  if (gk->D >= 0) {
    hipLaunchKernelGGL(gk->fn, dim3(nblk), dim3(gk->W * 64), 0, s, a);
  } else {  // two-pass fallback (one lane; row weights shared)
    TwoPassArgs t;
    t.X = ds->X; t.y = y; t.rw = ls.rw; t.z = a.z; t.r = ds->rvec; t.partial = a.partial;
    t.loss_partial = a.loss_partial; t.done = done; t.skip = skip; t.n = nr; t.ld = ds->ld;
    t.rows_base = a.rows_base; t.rows_rem = a.rows_rem; t.p2 = a.p2;
    hipLaunchKernelGGL(rowdot_kernel, dim3(nblk), dim3(256), 0, s, t);
    const unsigned tiles = (unsigned)((a.p2 + 512 * kTwoPassC - 1) / (512 * kTwoPassC));
    hipLaunchKernelGGL(xtr_kernel<kTwoPassC>, dim3(nblk, tiles), dim3(512), 0, s, t);
  }
  if (ev_stop) HIP_TRY(hipEventRecord(ev_stop, s));
  ReduceArgs ra;
  ra.partial = a.partial;
  ra.loss_partial = a.loss_partial;
  ra.g = ds->g;
  ra.done = done;
  ra.skip = skip;
  ra.nblk = nblk;
  ra.nblk_loss = nblk;
  ra.n_lanes = B;
  ra.ld = ds->ld;
  for (int l = 0; l < kMaxLanes; ++l) {
    const double ne = ls.n_eff[l] > 0 ? ls.n_eff[l] : (double)ds->n_global;
    ra.scale[l] = 1.0 / ne;
    ra.loss_scale[l] = 0.5 / ne;
  }
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)(ds->ld / 16 + 1), B), dim3(256), 0, s, ra);
  if (row_sharded(ds))  // also with one rank: keeps the RCCL path exercised on a single GPU
    SLM_TRY(all_reduce_sum(ds->eng, ds->g, (size_t)B * (size_t)(ds->ld + 16)));
  return SLM_OK;
}

// The same for working-set solves through the split pass: residuals (from the gathered columns where a
// lane's point is supported on W, from X otherwise), then X^T r for all sixteen lane slots on one read
// of X.  ctl == nullptr: every lane takes its residual from X (slm_gradient with SLM_GRAD_SPLIT=1).
// Residuals from X for the lanes the working set does not serve: all sixteen lane slots in one read of
// the column-major copy on the matrix cores when that copy exists (working-set solves make it), otherwise
// the vector kernel, five lanes per read of X (one window per grid row; a window returns at once unless
// one of its lanes needs X) -- also the choice for calls of up to five lanes.  SLM_ROWDOT_RING=1/0 forces one.
void launch_rowdot(slm_dataset* ds, const SplitKernel* sk, int nblk, int B, SplitArgs& a, hipStream_t s) {
  // Measured at n = 100k, p = 5k (tools/rowdot_probe.py): matrix cores 0.75-0.80 ms whatever the lane count;
  // vector kernel 0.62 ms for one lane, 0.81 ms for five, 3.1 ms for sixteen (four reads of X).
  const int halves = (B + SPLIT_LANES - 1) / SPLIT_LANES;
  const bool ring = sk->rowdot != nullptr && halves == 1 && (knobs().rowdot_ring >= 0 ? knobs().rowdot_ring == 1 : B <= ROWDOT_LANES);
  a.lane0 = 0;
  if ((!ring || sk->rowdot == nullptr) && ds->XT && ds->XT_ready) {
    a.XT = ds->XT;
    // (thirty-two lanes: both halves against ONE read of the copy -- rowdot32_mfma_kernel; SLM_ROWDOT32=0: a read per half)
    // (seventeen to twenty lanes: the lanes beyond sixteen on the vector units beside the matrix cores' sixteen -- rowdot18 /
    //  rowdot20_mfma_kernel, as for X^T R; SLM_XTR_EXTRAS=0: both halves on the matrix cores)
    const bool extras = halves == 2 && B <= SPLIT_LANES + 4 && knobs().xtr_extras;
    if (extras && B <= SPLIT_LANES + 2) hipLaunchKernelGGL(rowdot18_mfma_kernel, dim3(nblk, 1), dim3(XZ_WAVES * 64), 0, s, a);
    else if (extras) hipLaunchKernelGGL(rowdot20_mfma_kernel, dim3(nblk, 1), dim3(XZ_WAVES * 64), 0, s, a);
    else if (halves == 2 && knobs().rowdot32) hipLaunchKernelGGL(rowdot32_mfma_kernel, dim3(nblk, 1), dim3(XZ_WAVES * 64), 0, s, a);
    else hipLaunchKernelGGL(rowdot_mfma_kernel, dim3(nblk, halves), dim3(XZ_WAVES * 64), 0, s, a);
  } else if (sk->rowdot != nullptr) {
    hipLaunchKernelGGL(sk->rowdot, dim3(nblk, (B + ROWDOT_LANES - 1) / ROWDOT_LANES), dim3(sk->W * 64), 0, s, a);
  }
  // (no ring variant and no column-major copy: split_usable() keeps such datasets off the split pass)
}

// The split pass needs a kernel for the residuals that come from X: a ring variant, or the column-major copy.
bool split_usable(slm_dataset* ds) {
  if (!ds->sk) return false;
  if (ds->sk->rowdot != nullptr) return true;
  if (ensure_xt(ds) != SLM_OK) return false;
  return ds->XT != nullptr;
}

int enqueue_gradient_split(slm_dataset* ds, const LaneSetup& ls, const double* y, const int* done,
                           const PathCtl* ctl, const WsArgs* wa, hipEvent_t ev_start,
                           hipEvent_t ev_stop, int64_t n_rows, bool unit_bracket, const int* skip) {
  hipStream_t s = ds->eng->stream;
  const SplitKernel* sk = ds->sk;
  const int nblk = ds->split_nblk;
  if (!ds->R) {  // (a plane per half of the lanes)
    SLM_TRY(dalloc(&ds->R, (size_t)ds->n * SPLIT_RSTRIDE * SPLIT_HALVES));
    HIP_TRY(hipMemsetAsync(ds->R, 0, sizeof(double) * (size_t)ds->n * SPLIT_RSTRIDE * SPLIT_HALVES, s));
  }
  const int halves = (ls.B + SPLIT_LANES - 1) / SPLIT_LANES;
  if (halves > 1 && !(ds->XT && ds->XT_ready)) return fail(SLM_ERR_UNSUPPORTED, "more than %d lanes need the column-major copy of X", SPLIT_LANES);
  SplitArgs a;
  memset(&a, 0, sizeof(a));
  a.X = ds->X; a.y = y; a.rw = ls.rw; a.rw_stride = ls.rw_stride; a.z = ds->z; a.R = ds->R;
  a.lane_slots = SPLIT_LANES * halves; a.r_plane = (int64_t)ds->n * SPLIT_RSTRIDE;
  a.partial = ds->partial; a.loss_partial = ds->loss_partial; a.done = done; a.ctl = ctl; a.skip = skip;
  if (wa) { a.XW = wa->XW; a.idx = wa->idx; a.ws = wa->ws; }
  const int64_t nr = n_rows > 0 ? n_rows : ds->n;
  a.n = nr; a.ld = ds->ld; a.rows_base = nr / nblk; a.rows_rem = nr % nblk;
  a.p2 = (int)(ds->ld / 2);
  a.n_lanes = ls.B;
  // (SLM_PROFILE_UNIT=1: the bracket of SLM_FLAG_PROFILE opens here -- the whole gradient unit, residuals and X^T R, not the
  //  stream over X alone: bench.py's roofline.gradient_unit_frac)
  const bool unit = ev_start != nullptr && (unit_bracket || knobs().profile_unit);
  if (unit) HIP_TRY(hipEventRecord(ev_start, s));
  launch_rowdot(ds, sk, nblk, ls.B, a, s);
  if (wa && ctl) {  // residuals from the gathered columns: matrix cores (SLM_RESID_VEC=1: a row per thread)
    if (knobs().resid_vec && halves == 1) hipLaunchKernelGGL(sk->resid, dim3(nblk), dim3(256), 0, s, a);
    else if (halves == 2 && knobs().resid32) hipLaunchKernelGGL(resid32_mfma_kernel, dim3(nblk, 1), dim3(RM_WAVES * 64), 0, s, a);  // (both halves on one read of the gathered columns)
    else hipLaunchKernelGGL(resid_mfma_kernel, dim3(nblk, halves), dim3(RM_WAVES * 64), 0, s, a);
  }
  // (SLM_FLAG_PROFILE brackets the kernel that streams X, the one the roofline is quoted on)
  if (ev_start && !unit) HIP_TRY(hipEventRecord(ev_start, s));
  const int xblk = launch_xtr(ds->eng->cus, a, s, n_rows > 0 && ctl != nullptr);  // (rows of a sample start: solve_core)
  if (ev_stop) HIP_TRY(hipEventRecord(ev_stop, s));
  ReduceArgs ra;
  ra.partial = ds->partial;
  ra.loss_partial = ds->loss_partial;
  ra.g = ds->g;
  ra.done = done;
  ra.nblk = xblk;
  ra.nblk_loss = nblk;
  ra.skip = skip;
  ra.n_lanes = a.lane_slots;  // partial rows are laid out for all lane slots of the split pass
  ra.ld = ds->ld;
  for (int l = 0; l < kMaxLanes; ++l) {
    const double ne = ls.n_eff[l] > 0 ? ls.n_eff[l] : (double)ds->n_global;
    ra.scale[l] = 1.0 / ne;
    ra.loss_scale[l] = 0.5 / ne;
  }
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)(ds->ld / 16 + 1), ls.B), dim3(256), 0, s, ra);
  if (row_sharded(ds))  // row-sharded: sum the gradients (and losses) of the row blocks over ranks
    SLM_TRY(all_reduce_sum(ds->eng, ds->g, (size_t)ls.B * (size_t)(ds->ld + 16)));
  return SLM_OK;
}

// The gradient of one pass from the Grams of the lanes' row sets (cov_kernels.hpp): g_l = G_s z_l - c_s, loss in g_l[ld].
// `entry_of[l]`: the lane's entry of ds->cov.  One read of a 8 ld^2-byte Gram per row set of the call instead of X.
int enqueue_gradient_cov(slm_dataset* ds, int B, const int* entry_of, const int* done, hipEvent_t ev_start,
                         hipEvent_t ev_stop, const PathCtl* ctl, const WsArgs* wa) {
  hipStream_t s = ds->eng->stream;
  const int64_t ld = ds->ld;
  const int halves = (B + SPLIT_LANES - 1) / SPLIT_LANES;  // (more than sixteen lanes: a plane of Z, a block of partial sums and a launch per half)
  hipLaunchKernelGGL(cov_pack_kernel, dim3((unsigned)((ld * SPLIT_RSTRIDE + 255) / 256), (unsigned)halves), dim3(256), 0, s, ds->z, ld, B, ds->cov_Z, done);
  if (ev_start) HIP_TRY(hipEventRecord(ev_start, s));
  // the row sets of the call, in order of their first lane
  CovBatch cb;
  memset(&cb, 0, sizeof(cb));
  int n_sets = 0, entry_of_set[SLM_MAX_LANES];
  for (int l = 0; l < B; ++l) {
    int st = -1;
    for (int k = 0; k < n_sets && st < 0; ++k)
      if (entry_of_set[k] == entry_of[l]) st = k;
    if (st < 0) {
      st = n_sets++;
      entry_of_set[st] = entry_of[l];
      const slm_dataset::CovEntry& e = ds->cov[(size_t)entry_of[l]];
      cb.G[st] = e.G; cb.c[st] = e.c; cb.yy[st] = e.yy;
    }
    cb.set_of[l] = st;
  }
  // partial sums: one block of [row blocks][16][ld] per row set (the gradient's own buffer holds one or two)
  const int64_t blocks_most = std::max<int64_t>(1, xtr_max_row_blocks(ds->eng->cus, ld) / 2);
  cb.part_stride = blocks_most * SPLIT_LANES * ld;
  cb.half_stride = (int64_t)n_sets * cb.part_stride;
  const int blocks = halves * n_sets;
  double* partial = ds->partial;
  if ((size_t)blocks * (size_t)cb.part_stride > ds->partial_elems) {
    if (ds->cov_partial_sets < blocks) {
      dfree(ds->cov_partial);
      ds->cov_partial_sets = 0;
      SLM_TRY(dalloc(&ds->cov_partial, (size_t)blocks * (size_t)cb.part_stride));
      ds->cov_partial_sets = blocks;
    }
    partial = ds->cov_partial;
  }
  SplitArgs a;
  memset(&a, 0, sizeof(a));
  a.R = ds->cov_Z; a.partial = partial; a.done = done;
  a.n = ld; a.ld = ld; a.p2 = (int)(ld / 2); a.n_lanes = B;
  // (points the model solver produced are zero outside the working set: only its rows of G are read then)
  if (ctl && wa && wa->ws && !knobs().cov_all_rows) { a.ctl = ctl; a.ws = wa->ws; a.idx = wa->idx; }
  // (more than sixteen lanes: both planes of Z against ONE read of every Gram -- cov_gz32_mfma_kernel)
  a.lane0 = 0;
  a.r_plane = halves > 1 ? ld * SPLIT_RSTRIDE : 0;
  const int xblk = launch_cov_gz(ds->eng->cus, a, s, cb, n_sets);
  CovFinishArgs f;
  f.partial = partial; f.z = ds->z; f.g = ds->g; f.done = done; f.nblk = xblk; f.ld = ld;
  hipLaunchKernelGGL(cov_reduce_kernel, dim3((unsigned)(ld / 16), (unsigned)B), dim3(256), 0, s, f, cb);
  hipLaunchKernelGGL(cov_loss_kernel, dim3((unsigned)B), dim3(256), 0, s, f, cb);
  if (ev_stop) HIP_TRY(hipEventRecord(ev_stop, s));
  return SLM_OK;
}

// fingerprints of row-weight vectors already on the device -> host (one small copy, one wait)
int cov_fingerprints(slm_dataset* ds, const double* const* w, int count, double* out /* [2 * count] */) {
  hipStream_t s = ds->eng->stream;
  if (!ds->cov_fp) SLM_TRY(dalloc(&ds->cov_fp, 2 * (size_t)kMaxLanes + 2));
  CovFpArgs fa;
  memset(&fa, 0, sizeof(fa));
  for (int u = 0; u < count; ++u) fa.w[u] = w[u];
  hipLaunchKernelGGL(cov_fingerprint_kernel, dim3((unsigned)count), dim3(1024), 0, s, fa, ds->n, ds->cov_fp);
  HIP_TRY(hipMemcpyAsync(out, ds->cov_fp, sizeof(double) * 2 * (size_t)count, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return SLM_OK;
}

int cov_find(const slm_dataset* ds, double fp1, double fp2, double n_eff) { return slm_host::find_by_fingerprint(ds->cov, fp1, fp2, n_eff); }

// E = features per thread of the one-workgroup-per-lane tail kernel (p <= 1024 * E).  Up to E = 6 everything a thread
// needs of its features stays in registers; rows of more than 6 144 columns take the streaming form of the same kernel,
// which keeps nothing per feature across a workgroup sum and so needs no scratch memory at any p.
void launch_tail(const TailArgs& ta, hipStream_t s) {
  const int E = (ta.p + TAIL_THREADS - 1) / TAIL_THREADS;
  const dim3 grid(ta.n_lanes);
#define SLM_TAIL_LAUNCH(N) hipLaunchKernelGGL(fista_tail_kernel<N>, grid, dim3(TAIL_THREADS), 0, s, ta)
  switch (E) {
    case 1: SLM_TAIL_LAUNCH(1); break;
    case 2: SLM_TAIL_LAUNCH(2); break;
    case 3: SLM_TAIL_LAUNCH(3); break;
    case 4: SLM_TAIL_LAUNCH(4); break;
    case 5: SLM_TAIL_LAUNCH(5); break;
    case 6: SLM_TAIL_LAUNCH(6); break;
    case 7: hipLaunchKernelGGL(fista_tail_stream_kernel<7>, grid, dim3(TAIL_THREADS), 0, s, ta); break;
    case 8: hipLaunchKernelGGL(fista_tail_stream_kernel<8>, grid, dim3(TAIL_THREADS), 0, s, ta); break;
    case 9: hipLaunchKernelGGL(fista_tail_stream_kernel<9>, grid, dim3(TAIL_THREADS), 0, s, ta); break;
    case 10: hipLaunchKernelGGL(fista_tail_stream_kernel<10>, grid, dim3(TAIL_THREADS), 0, s, ta); break;
    default: hipLaunchKernelGGL(fista_tail_stream_kernel<0>, grid, dim3(TAIL_THREADS), 0, s, ta);
  }
#undef SLM_TAIL_LAUNCH
}

int check_launch() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(SLM_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
  return SLM_OK;
}

// ------------------------------------------------------------------------------------------------
// Lipschitz constants: lambda_max(X^T W_l X)/n_eff_l for every lane of `ls` in one batched run
// ------------------------------------------------------------------------------------------------
// Power steps used for the seed L of a solve.  The spectral scheme only needs the right order of
// magnitude (it measures curvature along its own steps) and FISTA's curvature guard repairs an
// under-estimate, so a handful of passes is enough; slm_dataset_lipschitz() asks for more.
// Power steps on the sketch (a thirty-second of the rows) for working-set solves: ONE.  The sketch's lambda_max is 2-3 x the
// whole matrix's, and one step from the fixed start lands at about half of the sketch's: closer to the truth than three
// steps' estimate, 80 us cheaper per solve (a launch chain of nine), and these solves only use L for a first candidate
// and for fallback steps whose curvature guard repairs an under-estimate.  Round 3, same box, alternating: headline
// 4.17-4.25 ms with three steps, 4.13-4.16 with two, 4.09-4.12 with one; passes of the headline, configs 3 / 4 and the
// sparse-regime soak paths unchanged, dense-regime soak paths -3 ... +3 passes of 24-60 (SLM_L_SKETCH_ITERS).
static const int kPowerItersSketchDefault = 1;
int sketch_iters() {
  return knobs().l_sketch_iters;  // (SLM_L_SKETCH_ITERS; default kPowerItersSketchDefault)
}
static const int kPowerItersQuery = 16;
// (a thirty-second of the rows: the bound is looser than from a sixteenth -- lambda_max of a sketch grows as it
// shrinks -- and nothing downstream noticed down to a sixty-fourth, SLM_L_SKETCH_DIV; three steps on
// 3 125 of 100 000 rows cost 0.10 ms where a sixteenth cost 0.17)
int64_t sketch_rows(int64_t n) {
  return std::max<int64_t>(1, n / knobs().l_sketch_div);  // (SLM_L_SKETCH_DIV; 32)
}

// n_rows > 0: the operator of the first n_rows rows only, X_S^T W X_S / (n_eff n_rows / n).  Its largest
// eigenvalue is, in expectation, no smaller than that of the full operator (Jensen: lambda_max is
// convex and E G_S = G for exchangeable rows), so it serves as a cheap step-size bound where the
// iteration does not depend on a tight one (working-set solves); the curvature guards cover the rest.
int power_iteration(slm_dataset* ds, const LaneSetup& ls_in, double* L_out /*[B]*/, int iters, int64_t n_rows) {
  LaneSetup ls = ls_in;
  if (n_rows > 0) {
    for (int l = 0; l < kMaxLanes; ++l) {
      const double ne = ls.n_eff[l] > 0 ? ls.n_eff[l] : (double)ds->n_global;
      ls.n_eff[l] = ne * (double)n_rows / (double)ds->n;
    }
  }
  hipStream_t s = ds->eng->stream;
  if (knobs().power_iters > 0) iters = knobs().power_iters;
  hipLaunchKernelGGL(power_init_kernel, dim3(ls.B), dim3(TAIL_THREADS), 0, s, ds->z, (int)ds->p, ds->ld);
  for (int k = 0; k < iters; ++k) {
    if (ds->gk[ls.B - 1]) {
      SLM_TRY(enqueue_gradient(ds, ls, ds->yzero, nullptr, nullptr, nullptr, n_rows));
    } else {  // more lanes than the fused kernels serve (working-set solves): the split pass has sixteen
      if (!split_usable(ds)) return fail(SLM_ERR_UNSUPPORTED, "no %d-lane kernel for p = %lld", ls.B, (long long)ds->p);
      SLM_TRY(enqueue_gradient_split(ds, ls, ds->yzero, nullptr, nullptr, nullptr, nullptr, nullptr, n_rows));
    }
    PowerArgs pa;
    pa.g = ds->g;
    pa.v = ds->z;
    pa.lambda = ds->lambda;
    pa.p = (int)ds->p;
    pa.ld = ds->ld;
    hipLaunchKernelGGL(power_step_kernel, dim3(ls.B), dim3(TAIL_THREADS), 0, s, pa);
  }
  SLM_TRY(check_launch());
  if (!L_out) return SLM_OK;  // the caller consumes ds->lambda on the device (seed_step_kernel)
  double lam[SLM_MAX_LANES] = {};
  HIP_TRY(hipMemcpyAsync(lam, ds->lambda, sizeof(double) * ls.B, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  for (int l = 0; l < ls.B; ++l) {
    if (!std::isfinite(lam[l])) return fail(SLM_ERR_NON_FINITE, "power iteration produced a non-finite value");
    // ||A v|| after k steps under-estimates lambda_max by a few per cent on flat spectra; the margin
    // below plus the in-loop curvature guard (fista_tail_kernel) keep the step 1/L safe.
    double L = lam[l] * 1.08;
    // (zero: X == 0 on the rows used -- the whole matrix, or, for a sketch, a window the lane's row mask
    //  blanks out: the caller then repeats with all rows before settling for 1)
    if (!(L > 0.0)) L = (n_rows > 0) ? 0.0 : 1.0;
    L_out[l] = L;
  }
  return SLM_OK;
}

int estimate_lipschitz(slm_dataset* ds, double* L_out, int iters) {
  if (!ds->L_valid || ds->L_iters < iters) {
    double L[SLM_MAX_LANES];
    SLM_TRY(power_iteration(ds, default_lanes(ds, 1), L, iters));
    ds->L = L[0];
    ds->L_iters = iters;
    ds->L_valid = true;
  }
  *L_out = ds->L;
  return SLM_OK;
}

extern "C" int slm_dataset_lipschitz(slm_dataset* ds, double* L_out) {
  if (!ds || !L_out) return fail(SLM_ERR_BAD_ARG, "NULL argument");
  HIP_TRY(hipSetDevice(ds->eng->device));
  return estimate_lipschitz(ds, L_out, kPowerItersQuery);
}

// ------------------------------------------------------------------------------------------------
// in-place centring by the row-weighted means
// ------------------------------------------------------------------------------------------------
extern "C" int slm_dataset_center(slm_dataset* ds, double* x_mean_out, double* y_mean_out) {
  if (!ds) return fail(SLM_ERR_BAD_ARG, "dataset is NULL");
  HIP_TRY(hipSetDevice(ds->eng->device));
  hipStream_t s = ds->eng->stream;
  // (X and y change in place: the Grams of covariance passes, the Gram of all rows included, go with the old values)
  cov_pending_drop(ds);
  ds->cov.clear();
  ds->cov_all_hold.reset();
  ds->cov_all = nullptr;
  const int64_t n = ds->n, p = ds->p, ld = ds->ld;
  // sum w, sum w y -- of ALL rows: a row-sharded dataset adds its ranks' sums here and its ranks' X_r^T w_r
  // in the gradient launch below (one all-reduce each), so every rank subtracts the global means
  // (reference model/_base.py:216-222 on the whole matrix)
  hipLaunchKernelGGL(weighted_sums_kernel, dim3(1), dim3(1024), 0, s, ds->y, ds->rw, n, ds->lambda);
  if (row_sharded(ds)) SLM_TRY(all_reduce_sum(ds->eng, ds->lambda, 2));
  double sums[2] = {0.0, 0.0};
  HIP_TRY(hipMemcpyAsync(sums, ds->lambda, sizeof(double) * 2, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (!(sums[0] > 0.0)) return fail(SLM_ERR_BAD_ARG, "row weights sum to zero");
  const double ymean = sums[1] / sums[0];
  // x_mean = X^T w / sum w: the gradient kernel with z = 0 and y = -1 (borrowing yzero)
  const int blocks = (int)std::min<int64_t>(4096, (n + 255) / 256);
  hipLaunchKernelGGL(fill_kernel, dim3(blocks), dim3(256), 0, s, ds->yzero, n, -1.0);
  HIP_TRY(hipMemsetAsync(ds->z, 0, sizeof(double) * ld, s));
  LaneSetup ls = default_lanes(ds, 1);
  ls.n_eff[0] = sums[0];
  SLM_TRY(enqueue_gradient(ds, ls, ds->yzero, nullptr, nullptr, nullptr));
  HIP_TRY(hipMemsetAsync(ds->yzero, 0, sizeof(double) * n, s));
  // subtract (g holds x_mean; pad entries are exactly zero)
  const int cblocks = ds->eng->cus * 8;
  hipLaunchKernelGGL(center_kernel, dim3(cblocks), dim3(256), 0, s, ds->X, ds->y, n, p, ld, ds->g, ymean);
  SLM_TRY(check_launch());
  if (x_mean_out) HIP_TRY(hipMemcpyAsync(x_mean_out, ds->g, sizeof(double) * p, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (y_mean_out) *y_mean_out = ymean;
  ds->L_valid = false;
  ds->sketch_valid = false;
  ds->carry_valid = false;
  ds->XT_ready = false;  // X changed in place: the column-major copy is rebuilt on next use
  mg_invalidate(ds);     // ... and so is the model Gram
  return SLM_OK;
}

// ------------------------------------------------------------------------------------------------
// single gradient evaluation (tests, alpha_max, roofline probe)
// ------------------------------------------------------------------------------------------------
// a timed loop of launches between two events on the engine's stream: mean ms per repetition
template <typename Launch>
static int timed_launches(hipStream_t s, int reps, double* ms_out, Launch launch) {
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  HIP_TRY(hipEventRecord(e0, s));
  int rc = SLM_OK;
  for (int r = 0; r < reps && rc == SLM_OK; ++r) rc = launch();
  HIP_TRY(hipEventRecord(e1, s));
  HIP_TRY(hipEventSynchronize(e1));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
  *ms_out = (double)ms / reps;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  SLM_TRY(rc);
  return check_launch();
}

extern "C" int slm_gradient_ex(slm_dataset* ds, const double* z, const slm_gradient_opts* opts, double* g_out, double* loss_out,
                               int32_t reps, double* ms_out) {
  if (!ds) return fail(SLM_ERR_BAD_ARG, "dataset is NULL");
  slm_gradient_opts o;
  memset(&o, 0, sizeof(o));
  if (opts) o = *opts;
  if (o.route < 0 || o.route > 1) return fail(SLM_ERR_BAD_ARG, "route must be 0 (fused) or 1 (split pass), got %d", o.route);
  if (o.n_lanes > kMaxLanes || o.probe_lanes > kMaxLanes) return fail(SLM_ERR_BAD_ARG, "at most %d lanes", kMaxLanes);
  HIP_TRY(hipSetDevice(ds->eng->device));
  hipStream_t s = ds->eng->stream;
  HIP_TRY(hipMemsetAsync(ds->z, 0, sizeof(double) * ds->ld, s));
  if (z) HIP_TRY(hipMemcpyAsync(ds->z, z, sizeof(double) * ds->p, hipMemcpyHostToDevice, s));
  const bool use_split = o.route == 1 && split_usable(ds);
  if (use_split) SLM_TRY(ensure_xt(ds));  // (so that tests and probes reach rowdot_mfma_kernel; optional copy)
  // (the split pass's wider forms: n_lanes lanes all at z, the gradient of lane lane_out returned)
  int lanes_run = 1, lane_out = 0;
  if (use_split && ds->XT && ds->XT_ready && (size_t)ds->lane_cap >= (size_t)kMaxLanes) {
    lanes_run = std::max(1, (int)o.n_lanes);
    lane_out = std::min(lanes_run - 1, std::max(0, (int)o.lane_out));
  }
  auto spread_z = [&](int B) -> int {  // lanes 1 .. B - 1 stand where lane 0 stands
    for (int l = 1; l < B; ++l)
      HIP_TRY(hipMemcpyAsync(ds->z + (size_t)l * ds->ld, ds->z, sizeof(double) * ds->ld, hipMemcpyDeviceToDevice, s));
    return SLM_OK;
  };
  const LaneSetup ls = default_lanes(ds, lanes_run);
  SLM_TRY(spread_z(lanes_run));
  if (use_split) SLM_TRY(enqueue_gradient_split(ds, ls, ds->y, nullptr, nullptr, nullptr, nullptr, nullptr));
  else SLM_TRY(enqueue_gradient(ds, ls, ds->y, nullptr, nullptr, nullptr));
  SLM_TRY(check_launch());
  HIP_TRY(hipStreamSynchronize(s));
  const double* g_lane = ds->g + (size_t)lane_out * (size_t)(ds->ld + 16);
  if (g_out) HIP_TRY(hipMemcpy(g_out, g_lane, sizeof(double) * ds->p, hipMemcpyDeviceToHost));
  if (loss_out) HIP_TRY(hipMemcpy(loss_out, g_lane + ds->ld, sizeof(double), hipMemcpyDeviceToHost));
  if (!ms_out) return SLM_OK;
  // ---- the roofline probe: `reps` launches of the route's kernels, timed by HIP events on the engine's stream -----------
  *ms_out = 0.0;
  if (reps < 1) reps = 1;
  const int B = std::max(1, (int)o.probe_lanes);
  SLM_TRY(spread_z(B));
  const LaneSetup lb = default_lanes(ds, B);
  if (use_split) {  // residuals from X + X^T R (or X^T R alone)
    SLM_TRY(enqueue_gradient_split(ds, lb, ds->y, nullptr, nullptr, nullptr, nullptr, nullptr));  // warm; allocates R
    SplitArgs a;
    memset(&a, 0, sizeof(a));
    a.X = ds->X; a.y = ds->y; a.rw = lb.rw; a.rw_stride = 0; a.z = ds->z; a.R = ds->R;
    a.partial = ds->partial; a.loss_partial = ds->loss_partial;
    a.n = ds->n; a.ld = ds->ld; a.rows_base = ds->n / ds->split_nblk; a.rows_rem = ds->n % ds->split_nblk;
    a.p2 = (int)(ds->ld / 2); a.n_lanes = B;
    a.lane_slots = SPLIT_LANES * ((B + SPLIT_LANES - 1) / SPLIT_LANES); a.r_plane = (int64_t)ds->n * SPLIT_RSTRIDE;
    return timed_launches(s, reps, ms_out, [&]() -> int {
      if (!o.xtr_only) launch_rowdot(ds, ds->sk, ds->split_nblk, B, a, s);
      (void)launch_xtr(ds->eng->cus, a, s);
      return SLM_OK;
    });
  }
  const GradKernel* gk = ds->gk[B - 1];
  if (!gk) return fail(SLM_ERR_UNSUPPORTED, "no %d-lane kernel for p = %lld", B, (long long)ds->p);
  if (gk->D < 0)  // two-pass fallback: the pair of kernels through the common path
    return timed_launches(s, reps, ms_out, [&]() -> int { return enqueue_gradient(ds, lb, ds->y, nullptr, nullptr, nullptr); });
  const int nblk = ds->nblk[B - 1];
  GradArgs a;
  a.X = ds->X; a.y = ds->y; a.rw = lb.rw; a.z = ds->z; a.partial = ds->partial;
  a.loss_partial = ds->loss_partial; a.done = nullptr; a.n = ds->n; a.ld = ds->ld;
  a.rows_base = ds->n / nblk; a.rows_rem = ds->n % nblk; a.rw_stride = 0; a.p2 = (int)(ds->ld / 2);
  hipLaunchKernelGGL(gk->fn, dim3(nblk), dim3(gk->W * 64), 0, s, a);  // warm
  return timed_launches(s, reps, ms_out, [&]() -> int {
    hipLaunchKernelGGL(gk->fn, dim3(nblk), dim3(gk->W * 64), 0, s, a);
    return SLM_OK;
  });
}

extern "C" int slm_gradient(slm_dataset* ds, const double* z, double* g_out, double* loss_out, int32_t reps, double* ms_out) {
  return slm_gradient_ex(ds, z, nullptr, g_out, loss_out, reps, ms_out);
}

// ------------------------------------------------------------------------------------------------
// the read-only stream ceiling of this device, measured on the dataset's own copy of X
// ------------------------------------------------------------------------------------------------
extern "C" int slm_dataset_read_ceiling(slm_dataset* ds, int32_t reps, double* gbs_out, double* ms_out) {
  if (!ds || !gbs_out) return fail(SLM_ERR_BAD_ARG, "NULL argument");
  if (reps < 1) reps = 1;
  HIP_TRY(hipSetDevice(ds->eng->device));
  hipStream_t s = ds->eng->stream;
  const int64_t count2 = ds->n * ds->ld / 2;  // (ld is a multiple of 16)
  const int cus = ds->eng->cus;
  double* sink = nullptr;
  SLM_TRY(dalloc(&sink, (size_t)cus * 16));
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  typedef void (*Kernel)(const double*, int64_t, double*);
  struct Form { Kernel fn; int per_cu; const char* name; };
  const Form forms[] = {
      {read_stream_kernel<8, true>, 1, "1 workgroup per CU, 8 loads in flight, non-temporal"},
      {read_stream_kernel<8, true>, 2, "2 per CU, 8, non-temporal"},
      {read_stream_kernel<8, true>, 4, "4 per CU, 8, non-temporal"},
      {read_stream_kernel<16, true>, 1, "1 per CU, 16, non-temporal"},
      {read_stream_kernel<16, true>, 2, "2 per CU, 16, non-temporal"},
      {read_stream_kernel<8, false>, 2, "2 per CU, 8, cached"},
      {read_stream_kernel<8, false>, 4, "4 per CU, 8, cached"},
      {read_stream_kernel<4, true>, 8, "8 per CU, 4, non-temporal"},
      {read_stream_kernel<8, true, true>, 1, "1 per CU, 8, non-temporal, chunks in turn"},
      {read_stream_kernel<8, true, true>, 2, "2 per CU, 8, non-temporal, chunks in turn"},
      {read_stream_kernel<8, true, true>, 4, "4 per CU, 8, non-temporal, chunks in turn"},
      {read_stream_kernel<16, true, true>, 2, "2 per CU, 16, non-temporal, chunks in turn"},
      {read_stream_kernel<4, false, true>, 8, "8 per CU, 4, cached, chunks in turn"},
  };
  double best = 1e300;
  const char* best_name = "";
  const bool trace = knobs().trace != 0;
  for (const Form& f : forms) {
    const int blocks = cus * f.per_cu;
    hipLaunchKernelGGL(f.fn, dim3(blocks), dim3(256), 0, s, (const double*)ds->X, count2, sink);  // warm
    HIP_TRY(hipEventRecord(e0, s));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(f.fn, dim3(blocks), dim3(256), 0, s, (const double*)ds->X, count2, sink);
    HIP_TRY(hipEventRecord(e1, s));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    const double per = (double)ms / reps;
    if (trace) fprintf(stderr, "[slm] read stream, %s: %.4f ms per sweep = %.0f GB/s\n", f.name, per, 16.0 * (double)count2 / (per * 1e-3) / 1e9);
    if (per < best) {
      best = per;
      best_name = f.name;
    }
  }
  (void)best_name;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  dfree(sink);
  SLM_TRY(check_launch());
  if (ms_out) *ms_out = best;
  *gbs_out = 16.0 * (double)count2 / (best * 1e-3) / 1e9;
  return SLM_OK;
}

// ------------------------------------------------------------------------------------------------
// hold-out scoring: weighted SSE of m coefficient vectors, SLM_MAX_LANES per pass over X
// ------------------------------------------------------------------------------------------------
extern "C" int slm_eval_sse(slm_dataset* ds, const double* Z, int32_t m, const double* row_weight,
                            double* sse_out) {
  if (!ds || !Z || !sse_out) return fail(SLM_ERR_BAD_ARG, "NULL argument");
  if (m <= 0) return fail(SLM_ERR_BAD_ARG, "m must be positive");
  HIP_TRY(hipSetDevice(ds->eng->device));
  hipStream_t s = ds->eng->stream;
  const int64_t n = ds->n, p = ds->p, ld = ds->ld;
  for (int64_t k = 0; k < (int64_t)m * p; ++k)
    if (!std::isfinite(Z[k])) return fail(SLM_ERR_BAD_ARG, "Z contains a non-finite value");
  LaneSetup ls = default_lanes(ds, 1);
  if (row_weight) {
    for (int64_t i = 0; i < n; ++i)
      if (!(row_weight[i] >= 0.0) || !std::isfinite(row_weight[i]))
        return fail(SLM_ERR_BAD_ARG, "row_weight[%lld] is negative or not finite", (long long)i);
    if (!ds->rw_lanes) SLM_TRY(dalloc(&ds->rw_lanes, (size_t)ds->lane_cap * n));
    HIP_TRY(hipMemcpyAsync(ds->rw_lanes, row_weight, sizeof(double) * n, hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));
    ls.rw = ds->rw_lanes;
    ls.rw_stride = 0;  // every lane reads the same mask
  }
  for (int l = 0; l < kMaxLanes; ++l) ls.n_eff[l] = 0.5;  // loss_scale = 1/(2 n_eff) = 1  =>  g[ld] = SSE
  int maxB = SPLIT_LANES;
  while (maxB > 1 && !ds->gk[maxB - 1]) --maxB;
  std::vector<double> losses(kMaxLanes);
  // More vectors than a fused pass takes (four at p = 5 000): the residual half of the split pass forms X z - y for SIXTEEN
  // per read of X and leaves the blocks' sums of w e^2 behind -- all a score needs.  (The dense ends of a grid's paths,
  // whose joint support is beyond slm_eval_sse_sparse: 50 candidates of a fold in 4 reads instead of 13.)
  if (m > maxB && ds->sk != nullptr && !row_sharded(ds) && split_usable(ds) && !knobs().eval_fused &&
      (ds->sk->rowdot != nullptr || (ensure_xt(ds) == SLM_OK && ds->XT && ds->XT_ready))) {
    const int nblk = ds->split_nblk;
    if (!ds->R) {
      SLM_TRY(dalloc(&ds->R, (size_t)n * SPLIT_RSTRIDE * SPLIT_HALVES));
      HIP_TRY(hipMemsetAsync(ds->R, 0, sizeof(double) * (size_t)n * SPLIT_RSTRIDE * SPLIT_HALVES, s));
    }
    for (int32_t k0 = 0; k0 < m; k0 += SPLIT_LANES) {  // (sixteen vectors per read of the copy: one half of the lane slots)
      const int B = std::min<int32_t>(SPLIT_LANES, m - k0);
      HIP_TRY(hipMemsetAsync(ds->z, 0, sizeof(double) * SPLIT_LANES * ld, s));
      HIP_TRY(hipMemcpy2DAsync(ds->z, sizeof(double) * ld, Z + (size_t)k0 * p, sizeof(double) * p, sizeof(double) * p, B, hipMemcpyHostToDevice, s));
      SplitArgs a;
      memset(&a, 0, sizeof(a));
      a.X = ds->X; a.y = ds->y; a.rw = ls.rw; a.rw_stride = ls.rw_stride; a.z = ds->z; a.R = ds->R;
      a.partial = ds->partial; a.loss_partial = ds->loss_partial;
      a.n = n; a.ld = ld; a.rows_base = n / nblk; a.rows_rem = n % nblk;
      a.p2 = (int)(ld / 2);
      a.n_lanes = B;
      a.lane_slots = SPLIT_LANES; a.r_plane = (int64_t)n * SPLIT_RSTRIDE;
      launch_rowdot(ds, ds->sk, nblk, B, a, s);
      hipLaunchKernelGGL(sse_from_blocks_kernel, dim3(1), dim3(64), 0, s, ds->loss_partial, nblk, SPLIT_LANES, ds->partial);
      SLM_TRY(check_launch());
      HIP_TRY(hipMemcpyAsync(losses.data(), ds->partial, sizeof(double) * SPLIT_LANES, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
      for (int l = 0; l < B; ++l) sse_out[k0 + l] = losses[l];
    }
    return SLM_OK;
  }
  for (int32_t k0 = 0; k0 < m; k0 += maxB) {
    const int B = std::min<int32_t>(maxB, m - k0);  // kernel variants exist for every B <= maxB
    ls.B = B;
    HIP_TRY(hipMemsetAsync(ds->z, 0, sizeof(double) * SPLIT_LANES * ld, s));
    for (int l = 0; l < B; ++l)
      HIP_TRY(hipMemcpyAsync(ds->z + (size_t)l * ld, Z + (size_t)(k0 + l) * p, sizeof(double) * p,
                             hipMemcpyHostToDevice, s));
    SLM_TRY(enqueue_gradient(ds, ls, ds->y, nullptr, nullptr, nullptr));
    SLM_TRY(check_launch());
    for (int l = 0; l < B; ++l)
      HIP_TRY(hipMemcpyAsync(&losses[l], ds->g + (size_t)l * (ld + 16) + ld, sizeof(double),
                             hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    for (int l = 0; l < B; ++l) sse_out[k0 + l] = losses[l];
  }
  return SLM_OK;
}

// column-major copy of X for column gathers (see solve_core / ws_setup); optional (no memory: nullptr)
int ensure_xt(slm_dataset* ds) {
  hipStream_t s = ds->eng->stream;
  const int64_t n = ds->n, ld = ds->ld;
  const int64_t row_tiles = (n + 31) / 32;
  if (!ds->XT && !ds->XT_failed) {
    if (pool_malloc((void**)&ds->XT, sizeof(double) * (size_t)ld * (size_t)row_tiles * 32) != hipSuccess) {
      (void)hipGetLastError();
      ds->XT = nullptr;
      ds->XT_failed = true;
    }
  }
  if (ds->XT && !ds->XT_ready) {
    ds->XT_ready = true;
    ds->colnorm_ready = false;
    const dim3 grid((unsigned)row_tiles, (unsigned)((ld + 31) / 32));
    hipLaunchKernelGGL(tile_columns_kernel, grid, dim3(256), 0, s, (const double*)ds->X, n, ld, ds->XT);
  }
  // the column norms certified partial passes bound with (light_kernels.hpp): one read of the copy, kept beside it
  // (unweighted own rows of one device: the only datasets such passes serve)
  if (ds->XT && ds->XT_ready && !ds->colnorm_ready && !ds->rw && !row_sharded(ds) && knobs().light_pass) {
    if (!ds->colnorm && pool_malloc((void**)&ds->colnorm, sizeof(double) * (size_t)ld) != hipSuccess) {
      (void)hipGetLastError();
      ds->colnorm = nullptr;
    }
    if (ds->colnorm) {
      hipLaunchKernelGGL(colnorm_kernel, dim3((unsigned)((ld + 7) / 8)), dim3(256), 0, s, (const double*)ds->XT, n, ld, 1.0 / (double)ds->n_global,
                         ds->colnorm);
      ds->colnorm_ready = true;
    }
  }
  return SLM_OK;
}

extern "C" int slm_eval_sse_sparse(slm_dataset* ds, const int32_t* cols, int32_t n_cols, const double* Zs,
                                   int32_t m, const double* row_weight, double* sse_out) {
  if (!ds || !cols || !Zs || !sse_out) return fail(SLM_ERR_BAD_ARG, "NULL argument");
  if (m <= 0 || n_cols <= 0) return fail(SLM_ERR_BAD_ARG, "m and n_cols must be positive");
  if (n_cols > WS_KCAP) return fail(SLM_ERR_UNSUPPORTED, "more than %d columns: use slm_eval_sse", WS_KCAP);
  for (int k = 0; k < n_cols; ++k)
    if (cols[k] < 0 || cols[k] >= ds->p) return fail(SLM_ERR_BAD_ARG, "cols[%d] = %d outside [0, p)", k, cols[k]);
  slm_engine* eng = ds->eng;
  HIP_TRY(hipSetDevice(eng->device));
  hipStream_t s = eng->stream;
  const int64_t n = ds->n;
  if (row_weight) {
    for (int64_t i = 0; i < n; ++i)
      if (!(row_weight[i] >= 0.0) || !std::isfinite(row_weight[i]))
        return fail(SLM_ERR_BAD_ARG, "row_weight[%lld] is negative or not finite", (long long)i);
    if (!ds->rw_lanes) SLM_TRY(dalloc(&ds->rw_lanes, (size_t)ds->lane_cap * n));
    HIP_TRY(hipMemcpyAsync(ds->rw_lanes, row_weight, sizeof(double) * n, hipMemcpyHostToDevice, s));
  }
  // scratch shared with the working set (every solve re-initialises that state)
  ds->ws_carry_valid = false;  // (this call gathers its own columns into the working set's buffers)
  if (!ds->ws_idx) SLM_TRY(dalloc(&ds->ws_idx, WS_KCAP));
  if (!ds->ws_XW) SLM_TRY(dalloc(&ds->ws_XW, (size_t)n * WS_KCAP));
  SLM_TRY(ensure_xt(ds));
  const int nblk = (int)std::max<int64_t>(1, std::min<int64_t>(eng->cus * 4, (n + 255) / 256));
  // scratch of the scoring loop of a grid search, kept with the dataset (it used to be allocated and freed per
  // call): the coefficient block grows on demand, the per-workgroup partial sums have a fixed size
  if ((size_t)m * n_cols > ds->sse_cap) {
    dfree(ds->sse_Z);
    ds->sse_cap = 0;
    SLM_TRY(dalloc(&ds->sse_Z, (size_t)m * n_cols));
    ds->sse_cap = (size_t)m * n_cols;
  }
  if (!ds->sse_part) SLM_TRY(dalloc(&ds->sse_part, (size_t)eng->cus * 4 * SSE_M));
  double *dZ = ds->sse_Z, *dpart = ds->sse_part;
  int rc = SLM_OK;
  auto bail = [&](hipError_t e) {
    if (e != hipSuccess && rc == SLM_OK) rc = fail(SLM_ERR_HIP, "slm_eval_sse_sparse: %s", hipGetErrorString(e));
  };
  bail(hipMemcpyAsync(ds->ws_idx, cols, sizeof(int32_t) * n_cols, hipMemcpyHostToDevice, s));
  bail(hipMemcpyAsync(dZ, Zs, sizeof(double) * (size_t)m * n_cols, hipMemcpyHostToDevice, s));
  if (rc == SLM_OK) {
    GatherArgs ga;
    ga.X = ds->X; ga.XT = ds->XT; ga.n = n; ga.ld = ds->ld;
    ga.idx = ds->ws_idx; ga.K = n_cols; ga.XW = ds->ws_XW;
    hipLaunchKernelGGL(gather_cols_kernel, dim3((unsigned)std::min<int64_t>((n + 31) / 32, 1024), (unsigned)((n_cols + 31) / 32)),
                       dim3(256), 0, s, ga);
    std::vector<double> part((size_t)nblk * SSE_M);
    for (int v0 = 0; v0 < m && rc == SLM_OK; v0 += SSE_M) {
      SseArgs sa;
      sa.XW = ds->ws_XW; sa.y = ds->y; sa.rw = row_weight ? ds->rw_lanes : ds->rw;
      sa.Zs = dZ + (size_t)v0 * n_cols; sa.partial = dpart; sa.n = n; sa.K = n_cols;
      sa.m = std::min<int>(SSE_M, m - v0);
      (void)hipFuncSetAttribute((const void*)sse_sparse_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
      hipLaunchKernelGGL(sse_sparse_kernel, dim3(nblk), dim3(256), sizeof(double) * (size_t)n_cols * SSE_M, s, sa);
      bail(hipMemcpyAsync(part.data(), dpart, sizeof(double) * part.size(), hipMemcpyDeviceToHost, s));
      bail(hipStreamSynchronize(s));
      for (int v = 0; v < sa.m; ++v) {
        double t = 0.0;
        for (int b = 0; b < nblk; ++b) t += part[(size_t)b * SSE_M + v];
        sse_out[v0 + v] = t;
      }
    }
  }
  bail(hipStreamSynchronize(s));
  if (rc == SLM_OK) rc = check_launch();
  return rc;
}

// kernels that ask for more dynamic LDS than the default limit: the attribute is per device
int allow_big_lds(const void* fn, int device) {
  static std::mutex m;
  static std::vector<std::pair<const void*, int>> done;
  std::lock_guard<std::mutex> lk(m);
  for (auto& d : done)
    if (d.first == fn && d.second == device) return SLM_OK;
  HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, SM_LDS_BYTES));
  done.push_back({fn, device});
  return SLM_OK;
}

// ------------------------------------------------------------------------------------------------
// SparseGroupLasso(standardize=True): the operator splitting on chip (small_split_kernels.hpp)
// ------------------------------------------------------------------------------------------------
extern "C" int slm_solve_standardized_sgl(slm_dataset* ds, const double* a, const double* b, const slm_solve_opts* opts,
                                          double tol_inner, int32_t max_sweeps, const double* beta0, int32_t warm,
                                          double* beta_out, double* group_norms_out, slm_point_info* info) {
  if (!ds || !a || !b || !beta_out) return fail(SLM_ERR_BAD_ARG, "NULL argument");
  slm_engine* eng = ds->eng;
  HIP_TRY(hipSetDevice(eng->device));
  hipStream_t s = eng->stream;
  const int p = (int)ds->p, G = ds->G;
  const int64_t ld = ds->ld;
  const int gm = ds->max_group;
  if (!knobs().on_chip) return fail(SLM_ERR_UNSUPPORTED, "the on-chip solvers are switched off (SLM_ON_CHIP=0)");
  if (row_sharded(ds) || ds->rw || p > SM_PMAX || (double)ds->n * (double)ld > 131072.0)
    return fail(SLM_ERR_UNSUPPORTED, "the splitting runs on chip for unweighted, unsharded problems of p <= %d and n * ld <= 131072", SM_PMAX);
  // LDS: Gram matrix, three vectors, the groups' Cholesky factors; what is left stages the rows of the build and
  // then holds the partial products of three wavefronts
  const size_t fixed = sizeof(double) * ((size_t)p * p + 3 * (size_t)p + (size_t)p * gm);
  const size_t lds = (size_t)SM_LDS_BYTES;
  const int64_t stage = fixed + 64 < lds ? (int64_t)((lds - fixed - 64) / sizeof(double)) : 0;
  const int ps = 4 * ((p + 4) / 4);
  if (stage < 3 * (int64_t)p + 512 || stage < 4 * (int64_t)ps)  // (512: the head of the b-step's direct solves)
    return fail(SLM_ERR_UNSUPPORTED, "groups of up to %d columns at p = %d leave no room in LDS", gm, p);
  const size_t rec_off = 3 * (size_t)ld + 4;  // state: gamma [ld], u [ld], rho, valid, direct b-steps, factorisations; then beta_out [ld]; then the record
  const size_t n_state = rec_off + (sizeof(slm_point_info) + 7) / 8 + (size_t)ld;  // (+ group norms [ld])
  const size_t n_host = 3 * (size_t)ld + n_state;
  if (!ds->split_state) {
    SLM_TRY(dalloc(&ds->split_state, n_state));
    HIP_TRY(hipMemsetAsync(ds->split_state, 0, sizeof(double) * n_state, s));
  }
  if (!ds->h_split) {
    hipError_t eh = hipHostMalloc((void**)&ds->h_split, sizeof(double) * n_host, hipHostMallocDefault);
    if (eh != hipSuccess) return fail(SLM_ERR_OOM, "hipHostMalloc: %s", hipGetErrorString(eh));
    memset(ds->h_split, 0, sizeof(double) * n_host);
  }
  // in: a | b | beta0 -> lane 0 of a0 | b0 | beta (one transfer each from the page-locked stage)
  double* h = ds->h_split;
  memcpy(h, a, sizeof(double) * p);
  memcpy(h + ld, b, sizeof(double) * G);
  if (beta0) memcpy(h + 2 * ld, beta0, sizeof(double) * p);
  HIP_TRY(hipMemcpyAsync(ds->a0, h, sizeof(double) * p, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(ds->b0, h + ld, sizeof(double) * G, hipMemcpyHostToDevice, s));
  if (beta0) HIP_TRY(hipMemcpyAsync(ds->beta, h + 2 * ld, sizeof(double) * p, hipMemcpyHostToDevice, s));
  SplitSglArgs k;
  memset(&k, 0, sizeof(k));
  k.X = ds->X; k.y = ds->y; k.rw = nullptr; k.n = ds->n; k.ld = ld; k.p = p; k.G = G; k.singleton = ds->singleton;
  k.order = ds->order; k.gid = ds->gid; k.gstart = ds->gstart;
  k.a = ds->a0; k.b = ds->b0; k.beta0 = beta0 ? ds->beta : nullptr;
  k.state = ds->split_state;
  k.beta_out = ds->split_state + 2 * ld + 4;
  k.info = reinterpret_cast<slm_point_info*>(ds->split_state + rec_off);
  k.gn_out = ds->split_state + rec_off + (sizeof(slm_point_info) + 7) / 8;
  k.warm = warm ? 1 : 0;
  k.tol = opts && opts->tol > 0 ? opts->tol : 1e-8;
  k.tol_inner = tol_inner > 0 ? tol_inner : std::min(k.tol, 1e-10);
  k.inv_n = 1.0 / (double)ds->n_global;
  k.max_sweeps = max_sweeps > 0 ? max_sweeps : 500;
  k.max_iters = opts && opts->max_iter > 0 ? (int)std::min<int64_t>(opts->max_iter, 4000) : 4000;
  k.gmax = gm;
  k.stage_doubles = (int)stage;
  SLM_TRY(allow_big_lds((const void*)small_stdsgl_kernel, eng->device));
  hipLaunchKernelGGL(small_stdsgl_kernel, dim3(1), dim3(SM_THREADS), lds, s, k);
  SLM_TRY(check_launch());
  double* h_out = h + 3 * ld;  // a copy of everything behind gamma and u
  const size_t out_off = 2 * (size_t)ld;
  HIP_TRY(hipMemcpyAsync(h_out + out_off, ds->split_state + out_off, sizeof(double) * (n_state - out_off), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  slm_point_info rec;
  memcpy(&rec, h_out + rec_off, sizeof(rec));
  memcpy(beta_out, h_out + 2 * ld + 4, sizeof(double) * p);
  if (knobs().trace == 2)
      fprintf(stderr, "[slm] standardised sparse-group splitting on chip: %d sweeps, %d products, %d b-steps by a direct solve, %d factorisations, rho %.3e\n",
              rec.n_iter, rec.rejects, (int)h_out[2 * ld + 2], (int)h_out[2 * ld + 3], rec.L);
  if (group_norms_out) memcpy(group_norms_out, h_out + rec_off + (sizeof(slm_point_info) + 7) / 8, sizeof(double) * G);
  if (info) *info = rec;
  if (rec.status == SLM_ERR_NON_FINITE) return fail(SLM_ERR_NON_FINITE, "non-finite iterate (diverged or non-finite data)");
  return SLM_OK;
}

// ------------------------------------------------------------------------------------------------
// diagnostic: the model solver's dense SPD solve on its own
// ------------------------------------------------------------------------------------------------
extern "C" int slm_dense_spd_solve(slm_engine* eng, const double* H, int32_t m, const double* rhs, double* x_out,
                                   double* mu_out) {
  if (!eng || !H || !rhs || !x_out) return fail(SLM_ERR_BAD_ARG, "NULL argument");
  if (m < 1 || m > NT_MAXT * NT_B) return fail(SLM_ERR_BAD_ARG, "m must be in [1, %d] (got %d)", NT_MAXT * NT_B, m);
  for (int64_t e = 0; e < (int64_t)m * m; ++e)
    if (!std::isfinite(H[e])) return fail(SLM_ERR_BAD_ARG, "H contains a non-finite value");
  HIP_TRY(hipSetDevice(eng->device));
  hipStream_t s = eng->stream;
  double *dH = nullptr, *dv = nullptr, *scratch = nullptr;
  int* dst = nullptr;
  int rc = dalloc(&dH, (size_t)m * m);
  if (rc == SLM_OK) rc = dalloc(&dv, (size_t)2 * m + 1);
  if (rc == SLM_OK) rc = dalloc(&scratch, (size_t)NT_SCRATCH);
  if (rc == SLM_OK) rc = dalloc(&dst, 1);
  int status = 0;
  double mu = 0.0;
  auto bail = [&](hipError_t e) {
    if (e != hipSuccess && rc == SLM_OK) rc = fail(SLM_ERR_HIP, "slm_dense_spd_solve: %s", hipGetErrorString(e));
  };
  if (rc == SLM_OK) {
    bail(hipMemcpyAsync(dH, H, sizeof(double) * (size_t)m * m, hipMemcpyHostToDevice, s));
    bail(hipMemcpyAsync(dv, rhs, sizeof(double) * m, hipMemcpyHostToDevice, s));
    DenseSolveArgs a;
    a.H = dH; a.rhs = dv; a.x = dv + m; a.mu = dv + 2 * m; a.status = dst; a.scratch = scratch; a.m = m;
    hipLaunchKernelGGL(dense_spd_solve_kernel, dim3(1), dim3(TAIL_THREADS), 0, s, a);
    bail(hipGetLastError());
    bail(hipMemcpyAsync(x_out, dv + m, sizeof(double) * m, hipMemcpyDeviceToHost, s));
    bail(hipMemcpyAsync(&mu, dv + 2 * m, sizeof(double), hipMemcpyDeviceToHost, s));
    bail(hipMemcpyAsync(&status, dst, sizeof(int), hipMemcpyDeviceToHost, s));
    bail(hipStreamSynchronize(s));
  }
  dfree(dH); dfree(dv); dfree(scratch); dfree(dst);
  if (rc != SLM_OK) return rc;
  if (status != 0) return fail(SLM_ERR_BAD_ARG, "H is not numerically positive definite");
  if (mu_out) *mu_out = mu;
  return SLM_OK;
}


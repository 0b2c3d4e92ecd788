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
static const int kProfStride = 3;  // SLM_FLAG_PROFILE times every 3rd gradient launch (a working-set path has ~5: two of them;
                                   // an event pair costs ~12 us of stream around the launch it brackets)

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
static const SplitKernel kSplit[] = {SLM_SK(1, 3), SLM_SK(2, 3), SLM_SK(3, 3), SLM_SK(4, 3), SLM_SK(5, 2),
                                     {8, 10, SPLIT_LANES, 0, nullptr, resid_ws_kernel<SPLIT_LANES>}};
const SplitKernel* pick_split_kernel(int64_t p2) {
  const char* env = getenv("SLM_SPLIT");
  if (env && env[0] == '0') return nullptr;
  for (const auto& k : kSplit)
    if (64LL * k.W * k.C >= p2) return &k;
  return nullptr;
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
  double per_cu = 1.0;
  if (const char* e = getenv("SLM_XTR_WGS_PER_CU")) {  // (A/B runs: workgroups per CU, up to 2)
    const double f = atof(e);
    if (f > 0.0 && f <= 2.0) per_cu = f;
  }
  const bool wide = a.lane_slots > SPLIT_LANES;  // thirty-two lanes: both planes of R per row of X
  if (wide) per_cu = 1.0;                        // (its partial sums fill the buffer at one workgroup per CU)
  const slm_host::XtrGrid g = slm_host::xtr_grid(a.n, a.ld, XTR_CB, (int64_t)(xtr_max_row_blocks(cus, a.ld) * per_cu / 2.0));
  const int xb = g.xb, yb = g.yb;
  a.xrows = g.rows;
  // seventeen to twenty lanes: the lanes beyond sixteen on the vector units beside the sixteen on the matrix cores
  // (xtr18 / xtr20_mfma_kernel: the price of sixteen; SLM_XTR_EXTRAS=0: both halves on the matrix cores)
  const char* ex_env = getenv("SLM_XTR_EXTRAS");
  const int extra = wide && !(ex_env && ex_env[0] == '0') ? a.n_lanes - SPLIT_LANES : 0;
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
  const char* ring = getenv("SLM_GRAD_RING");
  const bool ring_off = ring && ring[0] == '0', ring_all = ring && ring[0] == '1';
  if (!ring_off && p2 > 256 && (B >= 3 || ring_all)) {
    for (const auto& k : kGradRing)
      if (k.B == B && 64LL * k.W * k.C >= p2) return &k;
  }
  const char* env = getenv("SLM_GRAD_CONFIG");
  if (env) {
    int W = 0, C = 0, R = 0;
    if (sscanf(env, "%d,%d,%d", &W, &C, &R) == 3) {
      for (const auto& k : kGradDefault)
        if (k.B == B && k.W == W && k.C == C && k.R == R && 64LL * W * C >= p2) return &k;
      for (const auto& k : kGradExtra)
        if (k.B == B && k.W == W && k.C == C && k.R == R && 64LL * W * C >= p2) return &k;
    }
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
                     hipEvent_t ev_start, hipEvent_t ev_stop, int64_t n_rows) {
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
    t.loss_partial = a.loss_partial; t.done = done; t.n = nr; t.ld = ds->ld;
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
static void launch_rowdot(slm_dataset* ds, const SplitKernel* sk, int nblk, int B, SplitArgs& a, hipStream_t s) {
  // Measured at n = 100k, p = 5k (tools/rowdot_probe.py): matrix cores 0.75-0.80 ms whatever the lane count;
  // vector kernel 0.62 ms for one lane, 0.81 ms for five, 3.1 ms for sixteen (four reads of X).
  const char* env = getenv("SLM_ROWDOT_RING");
  const int halves = (B + SPLIT_LANES - 1) / SPLIT_LANES;
  const bool ring = sk->rowdot != nullptr && halves == 1 && (env ? env[0] == '1' : B <= ROWDOT_LANES);
  a.lane0 = 0;
  if ((!ring || sk->rowdot == nullptr) && ds->XT && ds->XT_ready) {
    a.XT = ds->XT;
    // (thirty-two lanes: both halves against ONE read of the copy -- rowdot32_mfma_kernel; SLM_ROWDOT32=0: a read per half)
    const char* e32 = getenv("SLM_ROWDOT32");
    // (seventeen to twenty lanes: the lanes beyond sixteen on the vector units beside the matrix cores' sixteen -- rowdot18 /
    //  rowdot20_mfma_kernel, as for X^T R; SLM_XTR_EXTRAS=0: both halves on the matrix cores)
    const char* exs = getenv("SLM_XTR_EXTRAS");
    const bool extras = halves == 2 && B <= SPLIT_LANES + 4 && !(exs && exs[0] == '0');
    if (extras && B <= SPLIT_LANES + 2) hipLaunchKernelGGL(rowdot18_mfma_kernel, dim3(nblk, 1), dim3(XZ_WAVES * 64), 0, s, a);
    else if (extras) hipLaunchKernelGGL(rowdot20_mfma_kernel, dim3(nblk, 1), dim3(XZ_WAVES * 64), 0, s, a);
    else if (halves == 2 && !(e32 && e32[0] == '0')) hipLaunchKernelGGL(rowdot32_mfma_kernel, dim3(nblk, 1), dim3(XZ_WAVES * 64), 0, s, a);
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
                           hipEvent_t ev_stop, int64_t n_rows) {
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
  a.partial = ds->partial; a.loss_partial = ds->loss_partial; a.done = done; a.ctl = ctl;
  if (wa) { a.XW = wa->XW; a.idx = wa->idx; a.ws = wa->ws; }
  const int64_t nr = n_rows > 0 ? n_rows : ds->n;
  a.n = nr; a.ld = ds->ld; a.rows_base = nr / nblk; a.rows_rem = nr % nblk;
  a.p2 = (int)(ds->ld / 2);
  a.n_lanes = ls.B;
  // (SLM_PROFILE_UNIT=1: the bracket of SLM_FLAG_PROFILE opens here -- the whole gradient unit, residuals and X^T R, not the
  //  stream over X alone: bench.py's roofline.gradient_unit_frac)
  const bool unit = ev_start != nullptr && getenv("SLM_PROFILE_UNIT") != nullptr;
  if (unit) HIP_TRY(hipEventRecord(ev_start, s));
  launch_rowdot(ds, sk, nblk, ls.B, a, s);
  if (wa && ctl) {  // residuals from the gathered columns: matrix cores (SLM_RESID_VEC=1: a row per thread)
    const char* env = getenv("SLM_RESID_VEC");
    if (env && env[0] == '1' && halves == 1) hipLaunchKernelGGL(sk->resid, dim3(nblk), dim3(256), 0, s, a);
    else if (halves == 2 && getenv("SLM_NO_RESID32") == nullptr) hipLaunchKernelGGL(resid32_mfma_kernel, dim3(nblk, 1), dim3(RM_WAVES * 64), 0, s, a);  // (both halves on one read of the gathered columns)
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
static int enqueue_gradient_cov(slm_dataset* ds, int B, const int* entry_of, const int* done, hipEvent_t ev_start,
                                hipEvent_t ev_stop, const PathCtl* ctl = nullptr, const WsArgs* wa = nullptr) {
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
  if (ctl && wa && wa->ws && !getenv("SLM_COV_ALL_ROWS")) { a.ctl = ctl; a.ws = wa->ws; a.idx = wa->idx; }
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
static void launch_tail(const TailArgs& ta, hipStream_t s) {
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
static const int kPowerItersSolve = 2;
// Power steps on the sketch (a thirty-second of the rows) for working-set solves: ONE.  The sketch's lambda_max is 2-3 x the
// whole matrix's, and one step from the fixed start lands at about half of the sketch's: closer to the truth than three
// steps' estimate, 80 us cheaper per solve (a launch chain of nine), and these solves only use L for a first candidate
// and for fallback steps whose curvature guard repairs an under-estimate.  Round 3, same box, alternating: headline
// 4.17-4.25 ms with three steps, 4.13-4.16 with two, 4.09-4.12 with one; passes of the headline, configs 3 / 4 and the
// sparse-regime soak paths unchanged, dense-regime soak paths -3 ... +3 passes of 24-60 (SLM_L_SKETCH_ITERS).
static const int kPowerItersSketchDefault = 1;
static int sketch_iters() {
  if (const char* e = getenv("SLM_L_SKETCH_ITERS")) return std::max(1, std::min(16, atoi(e)));
  return kPowerItersSketchDefault;
}
static const int kPowerItersQuery = 16;
// (a thirty-second of the rows: the bound is looser than from a sixteenth -- lambda_max of a sketch grows as it
// shrinks -- and nothing downstream noticed down to a sixty-fourth, SLM_L_SKETCH_DIV; three steps on
// 3 125 of 100 000 rows cost 0.10 ms where a sixteenth cost 0.17)
static int64_t sketch_rows(int64_t n) {
  int div = 32;
  if (const char* e = getenv("SLM_L_SKETCH_DIV")) div = std::max(1, std::min(1024, atoi(e)));
  return std::max<int64_t>(1, n / div);
}

// n_rows > 0: the operator of the first n_rows rows only, X_S^T W X_S / (n_eff n_rows / n).  Its largest
// eigenvalue is, in expectation, no smaller than that of the full operator (Jensen: lambda_max is
// convex and E G_S = G for exchangeable rows), so it serves as a cheap step-size bound where the
// iteration does not depend on a tight one (working-set solves); the curvature guards cover the rest.
static int power_iteration(slm_dataset* ds, const LaneSetup& ls_in, double* L_out /*[B]*/, int iters,
                           int64_t n_rows = 0) {
  LaneSetup ls = ls_in;
  if (n_rows > 0) {
    for (int l = 0; l < kMaxLanes; ++l) {
      const double ne = ls.n_eff[l] > 0 ? ls.n_eff[l] : (double)ds->n_global;
      ls.n_eff[l] = ne * (double)n_rows / (double)ds->n;
    }
  }
  hipStream_t s = ds->eng->stream;
  if (const char* env = getenv("SLM_POWER_ITERS")) iters = std::max(2, atoi(env));
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

static int estimate_lipschitz(slm_dataset* ds, double* L_out, int iters) {
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
extern "C" int slm_gradient(slm_dataset* ds, const double* z, double* g_out, double* loss_out,
                            int32_t reps, double* ms_out) {
  if (!ds) return fail(SLM_ERR_BAD_ARG, "dataset is NULL");
  HIP_TRY(hipSetDevice(ds->eng->device));
  hipStream_t s = ds->eng->stream;
  HIP_TRY(hipMemsetAsync(ds->z, 0, sizeof(double) * ds->ld, s));
  if (z) HIP_TRY(hipMemcpyAsync(ds->z, z, sizeof(double) * ds->p, hipMemcpyHostToDevice, s));
  const char* split_env = getenv("SLM_GRAD_SPLIT");  // tests: take the split pass (residuals from X)
  const bool use_split = split_env && split_env[0] == '1' && split_usable(ds);
  if (use_split) SLM_TRY(ensure_xt(ds));  // (so that tests and probes reach rowdot_mfma_kernel; optional copy)
  // (tests of the split pass's wider forms: SLM_GRAD_LANES lanes all at z, the gradient of lane SLM_GRAD_LANE returned --
  //  a lane of the second half: xtr32_mfma_kernel's second plane, or the vector units' share of xtr18 / xtr20_mfma_kernel)
  int lanes_run = 1, lane_out = 0;
  if (use_split && ds->XT && ds->XT_ready && (size_t)ds->lane_cap >= (size_t)kMaxLanes) {
    if (const char* e = getenv("SLM_GRAD_LANES")) lanes_run = std::min(kMaxLanes, std::max(1, atoi(e)));
    if (const char* e = getenv("SLM_GRAD_LANE")) lane_out = std::min(lanes_run - 1, std::max(0, atoi(e)));
  }
  const LaneSetup ls = default_lanes(ds, lanes_run);
  for (int l = 1; l < lanes_run; ++l)
    HIP_TRY(hipMemcpyAsync(ds->z + (size_t)l * ds->ld, ds->z, sizeof(double) * ds->ld, hipMemcpyDeviceToDevice, s));
  if (use_split) SLM_TRY(enqueue_gradient_split(ds, ls, ds->y, nullptr, nullptr, nullptr, nullptr, nullptr));
  else SLM_TRY(enqueue_gradient(ds, ls, ds->y, nullptr, nullptr, nullptr));
  SLM_TRY(check_launch());
  HIP_TRY(hipStreamSynchronize(s));
  const double* g_lane = ds->g + (size_t)lane_out * (size_t)(ds->ld + 16);
  if (g_out) HIP_TRY(hipMemcpy(g_out, g_lane, sizeof(double) * ds->p, hipMemcpyDeviceToHost));
  if (loss_out) HIP_TRY(hipMemcpy(loss_out, g_lane + ds->ld, sizeof(double), hipMemcpyDeviceToHost));
  if (ms_out) {
    *ms_out = 0.0;
    if (reps < 1) reps = 1;
    // lanes used by the probe: SLM_PROBE_LANES (tuning), default 1
    int B = 1;
    if (const char* env = getenv("SLM_PROBE_LANES")) B = std::min(kMaxLanes, std::max(1, atoi(env)));
    if (use_split) {  // time the split pass (rowdot + xtr, or xtr alone with SLM_GRAD_SPLIT_XTR_ONLY=1)
      for (int l = 1; l < B; ++l)
        HIP_TRY(hipMemcpyAsync(ds->z + l * ds->ld, ds->z, sizeof(double) * ds->ld, hipMemcpyDeviceToDevice, s));
      const LaneSetup lb = default_lanes(ds, B);
      const bool xtr_only = getenv("SLM_GRAD_SPLIT_XTR_ONLY") != nullptr;
      SLM_TRY(enqueue_gradient_split(ds, lb, ds->y, nullptr, nullptr, nullptr, nullptr, nullptr));  // warm; allocates R
      SplitArgs a;
      memset(&a, 0, sizeof(a));
      a.X = ds->X; a.y = ds->y; a.rw = lb.rw; a.rw_stride = 0; a.z = ds->z; a.R = ds->R;
      a.partial = ds->partial; a.loss_partial = ds->loss_partial;
      a.n = ds->n; a.ld = ds->ld; a.rows_base = ds->n / ds->split_nblk; a.rows_rem = ds->n % ds->split_nblk;
      a.p2 = (int)(ds->ld / 2); a.n_lanes = B;
      a.lane_slots = SPLIT_LANES * ((B + SPLIT_LANES - 1) / SPLIT_LANES); a.r_plane = (int64_t)ds->n * SPLIT_RSTRIDE;
      hipEvent_t e0, e1;
      HIP_TRY(hipEventCreate(&e0));
      HIP_TRY(hipEventCreate(&e1));
      HIP_TRY(hipEventRecord(e0, s));
      for (int r = 0; r < reps; ++r) {
        if (!xtr_only) {
          launch_rowdot(ds, ds->sk, ds->split_nblk, B, a, s);
        }
        (void)launch_xtr(ds->eng->cus, a, s);
      }
      HIP_TRY(hipEventRecord(e1, s));
      HIP_TRY(hipEventSynchronize(e1));
      float ms = 0.f;
      HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
      *ms_out = (double)ms / reps;
      (void)hipEventDestroy(e0);
      (void)hipEventDestroy(e1);
      return check_launch();
    }
    if (!ds->gk[B - 1]) return fail(SLM_ERR_UNSUPPORTED, "no %d-lane kernel for p = %lld", B, (long long)ds->p);
    for (int l = 1; l < B; ++l)
      HIP_TRY(hipMemcpyAsync(ds->z + l * ds->ld, ds->z, sizeof(double) * ds->ld, hipMemcpyDeviceToDevice, s));
    const LaneSetup lb = default_lanes(ds, B);
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    const GradKernel* gk = ds->gk[B - 1];
    const int nblk = ds->nblk[B - 1];
    if (gk->D < 0) {  // two-pass fallback: time the pair of kernels through the common path
      hipEvent_t e0, e1;
      HIP_TRY(hipEventCreate(&e0));
      HIP_TRY(hipEventCreate(&e1));
      HIP_TRY(hipEventRecord(e0, s));
      for (int r = 0; r < reps; ++r) SLM_TRY(enqueue_gradient(ds, lb, ds->y, nullptr, nullptr, nullptr));
      HIP_TRY(hipEventRecord(e1, s));
      HIP_TRY(hipEventSynchronize(e1));
      float ms = 0.f;
      HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
      *ms_out = (double)ms / reps;
      (void)hipEventDestroy(e0);
      (void)hipEventDestroy(e1);
      return check_launch();
    }
    GradArgs a;
    a.X = ds->X; a.y = ds->y; a.rw = lb.rw; a.z = ds->z; a.partial = ds->partial;
    a.loss_partial = ds->loss_partial; a.done = nullptr; a.n = ds->n; a.ld = ds->ld;
    a.rows_base = ds->n / nblk; a.rows_rem = ds->n % nblk; a.rw_stride = 0; a.p2 = (int)(ds->ld / 2);
    hipLaunchKernelGGL(gk->fn, dim3(nblk), dim3(gk->W * 64), 0, s, a);  // warm
    HIP_TRY(hipEventRecord(e0, s));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(gk->fn, dim3(nblk), dim3(gk->W * 64), 0, s, a);
    HIP_TRY(hipEventRecord(e1, s));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    *ms_out = (double)ms / reps;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    SLM_TRY(check_launch());
  }
  return SLM_OK;
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
  const bool trace = getenv("SLM_TRACE") != nullptr;
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
  if (m > maxB && ds->sk != nullptr && !row_sharded(ds) && split_usable(ds) && getenv("SLM_EVAL_FUSED") == nullptr &&
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
    const dim3 grid((unsigned)row_tiles, (unsigned)((ld + 31) / 32));
    hipLaunchKernelGGL(tile_columns_kernel, grid, dim3(256), 0, s, (const double*)ds->X, n, ld, ds->XT);
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

// ------------------------------------------------------------------------------------------------
// path solves
// ------------------------------------------------------------------------------------------------
// shared_path: the lanes are contiguous, ordered ranges of ONE path (slm_solve_path_lanes): global
// point indices on the device and work stealing between lanes.
// Working-set refinement policy (see solve_core): 0 = never, 1 = when a path point turns out hard
// (small problems), 2 = from the first pass.
static int ws_policy(const slm_dataset* ds, uint32_t flags) {
  const char* env = getenv("SLM_WS");
  if (ds->max_group > 64 || ds->n < 4) return 0;
  if ((env && env[0] == '0') || (flags & SLM_FLAG_NO_WORKING_SET)) return 0;
  const bool big = (double)ds->n * (double)ds->ld >= 67108864.0;  // 2^26 doubles = 512 MiB
  return (big || (flags & SLM_FLAG_WORKING_SET) || (env && env[0] == '1')) ? 2 : 1;
}
// kernels that ask for more dynamic LDS than the default limit: the attribute is per device
static int allow_big_lds(const void* fn, int device) {
  static std::mutex m;
  static std::vector<std::pair<const void*, int>> done;
  std::lock_guard<std::mutex> lk(m);
  for (auto& d : done)
    if (d.first == fn && d.second == device) return SLM_OK;
  HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, SM_LDS_BYTES));
  done.push_back({fn, device});
  return SLM_OK;
}

// The on-chip solver (small_kernels.hpp) takes a call when the caller allows it (SLM_FLAG_ON_CHIP), the Gram matrix
// fits the LDS and nothing asks for a particular iteration of the general path.
static bool small_ok(const slm_dataset* ds, uint32_t flags) {
  if (!(flags & SLM_FLAG_ON_CHIP)) return false;
  if (flags & (SLM_FLAG_NO_RESTART | SLM_FLAG_PROFILE | SLM_FLAG_FISTA_ONLY | SLM_FLAG_WORKING_SET | SLM_FLAG_NO_WORKING_SET))
    return false;
  if (const char* env = getenv("SLM_ON_CHIP"))
    if (env[0] == '0') return false;
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
    if (ws_policy(ds, flags) == 2 && !row_sharded(ds) && ds->sk->rowdot != nullptr && getenv("SLM_NO_WIDE_LANES") == nullptr &&
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

static int solve_core(slm_dataset* ds, const slm_lane* lanes, int32_t n_lanes, const slm_solve_opts* opts,
                      slm_solve_stats* stats, bool shared_path, const slm_reweight* rules, int32_t* rounds_out) {
  if (!ds || !lanes) return fail(SLM_ERR_BAD_ARG, "NULL argument");
  HIP_TRY(hipSetDevice(ds->eng->device));  // (before anything that may allocate or launch: split_usable / ensure_xt below)
  if (rules) {
    // re-weighted rounds run inside the on-chip kernel, nowhere else: other problems keep their loop on the caller's side
    if (!rounds_out) return fail(SLM_ERR_BAD_ARG, "rounds_out is NULL");
    if (shared_path || !small_ok(ds, opts ? opts->flags : 0u))
      return fail(SLM_ERR_UNSUPPORTED, "re-weighted rounds need a problem the on-chip solver takes (p <= %d, n * ld <= 131072)", SM_PMAX);
    for (int l = 0; l < n_lanes && l < SLM_MAX_CELLS; ++l) {
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
    if (n_lanes < 1 || n_lanes > cap) return fail(SLM_ERR_BAD_ARG, "n_lanes must be in [1, %d] (got %d)", cap, n_lanes);
  }
  const int B = n_lanes;
  // the split pass costs four launches where the fused kernel costs one: take it when X is large (the
  // accumulate-only stream is then all that matters) or when only it has enough lanes
  // -- the latter also without the working set when X is large: sixteen lanes on the two matrix-core halves
  // (two reads of X per pass) move more problems per byte than four on the fused kernel (one read)
  const bool big_x = (double)ds->n * (double)ds->ld >= 67108864.0;
  // (rows beyond 5120 columns have no ring variant for the residuals that need X: every such residual is a
  //  second full read, of the column-major copy -- the split pass is worth it there only for more lanes than
  //  the fused kernel serves: measured on config 5's shape, one lane, 2.9 ms per pass against 2.4 ms fused)
  const bool wide = ds->sk != nullptr && ds->sk->rowdot == nullptr;
  // (the fused kernels' table stops at SLM_MAX_LANES; calls of more lanes exist on the on-chip route only)
  const GradKernel* gk_B = B <= kMaxLanes ? ds->gk[B - 1] : nullptr;
  const bool want_split = (ws_policy(ds, opts ? opts->flags : 0u) == 2 && !wide) ? (big_x || !gk_B) : (big_x && !gk_B);
  // (covariance passes are a form of the split pass: the flag asks for it whatever the size, where Grams exist)
  const bool want_cov = opts && (opts->flags & SLM_FLAG_COVARIANCE) && !ds->cov.empty() && !row_sharded(ds) && B <= kMaxLanes;
  const bool split = (want_split || want_cov) && split_usable(ds);
  // Shared path with the working set on from the start: the lanes take the points of the path in turn
  // (lane l: l, l + B, ...) instead of contiguous ranges.  Every lane then starts near alpha_max, where
  // the first working set (chosen from the gradient at zero) is enough, and all lanes move down the
  // path together, so W only ever has to cover one band of alphas; a contiguous split starts some lanes
  // cold at small alpha, whose first refinement misses features W could not know about (one extra pass).
  // (Only for per-feature penalties.  With group penalties the cold starts do not miss -- config 3: no
  // miss either way -- while looking a whole stride ahead pulls noise groups into W: 380 columns and
  // 10.9 ms per path against 250 columns and 10.3 ms with contiguous ranges.)
  const bool interleave = shared_path && ds->singleton && ws_policy(ds, opts ? opts->flags : 0u) == 2 &&
                          !getenv("SLM_NO_INTERLEAVE");
  if (!split && !gk_B && !small_ok(ds, opts ? opts->flags : 0u))
    return fail(SLM_ERR_UNSUPPORTED, "no %d-lane gradient kernel covers p = %lld", B, (long long)ds->p);
  // more than sixteen lanes: two halves on one read of X (xtr32_mfma_kernel) -- working-set solves on the split pass, all rows here
  if (B > SPLIT_LANES && !small_ok(ds, opts ? opts->flags : 0u) && (!split || row_sharded(ds) || ws_policy(ds, opts ? opts->flags : 0u) != 2))
    return fail(SLM_ERR_UNSUPPORTED, "%d lanes: more than %d need a working-set solve on the split pass of an unsharded dataset", B, SPLIT_LANES);
  if (split && B > ROWDOT_LANES) SLM_TRY(ensure_xt(ds));  // rowdot_mfma_kernel reads the column-major copy (optional)
  int64_t total_points = 0;
  bool any_rw = false, any_gn = false;
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
  slm_engine* eng = ds->eng;
  const bool sharded = row_sharded(ds);  // (a replica on an engine with a communicator -- grid mode -- is not)
  HIP_TRY(hipSetDevice(eng->device));
  hipStream_t s = eng->stream;
  // Uploads from the caller's buffers and from the dataset's staging area are asynchronous: whichever way this
  // function is left, the stream is drained first (on the normal path it already is: a no-op then).
  struct DrainOnExit {
    hipStream_t s;
    ~DrainOnExit() { (void)hipStreamSynchronize(s); }
  } drain_on_exit{s};
  const auto t_begin = std::chrono::steady_clock::now();
  auto t_mark = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); };
  double tr[6] = {0, 0, 0, 0, 0, 0};
  const int64_t p = ds->p, ld = ds->ld, n = ds->n;
  const int G = ds->G;

  slm_solve_opts o;
  memset(&o, 0, sizeof(o));
  if (opts) o = *opts;
  if (!(o.tol > 0.0)) o.tol = 1e-8;
  if (o.max_iter <= 0) o.max_iter = 10000;
  const bool profile = (o.flags & SLM_FLAG_PROFILE) != 0;

  // ---- per-lane row weights / scaling -----------------------------------------------------------
  LaneSetup ls = default_lanes(ds, B);
  double wmax[SLM_MAX_CELLS];  // largest row weight of each lane (< 0: unknown)
  for (int l = 0; l < kMaxCells; ++l) wmax[l] = ds->rw ? ds->rw_max : 1.0;
  double rw_fp[SLM_MAX_CELLS][2] = {};  // two checksums of each lane's row weights (carried starts, below)
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
  const double tr_rw = t_mark();
  bool custom_scale = false;
  for (int l = 0; l < B; ++l)
    if (lanes[l].n_eff > 0) {
      ls.n_eff[l] = (double)lanes[l].n_eff;
      custom_scale = true;
    }

  const bool small = small_ok(ds, o.flags);
  // ---- covariance passes: every row set of the call has its Gram (slm_dataset_covariance) --------------------------
  int cov_entry[SLM_MAX_CELLS] = {};
  bool cov_on = false;
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
  // ---- Lipschitz constants -----------------------------------------------------------------------
  double L[SLM_MAX_CELLS];
  double lipschitz_ms = 0.0;
  bool L_on_device = false;  // the estimate stays on the device (no host round trip before the first pass)
  bool L_kept = false;       // ... in the dataset's kept slot (lambda[lane_cap]) rather than in lane 0's
  // the sketch's estimate for the dataset's own rows and weights, computed once and kept on the device
  auto kept_sketch = [&]() -> int {
    if (ds->sketch_valid && !(o.flags & SLM_FLAG_FRESH_L) && getenv("SLM_NO_SKETCH_CACHE") == nullptr) return SLM_OK;
    SLM_TRY(power_iteration(ds, default_lanes(ds, 1), nullptr, sketch_iters(), sketch_rows(ds->n)));
    HIP_TRY(hipMemcpyAsync(ds->lambda + ds->lane_cap, ds->lambda, sizeof(double), hipMemcpyDeviceToDevice, eng->stream));
    ds->sketch_valid = true;
    return SLM_OK;
  };
  double L_factor[SLM_MAX_CELLS];  // ... and lane l uses L_factor[l] times it
  for (int l = 0; l < kMaxCells; ++l) L_factor[l] = 1.0;
  if (o.L > 0.0 || small) {  // (the on-chip solver bounds its own steps from the Gram matrix)
    for (int l = 0; l < B; ++l) L[l] = o.L > 0.0 ? o.L : 1.0;
  } else {
    const auto t0 = std::chrono::steady_clock::now();
    bool ran = false;
    // working-set solves barely use L (first candidate, fallback steps): a bound from the first thirty-second
    // of the rows, three power steps, costs a sixth of the two full passes
    const bool sketch = ws_policy(ds, o.flags) == 2 && n >= 65536 && !getenv("SLM_NO_L_SKETCH");
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
  bool carry = false;
  if (ds->carry_valid && !small && !shared_path && B <= ds->carry_lanes && !(o.flags & SLM_FLAG_COLD_START) &&
      getenv("SLM_NO_CARRY") == nullptr) {
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
  int ws_set_of[SLM_MAX_LANES] = {}, ws_set_lane[SLM_MAX_LANES] = {}, ws_n_sets = 0;
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
  bool ws_carry = carry && ds->ws_carry_valid && ws_policy(ds, o.flags) == 2 && ws_n_sets == ds->ws_carry_sets &&
                  ds->ws_sets >= ws_n_sets && ds->ws_carry_cov == cov_on && getenv("SLM_NO_WS_CARRY") == nullptr;
  for (int l = 0; l < B && ws_carry; ++l) ws_carry = ws_set_of[l] == ds->ws_carry_set_of[l];
  ds->ws_carry_valid = false;
  PathCtl* h = ds->h_stage;  // (lives as long as the dataset: the upload below is asynchronous)
  memset(h, 0, sizeof(ds->h_stage));
  SetupArgs su;
  memset(&su, 0, sizeof(su));
  su.beta = ds->beta; su.z = ds->z; su.zprev = ds->zprev; su.gprev = ds->gprev;
  su.g = ds->g;
  su.carry = carry ? 1 : 0;
  if (carry)
    for (int l = 0; l < B; ++l) su.carry_loss[l] = ds->carry_lane[l].loss;
  su.a0 = ds->a0; su.b0 = ds->b0; su.d0 = ds->d0;
  // (small solves keep their per-point records inside the control block: one blocking copy less at the end)
  const bool infos_in_snap = total_points <= kSnapInfos;
  slm_point_info* d_infos = infos_in_snap ? ds->dctl->infos : ds->infos;
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
      const slm_host::LaneWalk w = slm_host::interleaved_walk(l, B, total_points, getenv("SLM_NO_TAIL_BAND") == nullptr,
                                                              getenv("SLM_NO_SLACK_DEEP") == nullptr);
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
    HIP_TRY(hipMemcpyAsync(ds->pts, ds->h_pts, sizeof(slm_path_point) * (size_t)total_points, hipMemcpyHostToDevice, s));
  }
  hipLaunchKernelGGL(solve_setup_kernel, dim3(128), dim3(256), 0, s, su);
  HIP_TRY(hipMemcpyAsync(ds->ctl, h, sizeof(PathCtl) * B, hipMemcpyHostToDevice, s));
  static_assert(offsetof(DevCtl, lane) >= offsetof(DevCtl, ws) + sizeof(WsCtl) && offsetof(DevCtl, g) == 0, "g, ws, lane");
  // stop words and working-set counters (a working set taken over from the solve before keeps its block: ws_setup)
  HIP_TRY(hipMemsetAsync(ds->dctl, 0, ws_carry ? offsetof(DevCtl, ws) : offsetof(DevCtl, lane), s));
  if (L_on_device) {  // the power steps are still in flight: their result goes into the control blocks on the device
    SeedArgs sa;
    sa.ctl = ds->ctl; sa.lambda = L_kept ? ds->lambda + ds->lane_cap : ds->lambda; sa.n_lanes = B; sa.margin = 1.08;
    for (int l = 0; l < kMaxLanes; ++l) sa.factor[l] = L_factor[l];
    hipLaunchKernelGGL(seed_step_kernel, dim3(1), dim3(64), 0, s, sa);
  }
  tr[0] = t_mark();
  // (no wait here: the caller's buffers outlive the call, the control blocks are staged in the dataset, and
  //  everything the host still has to prepare overlaps with the step-size seed running on the device)
  tr[1] = t_mark();

  TailArgs ta;
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

  // results: lanes whose host buffers follow each other (the ranges of one shared path do) travel in one
  // copy -- a device-to-host copy into pageable memory costs ~40 us before the first byte moves.  Queued on the
  // solve's stream; the caller waits for it.
  auto enqueue_result_copies = [&]() -> int {
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
  };
  // ---- problems that fit a workgroup: one launch for the whole call (small_kernels.hpp) -------------------------
  if (small) {
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
    bool staged_out = (size_t)total_points * (size_t)p <= kSmallOutDoubles && !any_gn && getenv("SLM_NO_SMALL_STAGE") == nullptr;
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
    if (unconverged && getenv("SLM_ON_CHIP_NO_FALLBACK") == nullptr) {  // (the variable: diagnostics -- the on-chip records as they are)
      if (const char* trc = getenv("SLM_TRACE"))
        if (trc[0] == '2') fprintf(stderr, "[slm] on-chip solve gave a point up after %.3f ms (%lld products): the general path takes the call\n", t_small, (long long)sweeps);
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
    if (const char* trc = getenv("SLM_TRACE"))
      if (trc[0] == '2') {
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
  bool use_ws = false, ws_late = false;
  WsArgs wa;
  memset(&wa, 0, sizeof(wa));
  {
    const int pol = ws_policy(ds, o.flags);
    use_ws = pol == 2;
    // (row-sharded: the switch would change the collectives of a pass on the strength of one rank's state)
    ws_late = pol == 1 && !sharded;
  }
  if (sharded && !ds->stop_words) SLM_TRY(dalloc(&ds->stop_words, STOP_WORDS));
  auto ws_setup = [&](bool late) -> int {
    // lanes with the same row weights (same host pointer: the folds of a CV grid) and the same 1/n
    // scaling share one Gram
    const int* set_of = ws_set_of;
    const int* set_lane = ws_set_lane;
    const int n_sets = ws_n_sets;
    const int ws_nblk = (int)std::max<int64_t>(1, std::min<int64_t>(eng->cus, n / 64));  // (2 MiB of partials each)
    // (each on its own: slm_eval_sse_sparse may already have brought idx and XW in)
    if (!ds->ws_idx) SLM_TRY(dalloc(&ds->ws_idx, WS_KCAP));
    if (!ds->ws_gs) SLM_TRY(dalloc(&ds->ws_gs, WS_KCAP));
    if (!ds->ws_gl) SLM_TRY(dalloc(&ds->ws_gl, WS_KCAP));
    if (!ds->ws_pos) SLM_TRY(dalloc(&ds->ws_pos, (size_t)ld));
    if (!ds->ws_score) SLM_TRY(dalloc(&ds->ws_score, (size_t)ld));
    if (!ds->ws_XW) SLM_TRY(dalloc(&ds->ws_XW, (size_t)n * WS_KCAP));
    if (!ds->ws_nt && !getenv("SLM_NO_DIRECT")) SLM_TRY(dalloc(&ds->ws_nt, (size_t)kMaxLanes * NT_SCRATCH));
    if (ds->ws_sets < n_sets) {
      dfree(ds->ws_part); dfree(ds->ws_G); dfree(ds->ws_Gx);
      ds->ws_sets = 0;
      if (sharded) SLM_TRY(dalloc(&ds->ws_Gx, (size_t)n_sets * WS_KCAP * WS_KCAP + STOP_WORDS));
      SLM_TRY(dalloc(&ds->ws_part, (size_t)ws_nblk * n_sets * WS_KCAP * WS_KCAP));  // (ws_nblk depends on n only)
      SLM_TRY(dalloc(&ds->ws_G, (size_t)n_sets * WS_KCAP * WS_KCAP));
      ds->ws_sets = n_sets;
    }
    // column-major copy of X (a layout of the data like the padded row-major one: depends on nothing
    // but X, kept for the life of the dataset; 2 ms for 4 GB).  Optional: without the memory for it
    // the gathers read the row-major X, one 64-byte sector per element.
    SLM_TRY(ensure_xt(ds));
    // (initialised on the device: a host-side copy would need the stream drained before its buffer goes away)
    if (late) HIP_TRY(hipMemsetAsync(ds->ws_ctl, 0, sizeof(WsCtl), s));  // (a fresh solve has cleared it already)
    if (ws_carry && !late) hipLaunchKernelGGL(ws_ctl_carry_kernel, dim3(1), dim3(256), 0, s, ds->ws_ctl, 24);
    else hipLaunchKernelGGL(ws_ctl_init_kernel, dim3(1), dim3(64), 0, s, ds->ws_ctl, 24);
    wa.ws = ds->ws_ctl;
    wa.idx = ds->ws_idx; wa.pos = ds->ws_pos; wa.gs = ds->ws_gs; wa.gl = ds->ws_gl;
    wa.score = ds->ws_score; wa.XW = ds->ws_XW; wa.part = ds->ws_part; wa.Gm = ds->ws_G;
    wa.nt = getenv("SLM_NO_DIRECT") ? nullptr : ds->ws_nt;
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
    wa.theta = 0.85;
    // measured on the headline path (tools/ws_sweep.py, 24 combinations within 8 % of each other):
    // theta 0.85 / look-ahead 2 / 16 newcomers per pass / 112 initial columns was the fastest
    wa.lookahead = 2;
    wa.append_max = 48;  // (interleaved lanes need the next band of the path at once; elsewhere 16 cost a pass now and then)
    // The first selection.  A path that walks down from alpha_max on interleaved lanes starts small: 112 columns (160: the
    // same; 208: 4.0 ms per headline path against 3.65).  Lanes that start cold at an alpha of their own (single fits, the
    // pieces of a grid's paths) have nothing that limits what is active at their first point: up to 256 columns -- a
    // selection cut short is repaired at 48 columns per pass, and every repair is a pass over X (one cold Lasso point with
    // 300 informative features: 3-5 passes from 112 columns, 2-4 from 256; 384 costs a sparse fit at a noise-level alpha
    // a millisecond of Gram and model solves on 330 noise columns, tools/single_fit_big.py).  Groups bring their features
    // in blocks: up to 384 (config 5's cold solve: 5 passes from 256 columns, 2 from 384 -- 13.2 -> 8.3 ms per fit of three
    // solves; configs 3 and 4 the same either way).
    wa.k_init = ds->singleton ? (shared_path ? 112 : 256) : 384;
    // tuning knobs (tools/ws_sweep.py)
    if (const char* th = getenv("SLM_WS_THETA")) {
      const double v = atof(th);
      if (v > 0.0 && v <= 1.0) wa.theta = v;
    }
    if (const char* e = getenv("SLM_WS_LOOKAHEAD")) wa.lookahead = std::max(0, std::min(64, atoi(e)));
    if (const char* e = getenv("SLM_WS_APPEND")) wa.append_max = std::max(1, std::min(WS_KCAP, atoi(e)));
    if (const char* e = getenv("SLM_WS_KINIT")) wa.k_init = std::max(16, std::min(WS_KCAP, atoi(e)));
    wa.bb_steps = 1;
    if (const char* e = getenv("SLM_WS_BB")) wa.bb_steps = atoi(e) != 0;
    wa.one_solver = 0;
    if (const char* e = getenv("SLM_WS_ONE_SOLVER")) wa.one_solver = atoi(e) != 0;
    wa.hard_call = getenv("SLM_HARD_CALLWIDE") != nullptr;
    wa.power_iters = 10;
    if (const char* e = getenv("SLM_WS_POWER_ITERS")) wa.power_iters = std::max(1, std::min(40, atoi(e)));
    wa.miss_factor = 4;
    if (const char* e = getenv("SLM_WS_MISS_FACTOR")) wa.miss_factor = std::max(1, std::min(8, atoi(e)));
    // (0.5 left a first working set of 56-112 columns to the luck of the bisection: 65 on one draw of the headline's law, 102 on
    //  another -- and the features of the first band's deepest points outside the small ones; 0.75: the soak law's twelve paths 92.6 -> 85.0 ms)
    // (per-feature penalties only: groups come in blocks and start from 384 columns -- config 3's first set 250 -> 300 columns at
    //  0.75, 2.79 -> 3.44 ms per path)
    // (and shared paths only: a single cold fit pays for the larger first set without a band of points to serve with it --
    //  1.45 -> 1.51 ms at 0.3 alpha_max, 2.46 -> 2.78 ms at 0.005, tools/single_fit_big.py)
    wa.fill = (ds->singleton && shared_path) ? 0.75 : 0.5;
    if (const char* e = getenv("SLM_WS_FILL")) wa.fill = std::max(0.1, std::min(1.0, atof(e)));
    return SLM_OK;
  };
  // no memory for the working-set buffers: the plain iteration still works (unless this solve runs
  // more lanes than the fused kernels serve, which only the split pass can do)
  auto ws_release = [&]() {
    ds->ws_carry_valid = false;
    dfree(ds->ws_idx); dfree(ds->ws_pos); dfree(ds->ws_gs); dfree(ds->ws_gl);
    dfree(ds->ws_score); dfree(ds->ws_XW); dfree(ds->ws_part); dfree(ds->ws_G); dfree(ds->ws_Gx); dfree(ds->ws_nt);
    ds->ws_sets = 0;
    (void)hipGetLastError();
  };
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
  const int* done_flag = &ds->gctl->done;
  // the gradient of one pass: split pass (sixteen lane slots, residuals from the gathered columns where
  // possible) when the working set runs from the start, the fused kernel otherwise
  auto enqueue_pass_gradient = [&](hipEvent_t e0, hipEvent_t e1) -> int {
    if (cov_on) return enqueue_gradient_cov(ds, B, cov_entry, done_flag, e0, e1, ds->ctl, use_ws ? &wa : nullptr);
    if (split) return enqueue_gradient_split(ds, ls, ds->y, done_flag, ds->ctl, &wa, e0, e1);
    return enqueue_gradient(ds, ls, ds->y, done_flag, e0, e1);
  };
  int ws_comm_rc = 0;  // first RCCL error of the per-pass Gram all-reduce (checked after each chunk)
  bool mg_handover = false;  // the rounds on the model Gram are on: tail points change hands (tail_handover_kernel)
  // everything that follows the gradient of one pass
  // (in two halves: behind the pass a solve is expected to end with, the second half waits for the verdict)
  auto enqueue_tail = [&]() {
    launch_tail(ta, s);
    if (shared_path && !interleave) hipLaunchKernelGGL(steal_kernel, dim3(1), dim3(256), 0, s, ta);
    // (the dense end of an interleaved path: finished lanes take over tail points their owners have not started)
    if (shared_path && interleave && mg_handover) hipLaunchKernelGGL(tail_handover_kernel, dim3(1), dim3(64), 0, s, ta);
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
  };
  bool fix_start = false;  // the refinement being queued follows the pass on the row sample (sample start, below)
  auto enqueue_refinement = [&]() {
    if (use_ws) {
      {
        const int bs = ds->singleton ? 256 : 64;
        const int64_t items = ds->singleton ? p : 16 * (int64_t)G;  // groups: one thread per (group, lane)
        hipLaunchKernelGGL(ws_score_kernel, dim3((unsigned)((items + bs - 1) / bs)), dim3(bs), 0, s, ta, wa);
      }
      hipLaunchKernelGGL(ws_select_kernel, dim3(1), dim3(WS_THREADS), 0, s, ta, wa);
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
      if (wa.one_solver && wa.nt) {
      } else if (ds->singleton) hipLaunchKernelGGL((ws_solve_kernel<false, false>), dim3(B), dim3(WS_THREADS), 0, s, ta, wa);
      else hipLaunchKernelGGL((ws_solve_kernel<true, false>), dim3(B), dim3(WS_THREADS), 0, s, ta, wa);
      if (wa.nt) {
        if (ds->singleton) hipLaunchKernelGGL((ws_solve_kernel<false, true>), dim3(B), dim3(WS_THREADS), 0, s, ta, wa);
        else hipLaunchKernelGGL((ws_solve_kernel<true, true>), dim3(B), dim3(WS_THREADS), 0, s, ta, wa);
      }
    }
  };

  // ---- queue iterations; the device decides when each point / lane / the solve is finished ------
  int chunk = o.check_every;
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
  const int64_t max_total = (int64_t)max_points * o.max_iter;
  int64_t enq = 0;
  int slot = 0;
  bool pending[2] = {false, false};
  bool done = false;

  // (hipGraph replay of a chunk of passes was tried in round 1 and removed: the loop is bound by the ~1.5 us
  //  dependent-kernel boundaries on the device, not by host launches -- 18.7 against 16.9 us per three-kernel pass on
  //  small problems -- and instantiation cost 0.6 ms per solve)
  tr[2] = t_mark();
  // Working-set solves from the start verify one point per lane and pass, after the pass at zero: the
  // queue is cut to end exactly there, and polls go pass by pass after it (a miss adds a pass or two).
  // Without this a 5-pass path drags three queued no-op passes behind it (12 launches each).
  int64_t expected = 0;
  if (use_ws && !ws_late && o.check_every <= 0) {
    int64_t most = 0;
    for (int l = 0; l < B; ++l) {
      int64_t mine = lanes[l].n_points;
      if (shared_path && interleave) mine = slm_host::interleaved_points(slm_host::LaneWalk{h[l].pt_lo, h[l].n_points, h[l].stride, h[l].tail_pt});
      most = std::max<int64_t>(most, mine);
    }
    expected = 1 + most;
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
  int64_t n_sample = 0;
  {
    bool cold = true;
    for (int l = 0; l < B; ++l) cold = cold && lanes[l].beta0 == nullptr;
    if (cold && (shared_path || getenv("SLM_SAMPLE_START_ALL")) && use_ws && !ws_late && !sharded && !cov_on && split && !any_rw && !ds->rw &&
        !custom_scale && expected > 0 && o.max_iter >= 4 && !(o.flags & SLM_FLAG_FISTA_ONLY) && getenv("SLM_NO_SAMPLE_START") == nullptr) {
      int64_t least = 65536;  // (below it a pass costs little more than the launches of the sample's)
      if (const char* e = getenv("SLM_SAMPLE_START_MIN_ROWS")) least = std::max<int64_t>(64, atoll(e));  // (tests)
      // A quarter of the rows (round 4: an eighth).  The sample has to rank the features of the first band's DEEPEST point
      // above the noise features: a gradient entry of the sample carries noise sd(y) / sqrt(rows) -- on the headline's law
      // 3.7 from an eighth of the rows, 2.6 from a quarter -- and the largest of 5 000 noise entries is 3.7 sd: from an
      // eighth the features entering at point 16-17 of eighteen lanes (|beta| about 9) sit INSIDE the noise features' range
      // (67 of those above 9), from a quarter above it (2).  Measured over eight draws of the headline's law
      // (tools/headline_data_seeds.py): eighteen lanes 34 passes / 4.43 ms per path on an eighth, 30 / 3.75 on a quarter,
      // 27 / 3.39 on a half; sixteen lanes 4.11 / 3.89 / 3.78; the bench's own draw 2.86 either way; the soak law's twelve
      // 91.6 -> 92.8 ms in total (a half: 98.2).  SLM_SAMPLE_DIV sets the divisor.
      int div = 4;
      if (const char* e = getenv("SLM_SAMPLE_DIV")) div = std::max(1, std::min(64, atoi(e)));
      if (n >= least) n_sample = n / div;
    }
  }
  // ---- model Gram (mg_kernels.hpp) -----------------------------------------------------------------------------------
  // Lanes whose solutions outgrow the working set used to finish with plain steps, two reads of X each.  When a snapshot
  // shows that this has begun (WsCtl::outgrown: a selection did not fit the working set's 512 columns), the
  // model Gram of the dataset is built -- once, it stays with the dataset -- and every later pass is followed by a round
  // of proximal-gradient steps on it for the lanes the working set does not serve (mg_enqueue_round).  From then on the
  // host looks at every pass's snapshot before it queues the next: a round is sized by what the last one needed.
  const char* mg_env = getenv("SLM_MG");
  const bool mg_forced = mg_env != nullptr && mg_env[0] == '2';  // (tests: any size, from the first snapshot on)
  // (the model Grams of a dataset, fp32, are kept within 3 GB: sixteen row sets at p = 5 000, seven at 10 000)
  const int mg_cap = slm_host::model_gram_cap(ld, 3.0e9, kMgEntries);
  const bool mg_ok = use_ws && split && (big_x || mg_forced) && !sharded && !cov_on && !(o.flags & SLM_FLAG_NO_MODEL_GRAM) &&
                     (size_t)ds->lane_cap >= (size_t)kMaxLanes && ws_n_sets <= mg_cap && mg_possible(ds);
  if (mg_ok && mg_forced) expected = 0;  // (tests: polled from the first chunk on, so that short solves reach the rounds too)
  // A dataset that already holds the model Gram of every row set of this call (an earlier solve outgrew the working set
  // and built them: the same path again, a refit, the next search on the data) will be served by the rounds the moment
  // its selection stops fitting: the working set then stays as it is from the first overflow on, instead of being
  // selected, gathered and multiplied afresh once (2-3 ms at 500 columns) before the host has seen the counter.
  // (lanes on the dataset's own rows only: other row sets are told apart by fingerprints, a kernel and a round trip)
  if (mg_ok && !mg_forced && !ds->mg.empty() && getenv("SLM_NO_MG_KEEP") == nullptr) {
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
  bool mg_on = false;
  int mg_inner = 20;
  double mg_build_ms = 0.0;
  int mg_entry_of_set[SLM_MAX_LANES] = {};
  // the model Gram of every row set of the call (ws_set_of: lanes with the same row weights and scaling share one), found
  // by the fingerprint of the row weights as the lanes brought them, or built
  int mg_built = 0;  // model Grams this call had to build
  auto mg_sets = [&]() -> int {
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
  };
  auto mg_wanted = [&](const DevCtl& c) -> bool {
    if (mg_forced) return true;
    // (capacity, not difficulty: a lane that spends passes on an ill-conditioned face inside the working set is served by
    //  the model solver's direct steps, and the set must stay free to be selected afresh there)
    return c.ws.outgrown > 0 || c.ws.overflows > 0 || c.ws.disabled != 0;
  };
  auto mg_consider = [&](const DevCtl& c) -> int {
    if (!mg_ok || mg_on || !mg_wanted(c)) return SLM_OK;
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = mg_sets();
    if (rc == SLM_OK) {
      mg_on = true;
      mg_handover = getenv("SLM_NO_HANDOVER") == nullptr;
      wa.keep_full = 1;  // (from here on the working set serves what it holds: enqueue_refinement passes wa by value)
      if (mg_built > 0) mg_build_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    } else if (rc != SLM_ERR_OOM) {
      return rc;
    }
    return SLM_OK;  // (no memory for it: the solve goes on with plain steps)
  };
  const int64_t prof_off = n_sample > 0 ? 1 : 0;  // (the pass on the sample is no launch of the roofline's kernel on X)
  bool results_queued = false, results_final = false;  // the result copies were queued early / and hold the final state
  int final_slot = 0;          // the snapshot in which the host saw `done`
  bool deferred = false;       // the refinement behind the last queued pass has not been queued yet
  // SLM_TRACE=3: one line per polled snapshot -- where every lane stands, the working set, the model Gram's rounds
  // (with SLM_TRACE_POLL=1 every pass is polled: a diagnostic, the queue then drains between passes)
  const char* trc3 = getenv("SLM_TRACE");
  const bool trace3 = trc3 != nullptr && trc3[0] == '3';
  if (trace3 && getenv("SLM_TRACE_POLL") != nullptr && expected > 0) expected = n_sample > 0 ? 2 : 1;  // (the sample pass's refinement is never held back)
  auto trace_pass = [&](const DevCtl& now) {
    if (!trace3) return;
    fprintf(stderr, "[slm] pass %lld at %.3f ms: K %d builds %d appends %d misses %d stale %d refined %d | mg on %d rounds %d inner %d most %d rej %d | lanes (point.iter/flags):",
            (long long)enq, t_mark(), now.ws.Kreal, now.ws.builds, now.ws.appends, now.ws.misses, now.ws.stale, now.ws.refined, (int)mg_on,
            now.mg.rounds, now.mg.inner_iters, now.mg.most_iters, now.mg.rejected);
    for (int l = 0; l < B; ++l)
      fprintf(stderr, " %d.%d%s%s%s", now.lane[l].point, now.lane[l].iter, now.lane[l].done ? "d" : "", now.lane[l].zsup ? "w" : "",
              l < SLM_MAX_LANES && now.mg.lane[l].active ? "m" : "");
    fprintf(stderr, "\n");
  };
  while (!done) {
    {
      // (a solve with an expected end queues all of its passes at once: launches behind the device-side stop flag return
      //  at once, and every snapshot in between -- a copy, an event, 6 us of idle stream around them -- told the host
      //  nothing it acts on)
      const int this_chunk = mg_on ? 1 : (expected <= 0 ? chunk : (enq < expected ? (int)std::min<int64_t>(64, expected - enq) : 1));
      for (int i = 0; i < this_chunk; ++i) {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (profile && enq >= prof_off && (enq - prof_off) % kProfStride == 0) {  // sampled: an event pair costs ~8 us of stream time
          const int64_t slot_id = (enq - prof_off) / kProfStride;
          while ((int64_t)ds->prof.size() < 2 * (slot_id + 1)) {
            hipEvent_t ev;
            HIP_TRY(hipEventCreate(&ev));
            ds->prof.push_back(ev);
          }
          e0 = ds->prof[2 * slot_id];
          e1 = ds->prof[2 * slot_id + 1];
        }
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
        // behind the pass the solve is expected to end with, the six launches of the refinement would only find
        // out that there is nothing left to refine (30 us): they follow once the snapshot says otherwise
        deferred = (expected > 0 && enq == expected && !sharded) || mg_on;
        if (!deferred) enqueue_refinement();
        fix_start = false;
      }
      SLM_TRY(check_launch());
      if (ws_comm_rc != 0) return ws_comm_rc;  // (all_reduce_sum has set the message)
      HIP_TRY(hipMemcpyAsync(&ds->hctl[slot].c, ds->dctl, sizeof(DevCtl), hipMemcpyDeviceToHost, s));
    }
    // The pass the solve is expected to end with: the host waits for THIS chunk instead of queueing another pass
    // behind it -- when the solve does end there (the usual case) the snapshot is final and only the coefficients
    // remain to be fetched.  Polling one chunk behind cost a queued pass that returned at once (eighteen launches,
    // 0.09 ms) and four blocking copies (0.2 ms of host round trips) on every 5 ms path.  A solve that overruns
    // gets a few more passes polled this way, then the pipelined polls.
    const bool at_end = mg_on || (expected > 0 && enq >= expected && (enq < expected + 4 || (trace3 && getenv("SLM_TRACE_POLL") != nullptr)));
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
    stats->grad_launches = passes - (carry ? 1 : 0) - prof_off;  // launches over the data that did work (every launch serves all lanes)
    stats->grad_ms_total = 0.0;
    stats->grad_timed = 0;
    if (profile) {
      double tot = 0.0;
      int64_t cnt = 0;
      // iterations 0, kProfStride, 2 kProfStride, ... below `passes` did real work and were timed
      for (int64_t k = 0; k * kProfStride < passes - prof_off && 2 * k + 1 < (int64_t)ds->prof.size(); ++k) {
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
      const char* trc = getenv("SLM_TRACE");
      if (trc && trc[0] == '2') {
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
  if (const char* trc = getenv("SLM_TRACE"))  // 1: slow solves only, 2: every solve (cumulative ms since entry)
    if (tr[4] > 15.0 || trc[0] == '2')
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
  const bool interleaved = ds->singleton && !getenv("SLM_NO_INTERLEAVE");  // (solve_core: per-feature penalties take the points in turn)
  int B = slm_host::auto_path_lanes(n_points, cap, big, !interleaved);
  if (const char* e = getenv("SLM_AUTO_LANES")) B = std::max(1, std::min<int>(std::min(atoi(e), cap), n_points));  // (A/B runs)
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
  if (const char* env = getenv("SLM_ON_CHIP"))
    if (env[0] == '0') return fail(SLM_ERR_UNSUPPORTED, "the on-chip solvers are switched off (SLM_ON_CHIP=0)");
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
  if (const char* trc = getenv("SLM_TRACE"))
    if (trc[0] == '2')
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


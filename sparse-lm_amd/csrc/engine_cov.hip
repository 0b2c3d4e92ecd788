// Host side of the MI355X fit engine, third unit: the Grams of covariance passes (SLM_FLAG_COVARIANCE) -- one row set at a
// time, the folds of a K-fold split at once, and, for replicas among ranks, each rank's share of the rows with one sum over
// the ranks per fold (grid mode's only collective; the reference dispatches whole fits,
// /root/reference/src/sparselm/model_selection.py:273,304-323).
#include "engine_internal.hpp"

// ------------------------------------------------------------------------------------------------
// covariance passes: the Gram of a row set
// ------------------------------------------------------------------------------------------------
// C = A^T A for the row-major rows x ld block A, the square with its mirror.  One row set at a time used to mean
// cov_syrk_kernel<3>, a 96-column tile pair per workgroup over ALL the rows: 56 ms for the 100 000 x 5 000 of the headline
// (46 TFLOP/s) and 12 ms whatever the width below 1 600 columns (a dozen tile pairs do not fill the device).  The rows are
// cut into up to sixteen chunks instead, the batched kernel of the folds (cov_syrk_packed_kernel: 69 TFLOP/s, the batch is
// what fills the device) multiplies each, and the packed triangles are summed in a fixed order and unpacked.  Chunks are
// multiples of sixteen rows read in place -- the ring's last prefetch then reads rows that follow in the block and are never
// multiplied -- but for the last, which is copied behind zero rows (cov_syrk_padded_rows).  SLM_COV_TILE=3/4: the old kernel
// (tests compare the two).
static int cov_gram(slm_dataset* ds, const double* A, int64_t rows, double* C) {
  slm_engine* eng = ds->eng;
  hipStream_t s = eng->stream;
  const int64_t ld = ds->ld;
  if (rows < 1) {
    HIP_TRY(hipMemsetAsync(C, 0, sizeof(double) * (size_t)ld * ld, s));
    return SLM_OK;
  }
  if (knobs().cov_tile != 0) {
    const int side = knobs().cov_tile == 3 ? 3 : 4;  // (96 or 128 columns per workgroup)
    const int nt = (int)((ld + 32 * side - 1) / (32 * side));
    const dim3 grid((unsigned)(nt * (nt + 1) / 2));
    if (side == 3) hipLaunchKernelGGL(cov_syrk_kernel<3>, grid, dim3(256), 0, s, A, rows, ld, C);
    else hipLaunchKernelGGL(cov_syrk_kernel<4>, grid, dim3(256), 0, s, A, rows, ld, C);
    return check_launch();
  }
  const int n_tiles = cov_syrk_tiles(ld);
  const int wgs = (n_tiles + 3) / 4;
  int B = (int)std::min<int64_t>(kMaxLanes, std::max<int64_t>(4, (4 * (int64_t)eng->cus + wgs - 1) / wgs));  // ~four rounds of workgroups
  B = (int)std::max<int64_t>(1, std::min<int64_t>(B, rows / 64));
  const int64_t chunk = B > 1 ? (rows / B) / 16 * 16 : 0;            // rows of chunks 0 .. B - 2 (in place)
  const int64_t last = rows - (int64_t)(B - 1) * chunk;              // rows of the last one (copied, padded)
  const size_t tri = ((size_t)ld * (size_t)(ld + 1) / 2 + 15) / 16 * 16;
  const int64_t last_pad = cov_syrk_padded_rows(last);
  double *packed = nullptr, *tail = nullptr, *sum = nullptr;
  int rc = dalloc(&packed, (size_t)B * tri);
  if (rc == SLM_OK) rc = dalloc(&tail, (size_t)last_pad * (size_t)ld);
  if (rc == SLM_OK && B > 1) rc = dalloc(&sum, tri);
  if (rc == SLM_OK) {
    const double* src = A + (size_t)(B - 1) * (size_t)chunk * (size_t)ld;
    hipError_t he = hipMemcpyAsync(tail, src, sizeof(double) * (size_t)last * (size_t)ld, hipMemcpyDeviceToDevice, s);
    if (he == hipSuccess && last_pad > last)
      he = hipMemsetAsync(tail + (size_t)last * (size_t)ld, 0, sizeof(double) * (size_t)(last_pad - last) * (size_t)ld, s);
    if (he != hipSuccess) rc = fail(SLM_ERR_HIP, "covariance build: %s", hipGetErrorString(he));
  }
  if (rc == SLM_OK) {
    SyrkBatch sb;
    memset(&sb, 0, sizeof(sb));
    for (int b = 0; b < B; ++b) {
      sb.A[b] = b + 1 < B ? A + (size_t)b * (size_t)chunk * (size_t)ld : tail;
      sb.rows[b] = b + 1 < B ? chunk : last;
      sb.P[b] = packed + (size_t)b * tri;
    }
    hipLaunchKernelGGL(cov_syrk_packed_kernel, dim3((unsigned)wgs, (unsigned)B), dim3(256), 0, s, sb, ld, n_tiles);
    const double* total = packed;
    if (B > 1) {
      hipLaunchKernelGGL(cov_sum_kernel, dim3(1024), dim3(256), 0, s, packed, B, (int64_t)tri, (int64_t)tri, sum);
      total = sum;
    }
    const int64_t nt32 = (ld + 31) / 32;
    hipLaunchKernelGGL(cov_unpack_kernel, dim3((unsigned)(nt32 * (nt32 + 1) / 2)), dim3(256), 0, s, total, (const double*)nullptr, 1.0, ld, C);
    rc = check_launch();
  }
  (void)hipStreamSynchronize(s);  // (the staging blocks go back to the pool below)
  dfree(packed);
  dfree(tail);
  dfree(sum);
  return rc;
}

// the Gram of the rows `rows_host[0..count)` of X (gathered into a block of its own), unscaled, into C
static int cov_gram_of_rows(slm_dataset* ds, const std::vector<int64_t>& rows_host, double* C) {
  hipStream_t s = ds->eng->stream;
  const int64_t ld = ds->ld;
  if (rows_host.empty()) {
    HIP_TRY(hipMemsetAsync(C, 0, sizeof(double) * (size_t)ld * ld, s));
    return SLM_OK;
  }
  int64_t* rows = nullptr;
  double* block = nullptr;
  int rc = dalloc(&rows, rows_host.size());
  if (rc == SLM_OK) rc = dalloc(&block, rows_host.size() * (size_t)ld);
  if (rc == SLM_OK) {
    hipError_t he = hipMemcpyAsync(rows, rows_host.data(), sizeof(int64_t) * rows_host.size(), hipMemcpyHostToDevice, s);
    if (he != hipSuccess) rc = fail(SLM_ERR_HIP, "covariance build: %s", hipGetErrorString(he));
  }
  if (rc == SLM_OK) {
    hipLaunchKernelGGL(cov_rows_kernel, dim3((unsigned)rows_host.size()), dim3(256), 0, s, ds->X, ld, rows, nullptr,
                       (int64_t)rows_host.size(), block);
    rc = cov_gram(ds, block, (int64_t)rows_host.size(), C);
  }
  (void)hipStreamSynchronize(s);  // (the staging blocks go back below; the index list is host memory of the caller)
  dfree(rows);
  dfree(block);
  return rc;
}

// files the entry of a row set whose scaled Gram G is ready: c = X^T W y / n and y^T W y / n from a standard pass at z = 0.
// Takes G over (it goes back to the pool if anything fails).
static int cov_file_entry(slm_dataset* ds, const double* wdev, double n_eff, const double fp[2], double* G) {
  hipStream_t s = ds->eng->stream;
  const int64_t ld = ds->ld;
  slm_dataset::CovEntry e;
  e.n_eff = n_eff; e.fp1 = fp[0]; e.fp2 = fp[1];
  e.G = G;
  struct EntryGuard {  // (whichever way this function is left before the entry is filed, its blocks go back)
    slm_dataset::CovEntry* e;
    hipStream_t s;
    ~EntryGuard() {
      if (!e) return;
      (void)hipStreamSynchronize(s);
      dfree(e->G);
      dfree(e->c);
    }
  } guard{&e, s};
  SLM_TRY(dalloc(&e.c, (size_t)ld));
  LaneSetup ls = default_lanes(ds, 1);
  ls.rw = wdev;
  ls.rw_stride = 0;
  ls.n_eff[0] = n_eff;
  HIP_TRY(hipMemsetAsync(ds->z, 0, sizeof(double) * ld, s));
  if (ds->gk[0]) SLM_TRY(enqueue_gradient(ds, ls, ds->y, nullptr, nullptr, nullptr));
  else SLM_TRY(enqueue_gradient_split(ds, ls, ds->y, nullptr, nullptr, nullptr, nullptr, nullptr));
  hipLaunchKernelGGL(cov_linear_kernel, dim3((unsigned)((ld + 255) / 256)), dim3(256), 0, s, ds->g, ld, e.c, ds->cov_fp);
  SLM_TRY(check_launch());
  HIP_TRY(hipMemcpyAsync(&e.yy, ds->cov_fp, sizeof(double), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  guard.e = nullptr;
  // (at most sixteen Grams per dataset -- 3.2 GB at p = 5 000 -- the oldest goes first: searches with fresh CV splits on a
  //  cached dataset would otherwise pile them up)
  if (ds->cov.size() >= 16) ds->cov.erase(ds->cov.begin());
  e.hold = std::make_shared<slm_dataset::CovBlocks>();
  e.hold->G = e.G;
  e.hold->c = e.c;
  ds->cov.push_back(e);
  return SLM_OK;
}

static int cov_checks(slm_dataset* ds) {
  if (!ds) return fail(SLM_ERR_BAD_ARG, "dataset is NULL");
  if (row_sharded(ds)) return fail(SLM_ERR_UNSUPPORTED, "covariance passes are not built for row-sharded datasets (replicas: slm_dataset_set_replicated)");
  if (!split_usable(ds)) return fail(SLM_ERR_UNSUPPORTED, "covariance passes ride on the split pass (rows of up to 10 240 columns)");
  return SLM_OK;
}

static int cov_ensure_all(slm_dataset* ds) {
  if (ds->cov_all) return SLM_OK;
  double* all = nullptr;
  SLM_TRY(dalloc(&all, (size_t)ds->ld * ds->ld));
  const int rc = cov_gram(ds, ds->X, ds->n, all);
  if (rc != SLM_OK) {  // (published only once the product is under way: a failed one must not stand in as the minuend)
    (void)hipStreamSynchronize(ds->eng->stream);
    dfree(all);
    return rc;
  }
  ds->cov_all_hold = std::make_shared<slm_dataset::CovBlocks>();
  ds->cov_all_hold->G = all;
  ds->cov_all = all;
  return SLM_OK;
}

extern "C" int slm_dataset_covariance(slm_dataset* ds, const double* row_weight, int64_t n_eff_in) {
  SLM_TRY(cov_checks(ds));
  slm_engine* eng = ds->eng;
  HIP_TRY(hipSetDevice(eng->device));
  hipStream_t s = eng->stream;
  const int64_t n = ds->n, ld = ds->ld;
  const double n_eff = n_eff_in > 0 ? (double)n_eff_in : (double)ds->n_global;
  struct Temps {
    double *a = nullptr, *b = nullptr;
    hipStream_t s;
    ~Temps() {
      (void)hipStreamSynchronize(s);
      dfree(a);
      dfree(b);
    }
  } tmp;
  tmp.s = s;
  if (row_weight) {
    SLM_TRY(dalloc(&tmp.a, (size_t)n));
    HIP_TRY(hipMemcpyAsync(tmp.a, row_weight, sizeof(double) * n, hipMemcpyHostToDevice, s));
  }
  const double* wdev = row_weight ? tmp.a : ds->rw;
  double fp[2];
  SLM_TRY(cov_fingerprints(ds, &wdev, 1, fp));
  if (cov_find(ds, fp[0], fp[1], n_eff) >= 0) return SLM_OK;
  // what kind of weights: none, a 0/1 mask (the Gram of all rows minus the Gram of the rows left out: a fifth of the
  // work for a fold of five), or anything else (rows scaled by sqrt(w) into a copy)
  std::vector<double> hw;
  const double* w_host = row_weight;
  if (!row_weight && ds->rw) {
    hw.resize((size_t)n);
    HIP_TRY(hipMemcpy(hw.data(), ds->rw, sizeof(double) * n, hipMemcpyDeviceToHost));
    w_host = hw.data();
  }
  bool binary = true;
  std::vector<int64_t> zeros;
  if (w_host)
    for (int64_t i = 0; i < n; ++i) {
      if (w_host[i] == 0.0) zeros.push_back(i);
      else if (w_host[i] != 1.0) binary = false;
    }
  double* G = nullptr;
  SLM_TRY(dalloc(&G, (size_t)ld * ld));
  const unsigned cgrid = (unsigned)std::min<int64_t>(4096, (ld * ld + 255) / 256);
  int rc = SLM_OK;
  if (!w_host || (binary && (int64_t)zeros.size() * 2 <= n)) {
    rc = cov_ensure_all(ds);
    if (rc == SLM_OK && !zeros.empty()) rc = cov_gram_of_rows(ds, zeros, G);
    if (rc == SLM_OK)
      hipLaunchKernelGGL(cov_combine_kernel, dim3(cgrid), dim3(256), 0, s, ds->cov_all, zeros.empty() ? nullptr : G, 1.0 / n_eff,
                         ld * ld, G);
  } else {
    rc = dalloc(&tmp.b, (size_t)n * (size_t)ld);
    if (rc == SLM_OK) {
      hipLaunchKernelGGL(cov_rows_kernel, dim3((unsigned)n), dim3(256), 0, s, ds->X, ld, nullptr, wdev, n, tmp.b);
      rc = cov_gram(ds, tmp.b, n, G);
    }
    if (rc == SLM_OK) hipLaunchKernelGGL(cov_combine_kernel, dim3(cgrid), dim3(256), 0, s, G, nullptr, 1.0 / n_eff, ld * ld, G);
  }
  if (rc != SLM_OK) {
    (void)hipStreamSynchronize(s);
    dfree(G);
    return rc;
  }
  return cov_file_entry(ds, wdev, n_eff, fp, G);
}

// The folds of a K-fold split at once.  Their test rows are a partition of the rows, so the Gram of ALL rows is the sum of
// the test rows' Grams: K products over n / K rows each -- one pass' worth of products in all -- instead of that plus a
// product over all n rows.  The same sums give the linear terms: X^T W_f y = (sum_g t_g - t_f) with t_g = X_g^T y_g of fold
// g's test rows (xtr_mfma_kernel on the gathered block and its targets), and y^T W_f y alike -- no pass over X at all.
// Part f = [Gram of the test rows | t_f | y_f . y_f] is one stretch of `stride` doubles of one block, so that
//   * a REPLICA on an engine with a communicator (grid mode: slm_dataset_set_replicated) builds the parts from ITS n_ranks-th
//     of the rows only and the ranks sum them -- one all-reduce per part, on the engine's second stream, entered as soon as
//     the part is built while the next one is still being multiplied;
//   * everything behind the parts -- (all - part_f) / n_f, the entries -- is the same with and without ranks.
// Anything that is not such a partition (masks that overlap or leave rows out, weights that are not 0/1, Grams that exist
// already) is built mask by mask (slm_dataset_covariance), by every rank for itself.
struct slm_dataset::CovPending {
  double *big = nullptr, *all = nullptr;  // [count][stride] parts; [stride] their sum -- packed triangle | t | y.y, see cov_folds_begin
  size_t stride = 0, tri = 0;             // doubles per part; of which the packed triangle (rounded up to 16)
  int count = 0;
  std::vector<double> n_eff;              // per fold
  double *block = nullptr, *R16 = nullptr, *wdev = nullptr;  // staging: the folds' test rows (padded), targets, the masks
  int64_t* rows = nullptr;                // the folds' test-row indices, one list after the other
  hipEvent_t built = nullptr;             // the parts are complete on the engine's stream
  hipEvent_t uploaded = nullptr;          // the caller's masks have left the host
};

void cov_pending_drop(slm_dataset* ds) {
  slm_dataset::CovPending* q = ds->cov_pend;
  if (!q) return;
  (void)hipStreamSynchronize(ds->eng->stream);
  if (ds->eng->comm_stream) (void)hipStreamSynchronize(ds->eng->comm_stream);
  dfree(q->big); dfree(q->all); dfree(q->block); dfree(q->R16); dfree(q->rows); dfree(q->wdev);
  if (q->built) (void)hipEventDestroy(q->built);
  if (q->uploaded) (void)hipEventDestroy(q->uploaded);
  delete q;
  ds->cov_pend = nullptr;
}

// 1: the zeros of the masks partition the rows (zeros[f] = test rows of fold f), 0: they do not
static int cov_partition(const slm_dataset* ds, const double* const* row_weights, const int64_t* n_effs, int count,
                         std::vector<std::vector<int64_t>>& zeros) {
  const int64_t n = ds->n;
  bool partition = count >= 2 && ds->cov_all == nullptr && ds->cov.empty();
  zeros.assign((size_t)count, {});
  if (partition) {
    std::vector<unsigned char> seen((size_t)n, 0);
    for (int f = 0; f < count && partition; ++f) {
      const double* w = row_weights[f];
      partition = w != nullptr && n_effs[f] > 0;
      for (int64_t i = 0; i < n && partition; ++i) {
        if (w[i] == 0.0) {
          partition = !seen[(size_t)i];
          seen[(size_t)i] = 1;
          zeros[(size_t)f].push_back(i);
        } else if (w[i] != 1.0) {
          partition = false;
        }
      }
    }
    for (int64_t i = 0; i < n && partition; ++i) partition = seen[(size_t)i] != 0;
  }
  return partition ? 1 : 0;
}

// Queues the parts of this rank's rows on the engine's stream and returns without waiting for anything; *started = 0 when
// the masks are no partition (nothing queued).  Part f, `stride` doubles: the PACKED lower triangle of X_f^T X_f (test rows of
// fold f among this rank's; cov_syrk_packed_kernel, all folds in one launch), then t_f = X_f^T y_f [ld] and y_f . y_f [16].
static int cov_folds_begin(slm_dataset* ds, const double* const* row_weights, const int64_t* n_effs, int32_t count, int* started) {
  *started = 0;
  SLM_TRY(cov_checks(ds));
  if (!row_weights || !n_effs || count < 1 || count > kMaxLanes) return fail(SLM_ERR_BAD_ARG, "between 1 and %d row sets", kMaxLanes);
  if (ds->cov_pend) return fail(SLM_ERR_BAD_ARG, "a fold build is already under way on this dataset (finish it first)");
  slm_engine* eng = ds->eng;
  HIP_TRY(hipSetDevice(eng->device));
  hipStream_t s = eng->stream;
  const int64_t n = ds->n, ld = ds->ld;
  std::vector<std::vector<int64_t>> zeros;
  if (!cov_partition(ds, row_weights, n_effs, count, zeros)) return SLM_OK;
  // this rank's rows: all of them, or -- a replica among ranks -- a contiguous n_ranks-th
  int64_t lo = 0, hi = n;
  if (ds->replicated && eng->sharded()) {
    const int64_t base = n / eng->n_ranks, rem = n % eng->n_ranks;
    lo = eng->rank * base + std::min<int64_t>(eng->rank, rem);
    hi = lo + base + (eng->rank < rem ? 1 : 0);
  }
  std::vector<int64_t> rows_host, first((size_t)count + 1, 0), at_row((size_t)count + 1, 0);
  int64_t most = 1;
  for (int f = 0; f < count; ++f) {
    for (int64_t i : zeros[(size_t)f])
      if (i >= lo && i < hi) rows_host.push_back(i);
    first[(size_t)f + 1] = (int64_t)rows_host.size();
    const int64_t m = first[(size_t)f + 1] - first[(size_t)f];
    most = std::max(most, m);
    at_row[(size_t)f + 1] = at_row[(size_t)f] + cov_syrk_padded_rows(m);  // (the fold's rows in the block, zeros behind them)
  }
  slm_dataset::CovPending* q = new slm_dataset::CovPending();
  ds->cov_pend = q;
  struct Guard {  // (whichever way this function is left before the parts are queued, the blocks go back)
    slm_dataset* ds;
    ~Guard() { if (ds) cov_pending_drop(ds); }
  } guard{ds};
  q->count = count;
  q->tri = ((size_t)ld * (size_t)(ld + 1) / 2 + 15) / 16 * 16;
  q->stride = q->tri + (size_t)ld + 16;
  q->n_eff.resize((size_t)count);
  // the masks go to the device for their fingerprints (read back by cov_folds_finish: nothing here waits)
  SLM_TRY(dalloc(&q->wdev, (size_t)count * n));
  {
    CovFpArgs fa;
    memset(&fa, 0, sizeof(fa));
    for (int f = 0; f < count; ++f) {
      q->n_eff[(size_t)f] = (double)n_effs[f];
      fa.w[f] = q->wdev + (size_t)f * n;
      HIP_TRY(hipMemcpyAsync(q->wdev + (size_t)f * n, row_weights[f], sizeof(double) * n, hipMemcpyHostToDevice, s));
    }
    if (!ds->cov_fp) SLM_TRY(dalloc(&ds->cov_fp, 2 * (size_t)kMaxLanes + 2));
    hipLaunchKernelGGL(cov_fingerprint_kernel, dim3((unsigned)count), dim3(1024), 0, s, fa, n, ds->cov_fp);
  }
  SLM_TRY(dalloc(&q->big, (size_t)count * q->stride));
  SLM_TRY(dalloc(&q->all, q->stride));
  SLM_TRY(dalloc(&q->block, (size_t)at_row[(size_t)count] * (size_t)ld));
  SLM_TRY(dalloc(&q->R16, (size_t)most * SPLIT_RSTRIDE));
  SLM_TRY(dalloc(&q->rows, std::max<size_t>(1, rows_host.size())));
  if (!rows_host.empty())
    HIP_TRY(hipMemcpyAsync(q->rows, rows_host.data(), sizeof(int64_t) * rows_host.size(), hipMemcpyHostToDevice, s));
  // (host memory -- the caller's masks, the row list -- is borrowed for the duration of the call only: the uploads sit at the
  //  head of the stream and are long through when everything behind them has been queued; that, not the products, is waited for)
  HIP_TRY(hipEventCreateWithFlags(&q->uploaded, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(q->uploaded, s));
  SyrkBatch sb;
  memset(&sb, 0, sizeof(sb));
  for (int f = 0; f < count; ++f) {
    const int64_t m = first[(size_t)f + 1] - first[(size_t)f];
    double* blk = q->block + (size_t)at_row[(size_t)f] * (size_t)ld;
    hipLaunchKernelGGL(cov_rows_pad_kernel, dim3((unsigned)(at_row[(size_t)f + 1] - at_row[(size_t)f])), dim3(256), 0, s, ds->X, ld,
                       q->rows + first[(size_t)f], m, blk);
    sb.A[f] = blk;
    sb.rows[f] = m;
    sb.P[f] = q->big + (size_t)f * q->stride;
  }
  const int n_tiles = cov_syrk_tiles(ld);
  hipLaunchKernelGGL(cov_syrk_packed_kernel, dim3((unsigned)((n_tiles + 3) / 4), (unsigned)count), dim3(256), 0, s, sb, ld, n_tiles);
  for (int f = 0; f < count; ++f) {
    double* lin = q->big + (size_t)f * q->stride + q->tri;  // t_f [ld], then y_f . y_f [16]
    const int64_t m = first[(size_t)f + 1] - first[(size_t)f];
    if (m < 1) {
      HIP_TRY(hipMemsetAsync(lin, 0, sizeof(double) * ((size_t)ld + 16), s));
      continue;
    }
    const int64_t* rows = q->rows + first[(size_t)f];
    // t_f = X_f^T y_f: the second half of the split pass on (block, [y_f, 0 ...])
    hipLaunchKernelGGL(cov_targets_kernel, dim3((unsigned)((m * SPLIT_RSTRIDE + 255) / 256)), dim3(256), 0, s, ds->y, rows, m, q->R16);
    SplitArgs a;
    memset(&a, 0, sizeof(a));
    a.X = sb.A[f]; a.R = q->R16; a.partial = ds->partial; a.n = m; a.ld = ld; a.p2 = (int)(ld / 2); a.n_lanes = 1;
    const int xblk = launch_xtr(eng->cus, a, s);
    hipLaunchKernelGGL(cov_xty_kernel, dim3((unsigned)((ld + 255) / 256)), dim3(256), 0, s, ds->partial, xblk, ld, lin);
    hipLaunchKernelGGL(cov_yy_kernel, dim3(1), dim3(1024), 0, s, ds->y, rows, m, lin + ld);
  }
  SLM_TRY(check_launch());
  HIP_TRY(hipEventCreateWithFlags(&q->built, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(q->built, s));
  HIP_TRY(hipEventSynchronize(q->uploaded));
  guard.ds = nullptr;
  *started = 1;
  return SLM_OK;
}

// sums the parts over the ranks (replicas among ranks), forms the folds' Grams and files the entries
static int cov_folds_finish(slm_dataset* ds) {
  slm_dataset::CovPending* q = ds->cov_pend;
  if (!q) return fail(SLM_ERR_BAD_ARG, "no fold build is under way on this dataset");
  slm_engine* eng = ds->eng;
  HIP_TRY(hipSetDevice(eng->device));
  hipStream_t s = eng->stream;
  const int64_t ld = ds->ld;
  struct Guard {
    slm_dataset* ds;
    double* full = nullptr;
    ~Guard() {
      cov_pending_drop(ds);  // (waits for the stream)
      dfree(full);
    }
  } guard{ds};
  if (ds->replicated && eng->sharded()) {
    // on the engine's second stream (RCCL's kernels then never sit between two kernels of a solve on this stream)
    if (!eng->comm_stream) HIP_TRY(hipStreamCreateWithFlags(&eng->comm_stream, hipStreamNonBlocking));
    if (!eng->comm_ev) HIP_TRY(hipEventCreateWithFlags(&eng->comm_ev, hipEventDisableTiming));
    hipStream_t cs = eng->comm_stream;
    HIP_TRY(hipStreamWaitEvent(cs, q->built, 0));
    for (int f = 0; f < q->count; ++f) SLM_TRY(all_reduce_sum(eng, q->big + (size_t)f * q->stride, q->stride, cs));
    HIP_TRY(hipEventRecord(eng->comm_ev, cs));
    HIP_TRY(hipStreamWaitEvent(s, eng->comm_ev, 0));
  }
  const unsigned cgrid = (unsigned)std::min<int64_t>(4096, ((int64_t)q->stride + 255) / 256);
  hipLaunchKernelGGL(cov_sum_kernel, dim3(cgrid), dim3(256), 0, s, q->big, q->count, (int64_t)q->stride, (int64_t)q->stride, q->all);
  // the squares a pass reads: G_f = (all - part_f) / n_f mirrored out of the packed triangles, c_f and y^T W_f y behind it
  const size_t fstride = (size_t)ld * ld + (size_t)ld + 16;
  SLM_TRY(dalloc(&guard.full, (size_t)q->count * fstride));
  double* all_sq = nullptr;
  SLM_TRY(dalloc(&all_sq, (size_t)ld * ld));
  const int64_t nt32 = (ld + 31) / 32;
  const dim3 ugrid((unsigned)(nt32 * (nt32 + 1) / 2));
  for (int f = 0; f < q->count; ++f) {
    const double* part = q->big + (size_t)f * q->stride;
    double* Gf = guard.full + (size_t)f * fstride;
    const double sc = 1.0 / q->n_eff[(size_t)f];
    hipLaunchKernelGGL(cov_unpack_kernel, ugrid, dim3(256), 0, s, q->all, part, sc, ld, Gf);
    hipLaunchKernelGGL(cov_combine_kernel, dim3((unsigned)((ld + 16 + 255) / 256)), dim3(256), 0, s, q->all + q->tri, part + q->tri, sc,
                       (int64_t)ld + 16, Gf + (size_t)ld * ld);
  }
  hipLaunchKernelGGL(cov_unpack_kernel, ugrid, dim3(256), 0, s, q->all, (const double*)nullptr, 1.0, ld, all_sq);
  double yy[SLM_MAX_LANES] = {}, fp[2 * SLM_MAX_LANES] = {};
  int rc = check_launch();
  hipError_t he = hipSuccess;
  for (int f = 0; f < q->count && he == hipSuccess; ++f)
    he = hipMemcpyAsync(&yy[f], guard.full + (size_t)f * fstride + (size_t)ld * ld + ld, sizeof(double), hipMemcpyDeviceToHost, s);
  if (he == hipSuccess) he = hipMemcpyAsync(fp, ds->cov_fp, sizeof(double) * 2 * (size_t)q->count, hipMemcpyDeviceToHost, s);
  if (he == hipSuccess) he = hipStreamSynchronize(s);
  if (rc == SLM_OK && he != hipSuccess) rc = fail(SLM_ERR_HIP, "covariance build: %s", hipGetErrorString(he));
  if (rc != SLM_OK) {
    (void)hipStreamSynchronize(s);
    dfree(all_sq);
    return rc;
  }
  // the entries share the block of the squares; the Gram of all rows stays for later single masks
  auto hold = std::make_shared<slm_dataset::CovBlocks>();
  hold->G = guard.full;
  guard.full = nullptr;
  ds->cov_all_hold = std::make_shared<slm_dataset::CovBlocks>();
  ds->cov_all_hold->G = all_sq;
  ds->cov_all = all_sq;
  for (int f = 0; f < q->count; ++f) {
    if (cov_find(ds, fp[2 * f], fp[2 * f + 1], q->n_eff[(size_t)f]) >= 0) continue;  // (the same mask twice)
    slm_dataset::CovEntry e;
    e.hold = hold;
    e.G = hold->G + (size_t)f * fstride;
    e.c = e.G + (size_t)ld * ld;
    e.yy = yy[f];
    e.n_eff = q->n_eff[(size_t)f];
    e.fp1 = fp[2 * f];
    e.fp2 = fp[2 * f + 1];
    if (ds->cov.size() >= 16) ds->cov.erase(ds->cov.begin());
    ds->cov.push_back(e);
  }
  return SLM_OK;
}

extern "C" int slm_dataset_covariance_folds_begin(slm_dataset* ds, const double* const* row_weights, const int64_t* n_effs, int32_t count,
                                                  int32_t* started_out) {
  if (!ds) return fail(SLM_ERR_BAD_ARG, "dataset is NULL");
  int started = 0;
  SLM_TRY(cov_folds_begin(ds, row_weights, n_effs, count, &started));
  if (started_out) *started_out = started;
  return SLM_OK;
}

extern "C" int slm_dataset_covariance_folds_finish(slm_dataset* ds) {
  if (!ds) return fail(SLM_ERR_BAD_ARG, "dataset is NULL");
  return cov_folds_finish(ds);
}

extern "C" int slm_dataset_covariance_folds(slm_dataset* ds, const double* const* row_weights, const int64_t* n_effs, int32_t count) {
  if (!ds) return fail(SLM_ERR_BAD_ARG, "dataset is NULL");
  int started = 0;
  SLM_TRY(cov_folds_begin(ds, row_weights, n_effs, count, &started));
  if (started) return cov_folds_finish(ds);
  for (int f = 0; f < count; ++f) SLM_TRY(slm_dataset_covariance(ds, row_weights[f], n_effs[f]));
  return SLM_OK;
}

extern "C" int slm_dataset_covariance_clear(slm_dataset* ds) {
  if (!ds) return fail(SLM_ERR_BAD_ARG, "dataset is NULL");
  HIP_TRY(hipSetDevice(ds->eng->device));
  HIP_TRY(hipStreamSynchronize(ds->eng->stream));
  cov_pending_drop(ds);
  ds->cov.clear();
  ds->cov_all_hold.reset();
  ds->cov_all = nullptr;
  return SLM_OK;
}

extern "C" int slm_dataset_set_replicated(slm_dataset* ds, int32_t replicated) {
  if (!ds) return fail(SLM_ERR_BAD_ARG, "dataset is NULL");
  ds->replicated = replicated != 0;
  ds->L_valid = false;
  ds->sketch_valid = false;
  ds->carry_valid = false;
  return SLM_OK;
}

// Diagnostic: entry `index` (oldest first) of the Grams kept with the dataset, to the host
extern "C" int slm_dataset_covariance_download(slm_dataset* ds, int32_t index, double* G_out, double* c_out, double scalars_out[4]) {
  if (!ds) return fail(SLM_ERR_BAD_ARG, "dataset is NULL");
  if (index < 0 || index >= (int32_t)ds->cov.size()) return fail(SLM_ERR_BAD_ARG, "Gram %d of %d", index, (int)ds->cov.size());
  HIP_TRY(hipSetDevice(ds->eng->device));
  HIP_TRY(hipStreamSynchronize(ds->eng->stream));
  const slm_dataset::CovEntry& e = ds->cov[(size_t)index];
  const int64_t p = ds->p, ld = ds->ld;
  if (G_out) HIP_TRY(hipMemcpy2D(G_out, sizeof(double) * p, e.G, sizeof(double) * ld, sizeof(double) * p, (size_t)p, hipMemcpyDeviceToHost));
  if (c_out) HIP_TRY(hipMemcpy(c_out, e.c, sizeof(double) * p, hipMemcpyDeviceToHost));
  if (scalars_out) {
    scalars_out[0] = e.yy; scalars_out[1] = e.n_eff; scalars_out[2] = e.fp1; scalars_out[3] = e.fp2;
  }
  return SLM_OK;
}

extern "C" int slm_dataset_covariance_count(slm_dataset* ds, int32_t* count_out) {
  if (!ds || !count_out) return fail(SLM_ERR_BAD_ARG, "NULL argument");
  *count_out = (int32_t)ds->cov.size();
  return SLM_OK;
}


// Host side of the MI355X fit engine, fourth unit: the model Gram (mg_kernels.hpp) -- its build on the fp16 matrix cores
// and the launches of an inner round.  What it stands in for is the part of the cvxpy solve
// (/root/reference/src/sparselm/model/_base.py:512-519) that the working set cannot serve: path points with more
// non-zeros than its 512 columns.
#include "engine_internal.hpp"

// Where the model Gram can exist: rows short enough for its 4 ld^2 bytes (fp32: 1 GB at the 16 384 columns of MG_MAX_LD), all
// rows on this device (a row-sharded dataset would have to all-reduce 4 ld^2 bytes per row set: not built).
bool mg_possible(const slm_dataset* ds) {
  if (knobs().mg == 0) return false;
  if (ds->mg_failed) return false;
  if (ds->ld > MG_MAX_LD || ds->n < 64) return false;
  if (row_sharded(ds)) return false;
  return true;
}

void mg_invalidate(slm_dataset* ds) {
  for (auto& e : ds->mg) dfree(e.G);
  ds->mg.clear();
}

void mg_free(slm_dataset* ds) {
  mg_invalidate(ds);
  dfree(ds->mg_vec);
  dfree(ds->mg_Z);
}

// G~ = X^T W X / n_eff into G.  Queued on the engine's stream: column maxima (one read of X), the fp16 operand with the
// square roots of the row weights folded in (one read of the column-major copy), the product in chunks of rows, the sum of
// the chunks.  The operand and the chunks' partial tiles are scratch; the stream is drained before they go back to the pool.
static int mg_build(slm_dataset* ds, const double* w, double n_eff, float* G) {
  slm_engine* eng = ds->eng;
  hipStream_t s = eng->stream;
  const int64_t n = ds->n, ld = ds->ld;
  SLM_TRY(ensure_xt(ds));
  if (!ds->XT || !ds->XT_ready) return fail(SLM_ERR_OOM, "no column-major copy of X: the model Gram is not built");
  const int64_t row_tiles = (n + 31) / 32;
  const int64_t n_pad = (n + MG_BK - 1) / MG_BK * MG_BK;
  const int64_t p_pad = (ld + MG_TILE - 1) / MG_TILE * MG_TILE;
  const int side = (int)(p_pad / MG_TILE);
  const int n_tiles = slm_host::triangle_tiles(side);
  const int64_t k_chunk = MG_CHUNK_ROWS;
  const int n_chunks = (int)((n_pad + k_chunk - 1) / k_chunk);
  unsigned long long* cmax = nullptr;
  _Float16* XTh = nullptr;
  float* P = nullptr;
  int rc = SLM_OK;
  auto need = [&](int r) {
    if (rc == SLM_OK) rc = r;
  };
  need(dalloc(&cmax, (size_t)ld));
  need(dalloc(&XTh, (size_t)p_pad * (size_t)n_pad));
  need(dalloc(&P, (size_t)n_chunks * (size_t)n_tiles * MG_TILE * MG_TILE));
  if (rc != SLM_OK) {
    dfree(cmax); dfree(XTh); dfree(P);
    (void)hipGetLastError();
    return rc;
  }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  bool timed = knobs().trace != 0;
  if (timed) {  // (a diagnostic: an event that cannot be made or recorded switches the timing off, nothing else)
    timed = hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess && hipEventRecord(e0, s) == hipSuccess;
    if (!timed) {
      (void)hipGetLastError();
      if (e0) (void)hipEventDestroy(e0);
      if (e1) (void)hipEventDestroy(e1);
      e0 = e1 = nullptr;
    }
  }
  (void)hipMemsetAsync(cmax, 0, sizeof(unsigned long long) * (size_t)ld, s);
  {
    const int64_t rows_per_block = 64;
    hipLaunchKernelGGL(mg_colmax_kernel, dim3((unsigned)((n + rows_per_block - 1) / rows_per_block)), dim3(256), 0, s, (const double*)ds->X, n, ld,
                       rows_per_block, cmax);
  }
  {
    MgConvArgs c;
    c.XT = ds->XT; c.cmax = cmax; c.rw = w; c.n = n; c.ld = ld; c.row_tiles = row_tiles; c.XTh = XTh; c.n_pad = n_pad; c.p_pad = p_pad;
    hipLaunchKernelGGL(mg_convert_kernel, dim3((unsigned)(n_pad / 64), (unsigned)((p_pad + 255) / 256)), dim3(256), 0, s, c);
  }
  {
    MgSyrkArgs a;
    a.M = XTh; a.n_pad = n_pad; a.k_chunk = k_chunk; a.n_tiles = n_tiles; a.n_chunks = n_chunks; a.P = P;
    if (knobs().mg_syrk == 0) hipLaunchKernelGGL  /* (SLM_MG_SYRK=0: the register-staged form, for A/B runs) */(mg_syrk_f16_kernel, dim3((unsigned)((int64_t)n_tiles * n_chunks)), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(mg_syrk_f16_dma_kernel, dim3((unsigned)((int64_t)n_tiles * n_chunks)), dim3(256), 0, s, a);
  }
  {
    MgReduceArgs r;
    r.P = P; r.cmax = cmax; r.n_tiles = n_tiles; r.n_chunks = n_chunks; r.ld = ld; r.inv_n = 1.0 / n_eff; r.G = G;
    hipLaunchKernelGGL(mg_reduce_kernel, dim3((unsigned)n_tiles, 16), dim3(256), 0, s, r);
  }
  rc = check_launch();
  if (timed && hipEventRecord(e1, s) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    timed = false;
  }
  hipError_t e = hipStreamSynchronize(s);
  dfree(cmax); dfree(XTh); dfree(P);
  if (timed) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess) ds->mg_build_ms = ms;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    fprintf(stderr, "[slm] model Gram: %lld x %lld from %lld rows in %d chunks, %d tiles, %.3f ms on the device\n", (long long)ld, (long long)ld,
            (long long)n, n_chunks, n_tiles, ds->mg_build_ms);
  }
  if (e != hipSuccess) return fail(SLM_ERR_HIP, "model Gram: %s", hipGetErrorString(e));
  return rc;
}

int mg_ensure(slm_dataset* ds, const double* w, double n_eff, bool own, double fp1, double fp2, int* entry_out) {
  for (size_t i = 0; i < ds->mg.size(); ++i) {
    const slm_dataset::MgEntry& e = ds->mg[i];
    if (e.n_eff == n_eff && (own ? e.own : (!e.own && e.fp1 == fp1 && e.fp2 == fp2))) {
      *entry_out = (int)i;
      return SLM_OK;
    }
  }
  const int64_t ld = ds->ld;
  // The cap (kMgEntries, and 3 GB of fp32 Grams: host_logic.hpp model_gram_cap) is this function's to keep, whoever calls: a
  // solve makes room for ALL the row sets of its call before the first build (PathCall::mg_sets drops every entry when they
  // do not fit beside the ones kept -- entries of one call must not evict each other), slm_dataset_model_gram builds one more
  // own-rows entry at a time: here the OLDEST entry goes when the dataset is full.
  while ((int)ds->mg.size() >= slm_host::model_gram_cap(ld, 3.0e9, kMgEntries) && !ds->mg.empty()) {
    dfree(ds->mg.front().G);
    ds->mg.erase(ds->mg.begin());
  }
  int rc = SLM_OK;
  if (!ds->mg_vec) rc = dalloc(&ds->mg_vec, 5 * (size_t)kMaxLanes * (size_t)ld);
  if (rc == SLM_OK && !ds->mg_Z) {  // (a plane per half of the lanes; its own block: a covariance pass's Z has one)
    rc = dalloc(&ds->mg_Z, (size_t)ld * SPLIT_RSTRIDE * SPLIT_HALVES);
    if (rc == SLM_OK) (void)hipMemsetAsync(ds->mg_Z, 0, sizeof(double) * (size_t)ld * SPLIT_RSTRIDE * SPLIT_HALVES, ds->eng->stream);
  }
  slm_dataset::MgEntry ne;
  if (rc == SLM_OK) rc = dalloc(&ne.G, (size_t)ld * (size_t)ld);
  if (rc == SLM_OK) rc = mg_build(ds, own ? ds->rw : w, n_eff, ne.G);
  if (rc != SLM_OK) {  // no memory: the solve goes on as it would have without the model
    dfree(ne.G);
    if (rc == SLM_ERR_OOM) ds->mg_failed = true;
    (void)hipGetLastError();
    return rc;
  }
  ne.fp1 = fp1; ne.fp2 = fp2; ne.n_eff = n_eff; ne.own = own;
  ds->mg.push_back(ne);
  *entry_out = (int)ds->mg.size() - 1;
  return SLM_OK;
}

#define SLM_MG_STEP(N) hipLaunchKernelGGL(mg_step_kernel<N>, dim3((unsigned)n_lanes), dim3(TAIL_THREADS), 0, s, ta, m)

// One round for the lanes the working set did not serve: begin, `inner_iters` x (products G~_s D for the row sets of the call,
// their sums over the row blocks, step), finish.  Everything returns at once for lanes that take no part, and the products
// return at once when no lane iterates any more.
int mg_enqueue_round(slm_dataset* ds, const TailArgs& ta, int n_lanes, int inner_iters, const int* done, int n_sets, const int* entry_of_set,
                     const int* set_of) {
  hipStream_t s = ds->eng->stream;
  const int64_t ld = ds->ld;
  SplitArgs a;
  memset(&a, 0, sizeof(a));
  a.done = done;
  a.n = ld; a.ld = ld; a.p2 = (int)(ld / 2); a.n_lanes = n_lanes;
  const slm_host::XtrGrid g = slm_host::xtr_grid(ld, ld, XTR_CB, xtr_max_row_blocks(ds->eng->cus, ld) / 2);  // (as a covariance pass's)
  const int xb = g.xb, yb = g.yb;
  a.xrows = g.rows;
  a.xrows_ws = 0;
  CovBatch cb;
  memset(&cb, 0, sizeof(cb));
  for (int st = 0; st < n_sets; ++st) cb.G[st] = reinterpret_cast<const double*>(ds->mg[(size_t)entry_of_set[st]].G);  // (fp32: cov_gz_body<H, float>)
  for (int l = 0; l < kMaxLanes; ++l) cb.set_of[l] = l < n_lanes ? set_of[l] : 0;
  cb.part_stride = (int64_t)yb * SPLIT_LANES * ld;
  // partial sums: one block of [row blocks][16][ld] per half of the lanes and row set (the gradient's own buffer holds two)
  const int halves = (n_lanes + SPLIT_LANES - 1) / SPLIT_LANES;
  const int blocks = halves * n_sets;
  double* partial = ds->partial;
  if ((size_t)blocks * (size_t)cb.part_stride > ds->partial_elems) {
    if (ds->cov_partial_sets < blocks || !ds->cov_partial) {
      dfree(ds->cov_partial);
      ds->cov_partial_sets = 0;
      // (sized like a covariance pass's: enqueue_gradient_cov shares the buffer)
      const int64_t blocks_most = std::max<int64_t>(1, xtr_max_row_blocks(ds->eng->cus, ld) / 2);
      SLM_TRY(dalloc(&ds->cov_partial, (size_t)blocks * (size_t)(blocks_most * SPLIT_LANES * ld)));
      ds->cov_partial_sets = blocks;
    }
    partial = ds->cov_partial;
  }
  MgArgs m;
  memset(&m, 0, sizeof(m));
  m.mg = &ds->dctl->mg;
  m.ws = ds->ws_ctl;
  m.partial = partial;
  m.part_stride = cb.part_stride;
  m.n_sets = n_sets;
  m.z_plane = (int64_t)ld * SPLIT_RSTRIDE;
  for (int l = 0; l < kMaxLanes; ++l) m.set_of[l] = cb.set_of[l];
  m.nblk = yb;
  m.Z = ds->mg_Z;
  m.x = ds->mg_vec;
  m.v = ds->mg_vec + (size_t)kMaxLanes * ld;
  m.vprev = ds->mg_vec + 2 * (size_t)kMaxLanes * ld;
  m.gvprev = ds->mg_vec + 3 * (size_t)kMaxLanes * ld;
  m.gd = ds->mg_vec + 4 * (size_t)kMaxLanes * ld;
  hipLaunchKernelGGL(mg_begin_kernel, dim3((unsigned)n_lanes), dim3(TAIL_THREADS), 0, s, ta, m);
  const int E = (ta.p + TAIL_THREADS - 1) / TAIL_THREADS;
  for (int it = 0; it < inner_iters; ++it) {
    if (halves == 2) {  // both halves' planes of Z against ONE read of every Gram
      a.R = ds->mg_Z; a.r_plane = m.z_plane; a.partial = partial;
      cb.half_stride = (int64_t)n_sets * cb.part_stride;
      hipLaunchKernelGGL(mg_gz32_kernel, dim3(xb, yb, (unsigned)n_sets), dim3(XTR_WAVES * 64), 0, s, a, cb, (const MgCtl*)m.mg);
    } else {
      a.R = ds->mg_Z; a.partial = partial;
      hipLaunchKernelGGL(mg_gz_kernel, dim3(xb, yb, (unsigned)n_sets), dim3(XTR_WAVES * 64), 0, s, a, cb, (const MgCtl*)m.mg, 0);
    }
    hipLaunchKernelGGL(mg_gsum_kernel, dim3((unsigned)((ld + 255) / 256), (unsigned)n_lanes), dim3(256), 0, s, m, ld, done);
    switch (E) {
      case 1: SLM_MG_STEP(1); break;
      case 2: SLM_MG_STEP(2); break;
      case 3: SLM_MG_STEP(3); break;
      case 4: SLM_MG_STEP(4); break;
      case 5: SLM_MG_STEP(5); break;
      case 6: SLM_MG_STEP(6); break;
      case 7: SLM_MG_STEP(7); break;
      case 8: SLM_MG_STEP(8); break;
      case 9: SLM_MG_STEP(9); break;
      case 10: SLM_MG_STEP(10); break;
      default: SLM_MG_STEP(0);
    }
  }
  hipLaunchKernelGGL(mg_finish_kernel, dim3((unsigned)n_lanes), dim3(TAIL_THREADS), 0, s, ta, m);
  return SLM_OK;
}
#undef SLM_MG_STEP

extern "C" int slm_dataset_model_gram(slm_dataset* ds, double* G_out) {
  if (!ds) return fail(SLM_ERR_BAD_ARG, "dataset is NULL");
  HIP_TRY(hipSetDevice(ds->eng->device));
  if (row_sharded(ds) || ds->ld > MG_MAX_LD || ds->n < 64)
    return fail(SLM_ERR_UNSUPPORTED, "the model Gram is built for unsharded datasets of 64 rows or more and p <= %d", MG_MAX_LD);
  ds->mg_failed = false;
  int entry = -1;
  SLM_TRY(mg_ensure(ds, nullptr, (double)ds->n_global, true, 0.0, 0.0, &entry));
  if (G_out) {
    HIP_TRY(hipStreamSynchronize(ds->eng->stream));
    const size_t count = (size_t)ds->ld * (size_t)ds->ld;
    std::vector<float> host(count);  // (stored in fp32; the caller's array is fp64)
    HIP_TRY(hipMemcpy(host.data(), ds->mg[(size_t)entry].G, sizeof(float) * count, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < count; ++i) G_out[i] = (double)host[i];
  }
  return SLM_OK;
}

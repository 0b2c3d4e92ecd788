// Host side of the MI355X fit engine, fourth unit: the model Gram (mg_kernels.hpp) -- its build on the fp16 matrix cores
// and the launches of an inner round.  What it stands in for is the part of the cvxpy solve
// (/root/reference/src/sparselm/model/_base.py:512-519) that the working set cannot serve: path points with more
// non-zeros than its 512 columns.
#include "engine_internal.hpp"

// Where the model Gram can exist: a matrix worth a pass (the same bound as the working set's), rows short enough for the
// 8 ld^2 bytes, the dataset's own rows (a row-sharded dataset would have to all-reduce 8 ld^2 bytes: not built).
bool mg_possible(const slm_dataset* ds) {
  if (const char* e = getenv("SLM_MG"))
    if (e[0] == '0') return false;
  if (ds->mg_failed) return false;
  if (ds->ld > MG_MAX_LD || ds->n < 64) return false;
  if (row_sharded(ds)) return false;
  return true;
}

void mg_invalidate(slm_dataset* ds) { ds->mg_ready = false; }

void mg_free(slm_dataset* ds) {
  dfree(ds->mg_G);
  dfree(ds->mg_vec);
  ds->mg_ready = false;
}

// G~ for the dataset's rows and row weights, scaled by 1 / n_global like every gradient of the dataset's own lanes.
// Queued on the engine's stream: column maxima (one read of X), the fp16 operand (one read of the column-major copy),
// the product in chunks of rows, the sum of the chunks.  The operand and the chunks' partial tiles are scratch, given back
// to the pool behind the last kernel that reads them (the pool hands blocks out again in stream order of this engine only
// after the free -- which the stream wait below precedes).
int mg_build(slm_dataset* ds) {
  if (ds->mg_ready) return SLM_OK;
  slm_engine* eng = ds->eng;
  hipStream_t s = eng->stream;
  const int64_t n = ds->n, ld = ds->ld;
  SLM_TRY(ensure_xt(ds));
  if (!ds->XT || !ds->XT_ready) {
    ds->mg_failed = true;
    return fail(SLM_ERR_OOM, "no column-major copy of X: the model Gram is not built");
  }
  const int64_t row_tiles = (n + 31) / 32;
  const int64_t n_pad = (n + MG_BK - 1) / MG_BK * MG_BK;
  const int64_t p_pad = (ld + MG_TILE - 1) / MG_TILE * MG_TILE;
  const int side = (int)(p_pad / MG_TILE);
  const int n_tiles = side * (side + 1) / 2;
  const int64_t k_chunk = MG_CHUNK_ROWS;
  const int n_chunks = (int)((n_pad + k_chunk - 1) / k_chunk);
  unsigned long long* cmax = nullptr;
  _Float16* XTh = nullptr;
  float* P = nullptr;
  int rc = SLM_OK;
  auto need = [&](int r) {
    if (rc == SLM_OK) rc = r;
  };
  if (!ds->mg_G) need(dalloc(&ds->mg_G, (size_t)ld * (size_t)ld));
  if (!ds->mg_vec) need(dalloc(&ds->mg_vec, 4 * (size_t)kMaxLanes * (size_t)ld));
  if (!ds->cov_Z) need(dalloc(&ds->cov_Z, (size_t)ld * SPLIT_RSTRIDE));
  need(dalloc(&cmax, (size_t)ld));
  need(dalloc(&XTh, (size_t)p_pad * (size_t)n_pad));
  need(dalloc(&P, (size_t)n_chunks * (size_t)n_tiles * MG_TILE * MG_TILE));
  if (rc != SLM_OK) {  // no memory: the solve goes on as it would have without the model
    dfree(cmax); dfree(XTh); dfree(P);
    mg_free(ds);
    ds->mg_failed = true;
    (void)hipGetLastError();
    return rc;
  }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  const bool timed = getenv("SLM_TRACE") != nullptr;
  if (timed) {
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, s);
  }
  (void)hipMemsetAsync(cmax, 0, sizeof(unsigned long long) * (size_t)ld, s);
  (void)hipMemsetAsync(ds->cov_Z, 0, sizeof(double) * (size_t)ld * SPLIT_RSTRIDE, s);
  {
    const int64_t rows_per_block = 64;
    hipLaunchKernelGGL(mg_colmax_kernel, dim3((unsigned)((n + rows_per_block - 1) / rows_per_block)), dim3(256), 0, s, (const double*)ds->X, n, ld,
                       rows_per_block, cmax);
  }
  {
    MgConvArgs c;
    c.XT = ds->XT; c.cmax = cmax; c.rw = ds->rw; c.n = n; c.ld = ld; c.row_tiles = row_tiles; c.XTh = XTh; c.n_pad = n_pad; c.p_pad = p_pad;
    hipLaunchKernelGGL(mg_convert_kernel, dim3((unsigned)(n_pad / 64), (unsigned)((p_pad + 255) / 256)), dim3(256), 0, s, c);
  }
  {
    MgSyrkArgs a;
    a.M = XTh; a.n_pad = n_pad; a.k_chunk = k_chunk; a.n_tiles = n_tiles; a.n_chunks = n_chunks; a.P = P;
    hipLaunchKernelGGL(mg_syrk_f16_kernel, dim3((unsigned)((int64_t)n_tiles * n_chunks)), dim3(256), 0, s, a);
  }
  {
    MgReduceArgs r;
    r.P = P; r.cmax = cmax; r.n_tiles = n_tiles; r.n_chunks = n_chunks; r.ld = ld; r.inv_n = 1.0 / (double)ds->n_global; r.G = ds->mg_G;
    hipLaunchKernelGGL(mg_reduce_kernel, dim3((unsigned)n_tiles, 16), dim3(256), 0, s, r);
  }
  rc = check_launch();
  if (timed) (void)hipEventRecord(e1, s);
  // (the scratch goes back to the pool only when the kernels that read it are through)
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
  if (rc != SLM_OK) return rc;
  ds->mg_ready = true;
  return SLM_OK;
}

#define SLM_MG_STEP(N) hipLaunchKernelGGL(mg_step_kernel<N>, dim3((unsigned)n_lanes), dim3(TAIL_THREADS), 0, s, ta, m)

// One round for the lanes the working set did not serve: begin, `inner_iters` x (product G~ D, step), finish.  Everything
// returns at once for lanes that take no part, and the products return at once when no lane iterates any more.
int mg_enqueue_round(slm_dataset* ds, const TailArgs& ta, int n_lanes, int inner_iters, const int* done) {
  hipStream_t s = ds->eng->stream;
  const int64_t ld = ds->ld;
  SplitArgs a;
  memset(&a, 0, sizeof(a));
  a.R = ds->cov_Z; a.partial = ds->partial; a.done = done;
  a.n = ld; a.ld = ld; a.p2 = (int)(ld / 2); a.n_lanes = n_lanes;
  const int xb = (int)((ld + XTR_CB - 1) / XTR_CB);
  const int64_t want = std::max<int64_t>(1, xtr_max_row_blocks(ds->eng->cus, ld) / 2);
  int64_t rows = (ld + want - 1) / want;
  rows = (rows + 7) / 8 * 8;
  const int yb = (int)((ld + rows - 1) / rows);
  a.xrows = (int)rows;
  a.xrows_ws = 0;
  if ((size_t)yb * SPLIT_LANES * (size_t)ld > ds->partial_elems) return fail(SLM_ERR_UNSUPPORTED, "model Gram: partial buffer too small");
  CovBatch cb;
  memset(&cb, 0, sizeof(cb));
  cb.G[0] = ds->mg_G;
  cb.part_stride = 0;
  MgArgs m;
  memset(&m, 0, sizeof(m));
  m.mg = &ds->dctl->mg;
  m.ws = ds->ws_ctl;
  m.partial = ds->partial;
  m.nblk = yb;
  m.Z = ds->cov_Z;
  m.x = ds->mg_vec;
  m.v = ds->mg_vec + (size_t)kMaxLanes * ld;
  m.vprev = ds->mg_vec + 2 * (size_t)kMaxLanes * ld;
  m.gvprev = ds->mg_vec + 3 * (size_t)kMaxLanes * ld;
  hipLaunchKernelGGL(mg_begin_kernel, dim3((unsigned)n_lanes), dim3(TAIL_THREADS), 0, s, ta, m);
  const int E = (ta.p + TAIL_THREADS - 1) / TAIL_THREADS;
  for (int it = 0; it < inner_iters; ++it) {
    hipLaunchKernelGGL(mg_gz_kernel, dim3(xb, yb, 1), dim3(XTR_WAVES * 64), 0, s, a, cb, (const MgCtl*)m.mg);
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
  SLM_TRY(mg_build(ds));
  if (G_out) {
    HIP_TRY(hipStreamSynchronize(ds->eng->stream));
    HIP_TRY(hipMemcpy(G_out, ds->mg_G, sizeof(double) * (size_t)ds->ld * (size_t)ds->ld, hipMemcpyDeviceToHost));
  }
  return SLM_OK;
}

// Host side of the MI355X fit engine, first of three units -- errors, handles, device memory, communicators, datasets: the C ABI of
// include/slm_engine.h.  Replaces the cvxpy `problem.solve` call of
// /root/reference/src/sparselm/model/_base.py:512-519 (and the inner solve of
// model/_adaptive_lasso.py:213-215) with a device-resident FISTA state machine.
//
// gfx950 only; there is no CPU fallback anywhere in this file.
#include "engine_internal.hpp"

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;

int fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

// ------------------------------------------------------------------------------------------------
// RCCL, loaded lazily (single-GPU use never touches it)
// ------------------------------------------------------------------------------------------------
RcclApi g_rccl;

int load_rccl() {
  if (g_rccl.lib) return SLM_OK;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* lib = nullptr;
  for (const char* nm : names) {
    lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
    if (lib) break;
  }
  if (!lib) return fail(SLM_ERR_COMM, "cannot load librccl: %s", dlerror());
  g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(lib, "ncclGetUniqueId");
  g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(lib, "ncclCommInitRank");
  g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(lib, "ncclAllReduce");
  g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(lib, "ncclCommDestroy");
  g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(lib, "ncclGetErrorString");
  g_rccl.CommCount = (decltype(g_rccl.CommCount))dlsym(lib, "ncclCommCount");
  g_rccl.CommUserRank = (decltype(g_rccl.CommUserRank))dlsym(lib, "ncclCommUserRank");
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.CommDestroy)
    return fail(SLM_ERR_COMM, "librccl is missing required symbols");
  g_rccl.lib = lib;
  return SLM_OK;
}

// ------------------------------------------------------------------------------------------------
// all-reduce: RCCL, or the in-process communicator
// ------------------------------------------------------------------------------------------------
static __global__ void local_sum_kernel(double* out, const double* const* stage, int n_ranks, size_t count) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
    double t = stage[0][i];
    for (int r = 1; r < n_ranks; ++r) t += stage[r][i];  // fixed order: every rank gets the same bits
    out[i] = t;
  }
}

// host threads of all ranks meet; returns false on timeout
static bool local_meet(LocalComm* lc, int phase, long round) {
  std::unique_lock<std::mutex> lk(lc->m);
  lc->arrived[phase] += 1;
  const long want = (round + 1) * lc->n_ranks;
  lc->cv.notify_all();
  return lc->cv.wait_for(lk, std::chrono::duration<double>(lc->timeout_s), [&] { return lc->arrived[phase] >= want; });
}

// one round of the in-process exchange: count <= lc->cap doubles
static int local_all_reduce_round(slm_engine* eng, double* buf, size_t count, hipStream_t s) {
  LocalComm* lc = eng->local;
  const int r = eng->rank;
  const long round = lc->round_of[r]++;
  const int par = (int)(round & 1);
  hipError_t e = hipSuccess;
  // my staging area is free once every rank has consumed the round before the previous one of this parity
  if (round >= 2)
    for (int q = 0; q < lc->n_ranks && e == hipSuccess; ++q) e = hipStreamWaitEvent(s, lc->consumed[q][par], 0);
  if (e == hipSuccess) e = hipMemcpyAsync(lc->stage[r] + (size_t)par * lc->cap, buf, sizeof(double) * count, hipMemcpyDeviceToDevice, s);
  if (e == hipSuccess) e = hipEventRecord(lc->ready[r][par], s);
  if (e != hipSuccess) return fail(SLM_ERR_HIP, "in-process all-reduce (stage): %s", hipGetErrorString(e));
  if (!local_meet(lc, 0, round))
    return fail(SLM_ERR_COMM, "in-process all-reduce %ld: a rank did not arrive within %.0f s (mismatched collective counts?)",
                round, lc->timeout_s);
  for (int q = 0; q < lc->n_ranks && e == hipSuccess; ++q)
    if (q != r) e = hipStreamWaitEvent(s, lc->ready[q][par], 0);
  if (e == hipSuccess) {
    // (pointer table of this parity lives behind the staging areas of rank 0)
    const double* const* tab = reinterpret_cast<const double* const*>(lc->stage[0] + 2 * lc->cap) + (size_t)par * LocalComm::kMaxRanks;
    const int blocks = (int)std::min<size_t>(256, (count + 255) / 256);
    hipLaunchKernelGGL(local_sum_kernel, dim3(blocks), dim3(256), 0, s, buf, tab, lc->n_ranks, count);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipEventRecord(lc->consumed[r][par], s);
  if (e != hipSuccess) return fail(SLM_ERR_HIP, "in-process all-reduce (sum): %s", hipGetErrorString(e));
  // (the `consumed` events of this round must exist before any rank waits on them two rounds from now)
  if (!local_meet(lc, 1, round))
    return fail(SLM_ERR_COMM, "in-process all-reduce %ld: a rank did not finish within %.0f s", round, lc->timeout_s);
  return SLM_OK;
}

// sum `count` doubles at `buf` (device) over the ranks, in place, on stream `s` (nullptr: the engine's own).  The
// in-process communicator moves buffers beyond its staging area in rounds (the folds' Grams: 25 M doubles each).
int all_reduce_sum(slm_engine* eng, double* buf, size_t count, hipStream_t s) {
  if (!s) s = eng->stream;
  eng->collectives += 1;
  if (eng->comm) {
    const int e = g_rccl.AllReduce(buf, buf, count, kNcclFloat64, kNcclSum, eng->comm, s);
    if (e != 0) return fail(SLM_ERR_COMM, "ncclAllReduce failed: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "rccl error");
    return SLM_OK;
  }
  LocalComm* lc = eng->local;
  if (!lc) return SLM_OK;
  for (size_t at = 0; at < count; at += lc->cap) SLM_TRY(local_all_reduce_round(eng, buf + at, std::min(lc->cap, count - at), s));
  return SLM_OK;
}

// Large device blocks (a dataset's X, its column-major copy, the gathered columns: gigabytes each) are recycled: a freed
// block waits in a per-device list of its exact size, and the next dataset of that shape takes it instead of asking the
// driver.  hipMalloc right behind the hipFree of such blocks stalled for SECONDS now and then (the soak over 96 datasets
// of the headline shape, profiles/r02c_headline_soak.log: a 6-pass path in 2.4 s; 3.6 s in r02a) -- a fresh fit paying
// three hundred times its solve.  At most pool_idle_cap() bytes wait per process; when the driver has no memory left the
// waiting blocks are handed back and the allocation is tried again.  Nothing relies on a block's contents.
static const size_t kPoolMinBytes = (size_t)64 << 20;
// idle blocks kept per process: 24 GB unless SLM_DEVICE_POOL_GB says otherwise (0 turns the pool off) -- a block idle here is
// memory no other allocator on the GPU can have (another rank sharing the device, torch, RCCL's buffers): enough for the two
// or three blocks of the largest dataset shape seen lately, not a standing reservation
static size_t pool_idle_cap() {
  const double gb = knobs().device_pool_gb;
  return gb >= 0.0 ? (size_t)(gb * (double)((size_t)1 << 30)) : (size_t)24 << 30;
}
// (the ledger itself -- which block waits, which is evicted -- is host_logic.hpp's PoolLedger: no device call in it, run
//  under the sanitizers by tests/host_logic_test.cpp)
struct DevicePool {
  std::mutex m;
  slm_host::PoolLedger book;
};
static DevicePool g_pool;

hipError_t pool_malloc(void** out, size_t bytes) {
  if (bytes < kPoolMinBytes) return hipMalloc(out, bytes);
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lk(g_pool.m);
  if (void* p = g_pool.book.take(dev, bytes)) {
    *out = p;
    return hipSuccess;
  }
  hipError_t e = hipMalloc(out, bytes);
  if (e == hipErrorOutOfMemory && !g_pool.book.idle.empty()) {
    (void)hipGetLastError();
    for (void* p : g_pool.book.flush()) (void)hipFree(p);
    e = hipMalloc(out, bytes);
  }
  if (e == hipSuccess) g_pool.book.adopt(*out, dev, bytes);
  return e;
}

void pool_free(void* p) {
  std::vector<void*> evict;
  bool kept;
  {
    std::lock_guard<std::mutex> lk(g_pool.m);
    // (over the cap: the blocks that have waited longest go back to the driver first)
    kept = g_pool.book.give_back(p, pool_idle_cap(), knobs().device_pool, &evict);
  }
  for (void* q : evict) (void)hipFree(q);
  if (!kept) (void)hipFree(p);
}

// ------------------------------------------------------------------------------------------------
// library
// ------------------------------------------------------------------------------------------------
extern "C" int slm_abi_version(void) { return SLM_ABI_VERSION; }

// The environment is read HERE and nowhere else: once, when the first caller asks (host_logic.hpp: Knobs), and again when
// slm_reload_knobs says so (tests and A/B tools that change a variable after the library is loaded).
static std::mutex g_knobs_m;
static slm_host::Knobs g_knobs;
static bool g_knobs_loaded = false;
const slm_host::Knobs& knobs() {
  std::lock_guard<std::mutex> lk(g_knobs_m);
  if (!g_knobs_loaded) {
    g_knobs = slm_host::Knobs::from([](const char* name) -> const char* { return getenv(name); });
    g_knobs_loaded = true;
  }
  return g_knobs;
}
extern "C" int slm_reload_knobs(void) {
  std::lock_guard<std::mutex> lk(g_knobs_m);
  g_knobs = slm_host::Knobs::from([](const char* name) -> const char* { return getenv(name); });
  g_knobs_loaded = true;
  return SLM_OK;
}

extern "C" int slm_host_alloc(size_t bytes, void** out) {
  if (!out || bytes == 0) return fail(SLM_ERR_BAD_ARG, "slm_host_alloc: NULL out or zero size");
  *out = nullptr;
  void* ptr = nullptr;
  // (portable: the block serves whichever device the calling process's engines sit on)
  const hipError_t e = hipHostMalloc(&ptr, bytes, hipHostMallocPortable);
  if (e == hipErrorOutOfMemory || e == hipErrorMemoryAllocation) {
    (void)hipGetLastError();
    return fail(SLM_ERR_OOM, "no %zu bytes of page-locked host memory", bytes);
  }
  HIP_TRY(e);
  *out = ptr;
  return SLM_OK;
}

extern "C" int slm_host_free(void* ptr) {
  if (!ptr) return SLM_OK;
  HIP_TRY(hipHostFree(ptr));
  return SLM_OK;
}
extern "C" const char* slm_last_error(void) { return g_last_error.c_str(); }

extern "C" int slm_device_count(int* count_out) {
  if (!count_out) return fail(SLM_ERR_BAD_ARG, "count_out is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count_out = 0;
    return fail(SLM_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
  }
  *count_out = n;
  return SLM_OK;
}

// ------------------------------------------------------------------------------------------------
// engine
// ------------------------------------------------------------------------------------------------
extern "C" int slm_engine_create(int device_id, slm_engine** out) {
  if (!out) return fail(SLM_ERR_BAD_ARG, "out is NULL");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(SLM_ERR_NO_DEVICE, "no HIP device visible (%s); this engine has no CPU fallback",
                e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
  if (device_id < 0 || device_id >= n)
    return fail(SLM_ERR_BAD_ARG, "device_id %d out of range [0, %d)", device_id, n);
  HIP_TRY(hipSetDevice(device_id));
  slm_engine* eng = new slm_engine();
  eng->device = device_id;
  e = hipGetDeviceProperties(&eng->prop, device_id);
  if (e != hipSuccess) {
    delete eng;
    return fail(SLM_ERR_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
  }
  if (strncmp(eng->prop.gcnArchName, "gfx950", 6) != 0 && !knobs().allow_any_arch) {
    std::string arch = eng->prop.gcnArchName;
    delete eng;
    return fail(SLM_ERR_NO_DEVICE, "device %d is %s; this library only carries gfx950 (MI355X) code",
                device_id, arch.c_str());
  }
  eng->cus = eng->prop.multiProcessorCount;
  e = hipStreamCreateWithFlags(&eng->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    delete eng;
    return fail(SLM_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
  }
  *out = eng;
  return SLM_OK;
}

extern "C" int slm_comm_destroy(slm_engine* eng);

extern "C" int slm_engine_destroy(slm_engine* eng) {
  if (!eng) return SLM_OK;
  (void)hipSetDevice(eng->device);
  if (eng->sharded()) (void)slm_comm_destroy(eng);
  if (eng->comm_ev) (void)hipEventDestroy(eng->comm_ev);
  if (eng->comm_stream) (void)hipStreamDestroy(eng->comm_stream);
  if (eng->stream) (void)hipStreamDestroy(eng->stream);
  {  // (the recycled blocks do not outlive the engines that could use them)
    std::lock_guard<std::mutex> lk(g_pool.m);
    for (void* q : g_pool.book.retire_device(eng->device)) (void)hipFree(q);
  }
  delete eng;
  return SLM_OK;
}

extern "C" int slm_engine_synchronize(slm_engine* eng) {
  if (!eng) return fail(SLM_ERR_BAD_ARG, "engine is NULL");
  HIP_TRY(hipSetDevice(eng->device));
  HIP_TRY(hipStreamSynchronize(eng->stream));
  return SLM_OK;
}

extern "C" int slm_engine_device_info(slm_engine* eng, int64_t out[6], char* name_out, int name_len) {
  if (!eng || !out) return fail(SLM_ERR_BAD_ARG, "NULL argument");
  HIP_TRY(hipSetDevice(eng->device));
  size_t free_b = 0, total_b = 0;
  HIP_TRY(hipMemGetInfo(&free_b, &total_b));
  out[0] = eng->prop.multiProcessorCount;
  out[1] = (int64_t)eng->prop.maxSharedMemoryPerMultiProcessor;
  out[2] = (int64_t)total_b;
  out[3] = (int64_t)free_b;
  out[4] = eng->prop.warpSize;
  out[5] = eng->prop.clockRate;
  if (name_out && name_len > 0) {
    snprintf(name_out, (size_t)name_len, "%s (%s)", eng->prop.name, eng->prop.gcnArchName);
  }
  return SLM_OK;
}

// ------------------------------------------------------------------------------------------------
// dataset
// ------------------------------------------------------------------------------------------------
static void dataset_free(slm_dataset* ds) {
  if (!ds) return;
  dfree(ds->X); dfree(ds->y); dfree(ds->rw); dfree(ds->yzero); dfree(ds->rw_lanes); dfree(ds->rvec);
  dfree(ds->order); dfree(ds->gid); dfree(ds->gstart);
  dfree(ds->partial); dfree(ds->loss_partial); dfree(ds->R);
  dfree(ds->g); dfree(ds->z); dfree(ds->beta); dfree(ds->zprev); dfree(ds->gprev);
  dfree(ds->u); dfree(ds->gscale); dfree(ds->a0); dfree(ds->b0); dfree(ds->d0);
  dfree(ds->lambda); dfree(ds->dctl);
  dfree(ds->pts); dfree(ds->betas_out); dfree(ds->gn_out); dfree(ds->infos);
  dfree(ds->ws_idx); dfree(ds->ws_pos); dfree(ds->ws_gs); dfree(ds->ws_gl);
  dfree(ds->ws_score); dfree(ds->ws_XW); dfree(ds->ws_part); dfree(ds->ws_G); dfree(ds->ws_Gx); dfree(ds->XT); dfree(ds->ws_owner);
  dfree(ds->ws_nt); dfree(ds->stop_words); dfree(ds->sse_Z); dfree(ds->sse_part); dfree(ds->split_state);
  dfree(ds->colnorm); dfree(ds->lt_dR); dfree(ds->lt_d2); dfree(ds->lt_part); dfree(ds->lt_cols); dfree(ds->lt_stamp);
  cov_pending_drop(ds);
  ds->cov.clear();
  ds->cov_all_hold.reset();
  ds->cov_all = nullptr;
  dfree(ds->cov_Z); dfree(ds->cov_fp); dfree(ds->cov_partial);
  mg_free(ds);
  if (ds->h_split) (void)hipHostFree(ds->h_split);
  if (ds->hctl) (void)hipHostFree(ds->hctl);
  if (ds->h_small_out) (void)hipHostFree(ds->h_small_out);
  if (ds->h_vec) (void)hipHostFree(ds->h_vec);
  if (ds->h_stage) (void)hipHostFree(ds->h_stage);
  if (ds->h_pts) (void)hipHostFree(ds->h_pts);
  for (auto& e : ds->ev)
    if (e) (void)hipEventDestroy(e);
  for (auto& e : ds->prof) (void)hipEventDestroy(e);
  delete ds;
}

int set_singleton_groups(slm_dataset* ds) {
  const int p = (int)ds->p;
  std::vector<int> ident(p), start(p + 1);
  std::iota(ident.begin(), ident.end(), 0);
  std::iota(start.begin(), start.end(), 0);
  dfree(ds->order); dfree(ds->gid); dfree(ds->gstart); dfree(ds->gscale);
  SLM_TRY(dalloc(&ds->order, p));
  SLM_TRY(dalloc(&ds->gid, p));
  SLM_TRY(dalloc(&ds->gstart, p + 1));
  SLM_TRY(dalloc(&ds->gscale, (size_t)ds->lane_cap * p));
  HIP_TRY(hipMemcpy(ds->order, ident.data(), sizeof(int) * p, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(ds->gid, ident.data(), sizeof(int) * p, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(ds->gstart, start.data(), sizeof(int) * (p + 1), hipMemcpyHostToDevice));
  ds->G = p;
  ds->singleton = 1;
  ds->team = 1;
  ds->max_group = 1;
  return SLM_OK;
}

// Allocates everything that depends only on (n, p): padded X, vectors, work space.
static int dataset_alloc(slm_engine* eng, int64_t n, int64_t p, slm_dataset** out) {
  if (n <= 0 || p <= 0) return fail(SLM_ERR_BAD_ARG, "n and p must be positive (got %lld x %lld)",
                                    (long long)n, (long long)p);
  if (p > (int64_t)TAIL_THREADS * kMaxTailE)
    return fail(SLM_ERR_UNSUPPORTED, "p = %lld exceeds the supported %d columns", (long long)p,
                TAIL_THREADS * kMaxTailE);
  if (n > (int64_t)2000000000) return fail(SLM_ERR_UNSUPPORTED, "n too large");
  slm_dataset* ds = new slm_dataset();
  ds->eng = eng;
  ds->n = n;
  ds->p = p;
  ds->n_global = n;
  ds->ld = (p + 15) / 16 * 16;
  const int64_t ld = ds->ld;
  size_t partial_elems = 0, loss_elems = 0;
  for (int B = 1; B <= kMaxLanes; ++B) {
    const GradKernel* gk = pick_grad_kernel(ld / 2, B);
    ds->gk[B - 1] = gk;
    if (!gk) continue;
    int occ = 2;
    if (gk->D >= 0) {
      hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)gk->fn, gk->W * 64, 0);
      if (e != hipSuccess || occ < 1) occ = 1;
    }
    int per_cu = occ;
    if (knobs().grad_blocks_per_cu > 0) per_cu = knobs().grad_blocks_per_cu;
    int64_t nblk = (int64_t)eng->cus * per_cu;
    const int64_t steps = (n + gk->R - 1) / gk->R;
    nblk = std::max<int64_t>(1, std::min<int64_t>(nblk, steps));
    ds->nblk[B - 1] = (int)nblk;
    partial_elems = std::max(partial_elems, (size_t)nblk * B * (size_t)ld);
    loss_elems = std::max(loss_elems, (size_t)nblk * B);
  }
  if (!ds->gk[0]) {
    delete ds;
    return fail(SLM_ERR_UNSUPPORTED, "no gradient kernel covers p = %lld", (long long)p);
  }
  ds->sk = pick_split_kernel(ld / 2);
  if (ds->sk) {
    ds->split_nblk = (int)std::max<int64_t>(1, std::min<int64_t>(eng->cus, n));  // one ring workgroup per CU
    // (xtr_mfma_kernel's row blocks; the residual kernels' split_nblk blocks only write R and loss_partial)
    partial_elems = std::max(partial_elems, (size_t)xtr_max_row_blocks(eng->cus, ld) * SPLIT_LANES * (size_t)ld);
    loss_elems = std::max(loss_elems, (size_t)ds->split_nblk * SPLIT_LANES * SPLIT_HALVES);
  }

  int rc = SLM_OK;
  auto A = [&](int r) { if (rc == SLM_OK) rc = r; };
  ds->lane_cap = (p <= SM_PMAX && (double)n * (double)ld <= 131072.0) ? kMaxCells : kMaxLanes;
  const size_t ML = (size_t)ds->lane_cap;
  A(dalloc(&ds->X, (size_t)n * ld));
  A(dalloc(&ds->y, n));
  A(dalloc(&ds->yzero, n));
  if (ds->gk[0]->D < 0) A(dalloc(&ds->rvec, n));
  A(dalloc(&ds->partial, partial_elems));
  ds->partial_elems = partial_elems;
  A(dalloc(&ds->loss_partial, loss_elems));
  A(dalloc(&ds->g, ML * (ld + 16)));
  A(dalloc(&ds->z, ML * ld));
  A(dalloc(&ds->beta, ML * ld));
  A(dalloc(&ds->zprev, ML * ld));
  A(dalloc(&ds->gprev, ML * ld));
  A(dalloc(&ds->u, ML * ld));
  A(dalloc(&ds->a0, ML * ld));
  A(dalloc(&ds->b0, ML * ld));
  A(dalloc(&ds->d0, ML * ld));
  A(dalloc(&ds->lambda, ML + 1));  // (+ 1: the kept sketch estimate, slm_dataset::sketch_valid)
  A(dalloc(&ds->dctl, 1));
  if (rc == SLM_OK) {  // (addresses only: nothing is read through the device pointer here)
    ds->ctl = ds->dctl->lane;
    ds->gctl = &ds->dctl->g;
    ds->ws_ctl = &ds->dctl->ws;
  }
  if (rc == SLM_OK) {
    hipError_t e2 = hipHostMalloc((void**)&ds->hctl, 2 * sizeof(HostCtl), hipHostMallocDefault);
    if (e2 != hipSuccess) rc = fail(SLM_ERR_OOM, "hipHostMalloc: %s", hipGetErrorString(e2));
  }
  for (int k = 0; k < 2 && rc == SLM_OK; ++k) {
    hipError_t e2 = hipEventCreateWithFlags(&ds->ev[k], hipEventDisableTiming);
    if (e2 != hipSuccess) rc = fail(SLM_ERR_HIP, "hipEventCreate: %s", hipGetErrorString(e2));
  }
  if (rc == SLM_OK) rc = set_singleton_groups(ds);
  if (rc == SLM_OK) {
    hipStream_t s = eng->stream;
    hipError_t e3 = hipMemsetAsync(ds->X, 0, sizeof(double) * (size_t)n * ld, s);
    if (e3 == hipSuccess) e3 = hipMemsetAsync(ds->yzero, 0, sizeof(double) * n, s);
    if (e3 == hipSuccess) e3 = hipMemsetAsync(ds->z, 0, sizeof(double) * ML * ld, s);
    if (e3 == hipSuccess) e3 = hipMemsetAsync(ds->beta, 0, sizeof(double) * ML * ld, s);
    if (e3 == hipSuccess) e3 = hipMemsetAsync(ds->zprev, 0, sizeof(double) * ML * ld, s);
    if (e3 == hipSuccess) e3 = hipMemsetAsync(ds->gprev, 0, sizeof(double) * ML * ld, s);
    if (e3 == hipSuccess) e3 = hipMemsetAsync(ds->g, 0, sizeof(double) * ML * (ld + 16), s);
    if (e3 == hipSuccess) e3 = hipStreamSynchronize(s);
    if (e3 != hipSuccess) rc = fail(SLM_ERR_HIP, "memset: %s", hipGetErrorString(e3));
  }
  if (rc != SLM_OK) {
    dataset_free(ds);
    return rc;
  }
  *out = ds;
  return SLM_OK;
}

static int upload_row_weights(slm_dataset* ds, const double* rw_host) {
  ds->L_valid = false;
  ds->sketch_valid = false;
  ds->carry_valid = false;
  mg_invalidate(ds);
  if (!rw_host) {
    dfree(ds->rw);
    ds->rw_max = 1.0;
    return SLM_OK;
  }
  double top = 0.0;
  for (int64_t i = 0; i < ds->n; ++i) {
    if (!(rw_host[i] >= 0.0)) return fail(SLM_ERR_BAD_ARG, "row_weight[%lld] is negative or NaN", (long long)i);
    top = std::max(top, rw_host[i]);
  }
  ds->rw_max = top;
  if (!ds->rw) SLM_TRY(dalloc(&ds->rw, ds->n));
  HIP_TRY(hipMemcpy(ds->rw, rw_host, sizeof(double) * ds->n, hipMemcpyHostToDevice));
  return SLM_OK;
}

extern "C" int slm_dataset_create(slm_engine* eng, const double* X, int64_t n, int64_t p,
                                  int64_t row_stride, int64_t col_stride, const double* y,
                                  const double* row_weight, slm_dataset** out) {
  if (!eng || !X || !y || !out) return fail(SLM_ERR_BAD_ARG, "NULL argument");
  *out = nullptr;
  HIP_TRY(hipSetDevice(eng->device));
  const bool c_order = (col_stride == 1 || p == 1) && (row_stride >= p || n == 1);
  const bool f_order = !c_order && (row_stride == 1 || n == 1) && col_stride >= n;
  if (!c_order && !f_order)
    return fail(SLM_ERR_BAD_ARG, "X must be C- or F-contiguous along one axis (strides %lld, %lld)",
                (long long)row_stride, (long long)col_stride);
  slm_dataset* ds = nullptr;
  SLM_TRY(dataset_alloc(eng, n, p, &ds));
  int rc = SLM_OK;
  hipError_t e = hipSuccess;
  if (c_order) {
    const int64_t rs = (n == 1) ? p : row_stride;
    e = hipMemcpy2D(ds->X, sizeof(double) * ds->ld, X, sizeof(double) * rs, sizeof(double) * p, n,
                    hipMemcpyHostToDevice);
  } else {
    double* tmp = nullptr;
    const int64_t cs = (p == 1) ? n : col_stride;
    rc = dalloc(&tmp, (size_t)n * p);
    if (rc == SLM_OK) {
      e = hipMemcpy2D(tmp, sizeof(double) * n, X, sizeof(double) * cs, sizeof(double) * n, p,
                      hipMemcpyHostToDevice);
      if (e == hipSuccess) {
        dim3 grid((unsigned)((n + 31) / 32), (unsigned)((p + 31) / 32));
        hipLaunchKernelGGL(transpose_f2c_kernel, grid, dim3(256), 0, eng->stream, tmp, n, p, ds->X, ds->ld);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(eng->stream);
      }
      dfree(tmp);
    }
  }
  if (rc == SLM_OK && e != hipSuccess) rc = fail(SLM_ERR_HIP, "upload of X failed: %s", hipGetErrorString(e));
  if (rc == SLM_OK) {
    e = hipMemcpy(ds->y, y, sizeof(double) * n, hipMemcpyHostToDevice);
    if (e != hipSuccess) rc = fail(SLM_ERR_HIP, "upload of y failed: %s", hipGetErrorString(e));
  }
  if (rc == SLM_OK) rc = upload_row_weights(ds, row_weight);
  if (rc != SLM_OK) {
    dataset_free(ds);
    return rc;
  }
  *out = ds;
  return SLM_OK;
}

extern "C" int slm_dataset_create_device(slm_engine* eng, const double* dX, int64_t n, int64_t p,
                                         int64_t ld_in, const double* dy, const double* d_row_weight,
                                         slm_dataset** out) {
  if (!eng || !dX || !dy || !out) return fail(SLM_ERR_BAD_ARG, "NULL argument");
  *out = nullptr;
  if (ld_in < p) return fail(SLM_ERR_BAD_ARG, "ld (%lld) < p (%lld)", (long long)ld_in, (long long)p);
  HIP_TRY(hipSetDevice(eng->device));
  slm_dataset* ds = nullptr;
  SLM_TRY(dataset_alloc(eng, n, p, &ds));
  hipError_t e = hipMemcpy2DAsync(ds->X, sizeof(double) * ds->ld, dX, sizeof(double) * ld_in,
                                  sizeof(double) * p, n, hipMemcpyDeviceToDevice, eng->stream);
  if (e == hipSuccess)
    e = hipMemcpyAsync(ds->y, dy, sizeof(double) * n, hipMemcpyDeviceToDevice, eng->stream);
  int rc = SLM_OK;
  if (e == hipSuccess && d_row_weight) {
    ds->rw_max = -1.0;  // (not looked at on the host)
    rc = dalloc(&ds->rw, n);
    if (rc == SLM_OK)
      e = hipMemcpyAsync(ds->rw, d_row_weight, sizeof(double) * n, hipMemcpyDeviceToDevice, eng->stream);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(eng->stream);
  if (rc == SLM_OK && e != hipSuccess) rc = fail(SLM_ERR_HIP, "device copy failed: %s", hipGetErrorString(e));
  if (rc != SLM_OK) {
    dataset_free(ds);
    return rc;
  }
  *out = ds;
  return SLM_OK;
}

extern "C" int slm_dataset_nonfinite(slm_dataset* ds, int32_t* kind_out) {
  if (!ds || !kind_out) return fail(SLM_ERR_BAD_ARG, "NULL argument");
  HIP_TRY(hipSetDevice(ds->eng->device));
  hipStream_t s = ds->eng->stream;
  int* flags = nullptr;
  SLM_TRY(dalloc(&flags, 1));
  int rc = SLM_OK;
  hipError_t e = hipMemsetAsync(flags, 0, sizeof(int), s);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(nonfinite_kernel, dim3((unsigned)(ds->eng->cus * 8)), dim3(256), 0, s, ds->X, (int64_t)ds->n * ds->ld, flags);
    e = hipGetLastError();
  }
  int host = 0;
  if (e == hipSuccess) e = hipMemcpyAsync(&host, flags, sizeof(int), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e != hipSuccess) rc = fail(SLM_ERR_HIP, "scan of X failed: %s", hipGetErrorString(e));
  dfree(flags);
  *kind_out = host;
  return rc;
}

// A copy of (X, y, row weights) on another engine of the same device: a second stream's own dataset without a
// second trip over PCIe.  Group structure is not carried over (the caller sets it again).
extern "C" int slm_dataset_clone(slm_dataset* src, slm_engine* eng, slm_dataset** out) {
  if (!src || !eng || !out) return fail(SLM_ERR_BAD_ARG, "NULL argument");
  *out = nullptr;
  if (eng->device != src->eng->device)
    return fail(SLM_ERR_BAD_ARG, "the copy has to live on the device of the original (%d), not on %d", src->eng->device,
                eng->device);
  HIP_TRY(hipSetDevice(src->eng->device));
  HIP_TRY(hipStreamSynchronize(src->eng->stream));
  slm_dataset* ds = nullptr;
  SLM_TRY(slm_dataset_create_device(eng, src->X, src->n, src->p, src->ld, src->y, src->rw, &ds));
  ds->n_global = src->n_global;
  ds->rw_max = src->rw_max;
  // the Grams of covariance passes built so far are shared, not copied (same device, read-only)
  ds->cov = src->cov;
  ds->cov_all_hold = src->cov_all_hold;
  ds->cov_all = src->cov_all;
  *out = ds;
  return SLM_OK;
}

extern "C" int slm_dataset_create_synthetic(slm_engine* eng, int64_t n, int64_t p, uint64_t seed,
                                            int64_t row_offset, const double* coef, double noise_sd,
                                            slm_dataset** out) {
  if (!eng || !coef || !out) return fail(SLM_ERR_BAD_ARG, "NULL argument");
  *out = nullptr;
  HIP_TRY(hipSetDevice(eng->device));
  slm_dataset* ds = nullptr;
  SLM_TRY(dataset_alloc(eng, n, p, &ds));
  // coef rides in ds->u for the duration of the generation
  hipError_t e = hipMemcpy(ds->u, coef, sizeof(double) * p, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    const int blocks = eng->cus * 8;
    hipLaunchKernelGGL(synth_x_kernel, dim3(blocks), dim3(256), 0, eng->stream, ds->X, n, p, ds->ld, seed,
                       row_offset);
    hipLaunchKernelGGL(synth_y_kernel, dim3(blocks), dim3(256), 0, eng->stream, ds->X, n, p, ds->ld, ds->u,
                       noise_sd, seed, row_offset, ds->y);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(eng->stream);
  }
  if (e != hipSuccess) {
    dataset_free(ds);
    return fail(SLM_ERR_HIP, "synthetic generation failed: %s", hipGetErrorString(e));
  }
  *out = ds;
  return SLM_OK;
}

extern "C" int slm_dataset_destroy(slm_dataset* ds) {
  if (!ds) return SLM_OK;
  (void)hipSetDevice(ds->eng->device);
  (void)hipStreamSynchronize(ds->eng->stream);
  dataset_free(ds);
  return SLM_OK;
}

extern "C" int slm_dataset_shape(slm_dataset* ds, int64_t* n, int64_t* p, int64_t* ld) {
  if (!ds) return fail(SLM_ERR_BAD_ARG, "dataset is NULL");
  if (n) *n = ds->n;
  if (p) *p = ds->p;
  if (ld) *ld = ds->ld;
  return SLM_OK;
}

extern "C" int slm_dataset_download(slm_dataset* ds, double* X_out, double* y_out) {
  if (!ds) return fail(SLM_ERR_BAD_ARG, "dataset is NULL");
  HIP_TRY(hipSetDevice(ds->eng->device));
  HIP_TRY(hipStreamSynchronize(ds->eng->stream));
  if (X_out)
    HIP_TRY(hipMemcpy2D(X_out, sizeof(double) * ds->p, ds->X, sizeof(double) * ds->ld,
                        sizeof(double) * ds->p, ds->n, hipMemcpyDeviceToHost));
  if (y_out) HIP_TRY(hipMemcpy(y_out, ds->y, sizeof(double) * ds->n, hipMemcpyDeviceToHost));
  return SLM_OK;
}

extern "C" int slm_dataset_center(slm_dataset* ds, double* x_mean_out, double* y_mean_out);

extern "C" int slm_dataset_set_row_weights(slm_dataset* ds, const double* row_weight) {
  if (!ds) return fail(SLM_ERR_BAD_ARG, "dataset is NULL");
  HIP_TRY(hipSetDevice(ds->eng->device));
  HIP_TRY(hipStreamSynchronize(ds->eng->stream));
  return upload_row_weights(ds, row_weight);
}

// Replace the targets in place (X stays): y appears only in the residuals, so nothing cached with the
// dataset (column-major copy, step-size bound) depends on it.
extern "C" int slm_dataset_set_targets(slm_dataset* ds, const double* y) {
  if (!ds || !y) return fail(SLM_ERR_BAD_ARG, "NULL argument");
  for (int64_t i = 0; i < ds->n; ++i)
    if (!std::isfinite(y[i])) return fail(SLM_ERR_BAD_ARG, "y[%lld] is not finite", (long long)i);
  HIP_TRY(hipSetDevice(ds->eng->device));
  HIP_TRY(hipStreamSynchronize(ds->eng->stream));
  HIP_TRY(hipMemcpy(ds->y, y, sizeof(double) * ds->n, hipMemcpyHostToDevice));
  ds->carry_valid = false;
  // (the Grams of covariance passes carry X^T W y: gone with the old targets; the Gram of all rows depends on X alone)
  cov_pending_drop(ds);
  ds->cov.clear();
  return SLM_OK;
}

extern "C" int slm_dataset_set_global_rows(slm_dataset* ds, int64_t n_global) {
  if (!ds) return fail(SLM_ERR_BAD_ARG, "dataset is NULL");
  if (n_global < 1) return fail(SLM_ERR_BAD_ARG, "n_global must be positive (got %lld)", (long long)n_global);
  ds->n_global = n_global;
  ds->L_valid = false;
  ds->sketch_valid = false;
  ds->carry_valid = false;
  ds->colnorm_ready = false;  // (the norms are scaled by 1 / sqrt(n_global))
  mg_invalidate(ds);
  return SLM_OK;
}

extern "C" int slm_dataset_set_groups(slm_dataset* ds, const int32_t* gid, int32_t n_groups) {
  if (!ds) return fail(SLM_ERR_BAD_ARG, "dataset is NULL");
  // (the estimators set the groups before every fit -- a cached dataset may carry another estimator's: the same
  //  structure again costs nothing, where it used to cost four allocations and three blocking copies)
  if (!gid && ds->singleton && ds->h_gid.empty()) return SLM_OK;
  if (gid && !ds->singleton && n_groups == ds->G && ds->h_gid.size() == (size_t)ds->p &&
      memcmp(ds->h_gid.data(), gid, sizeof(int32_t) * (size_t)ds->p) == 0)
    return SLM_OK;
  HIP_TRY(hipSetDevice(ds->eng->device));
  HIP_TRY(hipStreamSynchronize(ds->eng->stream));
  ds->ws_carry_valid = false;  // (the working set's group tables belong to the old structure)
  if (!gid) {
    ds->h_gid.clear();
    return set_singleton_groups(ds);
  }
  const int p = (int)ds->p;
  if (n_groups <= 0 || n_groups > p)
    return fail(SLM_ERR_BAD_ARG, "n_groups = %d must be in [1, p = %d]", n_groups, p);
  std::vector<int> count(n_groups, 0);
  for (int j = 0; j < p; ++j) {
    if (gid[j] < 0 || gid[j] >= n_groups)
      return fail(SLM_ERR_BAD_ARG, "gid[%d] = %d outside [0, %d)", j, gid[j], n_groups);
    count[gid[j]]++;
  }
  std::vector<int> start(n_groups + 1, 0);
  int max_size = 1;
  for (int g = 0; g < n_groups; ++g) {
    start[g + 1] = start[g] + count[g];
    max_size = std::max(max_size, count[g]);
  }
  std::vector<int> order(p), fill(start.begin(), start.end() - 1);
  for (int j = 0; j < p; ++j) order[fill[gid[j]]++] = j;  // stable inside a group
  int team = 1;
  while (team < max_size && team < 64) team <<= 1;
  dfree(ds->order); dfree(ds->gid); dfree(ds->gstart); dfree(ds->gscale);
  SLM_TRY(dalloc(&ds->order, p));
  SLM_TRY(dalloc(&ds->gid, p));
  SLM_TRY(dalloc(&ds->gstart, n_groups + 1));
  SLM_TRY(dalloc(&ds->gscale, (size_t)ds->lane_cap * n_groups));
  HIP_TRY(hipMemcpy(ds->order, order.data(), sizeof(int) * p, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(ds->gid, gid, sizeof(int) * p, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(ds->gstart, start.data(), sizeof(int) * (n_groups + 1), hipMemcpyHostToDevice));
  ds->G = n_groups;
  ds->singleton = 0;
  ds->team = team;
  ds->max_group = max_size;
  ds->h_gid.assign(gid, gid + p);
  return SLM_OK;
}

// ------------------------------------------------------------------------------------------------
// row-sharded mode
// ------------------------------------------------------------------------------------------------
extern "C" int slm_comm_info(slm_engine* eng, int32_t* rank_out, int32_t* n_ranks_out) {
  if (!eng) return fail(SLM_ERR_BAD_ARG, "engine is NULL");
  int rank = 0, count = 1;
  if (eng->comm) {
    if (!g_rccl.CommCount || !g_rccl.CommUserRank) return fail(SLM_ERR_COMM, "librccl lacks ncclCommCount / ncclCommUserRank");
    RCCL_TRY(g_rccl.CommCount(eng->comm, &count));
    RCCL_TRY(g_rccl.CommUserRank(eng->comm, &rank));
  } else if (eng->local) {
    rank = eng->rank;
    count = eng->local->n_ranks;
  }
  if (rank_out) *rank_out = rank;
  if (n_ranks_out) *n_ranks_out = count;
  return SLM_OK;
}

extern "C" int slm_comm_collectives(slm_engine* eng, int64_t* count_out) {
  if (!eng || !count_out) return fail(SLM_ERR_BAD_ARG, "NULL argument");
  *count_out = eng->collectives;
  return SLM_OK;
}

// Measurement: `reps` all-reduces of `count` doubles on the engine's communicator, one after the other on its stream
// (after one warm-up), by HIP events: microseconds per collective -- the latency floor a row-sharded pass pays per
// collective (bench.py `rowshard.collective_us`).  Every rank of the communicator must call it with the same arguments.
extern "C" int slm_comm_all_reduce_probe(slm_engine* eng, int64_t count, int32_t reps, double* us_out) {
  if (!eng || !us_out) return fail(SLM_ERR_BAD_ARG, "NULL argument");
  if (!eng->sharded()) return fail(SLM_ERR_COMM, "the engine has no communicator");
  if (count < 1 || reps < 1) return fail(SLM_ERR_BAD_ARG, "count and reps must be positive");
  HIP_TRY(hipSetDevice(eng->device));
  double* buf = nullptr;
  SLM_TRY(dalloc(&buf, (size_t)count));
  HIP_TRY(hipMemsetAsync(buf, 0, sizeof(double) * (size_t)count, eng->stream));
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  int rc = all_reduce_sum(eng, buf, (size_t)count);
  if (rc == SLM_OK) {
    (void)hipEventRecord(e0, eng->stream);
    for (int r = 0; r < reps && rc == SLM_OK; ++r) rc = all_reduce_sum(eng, buf, (size_t)count);
    (void)hipEventRecord(e1, eng->stream);
  }
  hipError_t e = hipStreamSynchronize(eng->stream);
  float ms = 0.f;
  if (rc == SLM_OK && e == hipSuccess) (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  dfree(buf);
  if (rc != SLM_OK) return rc;
  if (e != hipSuccess) return fail(SLM_ERR_HIP, "all-reduce probe: %s", hipGetErrorString(e));
  *us_out = 1e3 * (double)ms / reps;
  return SLM_OK;
}

static void local_comm_release(LocalComm* lc) {
  bool last = false;
  {
    std::lock_guard<std::mutex> lk(lc->m);
    last = --lc->refs == 0;
  }
  if (!last) return;
  (void)hipSetDevice(lc->device);
  for (int r = 0; r < lc->n_ranks; ++r) {
    if (lc->stage[r]) (void)hipFree(lc->stage[r]);
    for (int k = 0; k < 2; ++k) {
      if (lc->ready[r][k]) (void)hipEventDestroy(lc->ready[r][k]);
      if (lc->consumed[r][k]) (void)hipEventDestroy(lc->consumed[r][k]);
    }
  }
  delete lc;
}

extern "C" int slm_comm_init_local(slm_engine** engines, int32_t n_ranks, double timeout_s) {
  if (!engines) return fail(SLM_ERR_BAD_ARG, "engines is NULL");
  if (n_ranks < 1 || n_ranks > LocalComm::kMaxRanks)
    return fail(SLM_ERR_BAD_ARG, "n_ranks must be in [1, %d] (got %d)", LocalComm::kMaxRanks, n_ranks);
  for (int r = 0; r < n_ranks; ++r) {
    if (!engines[r]) return fail(SLM_ERR_BAD_ARG, "engine %d is NULL", r);
    if (engines[r]->sharded()) return fail(SLM_ERR_BAD_ARG, "engine %d already has a communicator", r);
    if (engines[r]->device != engines[0]->device) return fail(SLM_ERR_BAD_ARG, "the engines of an in-process communicator share one device");
    for (int q = 0; q < r; ++q)
      if (engines[q] == engines[r]) return fail(SLM_ERR_BAD_ARG, "engine %d is listed twice", r);
  }
  HIP_TRY(hipSetDevice(engines[0]->device));
  LocalComm* lc = new LocalComm();
  lc->n_ranks = n_ranks;
  lc->device = engines[0]->device;
  lc->refs = n_ranks;
  if (timeout_s > 0.0) lc->timeout_s = timeout_s;
  // the largest exchange is the staged working-set Gram of every lane set
  lc->cap = (size_t)SLM_MAX_LANES * WS_KCAP * WS_KCAP + STOP_WORDS;
  const size_t tab_doubles = 2 * LocalComm::kMaxRanks;  // two pointer tables behind rank 0's staging areas
  hipError_t e = hipSuccess;
  for (int r = 0; r < n_ranks && e == hipSuccess; ++r) {
    e = hipMalloc((void**)&lc->stage[r], sizeof(double) * (2 * lc->cap + (r == 0 ? tab_doubles : 0)));
    for (int k = 0; k < 2 && e == hipSuccess; ++k) {
      e = hipEventCreateWithFlags(&lc->ready[r][k], hipEventDisableTiming);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&lc->consumed[r][k], hipEventDisableTiming);
    }
  }
  if (e == hipSuccess) {
    const double* tab[2][LocalComm::kMaxRanks] = {};
    for (int k = 0; k < 2; ++k)
      for (int r = 0; r < n_ranks; ++r) tab[k][r] = lc->stage[r] + (size_t)k * lc->cap;
    e = hipMemcpy(lc->stage[0] + 2 * lc->cap, tab, sizeof(tab), hipMemcpyHostToDevice);
  }
  if (e != hipSuccess) {
    lc->refs = 1;
    local_comm_release(lc);
    return fail(e == hipErrorOutOfMemory ? SLM_ERR_OOM : SLM_ERR_HIP, "slm_comm_init_local: %s", hipGetErrorString(e));
  }
  for (int r = 0; r < n_ranks; ++r) {
    engines[r]->local = lc;
    engines[r]->rank = r;
    engines[r]->n_ranks = n_ranks;
    engines[r]->collectives = 0;
  }
  return SLM_OK;
}

extern "C" int slm_comm_unique_id(uint8_t id_out[SLM_COMM_ID_BYTES]) {
  if (!id_out) return fail(SLM_ERR_BAD_ARG, "id_out is NULL");
  SLM_TRY(load_rccl());
  rcclUniqueId_t id;
  RCCL_TRY(g_rccl.GetUniqueId(&id));
  memcpy(id_out, id.internal, SLM_COMM_ID_BYTES);
  return SLM_OK;
}

extern "C" int slm_comm_init(slm_engine* eng, int32_t rank, int32_t n_ranks,
                             const uint8_t id_in[SLM_COMM_ID_BYTES]) {
  if (!eng || !id_in) return fail(SLM_ERR_BAD_ARG, "NULL argument");
  if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(SLM_ERR_BAD_ARG, "bad rank %d of %d", rank, n_ranks);
  if (eng->sharded()) return fail(SLM_ERR_BAD_ARG, "communicator already initialised");
  SLM_TRY(load_rccl());
  HIP_TRY(hipSetDevice(eng->device));
  rcclUniqueId_t id;
  memcpy(id.internal, id_in, SLM_COMM_ID_BYTES);
  RCCL_TRY(g_rccl.CommInitRank(&eng->comm, n_ranks, id, rank));
  eng->rank = rank;
  eng->n_ranks = n_ranks;
  eng->collectives = 0;
  return SLM_OK;
}

extern "C" int slm_comm_destroy(slm_engine* eng) {
  if (!eng) return fail(SLM_ERR_BAD_ARG, "engine is NULL");
  if (eng->comm) {
    (void)hipSetDevice(eng->device);
    (void)hipStreamSynchronize(eng->stream);
    if (g_rccl.CommDestroy) (void)g_rccl.CommDestroy(eng->comm);
    eng->comm = nullptr;
    eng->n_ranks = 1;
    eng->rank = 0;
  }
  if (eng->local) {
    (void)hipSetDevice(eng->device);
    (void)hipStreamSynchronize(eng->stream);
    LocalComm* lc = eng->local;
    eng->local = nullptr;
    eng->n_ranks = 1;
    eng->rank = 0;
    local_comm_release(lc);
  }
  return SLM_OK;
}


// Host side of the MI355X fit engine: handles, HBM layout, launch logic, the C ABI of
// include/slm_engine.h.  Replaces the cvxpy `problem.solve` call of
// /root/reference/src/sparselm/model/_base.py:512-519 (and the inner solve of
// model/_adaptive_lasso.py:213-215) with a device-resident FISTA state machine.
//
// gfx950 only; there is no CPU fallback anywhere in this file.
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/slm_engine.h"
#include "data_kernels.hpp"
#include "grad_kernel.hpp"
#include "tail_kernels.hpp"
#include "ws_kernels.hpp"
#include "split_kernels.hpp"
#include "small_kernels.hpp"
#include "small_split_kernels.hpp"
#include "cov_kernels.hpp"

using namespace slm;

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;

static int fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

#define HIP_TRY(expr)                                                                       \
  do {                                                                                      \
    hipError_t e__ = (expr);                                                                \
    if (e__ != hipSuccess)                                                                  \
      return fail(e__ == hipErrorOutOfMemory ? SLM_ERR_OOM : SLM_ERR_HIP, "%s failed: %s (%s:%d)", \
                  #expr, hipGetErrorString(e__), __FILE__, __LINE__);                       \
  } while (0)

#define SLM_TRY(expr)            \
  do {                           \
    int rc__ = (expr);           \
    if (rc__ != SLM_OK) return rc__; \
  } while (0)

// ------------------------------------------------------------------------------------------------
// RCCL, loaded lazily (single-GPU use never touches it)
// ------------------------------------------------------------------------------------------------
typedef struct { char internal[SLM_COMM_ID_BYTES]; } rcclUniqueId_t;
typedef void* rcclComm_t;
struct RcclApi {
  void* lib = nullptr;
  int (*GetUniqueId)(rcclUniqueId_t*) = nullptr;
  int (*CommInitRank)(rcclComm_t*, int, rcclUniqueId_t, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, rcclComm_t, hipStream_t) = nullptr;
  int (*CommDestroy)(rcclComm_t) = nullptr;
  int (*CommCount)(rcclComm_t, int*) = nullptr;
  int (*CommUserRank)(rcclComm_t, int*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
static RcclApi g_rccl;
static const int kNcclFloat64 = 8;  // ncclDouble
static const int kNcclSum = 0;      // ncclSum

static int load_rccl() {
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

#define RCCL_TRY(expr)                                                                      \
  do {                                                                                      \
    int e__ = (expr);                                                                       \
    if (e__ != 0)                                                                           \
      return fail(SLM_ERR_COMM, "%s failed: %s", #expr,                                     \
                  g_rccl.GetErrorString ? g_rccl.GetErrorString(e__) : "rccl error");       \
  } while (0)

// ------------------------------------------------------------------------------------------------
// gradient kernel table
// ------------------------------------------------------------------------------------------------
struct GradKernel {
  int W, C, R, B;
  int D;  // 0: rows wait in VGPRs (grad_fused_kernel); > 0: LDS ring, D rows in flight (grad_ring_kernel)
  void (*fn)(GradArgs);
};

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
static const int kMaxTailE = 64;  // tail kernel instantiations cover p <= 1024 * 64
static const int kProfStride = 3;  // SLM_FLAG_PROFILE times every 3rd gradient launch (a working-set path has ~5: two of them;
                                   // an event pair costs ~12 us of stream around the launch it brackets)
static const int64_t kMaxChunks = 64 * 8 * 10;  // largest row the fused kernel covers (p <= 10240)

// Rows longer than the fused kernels cover: two-pass fallback (D = -1), one lane, any p.
static const GradKernel kGradTwoPass = {8, 4, 2, 1, -1, nullptr};
static const int kTwoPassC = 4;  // column tile of xtr_kernel: 512 * 4 chunks = 4096 columns

// Split pass (split_kernels.hpp) for working-set solves: sixteen lanes per read of X.  The table is for
// rowdot_ring_kernel (rows of up to 5120 columns, D rows in flight as for the fused ring kernel); rows of
// 5 121 ... 10 240 columns (BASELINE config 5: p = 10 000) have no ring variant -- their LDS ring would not
// fit -- and take every residual that needs X from rowdot_mfma_kernel, which has no column limit but needs
// the column-major copy of X (`rowdot == nullptr`: the split pass is then only used when that copy exists).
struct SplitKernel {
  int W, C, B, D;
  void (*rowdot)(SplitArgs);
  void (*resid)(SplitArgs);
};
#define SLM_SK(C, D)                                                                                   \
  {8, C, SPLIT_LANES, D, rowdot_ring_kernel<8, C, ROWDOT_LANES, D>, resid_ws_kernel<SPLIT_LANES>}
static const SplitKernel kSplit[] = {SLM_SK(1, 3), SLM_SK(2, 3), SLM_SK(3, 3), SLM_SK(4, 3), SLM_SK(5, 2),
                                     {8, 10, SPLIT_LANES, 0, nullptr, resid_ws_kernel<SPLIT_LANES>}};
static const SplitKernel* pick_split_kernel(int64_t p2) {
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
static int xtr_max_row_blocks(int cus, int64_t ld) {  // (sizes the partial buffer: the larger of the two grids)
  const int xb = (int)((ld + XTR_CB - 1) / XTR_CB);
  return std::max(1, 2 * cus / xb);
}
// sets a.xrows; returns the number of row blocks (= blocks of `partial` to reduce)
static int launch_xtr(int cus, SplitArgs& a, hipStream_t s) {
  const int xb = (int)((a.ld + XTR_CB - 1) / XTR_CB);
  double per_cu = 1.0;
  if (const char* e = getenv("SLM_XTR_WGS_PER_CU")) {  // (A/B runs: workgroups per CU, up to 2)
    const double f = atof(e);
    if (f > 0.0 && f <= 2.0) per_cu = f;
  }
  const int64_t want = std::max<int64_t>(1, (int64_t)(xtr_max_row_blocks(cus, a.ld) * per_cu / 2.0));
  int64_t rows = (a.n + want - 1) / want;
  rows = (rows + 7) / 8 * 8;
  const int yb = (int)((a.n + rows - 1) / rows);  // <= want
  a.xrows = (int)rows;
  hipLaunchKernelGGL(xtr_mfma_kernel, dim3(xb, yb), dim3(XTR_WAVES * 64), 0, s, a);
  return yb;
}

// the product of a covariance pass (cov_gz_mfma_kernel): xtr_mfma_kernel's grid; when only the working set's rows are read a
// workgroup row takes the next multiple of four of WS_KCAP / row blocks list entries (at most 32: eight steps in registers)
static int launch_cov_gz(int cus, SplitArgs& a, hipStream_t s) {
  const int xb = (int)((a.ld + XTR_CB - 1) / XTR_CB);
  const int64_t want = std::max<int64_t>(1, xtr_max_row_blocks(cus, a.ld) / 2);
  int64_t rows = (a.n + want - 1) / want;
  rows = (rows + 7) / 8 * 8;
  const int yb = (int)((a.n + rows - 1) / rows);  // <= want
  a.xrows = (int)rows;
  const int per = ((WS_KCAP + yb - 1) / yb + 3) / 4 * 4;
  a.xrows_ws = (a.ctl != nullptr && per <= 32) ? per : 0;
  hipLaunchKernelGGL(cov_gz_mfma_kernel, dim3(xb, yb), dim3(XTR_WAVES * 64), 0, s, a);
  return yb;
}

static const GradKernel* pick_grad_kernel(int64_t p2, int B) {
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
// In-process communicator (slm_comm_init_local): the engines of ONE process on ONE device form the ranks
// of a row-sharded job, each driven by its own host thread.  The all-reduce is a rendezvous of their
// streams -- every rank copies its buffer into a staging area, the host threads meet, every rank waits for
// the others' copies (events) and adds the staged buffers in rank order, so all ranks get identical bits,
// like the ring all-reduce RCCL runs between GPUs.  It exists so that the multi-rank state machine (stop
// agreement, Gram exchange, sharded centring) can run -- and be tested -- where RCCL cannot form a group:
// several ranks on one GPU.  A rank that does not show up within `timeout_s` fails the collective with
// SLM_ERR_COMM on the ranks that did: mismatched collective counts are an error report, not a hang.
// ------------------------------------------------------------------------------------------------
struct LocalComm {
  static const int kMaxRanks = 8;
  int n_ranks = 0;
  int device = 0;
  size_t cap = 0;                   // doubles per rank of staging
  double* stage[kMaxRanks] = {};    // device
  hipEvent_t ready[kMaxRanks][2] = {};     // rank r's copy of round k (parity k & 1) is in its staging area
  hipEvent_t consumed[kMaxRanks][2] = {};  // rank r has finished reading every staging area of round k
  std::mutex m;
  std::condition_variable cv;
  long arrived[2] = {0, 0};         // host threads that have recorded `ready` / `consumed` for the round
  long round_of[kMaxRanks] = {};    // collectives each rank has entered
  int refs = 0;
  double timeout_s = 30.0;
};

struct slm_engine {
  int device = 0;
  hipStream_t stream = nullptr;
  hipDeviceProp_t prop;
  int cus = 0;
  // row-sharded mode: RCCL communicator, or the in-process one
  rcclComm_t comm = nullptr;
  LocalComm* local = nullptr;
  int rank = 0, n_ranks = 1;
  long collectives = 0;  // all-reduces this engine has entered (diagnostics, slm_comm_info)
  // a second stream for collectives that run beside the solve stream's kernels (the folds' Grams of a replicated
  // dataset: part f is summed over the ranks while part f + 1 is still being built), made on first use
  hipStream_t comm_stream = nullptr;
  hipEvent_t comm_ev = nullptr;
  bool sharded() const { return comm != nullptr || local != nullptr; }
};

__global__ void local_sum_kernel(double* out, const double* const* stage, int n_ranks, size_t count) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
    double t = stage[0][i];
    for (int r = 1; r < n_ranks; ++r) t += stage[r][i];  // fixed order: every rank gets the same bits
    out[i] = t;
  }
}

static int fail(int code, const char* fmt, ...);

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
static int all_reduce_sum(slm_engine* eng, double* buf, size_t count, hipStream_t s = nullptr) {
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

static const int kMaxLanes = SLM_MAX_LANES;

// Control words shared by all lanes of a solve.
struct GlobalCtl {
  int32_t done;        // every lane finished, or abort
  int32_t lanes_done;
  int32_t hard;        // most passes any lane has spent on one path point so far (TailArgs::gdone[2])
  int32_t done_local;  // row-sharded mode: this rank has finished (TailArgs::gdone[3]); `done` follows once all have
  int32_t diverged;    // row-sharded mode: the ranks' control blocks differ (TailArgs::gdone[4]); ends the solve
  int32_t pad_[3];
};
static const int kWsLateIters = 12;  // passes on one point after which a small problem gets the working set

static const int kSnapInfos = 64;
// Everything the host reads back about a solve sits in ONE device block -- the stop word, the lanes' control
// blocks and the working set's counters -- so a poll is one copy, and the snapshot in which the host sees
// `done` already holds the final statistics (after `done` every queued kernel returns at once: nothing in the
// block changes any more).  Reading them one by one at the end cost four blocking copies, ~0.1 ms of host
// round trips on a 5 ms path.
struct DevCtl {
  GlobalCtl g;
  WsCtl ws;  // (next to g: one fill clears both at the start of a solve)
  PathCtl lane[SLM_MAX_LANES];
  slm_point_info infos[kSnapInfos];  // the per-point records of solves of up to kSnapInfos points ride along
};
struct HostCtl {  // pinned snapshot the host polls
  DevCtl c;
};

static void pool_free(void* p);

struct slm_dataset {
  slm_engine* eng = nullptr;
  int64_t n = 0, p = 0, ld = 0, n_global = 0;
  // On an engine with a communicator a dataset is a row block of one tall matrix (every pass all-reduces its gradients),
  // unless it is marked as a REPLICA (slm_dataset_set_replicated: grid mode -- every rank holds all rows and solves its own
  // lanes; the communicator then only carries the folds' Grams, each rank building the part of an n_ranks-th of the rows)
  bool replicated = false;
  double *X = nullptr, *y = nullptr, *rw = nullptr, *yzero = nullptr;
  double rw_max = 1.0;  // largest row weight (1 without row weights; < 0: unknown -- weights handed over on the device)
  double* rw_lanes = nullptr;  // [kMaxLanes][n], allocated when a lane brings its own row weights
  double* rvec = nullptr;      // [n] residuals of the two-pass fallback
  // group structure (group-sorted permutation)
  int G = 0, singleton = 1, team = 1, max_group = 1;
  std::vector<int32_t> h_gid;  // the group index as last set (empty: singleton groups) -- setting it again is free
  int *order = nullptr, *gid = nullptr, *gstart = nullptr;
  // working-set refinement (ws_kernels.hpp), allocated on first use
  WsCtl* ws_ctl = nullptr;  // (inside dctl)
  int32_t *ws_idx = nullptr, *ws_pos = nullptr, *ws_gs = nullptr, *ws_gl = nullptr;
  double *ws_score = nullptr, *ws_XW = nullptr, *ws_part = nullptr, *ws_G = nullptr, *ws_Gx = nullptr;
  double* ws_nt = nullptr;  // [kMaxLanes][NT_SCRATCH] factors of the model solver's direct steps
  double *sse_Z = nullptr, *sse_part = nullptr;  // slm_eval_sse_sparse: coefficient block, partial sums
  size_t sse_cap = 0;
  PathCtl h_stage[SLM_MAX_LANES];  // host staging of the control blocks of the solve in flight
  // page-locked staging of what the lanes of a call bring (penalty vectors, warm starts: [4][kMaxLanes][ld]; path points):
  // one transfer per kind instead of one per lane and kind -- sixteen lanes x (a, warm start, points) were 48 transfers of a
  // few hundred bytes, 0.25 ms of submissions before a 0.2 ms call of the on-chip solver
  double* h_vec = nullptr;
  slm_path_point* h_pts = nullptr;
  int64_t h_pts_cap = 0;
  // covariance passes (cov_kernels.hpp): a Gram per row set, found again by the fingerprint of its row weights
  // (the blocks are shared with the copies of a dataset on further engines of its device -- slm_dataset_clone: the
  //  streams of a grid search -- and go back when the last holder lets go)
  struct CovBlocks {
    double *G = nullptr, *c = nullptr;
    ~CovBlocks() {
      if (G) pool_free(G);
      if (c) pool_free(c);
    }
  };
  struct CovEntry {
    std::shared_ptr<CovBlocks> hold;
    double *G = nullptr, *c = nullptr;  // = hold->G, hold->c
    double yy = 0.0, n_eff = 0.0, fp1 = 0.0, fp2 = 0.0;
  };
  std::vector<CovEntry> cov;
  struct CovPending;          // a fold build between slm_dataset_covariance_folds_begin and _finish
  CovPending* cov_pend = nullptr;
  std::shared_ptr<CovBlocks> cov_all_hold;
  double* cov_all = nullptr;  // X^T X of all rows, unscaled (the minuend of fold Grams), built on first use (= cov_all_hold->G)
  double* cov_Z = nullptr;    // [ld][16] the lanes' points, lane-minor
  double* cov_fp = nullptr;   // [2 * kMaxLanes + 2] fingerprints / scalars on their way to the host
  double* split_state = nullptr;  // [3 ld + 2 + record] slm_solve_standardized_sgl: gamma, u, rho, valid; outputs
  double* h_split = nullptr;      // its page-locked staging: a, b, warm start in; coefficients, group norms, record out
  double* stop_words = nullptr;  // [STOP_WORDS] row-sharded mode: the vector the ranks all-reduce after every pass
  double* XT = nullptr;  // column-major copy of X in tiles of 32 rows (tile_columns_kernel), built on first use
  bool XT_ready = false, XT_failed = false;
  int ws_sets = 0;  // Gram copies allocated
  // gradient launch, per lane count B = 1..kMaxLanes (index B-1); gk == nullptr => unsupported
  const GradKernel* gk[SLM_MAX_LANES] = {};
  int nblk[SLM_MAX_LANES] = {};
  const SplitKernel* sk = nullptr;  // split pass for working-set solves (nullptr: rows too long)
  int split_nblk = 0;
  double* R = nullptr;              // [n][SPLIT_RSTRIDE] residuals of the split pass, allocated on first use
  double *partial = nullptr, *loss_partial = nullptr;
  // iteration state: kMaxLanes copies, lane stride ld (g: ld + 16)
  double *g = nullptr, *z = nullptr, *beta = nullptr, *zprev = nullptr, *gprev = nullptr;
  double *u = nullptr, *gscale = nullptr, *a0 = nullptr, *b0 = nullptr, *d0 = nullptr;
  double* lambda = nullptr;  // [kMaxLanes]
  DevCtl* dctl = nullptr;    // the block below is made of:
  PathCtl* ctl = nullptr;    // [kMaxLanes]  (= dctl->lane)
  GlobalCtl* gctl = nullptr; //              (= &dctl->g)
  HostCtl* hctl = nullptr;   // pinned, 2 slots
  hipEvent_t ev[2] = {nullptr, nullptr};
  // path buffers (grown on demand), concatenated over lanes
  int64_t cap_points = 0, cap_gn = 0;
  slm_path_point* pts = nullptr;
  double *betas_out = nullptr, *gn_out = nullptr;
  slm_point_info* infos = nullptr;
  std::vector<hipEvent_t> prof;
  // cached Lipschitz constant (dataset row weights, dataset n_global)
  double L = 0.0;
  int L_iters = 0;
  bool L_valid = false;
};

static inline bool row_sharded(const slm_dataset* ds) { return ds->eng->sharded() && !ds->replicated; }

// Large device blocks (a dataset's X, its column-major copy, the gathered columns: gigabytes each) are recycled: a freed
// block waits in a per-device list of its exact size, and the next dataset of that shape takes it instead of asking the
// driver.  hipMalloc right behind the hipFree of such blocks stalled for SECONDS now and then (the soak over 96 datasets
// of the headline shape, profiles/r02c_headline_soak.log: a 6-pass path in 2.4 s; 3.6 s in r02a) -- a fresh fit paying
// three hundred times its solve.  At most pool_idle_cap() bytes wait per process; when the driver has no memory left the
// waiting blocks are handed back and the allocation is tried again.  Nothing relies on a block's contents.
static const size_t kPoolMinBytes = (size_t)64 << 20;
// idle blocks kept per process: 16 GB unless SLM_DEVICE_POOL_GB says otherwise (0 turns the pool off) -- a block idle here is
// memory no other allocator on the GPU can have (another rank sharing the device, torch, RCCL's buffers): enough for the two
// or three blocks of the largest dataset shape seen lately, not a standing reservation
static size_t pool_idle_cap() {
  static const size_t cap = [] {
    if (const char* e = getenv("SLM_DEVICE_POOL_GB")) return (size_t)(std::max(0.0, atof(e)) * (double)((size_t)1 << 30));
    return (size_t)16 << 30;
  }();
  return cap;
}
struct DevicePool {
  std::mutex m;
  std::vector<std::pair<int, std::pair<size_t, void*>>> idle;  // (device, (bytes, block))
  std::vector<std::pair<void*, std::pair<int, size_t>>> live;  // pooled blocks in use: block -> (device, bytes)
  size_t idle_bytes = 0;
};
static DevicePool g_pool;

static void pool_flush_locked() {
  for (auto& e : g_pool.idle) (void)hipFree(e.second.second);
  g_pool.idle.clear();
  g_pool.idle_bytes = 0;
}

static hipError_t pool_malloc(void** out, size_t bytes) {
  if (bytes < kPoolMinBytes) return hipMalloc(out, bytes);
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lk(g_pool.m);
  for (size_t i = 0; i < g_pool.idle.size(); ++i) {
    if (g_pool.idle[i].first == dev && g_pool.idle[i].second.first == bytes) {
      *out = g_pool.idle[i].second.second;
      g_pool.idle.erase(g_pool.idle.begin() + (long)i);
      g_pool.idle_bytes -= bytes;
      g_pool.live.push_back({*out, {dev, bytes}});
      return hipSuccess;
    }
  }
  hipError_t e = hipMalloc(out, bytes);
  if (e == hipErrorOutOfMemory && !g_pool.idle.empty()) {
    (void)hipGetLastError();
    pool_flush_locked();
    e = hipMalloc(out, bytes);
  }
  if (e == hipSuccess) g_pool.live.push_back({*out, {dev, bytes}});
  return e;
}

static void pool_free(void* p) {
  {
    std::lock_guard<std::mutex> lk(g_pool.m);
    for (size_t i = 0; i < g_pool.live.size(); ++i) {
      if (g_pool.live[i].first == p) {
        const int dev = g_pool.live[i].second.first;
        const size_t bytes = g_pool.live[i].second.second;
        g_pool.live.erase(g_pool.live.begin() + (long)i);
        if (!getenv("SLM_NO_DEVICE_POOL") && bytes <= pool_idle_cap()) {
          // (over the cap: the blocks that have waited longest go back to the driver first)
          while (g_pool.idle_bytes + bytes > pool_idle_cap() && !g_pool.idle.empty()) {
            (void)hipFree(g_pool.idle.front().second.second);
            g_pool.idle_bytes -= g_pool.idle.front().second.first;
            g_pool.idle.erase(g_pool.idle.begin());
          }
          g_pool.idle.push_back({dev, {bytes, p}});
          g_pool.idle_bytes += bytes;
          return;
        }
        break;
      }
    }
  }
  (void)hipFree(p);
}

template <typename T>
static int dalloc(T** out, size_t count) {
  *out = nullptr;
  if (count == 0) count = 1;
  HIP_TRY(pool_malloc((void**)out, count * sizeof(T)));
  return SLM_OK;
}
template <typename T>
static void dfree(T*& p) {
  if (p) pool_free(p);
  p = nullptr;
}

// ------------------------------------------------------------------------------------------------
// library
// ------------------------------------------------------------------------------------------------
extern "C" int slm_abi_version(void) { return SLM_ABI_VERSION; }

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
  if (strncmp(eng->prop.gcnArchName, "gfx950", 6) != 0 && !getenv("SLM_ALLOW_ANY_ARCH")) {
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
    bool other = false;
    for (auto& e : g_pool.live) other = other || e.second.first == eng->device;
    if (!other) {
      for (size_t i = 0; i < g_pool.idle.size();) {
        if (g_pool.idle[i].first == eng->device) {
          (void)hipFree(g_pool.idle[i].second.second);
          g_pool.idle_bytes -= g_pool.idle[i].second.first;
          g_pool.idle.erase(g_pool.idle.begin() + (long)i);
        } else {
          ++i;
        }
      }
    }
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
static void cov_pending_drop(slm_dataset* ds);
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
  dfree(ds->ws_score); dfree(ds->ws_XW); dfree(ds->ws_part); dfree(ds->ws_G); dfree(ds->ws_Gx); dfree(ds->XT);
  dfree(ds->ws_nt); dfree(ds->stop_words); dfree(ds->sse_Z); dfree(ds->sse_part); dfree(ds->split_state);
  cov_pending_drop(ds);
  ds->cov.clear();
  ds->cov_all_hold.reset();
  ds->cov_all = nullptr;
  dfree(ds->cov_Z); dfree(ds->cov_fp);
  if (ds->h_split) (void)hipHostFree(ds->h_split);
  if (ds->hctl) (void)hipHostFree(ds->hctl);
  if (ds->h_vec) (void)hipHostFree(ds->h_vec);
  if (ds->h_pts) (void)hipHostFree(ds->h_pts);
  for (auto& e : ds->ev)
    if (e) (void)hipEventDestroy(e);
  for (auto& e : ds->prof) (void)hipEventDestroy(e);
  delete ds;
}

static int set_singleton_groups(slm_dataset* ds) {
  const int p = (int)ds->p;
  std::vector<int> ident(p), start(p + 1);
  std::iota(ident.begin(), ident.end(), 0);
  std::iota(start.begin(), start.end(), 0);
  dfree(ds->order); dfree(ds->gid); dfree(ds->gstart); dfree(ds->gscale);
  SLM_TRY(dalloc(&ds->order, p));
  SLM_TRY(dalloc(&ds->gid, p));
  SLM_TRY(dalloc(&ds->gstart, p + 1));
  SLM_TRY(dalloc(&ds->gscale, (size_t)kMaxLanes * p));
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
    if (const char* env = getenv("SLM_GRAD_BLOCKS_PER_CU")) per_cu = std::max(1, atoi(env));
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
    loss_elems = std::max(loss_elems, (size_t)ds->split_nblk * SPLIT_LANES);
  }

  int rc = SLM_OK;
  auto A = [&](int r) { if (rc == SLM_OK) rc = r; };
  const size_t ML = kMaxLanes;
  A(dalloc(&ds->X, (size_t)n * ld));
  A(dalloc(&ds->y, n));
  A(dalloc(&ds->yzero, n));
  if (ds->gk[0]->D < 0) A(dalloc(&ds->rvec, n));
  A(dalloc(&ds->partial, partial_elems));
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
  A(dalloc(&ds->lambda, ML));
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
static void cov_pending_drop(slm_dataset* ds);

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
  SLM_TRY(dalloc(&ds->gscale, (size_t)kMaxLanes * n_groups));
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
// launches
// ------------------------------------------------------------------------------------------------
struct LaneSetup {
  int B = 1;
  const double* rw = nullptr;  // device row weights handed to the kernel
  int64_t rw_stride = 0;
  double n_eff[SLM_MAX_LANES] = {};
};

static LaneSetup default_lanes(slm_dataset* ds, int B) {
  LaneSetup ls;
  ls.B = B;
  ls.rw = ds->rw;
  ls.rw_stride = 0;
  for (int l = 0; l < kMaxLanes; ++l) ls.n_eff[l] = (double)ds->n_global;
  return ls;
}

// grad -> reduce (-> all-reduce) for B lanes on ONE pass over X:
// g_l = X^T W_l (X z_l - y) / n_eff_l in ds->g + l*(ld+16), loss_l in g_l[ld].
static int enqueue_gradient(slm_dataset* ds, const LaneSetup& ls, const double* y, const int* done,
                            hipEvent_t ev_start, hipEvent_t ev_stop, int64_t n_rows = 0) {
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
  const bool ring = sk->rowdot != nullptr && (env ? env[0] == '1' : B <= ROWDOT_LANES);
  a.lane0 = 0;
  if ((!ring || sk->rowdot == nullptr) && ds->XT && ds->XT_ready) {
    a.XT = ds->XT;
    hipLaunchKernelGGL(rowdot_mfma_kernel, dim3(nblk), dim3(XZ_WAVES * 64), 0, s, a);
  } else if (sk->rowdot != nullptr) {
    hipLaunchKernelGGL(sk->rowdot, dim3(nblk, (B + ROWDOT_LANES - 1) / ROWDOT_LANES), dim3(sk->W * 64), 0, s, a);
  }
  // (no ring variant and no column-major copy: split_usable() keeps such datasets off the split pass)
}

// The split pass needs a kernel for the residuals that come from X: a ring variant, or the column-major copy.
static int ensure_xt(slm_dataset* ds);
static bool split_usable(slm_dataset* ds) {
  if (!ds->sk) return false;
  if (ds->sk->rowdot != nullptr) return true;
  if (ensure_xt(ds) != SLM_OK) return false;
  return ds->XT != nullptr;
}

static int enqueue_gradient_split(slm_dataset* ds, const LaneSetup& ls, const double* y, const int* done,
                                  const PathCtl* ctl, const WsArgs* wa, hipEvent_t ev_start,
                                  hipEvent_t ev_stop, int64_t n_rows = 0) {
  hipStream_t s = ds->eng->stream;
  const SplitKernel* sk = ds->sk;
  const int nblk = ds->split_nblk;
  if (!ds->R) {
    SLM_TRY(dalloc(&ds->R, (size_t)ds->n * SPLIT_RSTRIDE));
    HIP_TRY(hipMemsetAsync(ds->R, 0, sizeof(double) * (size_t)ds->n * SPLIT_RSTRIDE, s));
  }
  SplitArgs a;
  memset(&a, 0, sizeof(a));
  a.X = ds->X; a.y = y; a.rw = ls.rw; a.rw_stride = ls.rw_stride; a.z = ds->z; a.R = ds->R;
  a.partial = ds->partial; a.loss_partial = ds->loss_partial; a.done = done; a.ctl = ctl;
  if (wa) { a.XW = wa->XW; a.idx = wa->idx; a.ws = wa->ws; }
  const int64_t nr = n_rows > 0 ? n_rows : ds->n;
  a.n = nr; a.ld = ds->ld; a.rows_base = nr / nblk; a.rows_rem = nr % nblk;
  a.p2 = (int)(ds->ld / 2);
  a.n_lanes = ls.B;
  launch_rowdot(ds, sk, nblk, ls.B, a, s);
  if (wa && ctl) {  // residuals from the gathered columns: matrix cores (SLM_RESID_VEC=1: a row per thread)
    const char* env = getenv("SLM_RESID_VEC");
    if (env && env[0] == '1') hipLaunchKernelGGL(sk->resid, dim3(nblk), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(resid_mfma_kernel, dim3(nblk), dim3(RM_WAVES * 64), 0, s, a);
  }
  // (SLM_FLAG_PROFILE brackets the kernel that streams X, the one the roofline is quoted on)
  if (ev_start) HIP_TRY(hipEventRecord(ev_start, s));
  const int xblk = launch_xtr(ds->eng->cus, a, s);
  if (ev_stop) HIP_TRY(hipEventRecord(ev_stop, s));
  ReduceArgs ra;
  ra.partial = ds->partial;
  ra.loss_partial = ds->loss_partial;
  ra.g = ds->g;
  ra.done = done;
  ra.nblk = xblk;
  ra.nblk_loss = nblk;
  ra.n_lanes = SPLIT_LANES;  // partial rows are laid out for all lane slots of the split pass
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
  hipLaunchKernelGGL(cov_pack_kernel, dim3((unsigned)((ld * SPLIT_RSTRIDE + 255) / 256)), dim3(256), 0, s, ds->z, ld, B, ds->cov_Z, done);
  if (ev_start) HIP_TRY(hipEventRecord(ev_start, s));
  bool seen[SLM_MAX_LANES] = {};
  for (int l = 0; l < B; ++l) {
    if (seen[l]) continue;
    uint32_t mask = 0;
    for (int m = l; m < B; ++m)
      if (entry_of[m] == entry_of[l]) {
        mask |= 1u << m;
        seen[m] = true;
      }
    const slm_dataset::CovEntry& e = ds->cov[(size_t)entry_of[l]];
    SplitArgs a;
    memset(&a, 0, sizeof(a));
    a.X = e.G; a.R = ds->cov_Z; a.partial = ds->partial; a.done = done;
    a.n = ld; a.ld = ld; a.p2 = (int)(ld / 2); a.n_lanes = B;
    // (points the model solver produced are zero outside the working set: only its rows of G are read then)
    if (ctl && wa && wa->ws && !getenv("SLM_COV_ALL_ROWS")) { a.ctl = ctl; a.ws = wa->ws; a.idx = wa->idx; }
    const int xblk = launch_cov_gz(ds->eng->cus, a, s);
    CovFinishArgs f;
    f.partial = ds->partial; f.c = e.c; f.z = ds->z; f.g = ds->g; f.done = done; f.nblk = xblk; f.ld = ld;
    f.lane_mask = mask; f.yy = e.yy;
    hipLaunchKernelGGL(cov_reduce_kernel, dim3((unsigned)(ld / 16), (unsigned)B), dim3(256), 0, s, f);
    hipLaunchKernelGGL(cov_loss_kernel, dim3((unsigned)B), dim3(256), 0, s, f);
  }
  if (ev_stop) HIP_TRY(hipEventRecord(ev_stop, s));
  return SLM_OK;
}

// fingerprints of row-weight vectors already on the device -> host (one small copy, one wait)
static int cov_fingerprints(slm_dataset* ds, const double* const* w, int count, double* out /* [2 * count] */) {
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

static int cov_find(const slm_dataset* ds, double fp1, double fp2, double n_eff) {
  for (size_t i = 0; i < ds->cov.size(); ++i)
    if (ds->cov[i].fp1 == fp1 && ds->cov[i].fp2 == fp2 && ds->cov[i].n_eff == n_eff) return (int)i;
  return -1;
}

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

static int check_launch() {
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
  ds->XT_ready = false;  // X changed in place: the column-major copy is rebuilt on next use
  return SLM_OK;
}

// ------------------------------------------------------------------------------------------------
// single gradient evaluation (tests, alpha_max, roofline probe)
// ------------------------------------------------------------------------------------------------
static int ensure_xt(slm_dataset* ds);

extern "C" int slm_gradient(slm_dataset* ds, const double* z, double* g_out, double* loss_out,
                            int32_t reps, double* ms_out) {
  if (!ds) return fail(SLM_ERR_BAD_ARG, "dataset is NULL");
  HIP_TRY(hipSetDevice(ds->eng->device));
  hipStream_t s = ds->eng->stream;
  const LaneSetup ls = default_lanes(ds, 1);
  HIP_TRY(hipMemsetAsync(ds->z, 0, sizeof(double) * ds->ld, s));
  if (z) HIP_TRY(hipMemcpyAsync(ds->z, z, sizeof(double) * ds->p, hipMemcpyHostToDevice, s));
  const char* split_env = getenv("SLM_GRAD_SPLIT");  // tests: take the split pass (residuals from X)
  const bool use_split = split_env && split_env[0] == '1' && split_usable(ds);
  if (use_split) SLM_TRY(ensure_xt(ds));  // (so that tests and probes reach rowdot_mfma_kernel; optional copy)
  if (use_split) SLM_TRY(enqueue_gradient_split(ds, ls, ds->y, nullptr, nullptr, nullptr, nullptr, nullptr));
  else SLM_TRY(enqueue_gradient(ds, ls, ds->y, nullptr, nullptr, nullptr));
  SLM_TRY(check_launch());
  HIP_TRY(hipStreamSynchronize(s));
  if (g_out) HIP_TRY(hipMemcpy(g_out, ds->g, sizeof(double) * ds->p, hipMemcpyDeviceToHost));
  if (loss_out) HIP_TRY(hipMemcpy(loss_out, ds->g + ds->ld, sizeof(double), hipMemcpyDeviceToHost));
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
    if (!ds->rw_lanes) SLM_TRY(dalloc(&ds->rw_lanes, (size_t)kMaxLanes * n));
    HIP_TRY(hipMemcpyAsync(ds->rw_lanes, row_weight, sizeof(double) * n, hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));
    ls.rw = ds->rw_lanes;
    ls.rw_stride = 0;  // every lane reads the same mask
  }
  for (int l = 0; l < kMaxLanes; ++l) ls.n_eff[l] = 0.5;  // loss_scale = 1/(2 n_eff) = 1  =>  g[ld] = SSE
  int maxB = kMaxLanes;
  while (maxB > 1 && !ds->gk[maxB - 1]) --maxB;
  std::vector<double> losses(kMaxLanes);
  for (int32_t k0 = 0; k0 < m; k0 += maxB) {
    const int B = std::min<int32_t>(maxB, m - k0);  // kernel variants exist for every B <= maxB
    ls.B = B;
    HIP_TRY(hipMemsetAsync(ds->z, 0, sizeof(double) * kMaxLanes * ld, s));
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
static int ensure_xt(slm_dataset* ds) {
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
    if (!ds->rw_lanes) SLM_TRY(dalloc(&ds->rw_lanes, (size_t)kMaxLanes * n));
    HIP_TRY(hipMemcpyAsync(ds->rw_lanes, row_weight, sizeof(double) * n, hipMemcpyHostToDevice, s));
  }
  // scratch shared with the working set (every solve re-initialises that state)
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
// set is on from the start
static int max_lanes_for(slm_dataset* ds, uint32_t flags) {
  if (small_ok(ds, flags)) return kMaxLanes;  // a workgroup per lane
  if ((ws_policy(ds, flags) == 2 || (double)ds->n * (double)ds->ld >= 67108864.0) && split_usable(ds)) return SPLIT_LANES;
  int B = kMaxLanes;
  while (B > 1 && !ds->gk[B - 1]) --B;
  return B;
}

static int solve_core(slm_dataset* ds, const slm_lane* lanes, int32_t n_lanes, const slm_solve_opts* opts,
                      slm_solve_stats* stats, bool shared_path);

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
                      slm_solve_stats* stats, bool shared_path) {
  if (!ds || !lanes) return fail(SLM_ERR_BAD_ARG, "NULL argument");
  if (n_lanes < 1 || n_lanes > kMaxLanes)
    return fail(SLM_ERR_BAD_ARG, "n_lanes must be in [1, %d] (got %d)", kMaxLanes, n_lanes);
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
  const bool want_split = (ws_policy(ds, opts ? opts->flags : 0u) == 2 && !wide) ? (big_x || !ds->gk[B - 1]) : (big_x && !ds->gk[B - 1]);
  // (covariance passes are a form of the split pass: the flag asks for it whatever the size, where Grams exist)
  const bool want_cov = opts && (opts->flags & SLM_FLAG_COVARIANCE) && !ds->cov.empty() && !row_sharded(ds);
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
  if (!split && !ds->gk[B - 1] && !small_ok(ds, opts ? opts->flags : 0u))
    return fail(SLM_ERR_UNSUPPORTED, "no %d-lane gradient kernel covers p = %lld", B, (long long)ds->p);
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
  double wmax[SLM_MAX_LANES];  // largest row weight of each lane (< 0: unknown)
  for (int l = 0; l < kMaxLanes; ++l) wmax[l] = ds->rw ? ds->rw_max : 1.0;
  if (any_rw) {
    if (!ds->rw_lanes) SLM_TRY(dalloc(&ds->rw_lanes, (size_t)kMaxLanes * n));
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
          HIP_TRY(hipMemcpyAsync(dst, ds->rw_lanes + (size_t)same * n, sizeof(double) * n, hipMemcpyDeviceToDevice, s));
          continue;
        }
        double top = 0.0;
        for (int64_t i = 0; i < n; ++i) {
          if (!(w[i] >= 0.0) || !std::isfinite(w[i]))
            return fail(SLM_ERR_BAD_ARG, "lane %d: row_weight[%lld] is negative or not finite", l, (long long)i);
          top = std::max(top, w[i]);
        }
        wmax[l] = top;
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
  int cov_entry[SLM_MAX_LANES] = {};
  bool cov_on = false;
  if (want_cov && split && !small) {
    const double* wdev[SLM_MAX_LANES];
    int uniq_of[SLM_MAX_LANES], first_lane[SLM_MAX_LANES], nu = 0;
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
      if (!ds->cov_Z) SLM_TRY(dalloc(&ds->cov_Z, (size_t)ld * SPLIT_RSTRIDE));
    }
  }
  // ---- Lipschitz constants -----------------------------------------------------------------------
  double L[SLM_MAX_LANES];
  double lipschitz_ms = 0.0;
  bool L_on_device = false;  // the estimate stays on the device (no host round trip before the first pass)
  double L_factor[SLM_MAX_LANES];  // ... and lane l uses L_factor[l] times it
  for (int l = 0; l < kMaxLanes; ++l) L_factor[l] = 1.0;
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
        SLM_TRY(power_iteration(ds, plain, nullptr, sketch_iters(), sketch_rows(n)));
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
        SLM_TRY(power_iteration(ds, default_lanes(ds, 1), nullptr, sketch_iters(), sketch_rows(n)));
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
  PathCtl* h = ds->h_stage;  // (lives as long as the dataset: the upload below is asynchronous)
  memset(h, 0, sizeof(ds->h_stage));
  SetupArgs su;
  memset(&su, 0, sizeof(su));
  su.beta = ds->beta; su.z = ds->z; su.zprev = ds->zprev; su.gprev = ds->gprev;
  su.a0 = ds->a0; su.b0 = ds->b0; su.d0 = ds->d0;
  // (small solves keep their per-point records inside the control block: one blocking copy less at the end)
  const bool infos_in_snap = total_points <= kSnapInfos;
  slm_point_info* d_infos = infos_in_snap ? ds->dctl->infos : ds->infos;
  su.infos = reinterpret_cast<unsigned char*>(d_infos);
  su.infos_bytes = (int64_t)(sizeof(slm_point_info) * total_points);
  static_assert(sizeof(slm_point_info) % 8 == 0, "infos are zeroed in 8-byte words");
  su.ld = ld; su.p = p; su.G = G; su.n_lanes = B; su.max_lanes = kMaxLanes;
  if (!ds->h_vec) {
    hipError_t eh = hipHostMalloc((void**)&ds->h_vec, sizeof(double) * 4 * (size_t)kMaxLanes * (size_t)ld, hipHostMallocDefault);
    if (eh != hipSuccess) return fail(SLM_ERR_OOM, "hipHostMalloc: %s", hipGetErrorString(eh));
    memset(ds->h_vec, 0, sizeof(double) * 4 * (size_t)kMaxLanes * (size_t)ld);
  }
  if (total_points > ds->h_pts_cap) {
    if (ds->h_pts) (void)hipHostFree(ds->h_pts);
    ds->h_pts = nullptr;
    ds->h_pts_cap = 0;
    hipError_t eh = hipHostMalloc((void**)&ds->h_pts, sizeof(slm_path_point) * (size_t)total_points, hipHostMallocDefault);
    if (eh != hipSuccess) return fail(SLM_ERR_OOM, "hipHostMalloc: %s", hipGetErrorString(eh));
    ds->h_pts_cap = total_points;
  }
  int up_lo[4] = {kMaxLanes, kMaxLanes, kMaxLanes, kMaxLanes}, up_hi[4] = {-1, -1, -1, -1};  // lanes that bring a, b, d, beta0
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
        memcpy(ds->h_vec + ((size_t)v * kMaxLanes + l) * ld, src[v], sizeof(double) * cnt[v]);
        up_lo[v] = std::min(up_lo[v], l);
        up_hi[v] = std::max(up_hi[v], l);
      }
    }
    memcpy(ds->h_pts + off, ln.points, sizeof(slm_path_point) * (size_t)ln.n_points);
    if (ln.beta0) {
      for (int64_t j = 0; j < p; ++j)
        if (!std::isfinite(ln.beta0[j])) return fail(SLM_ERR_BAD_ARG, "beta0[%lld] is not finite", (long long)j);
      memcpy(ds->h_vec + ((size_t)3 * kMaxLanes + l) * ld, ln.beta0, sizeof(double) * p);
      up_lo[3] = std::min(up_lo[3], l);
      up_hi[3] = std::max(up_hi[3], l);
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
      h[l].point = l;
      h[l].pt_lo = l;
      h[l].n_points = (int32_t)total_points;
      h[l].stride = B;
      // The points beyond the last full band (two of a 50-point path on sixteen lanes) go to the LAST lanes -- the
      // ones that have just solved their neighbours -- not to the first, which would reach them from sixteen points
      // up the path: there the features of the last decade of alpha cannot be told yet, the first verification
      // misses and a large append follows (0.33 ms on the headline path).
      const int64_t rem = total_points % B, n_reg = total_points - rem;
      if (rem > 0 && n_reg >= 2 * (int64_t)B && !getenv("SLM_NO_TAIL_BAND")) {
        h[l].n_points = (int32_t)n_reg;
        if (l >= B - rem) h[l].tail_pt = (int32_t)(n_reg + (l - (B - rem)));
      }
    } else if (shared_path) {  // global indices: [off, off + n_points)
      h[l].point = (int32_t)off;
      h[l].pt_lo = (int32_t)off;
      h[l].n_points = (int32_t)(off + ln.n_points);
    }
    h[l].zzero = ln.beta0 ? 0 : 1;
    h[l].mode = (o.flags & SLM_FLAG_FISTA_ONLY) ? 0 : 1;
    h[l].ak = 1.25 * L[l];  // a slightly short first step; the scheme measures its own curvature after it
    h[l].Lhat = 0.5 * L[l];  // a sure lower bound of lambda_max for the residual scaling
    off += ln.n_points;
  }
  {  // the staged rows, first to last lane that brings any (rows in between are filled by solve_setup_kernel afterwards)
    double* dev[4] = {ds->a0, ds->b0, ds->d0, ds->beta};
    for (int v = 0; v < 4; ++v)
      if (up_hi[v] >= 0)
        HIP_TRY(hipMemcpyAsync(dev[v] + (size_t)up_lo[v] * ld, ds->h_vec + ((size_t)v * kMaxLanes + up_lo[v]) * ld,
                               sizeof(double) * (size_t)(up_hi[v] - up_lo[v] + 1) * ld, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(ds->pts, ds->h_pts, sizeof(slm_path_point) * (size_t)total_points, hipMemcpyHostToDevice, s));
  }
  hipLaunchKernelGGL(solve_setup_kernel, dim3(128), dim3(256), 0, s, su);
  HIP_TRY(hipMemcpyAsync(ds->ctl, h, sizeof(PathCtl) * B, hipMemcpyHostToDevice, s));
  static_assert(offsetof(DevCtl, lane) >= offsetof(DevCtl, ws) + sizeof(WsCtl) && offsetof(DevCtl, g) == 0, "g, ws, lane");
  HIP_TRY(hipMemsetAsync(ds->dctl, 0, offsetof(DevCtl, lane), s));  // stop words and working-set counters
  if (L_on_device) {  // the power steps are still in flight: their result goes into the control blocks on the device
    SeedArgs sa;
    sa.ctl = ds->ctl; sa.lambda = ds->lambda; sa.n_lanes = B; sa.margin = 1.08;
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
    for (int l = 0; l < kMaxLanes; ++l) sm.inv_n[l] = 1.0 / (ls.n_eff[l] > 0 ? ls.n_eff[l] : (double)ds->n_global);
    sm.max_iters = (int)std::min<int64_t>((int64_t)o.max_iter, 1500);  // (products per point; then the general path takes over)
    sm.cold = (o.flags & SLM_FLAG_COLD_START) ? 1 : 0;
    // LDS: the Gram matrix, three vectors, and the rest as the stage of the rows while the matrix is built
    const size_t fixed = sizeof(double) * ((size_t)p * p + 3 * (size_t)p);
    const size_t lds = (size_t)SM_LDS_BYTES;
    sm.stage_doubles = (int)((lds - fixed - 64) / sizeof(double));
    SLM_TRY(allow_big_lds((const void*)small_solve_kernel, eng->device));
    hipLaunchKernelGGL(small_solve_kernel, dim3(B), dim3(SM_THREADS), lds, s, sm);
    SLM_TRY(check_launch());
    if (infos_in_snap) HIP_TRY(hipMemcpyAsync(&ds->hctl[0].c, ds->dctl, sizeof(DevCtl), hipMemcpyDeviceToHost, s));
    else HIP_TRY(hipMemcpyAsync(&ds->hctl[0].c, ds->dctl, offsetof(DevCtl, infos), hipMemcpyDeviceToHost, s));
    SLM_TRY(enqueue_result_copies());
    HIP_TRY(hipStreamSynchronize(s));
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
    int set_of[SLM_MAX_LANES] = {}, set_lane[SLM_MAX_LANES] = {};
    int n_sets = 0;
    for (int l = 0; l < B; ++l) {
      int found = -1;
      for (int m = 0; m < l && found < 0; ++m)
        if (lanes[m].row_weight == lanes[l].row_weight && lanes[m].n_eff == lanes[l].n_eff) found = set_of[m];
      if (found < 0) {
        found = n_sets;
        set_lane[n_sets++] = l;
      }
      set_of[l] = found;
    }
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
    hipLaunchKernelGGL(ws_ctl_init_kernel, dim3(1), dim3(64), 0, s, ds->ws_ctl, 24);
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
    wa.append_max = interleave ? 48 : 16;  // interleaved lanes need the next band of the path at once
    wa.k_init = ds->singleton ? 112 : 256;  // groups bring their features in blocks (config 3: 12.4 vs 29.6 ms per path)
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
    return SLM_OK;
  };
  // no memory for the working-set buffers: the plain iteration still works (unless this solve runs
  // more lanes than the fused kernels serve, which only the split pass can do)
  auto ws_release = [&]() {
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
  // everything that follows the gradient of one pass
  // (in two halves: behind the pass a solve is expected to end with, the second half waits for the verdict)
  auto enqueue_tail = [&]() {
    launch_tail(ta, s);
    if (shared_path && !interleave) hipLaunchKernelGGL(steal_kernel, dim3(1), dim3(256), 0, s, ta);
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
      if (shared_path && interleave) mine = (h[l].n_points - l + B - 1) / B + (h[l].tail_pt >= 0 ? 1 : 0);
      most = std::max<int64_t>(most, mine);
    }
    expected = 1 + most;
  }
  int final_slot = 0;          // the snapshot in which the host saw `done`
  bool deferred = false;       // the refinement behind the last queued pass has not been queued yet
  while (!done) {
    {
      // (a solve with an expected end queues all of its passes at once: launches behind the device-side stop flag return
      //  at once, and every snapshot in between -- a copy, an event, 6 us of idle stream around them -- told the host
      //  nothing it acts on)
      const int this_chunk = expected <= 0 ? chunk : (enq < expected ? (int)std::min<int64_t>(64, expected - enq) : 1);
      for (int i = 0; i < this_chunk; ++i) {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (profile && enq % kProfStride == 0) {  // sampled: an event pair costs ~8 us of stream time
          const int64_t slot_id = enq / kProfStride;
          while ((int64_t)ds->prof.size() < 2 * (slot_id + 1)) {
            hipEvent_t ev;
            HIP_TRY(hipEventCreate(&ev));
            ds->prof.push_back(ev);
          }
          e0 = ds->prof[2 * slot_id];
          e1 = ds->prof[2 * slot_id + 1];
        }
        SLM_TRY(enqueue_pass_gradient(e0, e1));
        enqueue_tail();
        ++enq;
        // behind the pass the solve is expected to end with, the six launches of the refinement would only find
        // out that there is nothing left to refine (30 us): they follow once the snapshot says otherwise
        deferred = expected > 0 && enq == expected && !sharded;
        if (!deferred) enqueue_refinement();
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
    const bool at_end = expected > 0 && enq >= expected && enq < expected + 4;
    HIP_TRY(hipEventRecord(ds->ev[slot], s));
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
      } else if (deferred) {  // the solve goes on: what was held back, then the next pass
        enqueue_refinement();
        deferred = false;
      }
    } else if (pending[other]) {
      HIP_TRY(hipEventSynchronize(ds->ev[other]));
      pending[other] = false;
      if (ds->hctl[other].c.g.done) {
        done = true;
        final_slot = other;
      }
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
  SLM_TRY(enqueue_result_copies());
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
    stats->grad_launches = passes;  // launches that did work (every launch serves all lanes)
    stats->grad_ms_total = 0.0;
    stats->grad_timed = 0;
    if (profile) {
      double tot = 0.0;
      int64_t cnt = 0;
      // iterations 0, kProfStride, 2 kProfStride, ... below `passes` did real work and were timed
      for (int64_t k = 0; k * kProfStride < passes && 2 * k + 1 < (int64_t)ds->prof.size(); ++k) {
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
        fprintf(stderr, "[slm] working set: %d model-solver iterations over %d refinements, %d direct steps (%d refused, %d of them not positive definite), K = %d\n",
                wc.inner_iters, wc.refined, wc.newton_steps, wc.newton_fails, wc.newton_nopd, wc.K);
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
  return SLM_OK;
}

// ------------------------------------------------------------------------------------------------
// covariance passes: the Gram of a row set
// ------------------------------------------------------------------------------------------------
// C = A^T A for the row-major rows x ld block A (cov_syrk_kernel: the square with its mirror, for one row set at a time;
// the folds of a K-fold split go through cov_syrk_packed_kernel, cov_folds_begin)
static int cov_gram(slm_dataset* ds, const double* A, int64_t rows, double* C) {
  slm_engine* eng = ds->eng;
  hipStream_t s = eng->stream;
  const int64_t ld = ds->ld;
  if (rows < 1) {
    HIP_TRY(hipMemsetAsync(C, 0, sizeof(double) * (size_t)ld * ld, s));
    return SLM_OK;
  }
  int side = cov_tile_for(ld, eng->cus);
  if (const char* e = getenv("SLM_COV_TILE")) side = atoi(e) == 3 ? 3 : 4;  // (A/B runs, tests: 96 or 128 columns per workgroup)
  const int nt = (int)((ld + 32 * side - 1) / (32 * side));
  const dim3 grid((unsigned)(nt * (nt + 1) / 2));
  if (side == 3) hipLaunchKernelGGL(cov_syrk_kernel<3>, grid, dim3(256), 0, s, A, rows, ld, C);
  else hipLaunchKernelGGL(cov_syrk_kernel<4>, grid, dim3(256), 0, s, A, rows, ld, C);
  return check_launch();
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

static void cov_pending_drop(slm_dataset* ds) {
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

extern "C" int slm_dataset_max_lanes(slm_dataset* ds, uint32_t flags, int32_t* max_lanes_out) {
  if (!ds || !max_lanes_out) return fail(SLM_ERR_BAD_ARG, "NULL argument");
  *max_lanes_out = max_lanes_for(ds, flags);
  return SLM_OK;
}

extern "C" int slm_solve_lanes(slm_dataset* ds, const slm_lane* lanes, int32_t n_lanes,
                               const slm_solve_opts* opts, slm_solve_stats* stats) {
  return solve_core(ds, lanes, n_lanes, opts, stats, false);
}

extern "C" int slm_solve_path_lanes(slm_dataset* ds, const slm_penalty* pen, const slm_path_point* points,
                                    int32_t n_points, int32_t n_lanes, const slm_solve_opts* opts,
                                    const double* beta0, double* betas_out, double* group_norms_out,
                                    slm_point_info* infos, slm_solve_stats* stats) {
  if (!ds || !points || !betas_out) return fail(SLM_ERR_BAD_ARG, "NULL argument");
  if (n_points <= 0) return fail(SLM_ERR_BAD_ARG, "n_points must be positive");
  int B = std::max(1, std::min<int>(std::min<int>(n_lanes, kMaxLanes), n_points));
  B = std::min(B, max_lanes_for(ds, opts ? opts->flags : 0u));  // no kernel variant for (p, B): fewer lanes
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

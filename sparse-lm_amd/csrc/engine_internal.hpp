// Shared by the translation units of the host side (engine.hip: handles, memory, communicators, datasets; engine_solve.hip:
// launches and the solve loop; engine_cov.hip: the Grams of covariance passes): types, error macros, prototypes.
// The kernels live in the *_kernels.hpp headers; non-template kernels are `static`, so every unit may include every header
// and carries code for exactly the kernels it launches.
#pragma once
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
#include "host_logic.hpp"
#include "data_kernels.hpp"
#include "grad_kernel.hpp"
#include "tail_kernels.hpp"
#include "ws_kernels.hpp"
#include "split_kernels.hpp"
#include "light_kernels.hpp"
#include "small_kernels.hpp"
#include "small_split_kernels.hpp"
#include "cov_kernels.hpp"
#include "mg_kernels.hpp"

using namespace slm;


// ------------------------------------------------------------------------------------------------
// errors (engine.hip)
// ------------------------------------------------------------------------------------------------
int fail(int code, const char* fmt, ...);
// the SLM_* environment knobs, read once per process (engine.hip; slm_reload_knobs reads them again)
const slm_host::Knobs& knobs();

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
extern RcclApi g_rccl;
static const int kNcclFloat64 = 8;  // ncclDouble
static const int kNcclSum = 0;      // ncclSum
int load_rccl();

#define RCCL_TRY(expr)                                                                      \
  do {                                                                                      \
    int e__ = (expr);                                                                       \
    if (e__ != 0)                                                                           \
      return fail(SLM_ERR_COMM, "%s failed: %s", #expr,                                     \
                  g_rccl.GetErrorString ? g_rccl.GetErrorString(e__) : "rccl error");       \
  } while (0)

// ------------------------------------------------------------------------------------------------
// gradient kernel tables and launches (engine_solve.hip)
// ------------------------------------------------------------------------------------------------
struct GradKernel {
  int W, C, R, B;
  int D;  // 0: rows wait in VGPRs (grad_fused_kernel); > 0: LDS ring, D rows in flight (grad_ring_kernel)
  void (*fn)(GradArgs);
};
struct SplitKernel {
  int W, C, B, D;
  void (*rowdot)(SplitArgs);
  void (*resid)(SplitArgs);
};
static const int kMaxTailE = 64;  // the tail kernels cover p <= 1024 * 64
static const int64_t kMaxChunks = 64 * 8 * 10;  // largest row the fused kernel covers (p <= 10240)
const SplitKernel* pick_split_kernel(int64_t p2);
int xtr_max_row_blocks(int cus, int64_t ld);
int launch_xtr(int cus, SplitArgs& a, hipStream_t s, bool sample = false);
const GradKernel* pick_grad_kernel(int64_t p2, int B);

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

int all_reduce_sum(slm_engine* eng, double* buf, size_t count, hipStream_t s = nullptr);

static const int kMaxLanes = SLM_MAX_LANES;
static const int kMaxCells = SLM_MAX_CELLS;  // lanes of a call the on-chip solver takes (slm_dataset::lane_cap)

// Control words shared by all lanes of a solve.
struct GlobalCtl {
  int32_t done;        // every lane finished, or abort
  int32_t lanes_done;
  int32_t hard;        // most passes any lane has spent on one path point so far (TailArgs::gdone[2])
  int32_t done_local;  // row-sharded mode: this rank has finished (TailArgs::gdone[3]); `done` follows once all have
  int32_t diverged;    // row-sharded mode: the ranks' control blocks differ (TailArgs::gdone[4]); ends the solve
  int32_t pad_[3];
};
static const int kWsMaxBuilds = 24;  // fresh selections of the working set per solve (WsCtl::max_builds; appends: eight times as many)
static const int kWsLateIters = 12;  // passes on one point after which a small problem gets the working set

static const int kSnapInfos = 64;
// Everything the host reads back about a solve sits in ONE device block -- the stop word, the lanes' control
// blocks and the working set's counters -- so a poll is one copy, and the snapshot in which the host sees
// `done` already holds the final statistics (after `done` every queued kernel returns at once: nothing in the
// block changes any more).  Reading them one by one at the end cost four blocking copies, ~0.1 ms of host
// round trips on a 5 ms path.
struct DevCtl {
  GlobalCtl g;
  MgCtl mg;  // model-Gram rounds (mg_kernels.hpp); cleared with g at the start of every solve
  LightCtl lt;  // certified partial passes (light_kernels.hpp); likewise
  WsCtl ws;  // (next to them: one fill clears all three at the start of a solve; a working set taken over keeps its block)
  PathCtl lane[SLM_MAX_CELLS];
  slm_point_info infos[kSnapInfos];  // the per-point records of solves of up to kSnapInfos points ride along
};
struct HostCtl {  // pinned snapshot the host polls
  DevCtl c;
};

void pool_free(void* p);

constexpr size_t kSmallOutDoubles = 32768;  // 256 KiB of staged coefficients per dataset (on-chip calls)

struct slm_dataset {
  slm_engine* eng = nullptr;
  int64_t n = 0, p = 0, ld = 0, n_global = 0;
  // On an engine with a communicator a dataset is a row block of one tall matrix (every pass all-reduces its gradients),
  // unless it is marked as a REPLICA (slm_dataset_set_replicated: grid mode -- every rank holds all rows and solves its own
  // lanes; the communicator then only carries the folds' Grams, each rank building the part of an n_ranks-th of the rows)
  bool replicated = false;
  // lanes the per-lane buffers are laid out for: SLM_MAX_LANES, or SLM_MAX_CELLS on datasets the on-chip solver can take
  // (p <= 128, n * ld <= 2^17: the buffers are tiny there, and a small grid search has more cells than sixteen)
  int lane_cap = SLM_MAX_LANES;
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
  int32_t* ws_owner = nullptr;  // [kMaxLanes][ws_nblk <= 512] ws_block_owner_kernel
  double* ws_nt = nullptr;  // [kMaxLanes][NT_SCRATCH] factors of the model solver's direct steps
  double *sse_Z = nullptr, *sse_part = nullptr;  // slm_eval_sse_sparse: coefficient block, partial sums
  size_t sse_cap = 0;
  PathCtl* h_stage = nullptr;  // [SLM_MAX_CELLS] page-locked, device-visible staging of the control blocks of the solve in
                               // flight: solve_begin_kernel fetches them itself (allocated on first use)
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
  double* cov_partial = nullptr;  // partial sums of a covariance pass over more row sets than `partial` holds
  int cov_partial_sets = 0;
  size_t partial_elems = 0;   // doubles in `partial`
  double* cov_fp = nullptr;   // [2 * kMaxLanes + 2] fingerprints / scalars on their way to the host
  double* split_state = nullptr;  // [3 ld + 2 + record] slm_solve_standardized_sgl: gamma, u, rho, valid; outputs
  double* h_split = nullptr;      // its page-locked staging: a, b, warm start in; coefficients, group norms, record out
  double* stop_words = nullptr;  // [STOP_WORDS] row-sharded mode: the vector the ranks all-reduce after every pass
  double* XT = nullptr;  // column-major copy of X in tiles of 32 rows (tile_columns_kernel), built on first use
  bool XT_ready = false, XT_failed = false;
  // certified partial passes (light_kernels.hpp): the column norms ||X_j|| / sqrt(n_global), built with the copy; the moves'
  // residual changes, their block sums, the borderline columns and their partial products -- allocated on first use
  double* colnorm = nullptr;
  bool colnorm_ready = false;
  double *lt_dR = nullptr, *lt_d2 = nullptr, *lt_part = nullptr;
  int32_t *lt_cols = nullptr, *lt_stamp = nullptr;
  // model Gram (mg_kernels.hpp, engine_mg.hip): G~ ~ X^T W X / n_global of the dataset's own rows and weights from an fp16
  // product, built when a solve's lanes outgrow the working set and kept for the later solves of the dataset
  struct MgEntry {           // one per row set, found again like the Grams of covariance passes: by the fingerprint of its row weights
    float* G = nullptr;      // [ld][ld], fp32: the product it came from is good to 1e-4 -- half the bytes of every step of a round
    double fp1 = 0.0, fp2 = 0.0, n_eff = 0.0;
    bool own = false;        // the dataset's own rows and weights (no fingerprint: lanes that bring neither weights nor scaling)
  };
  std::vector<MgEntry> mg;   // oldest first; at most model_gram_cap() of them (mg_ensure: the oldest goes when a single build needs the
                             // room; a solve whose row sets do not fit beside the kept ones drops them all first: PathCall::mg_sets)
  double* mg_vec = nullptr;  // [5][kMaxLanes][ld]: iterate, evaluation point, the last point / model gradient of the inner iteration, the product
  double* mg_Z = nullptr;    // [halves][ld][16] the lanes' moves D = v - z0, lane-minor (the product's B operand)
  bool mg_failed = false;
  double mg_build_ms = 0.0;  // device time of the last build (events)
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
  double* h_small_out = nullptr;  // pinned, device-visible: the coefficients of an on-chip call are written here by the kernel
                                  // itself (kSmallOutDoubles; solve_core: a copy command into the caller's pageable array
                                  // costs more than the kernel's stores across the bus)
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
  bool sketch_valid = false;  // lambda[lane_cap] holds the sketch's estimate for the dataset's own rows and weights (solve_core)
  // Carried start (solve_core): what the last solve left on the device -- per lane the point zprev, its gradient gprev
  // and its loss -- described on the host so that the next solve can tell whether it starts exactly there.
  struct CarryLane {
    int64_t n_eff = 0;
    bool has_rw = false;
    double fp[2] = {0.0, 0.0};  // checksums of the lane's row weights (has_rw)
    double loss = 0.0;
  };
  bool carry_valid = false;
  int carry_lanes = 0;
  // ... and the working set it ended with (WsCtl, idx / pos, XW, the Grams): a carried start may take it over when the
  // lanes form the same row sets (ws_ctl_carry_kernel)
  bool ws_carry_valid = false;
  bool ws_carry_cov = false;   // its Grams were sub-matrices of the row sets' Grams (covariance passes)
  int ws_carry_sets = 0;
  int ws_carry_set_of[SLM_MAX_LANES] = {};
  CarryLane carry_lane[SLM_MAX_LANES];
  std::vector<double> carry_out;  // [carry_lanes][p]: the solutions the last solve reported
};

static inline bool row_sharded(const slm_dataset* ds) { return ds->eng->sharded() && !ds->replicated; }

// large device blocks are recycled (DevicePool, engine.hip)
hipError_t pool_malloc(void** out, size_t bytes);

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
// across the units
// ------------------------------------------------------------------------------------------------
struct LaneSetup {
  int B = 1;
  const double* rw = nullptr;  // device row weights handed to the kernel
  int64_t rw_stride = 0;
  double n_eff[SLM_MAX_CELLS] = {};
};
LaneSetup default_lanes(slm_dataset* ds, int B);
int enqueue_gradient(slm_dataset* ds, const LaneSetup& ls, const double* y, const int* done, hipEvent_t ev_start, hipEvent_t ev_stop,
                     int64_t n_rows = 0, const int* skip = nullptr);
int enqueue_gradient_split(slm_dataset* ds, const LaneSetup& ls, const double* y, const int* done, const PathCtl* ctl, const WsArgs* wa,
                           hipEvent_t ev_start, hipEvent_t ev_stop, int64_t n_rows = 0, bool unit_bracket = false, const int* skip = nullptr);
bool split_usable(slm_dataset* ds);
int ensure_xt(slm_dataset* ds);
int check_launch();
// (engine_solve.hip, used by the solve loop of engine_path.hip)
void launch_rowdot(slm_dataset* ds, const SplitKernel* sk, int nblk, int B, SplitArgs& a, hipStream_t s);
int enqueue_gradient_cov(slm_dataset* ds, int B, const int* entry_of, const int* done, hipEvent_t ev_start, hipEvent_t ev_stop,
                         const PathCtl* ctl = nullptr, const WsArgs* wa = nullptr);
void launch_tail(const TailArgs& ta, hipStream_t s);
int sketch_iters();
int64_t sketch_rows(int64_t n);
int power_iteration(slm_dataset* ds, const LaneSetup& ls_in, double* L_out /*[B]*/, int iters, int64_t n_rows = 0);
int estimate_lipschitz(slm_dataset* ds, double* L_out, int iters);
int allow_big_lds(const void* fn, int device);
static const int kProfStride = 3;  // SLM_FLAG_PROFILE times every 3rd gradient launch (a working-set path has ~5: two of them;
                                   // an event pair costs ~12 us of stream around the launch it brackets)
static const int kPowerItersSolve = 2;  // power steps for the seed L of a solve (see engine_solve.hip)
int cov_fingerprints(slm_dataset* ds, const double* const* w, int count, double* out /* [2 * count] */);
int cov_find(const slm_dataset* ds, double fp1, double fp2, double n_eff);
void cov_pending_drop(slm_dataset* ds);  // (engine_cov.hip)
int set_singleton_groups(slm_dataset* ds);
// model Gram (engine_mg.hip)
static const int kMgEntries = 8;
bool mg_possible(const slm_dataset* ds);
// the model Gram of a row set -- w: its row weights on the device (nullptr: the dataset's own), n_eff its scaling, (fp1, fp2) the
// fingerprint of w (ignored for the dataset's own) -- built if it is not there yet; entry_out: its index in ds->mg
int mg_ensure(slm_dataset* ds, const double* w, double n_eff, bool own, double fp1, double fp2, int* entry_out);
void mg_invalidate(slm_dataset* ds);    // X, the row weights or the scaling changed
void mg_free(slm_dataset* ds);
// one round: begin, inner_iters x (products, sums, step), finish.  entry_of_set[s]: the entry of the s-th row set of the call,
// set_of[l] the row set of lane l
int mg_enqueue_round(slm_dataset* ds, const TailArgs& ta, int n_lanes, int inner_iters, const int* done, int n_sets, const int* entry_of_set,
                     const int* set_of);

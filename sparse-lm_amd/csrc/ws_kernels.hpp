// Working-set refinement: Gram-assisted prediction of the next iterate between two passes over X.
//
// f(b) = 1/(2n) ||X b - y||^2_W is quadratic, so once the gradient g0 = grad f(z0) at ONE point is
// known (the fused pass just delivered it), the gradient at any b that differs from z0 only on a
// small column set W is exact without touching X again:
//
//     grad f(b)_W = g0_W + G_WW (b - z0)_W,        G_WW = X_W^T diag(w) X_W / n     (K x K, K <= 512)
//
// The kernels below (i) pick W on the device (every coefficient that is non-zero in any lane plus the
// features / groups whose gradient is within a factor theta of entering at the loosest penalty of
// the lane's range), (ii) gather the K columns into a compact n x K matrix, (iii) build G_WW with
// f64 MFMA (the one GEMM-shaped piece of this path: 2 n K^2 flop), and (iv) after every pass let one
// workgroup per lane minimise the penalised quadratic model over W (FISTA on K unknowns, the matrix
// stays in L2) and move the lane's next evaluation point there.  The next pass over X then VERIFIES
// that point with the true gradient under the unchanged stopping rule of fista_tail_kernel, so the
// meaning of `tol` and every reported solution are exactly those of the plain iteration; a point of
// a path costs one pass instead of three to seven.  Coordinates outside W are never touched: if a
// lane's iterate moves there (a feature outside W wants to enter) the refinement is skipped for
// that lane and W is rebuilt.
//
// Counterpart in the reference: none (cvxpy hands the whole problem to an interior-point solver,
// src/sparselm/model/_base.py:512-519); the idea is the covariance-update / working-set strategy
// of coordinate-descent Lasso solvers, restated for the proximal-gradient state machine.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tail_kernels.hpp"
#include "newton_kernels.hpp"

namespace slm {

constexpr int WS_KCAP = 512;        // capacity of the working set (leading dimension of G and XW)
constexpr int WS_KLDS = 112;        // up to this many columns the Gram lives in LDS during the model solve
constexpr int WS_TILES = WS_KCAP / 16; // 16x16 tiles per side of the Gram
constexpr int WS_THREADS = 1024;
constexpr int WS_INNER_MAX = 400;   // inner iterations per refinement
constexpr int WS_BB_ITERS = 14;         // spectral steps the model solver opens with
constexpr double WS_BB_MAX_STEP = 20.0; // longest of them, in steps of 1/L
constexpr double WS_INNER_TOL = 0.05;  // inner stop: residual <= WS_INNER_TOL * tol * ||b||
constexpr int WS_MAX_REPEATS = 6;   // refinements of one path point before the lane iterates plainly
// Direct step of the model solver (newton_kernels.hpp): when the proximal-gradient iteration on the model has
// not converged after WS_NEWTON_AFTER iterations -- or the curvature it measures along its own moves says
// the face is ill-conditioned (Rayleigh quotient below WS_NEWTON_RQ of lambda_max) -- a projected Newton step
// on the free coordinates (Cholesky solve, see `direct_step`) replaces further iterations; one prox-gradient
// step after it tests convergence, and the next direct step follows at once if that fails.
constexpr int WS_NEWTON_AFTER = 16;
constexpr int WS_NEWTON_MAX = 64;    // direct steps per refinement
constexpr int WS_AFTER_DIRECT = 40;  // iterations left to a refinement whose direct steps are used up: on a face that
                                     // needed them more would not converge either -- the next pass re-expands the model
                                     // at the point reached and the refinement resumes from there
constexpr double WS_NEWTON_RQ = 0.05;
constexpr int WS_NEWTON_REFUSALS = 4; // refused direct steps after which a refinement stops trying
constexpr double WS_NEWTON_ENTER = 0.3;
constexpr int WS_NEWTON_RESOLVE = 5; // solves per direct step while the free set is being made consistent
static_assert(NT_MAXT * NT_B == WS_KCAP, "the direct solve covers a full working set");

struct WsCtl {
  int32_t request;    // (re)build W at the next opportunity
  int32_t building;   // select changed W in this pass: gather / gram / reduce must run
  int32_t valid;      // G matches idx
  int32_t K;          // columns in use, padded to a multiple of 16 (padding entries have idx = -1)
  int32_t Kreal;      // columns in use without the padding
  int32_t k_new;      // first position whose column / Gram rows this build has to produce (0 = all)
  int32_t builds;     // full selections built
  int32_t appends;    // builds that only appended columns to the existing W
  int32_t max_builds;
  int32_t disabled;   // the non-zero coefficients alone kept exceeding WS_KCAP
  int32_t misses;     // times a lane's plain step left W
  int32_t refined;    // refinements applied (all lanes)
  int32_t counter;    // last-workgroup detection of the reduce kernel
  int32_t stale;      // a lane left W and the build budget is spent: such lanes iterate plainly
  int32_t overflows;  // builds skipped because the non-zero coefficients alone exceed WS_KCAP
  int32_t sweep_new;  // ws_score_kernel -> ws_select_kernel: features of newcomers at theta ...
  int32_t sweep_miss; // ... and coordinates a plain step moved outside W (both reset by ws_select_kernel)
  int32_t staged;     // row-sharded mode: the local Gram parts sit in the staging matrix, waiting for the
                      // all-reduce and ws_publish_kernel
  int32_t inner_iters;  // model-solver iterations of all refinements (diagnostics: SLM_TRACE=2)
  int32_t hard;         // direct steps were needed earlier in this call: refinements start with them
  int32_t hard_next;    // ... as recorded during the current pass (ws_select_kernel publishes it between passes)
  int32_t newton_steps; // direct (Cholesky) steps taken by the model solver, all lanes
  int32_t newton_fails; // ... refused: face not positive definite, or the step did not lower the model
  int32_t newton_nopd;  // ... of which: Cholesky pivot below the floor
  int32_t newton_trial[4];  // accepted trial of a direct step: full step, 1/2, 1/4, up to the first sign change
  int32_t newton_ref[4];    // refused steps: no sign-consistent segment (t = 0), slope <= 0, curvature <= 0, no decrease
  int32_t newton_factors;   // Cholesky factorisations (a direct step re-solves when zero coordinates move the wrong way)
  int32_t newton_unknowns;  // sum of their sizes
  // device clock ticks (wall_clock64: 100 MHz) spent by each lane's workgroup in the parts of its direct steps
  // (SLM_TRACE=2 prints them): matvec + free set, assembly, factorisation, solve, trial points, mu
  unsigned long long nt_ticks[SLM_MAX_LANES][6];
  int32_t nt_factors[SLM_MAX_LANES];
  // ... and in the parts of lane 0's refinements: set-up, lambda_max of a new Gram, start value, the iteration,
  // acceptance + write-back (SLM_TRACE=2)
  unsigned long long solve_ticks[5];
  int32_t want_full[SLM_MAX_LANES];  // lanes the light model solver left to the one with direct steps (this pass)
  int32_t iters_hist[32];  // (SLM_TRACE=2) model solves by their number of iterations (last bin: 31 or more)
  unsigned long long lane_ticks[SLM_MAX_LANES];  // (SLM_TRACE=2) ws_solve_kernel, entry to exit, per lane
  int32_t last_point[SLM_MAX_LANES];  // path point of each lane's last refinement ...
  int32_t repeats[SLM_MAX_LANES];     // ... and how many times in a row it was that point WITH the same columns
  int32_t last_cols[SLM_MAX_LANES];   // columns W held at each lane's last refinement (growth resets the count)
  double Lw[SLM_MAX_LANES];  // lambda_max estimate per Gram (0 = not yet computed)
  int32_t carried;    // this solve took over the working set of the solve before it (ws_ctl_carry_kernel): its first
                      // selection may append as much as a fresh one would choose
  int32_t outgrown;   // selections that did not fit: appends that would pass WS_KCAP, non-zeros alone beyond it -- the solve's
                      // lanes are outgrowing the working set (solve_core turns the model-Gram rounds on, mg_kernels.hpp)
  int32_t served[SLM_MAX_LANES];  // lanes the model solver moved since the model-Gram rounds last looked (mg_begin_kernel,
                                  // mg_kernels.hpp: what the working set serves is not served twice)
  int32_t hard_lane[SLM_MAX_LANES];  // a refinement of THIS lane needed direct steps: its next ones start with them.  (Round 5:
                                     // per lane, not per call -- one refinement of 31 iterations among the 84 of a path sent
                                     // every later one of all eighteen lanes through two factorisations of 200 unknowns, 0.4 ms
                                     // each where eight iterations take 20 us: 26 ms for an 8-pass path, soak seed 29)
};

struct WsArgs {
  WsCtl* ws;
  int32_t* idx;    // [WS_KCAP] feature of working-set position k, or -1
  int32_t* pos;    // [ld] position of feature j in W, or -1
  int32_t* gs;     // [WS_KCAP] first position of k's group
  int32_t* gl;     // [WS_KCAP] members of k's group
  double* score;   // [ld] scratch: entry score per item
  double* XW;      // [n][WS_KCAP] gathered columns
  double* part;    // [n_sets][tile][nblk][16 x 16] partial Grams
  double* Gm;      // [n_sets][WS_KCAP * WS_KCAP]
  double* nt;      // [SLM_MAX_LANES][NT_SCRATCH] factor of each lane's direct solve (nullptr: no direct solves)
  double* Gx;      // row-sharded mode: [n_sets][WS_KCAP * WS_KCAP] staging, zeroed every pass, summed over
                   // ranks between ws_gram_reduce_kernel and ws_publish_kernel (nullptr otherwise)
  const double* X;
  const double* XT;   // column-major copy of X in tiles of 32 rows (tile_columns_kernel; built once per dataset)
  int64_t n, ld;
  const double* rw;   // row weights per set, or nullptr (all ones)
  int64_t rw_stride;
  double inv_n[SLM_MAX_LANES];        // per SET: 1 / n_eff
  int32_t set_of[SLM_MAX_LANES];      // Gram of lane l (lanes with the same row weights and scaling share one)
  int32_t set_lane[SLM_MAX_LANES];    // a lane of set s (whose row weights the Gram kernel reads)
  int32_t n_sets;     // distinct (row weights, 1/n scaling) among the lanes
  int32_t nblk;
  const int32_t* owner;  // [n_sets][nblk] whose partial Gram of a row block a set sums up (ws_block_owner_kernel), or nullptr: its own
  double theta;
  int32_t lookahead;   // path points ahead whose penalty decides what enters W now
  int32_t append_max;  // newcomers appended per pass (the likeliest first)
  int32_t k_init;      // a fresh selection is cut down to this size (or to its non-zeros)
  int32_t bb_steps;    // the model solver opens with spectral steps (SLM_WS_BB=0: accelerated steps throughout)
  int32_t one_solver;  // SLM_WS_ONE_SOLVER=1: every lane goes to the solver with direct steps (measurements)
  int32_t miss_factor; // an append after a miss may take up to this many times append_max (4) ...
  int32_t miss_div;    // ... one more append_max for every miss_div coordinates the plain steps moved outside W
  double fill;         // a selection cut down to a cap stops bisecting its threshold once it holds this share of the cap
  int32_t power_iters; // power steps for lambda_max of a new Gram (SLM_WS_POWER_ITERS)
  int32_t hard_call;   // SLM_HARD_CALLWIDE=1 (A/B runs): direct steps once needed start every later refinement of the CALL
  int32_t keep_full;   // a selection that does not fit leaves W as it is (`stale`: the lanes it no longer covers are served by
                       // the model-Gram rounds) instead of selecting, gathering and multiplying afresh pass after pass
};

// The first words of the control block (request ... hard_next) and the call's stop word in ONE round trip: the first
// WS_HEAD_WORDS + 1 threads fetch a word each into `hd`; the caller's barrier makes them visible.  (A kernel that reads
// five words behind five tests waits five times; worth 1.2 us of ws_score_kernel's 15.6, nothing measurable in
// ws_select_kernel, which keeps its plain reads.)
constexpr int WS_HEAD_WORDS = 24;
static_assert(offsetof(WsCtl, hard_next) / 4 < WS_HEAD_WORDS, "the words the pass kernels decide on");
#define WS_HEAD(hd, field) ((hd)[offsetof(WsCtl, field) / 4])
__device__ __forceinline__ void ws_head_load(const WsCtl* ws, const int* gdone, int32_t* hd) {
  const int tid = threadIdx.x;
  if (tid < WS_HEAD_WORDS) hd[tid] = reinterpret_cast<const int32_t*>(ws)[tid];
  else if (tid == WS_HEAD_WORDS) hd[tid] = gdone ? gdone[0] : 0;
}

// state of a fresh solve (the block is zeroed first): a build is requested, no lane has been refined yet
static __global__ void ws_ctl_init_kernel(WsCtl* ws, int max_builds) {
  if (threadIdx.x == 0) {
    ws->request = 1;
    ws->max_builds = max_builds;
  }
  if (threadIdx.x < SLM_MAX_LANES) ws->last_point[threadIdx.x] = -1;
}

// Everything a solve sets up on the device before its first pass, in ONE launch (round 6): the vectors (solve_setup_body), the
// path points and the lanes' control blocks -- fetched by the kernel from the page-locked staging the host has just filled, not
// by copy commands --, the head of the control block zeroed, the step-size seed written into the lanes' blocks
// (seed_step_kernel's work) and the working set's state started (ws_ctl_init_kernel's).  They were six commands of the stream,
// 4-6 us each, in front of every solve: a third of the 0.1 ms the device spent before the first pass of a headline path.
struct BeginArgs {
  const PathCtl* h_ctl;  // [n_lanes] page-locked, device-visible
  PathCtl* ctl;
  const slm_path_point* h_pts;  // [n_pts] page-locked, device-visible
  slm_path_point* pts;
  int64_t n_pts;
  int32_t* head;       // the control block's first words (GlobalCtl, MgCtl[, WsCtl]) ...
  int32_t head_words;  // ... this many of them are zeroed
  const double* lambda;  // step-size estimate on the device (nullptr: the host has written L into h_ctl)
  double margin;
  double factor[SLM_MAX_LANES];
  WsCtl* ws;           // nullptr: the working set's state is left alone (not used, taken over, or set up later)
  int32_t max_builds;
  int32_t n_lanes;
};
static __global__ __launch_bounds__(256) void solve_begin_kernel(SetupArgs s, BeginArgs b) {
  solve_setup_body(s);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  {  // path points: four doubles each
    static_assert(sizeof(slm_path_point) == 4 * sizeof(double), "points travel as doubles");
    const double* src = reinterpret_cast<const double*>(b.h_pts);
    double* dst = reinterpret_cast<double*>(b.pts);
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < 4 * b.n_pts; e += stride) dst[e] = src[e];
  }
  if (blockIdx.x != 0) return;
  for (int e = threadIdx.x; e < b.head_words; e += 256) b.head[e] = 0;
  {
    static_assert(sizeof(PathCtl) % sizeof(int32_t) == 0, "control blocks travel as words");
    const int32_t* src = reinterpret_cast<const int32_t*>(b.h_ctl);
    int32_t* dst = reinterpret_cast<int32_t*>(b.ctl);
    const int words = b.n_lanes * (int)(sizeof(PathCtl) / sizeof(int32_t));
    for (int e = threadIdx.x; e < words; e += 256) dst[e] = src[e];
  }
  __syncthreads();
  if (b.lambda != nullptr && (int)threadIdx.x < b.n_lanes && threadIdx.x < SLM_MAX_LANES) {  // (seed_step_kernel)
    const int l = threadIdx.x;
    double L0 = b.lambda[0] * b.margin;
    if (L0 <= 0.0) L0 = 1.0;  // X == 0
    const double L = L0 * b.factor[l];
    b.ctl[l].L = L;
    b.ctl[l].ak = 1.25 * L;
    b.ctl[l].Lhat = 0.5 * L0;
  }
  if (b.ws != nullptr) {  // (ws_ctl_init_kernel; the block was zeroed above)
    if (threadIdx.x == 0) {
      b.ws->request = 1;
      b.ws->max_builds = b.max_builds;
    }
    if (threadIdx.x < SLM_MAX_LANES) b.ws->last_point[threadIdx.x] = -1;
  }
}

// state of a solve that starts where the dataset's last solve ended (solve_core: carried start, same lanes, same row
// sets): that solve's working set is still in place -- the columns' indices and positions, the gathered columns, the
// Grams, which depend on X and the rows alone -- and the penalty has changed, not the data.  The counters and per-solve
// records start afresh; W, its size and the Grams' lambda_max stay, and the first selection appends what the new
// penalty lets in instead of choosing, gathering and multiplying everything again (config 5's re-weighted solves: 0.8 ms
// of gather + Gram + reduce each).  One workgroup of 256.
static __global__ __launch_bounds__(256) void ws_ctl_carry_kernel(WsCtl* ws, int max_builds) {
  __shared__ int keep_i[2];
  __shared__ double keep_L[SLM_MAX_LANES];
  if (threadIdx.x == 0) {
    keep_i[0] = ws->K;
    keep_i[1] = ws->Kreal;
  }
  if (threadIdx.x < SLM_MAX_LANES) keep_L[threadIdx.x] = ws->Lw[threadIdx.x];
  __syncthreads();
  int32_t* words = reinterpret_cast<int32_t*>(ws);
  for (int e = threadIdx.x; e < (int)(sizeof(WsCtl) / sizeof(int32_t)); e += 256) words[e] = 0;
  __syncthreads();
  if (threadIdx.x == 0) {
    ws->K = keep_i[0];
    ws->Kreal = keep_i[1];
    ws->valid = 1;
    ws->carried = 1;
    ws->max_builds = max_builds;
  }
  if (threadIdx.x < SLM_MAX_LANES) {
    ws->Lw[threadIdx.x] = keep_L[threadIdx.x];
    ws->last_point[threadIdx.x] = -1;
  }
}

// exclusive prefix sum of one int per thread over the 1024-thread workgroup
__device__ __forceinline__ int block_excl_scan(int v, int* wave_tot /*[16]*/, int* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(inc, off, 64);
    if (lane >= off) inc += t;
  }
  __syncthreads();
  if (lane == 63) wave_tot[wave] = inc;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < TAIL_WAVES; ++w) {
    const int t = wave_tot[w];
    if (w < wave) base += t;
    tot += t;
  }
  *total = tot;
  return base + inc - v;
}

// ---------------------------------------------------------------------------------------------
// (i-a) one sweep over the items (features, or groups), one thread each: entry score (max over lanes;
// +inf marks items that are non-zero at a lane's expansion point), the feature count of newcomers at
// the default threshold, and whether the plain step of any lane moved a coordinate outside W
// (z = candidate / extrapolated point just produced by the tail kernel, zprev = the point it was
// produced from: they differ outside W exactly when a feature outside W wants to enter).
// Per-lane constants: is the lane walking a path, and the penalty scales w.lookahead points ahead of
// where it is (ranges may move between lanes in shared-path mode, so the look-ahead runs to the end of
// the path there).  Features that would enter by then are taken now; later ones are appended when
// their time comes.  (Inside the one-workgroup select kernel this sweep cost 45-60 us per pass.)
// ---------------------------------------------------------------------------------------------
// (the body, for the workgroup that scores items chunk * blockDim.x ...: ws_score_kernel runs it per blockIdx.x in workgroups
//  of 256 (features) or 64 (groups) threads)
__device__ __forceinline__ void ws_score_body(TailArgs a, WsArgs w, const int chunk) {
  WsCtl* ws = w.ws;
  const int tid = threadIdx.x;
  // The words this kernel decides on, fetched TOGETHER (ws_head_load) instead of one after the other behind the tests they
  // feed (`a || b` is two dependent round trips), and the lanes' point counts by a thread each instead of a loop per thread.
  __shared__ int32_t hd[WS_HEAD_WORDS + 1];
  __shared__ int lane_np[SLM_MAX_LANES];
  ws_head_load(ws, a.gdone, hd);
  if (tid < SLM_MAX_LANES) lane_np[tid] = tid < a.n_lanes ? a.ctl[tid].n_points : 0;  // (workgroups of 64 threads score groups)
  __syncthreads();
  if (hd[WS_HEAD_WORDS] != 0 || WS_HEAD(hd, disabled)) return;
  const bool had_w = WS_HEAD(hd, valid) != 0;
  if (!WS_HEAD(hd, request) && !had_w) return;
  if (WS_HEAD(hd, builds) >= WS_HEAD(hd, max_builds) || WS_HEAD(hd, appends) >= 8 * WS_HEAD(hd, max_builds)) return;
  const bool singleton = a.singleton != 0;
  const int nitems = singleton ? a.p : a.G;
  __shared__ int lane_live[SLM_MAX_LANES];
  __shared__ double lane_sa[SLM_MAX_LANES], lane_sb[SLM_MAX_LANES];
  __shared__ int cnt[2];
  if (tid < SLM_MAX_LANES) {
    int live = 0;
    double sa = 0.0, sb = 0.0;
    if (tid < a.n_lanes) {
      const PathCtl* c = a.ctl + tid;
      live = !(c->done || c->idle);
      int path_end = 0;
      for (int l = 0; l < SLM_MAX_LANES; ++l) path_end = max(path_end, lane_np[l]);
      const int end = a.steal ? path_end : c->pt_off + c->n_points;
      // (interleaved lanes jump `stride` points at a time: look at least as far as the next one)
      int look = min(c->pt_off + c->point + max(w.lookahead, c->stride), end - 1);
      if (c->tail_pt >= 0 && c->point + c->stride >= c->n_points) look = c->pt_off + c->tail_pt;  // (its next point)
      const slm_path_point pe = a.pts[look < 0 ? 0 : look];
      sa = pe.sa;
      sb = pe.sb;
    }
    lane_live[tid] = live;
    lane_sa[tid] = sa;
    lane_sb[tid] = sb;
  }
  if (tid == 0) cnt[0] = cnt[1] = 0;
  __syncthreads();

  const double inf = __builtin_huge_val();
  int n_new = 0, n_miss = 0;
  // features: one thread each.  Groups: SIXTEEN threads each, one per lane (a thread then walks the members
  // of its group for one lane: 10 x 4 scattered loads instead of 16 x 10 x 4; the lanes' scores meet in a
  // max over the 16 threads) -- 64-thread workgroups, four groups each.
  const int gt = chunk * (int)blockDim.x + tid;
  const int it = singleton ? gt : gt >> 4;
  if (singleton) {
    if (it < nitems) {
      double sc = 0.0;
      const int j = it;
      const bool in_w = had_w && w.pos[j] >= 0;
      // all loads of all lanes first (unconditional, clamped to a lane that exists), then the arithmetic: with
      // the tests between them every lane cost two dependent round trips, 32 in a row (21 us per pass)
      // (sixteen lanes at a time: a call of thirty-two is two such rounds -- all of them in registers at once would be 320)
      for (int l0 = 0; l0 < a.n_lanes; l0 += 16) {
        double zp[16], zz[16], aa[16], bb[16], gg[16];
#pragma unroll
        for (int l = 0; l < 16; ++l) {
          const int ll = l0 + l < a.n_lanes ? l0 + l : 0;
          const int64_t off = (int64_t)ll * a.ld;
          zp[l] = a.zprev[off + j];
          zz[l] = a.z[off + j];
          aa[l] = a.a0[off + j];
          bb[l] = a.b0[off + j];
          gg[l] = a.g[(int64_t)ll * (a.ld + 16) + j];
        }
#pragma unroll
        for (int l = 0; l < 16; ++l) {
          if (l0 + l >= a.n_lanes || !lane_live[l0 + l]) continue;
          if (!in_w && had_w && zz[l] != zp[l]) n_miss += 1;
          if (zp[l] != 0.0) {
            sc = inf;
          } else {
            const double thr = lane_sa[l0 + l] * aa[l] + lane_sb[l0 + l] * bb[l];
            sc = fmax(sc, thr > 0.0 ? fabs(gg[l]) / thr : inf);
          }
        }
      }
      if (sc >= w.theta && !in_w) n_new += 1;
      w.score[it] = sc;
    }
  } else {
    static_assert(SLM_MAX_LANES % 16 == 0, "16-thread teams: a thread per (group, lane of a half)");
    const int l16 = tid & 15;
    const bool have = it < nitems;  // (a team is all in or all out)
    double sc = 0.0;
    bool in_w = false;
    int k0 = 0, k1 = 0;
    if (have) {
      k0 = a.gstart[it];
      k1 = a.gstart[it + 1];
      in_w = had_w && w.pos[a.order[k0]] >= 0;
    }
    for (int l = l16; have && l < a.n_lanes; l += 16) {  // (a call of thirty-two lanes: two lanes per thread)
      if (lane_live[l]) {
        const int64_t off = (int64_t)l * a.ld;
        const double* g = a.g + (int64_t)l * (a.ld + 16);
        double num = 0.0, rmax = 0.0;
        bool act = false;
        for (int k = k0; k < k1; ++k) {
          const int j = a.order[k];
          const double zp = a.zprev[off + j], zj = a.z[off + j], aj = a.a0[off + j], gj = g[j];  // (one round of loads)
          if (!in_w && had_w && zj != zp) n_miss += 1;
          act = act || zp != 0.0;
          const double thr = lane_sa[l] * aj;
          const double m = fmax(fabs(gj) - thr, 0.0);
          num = __builtin_fma(m, m, num);
          rmax = fmax(rmax, thr > 0.0 ? fabs(gj) / thr : inf);
        }
        const double den = lane_sb[l] * a.b0[off + it];
        const double r = den > 0.0 ? sqrt(num) / den : rmax;
        sc = fmax(sc, act ? inf : r);
      }
    }
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) sc = fmax(sc, __shfl_xor(sc, o, 64));
    if (have && l16 == 0) {
      if (sc >= w.theta && !in_w) n_new += k1 - k0;
      w.score[it] = sc;
    }
  }
  // integer counts: the order of the atomic additions does not matter
  if (n_new) atomicAdd(&cnt[0], n_new);
  if (n_miss) atomicAdd(&cnt[1], n_miss);
  __syncthreads();
  if (tid == 0) {
    if (cnt[0]) atomicAdd(&ws->sweep_new, cnt[0]);
    if (cnt[1]) atomicAdd(&ws->sweep_miss, cnt[1]);
  }
}

static __global__ __launch_bounds__(256) void ws_score_kernel(TailArgs a, WsArgs w) { ws_score_body(a, w, (int)blockIdx.x); }

// ---------------------------------------------------------------------------------------------
// (i-b) choose W.  One workgroup.  Runs after ws_score_kernel in every pass; returns at once unless a
// build was requested, a lane's plain step left the current W, or there are newcomers.  With a valid W the newcomers are
// APPENDED (their columns and Gram rows are all that has to be produced); a fresh selection is made
// at the start of a solve and when the appended set would exceed WS_KCAP.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void ws_select_body(TailArgs a, WsArgs w) {
  __shared__ double red[2][TAIL_WAVES];
  __shared__ int wave_tot[TAIL_WAVES];
  WsCtl* ws = w.ws;
  // `served` is this pass's: set again by the model solver (launched behind this kernel) for the lanes it moves.  Left
  // standing from pass to pass, the flags of the last pass the working set served everyone made the FIRST round on the
  // model Gram sit out -- a whole pass over X between the overflow and the first proposal.
  if (threadIdx.x < SLM_MAX_LANES) ws->served[threadIdx.x] = 0;
  if (a.gdone[0] != 0 || ws->disabled) return;
  const int tid = threadIdx.x;
  const int p = a.p, G = a.G;
  const bool singleton = a.singleton != 0;
  const int nitems = singleton ? p : G;
  const bool had_w = ws->valid != 0;

  if (tid == 0 && ws->hard_next) ws->hard = 1;
  const bool requested = ws->request != 0;
  if (!requested && !had_w) return;
  if (ws->builds >= ws->max_builds || ws->appends >= 8 * ws->max_builds) {
    // budget spent: keep the W we have; lanes whose plain step leaves it iterate plainly (ws_solve_kernel
    // checks that per lane once `stale` is set)
    if (tid == 0) {
      ws->request = 0;
      ws->stale = 1;
    }
    return;
  }

  // scores and the two counts come from ws_score_kernel (many workgroups, launched just before)
  double sweep[2] = {(double)ws->sweep_new, (double)ws->sweep_miss};
  const bool carried = ws->carried != 0;  // (first selection of a solve on its predecessor's W)
  __syncthreads();
  if (tid == 0) {
    ws->sweep_new = 0;
    ws->sweep_miss = 0;
    ws->carried = 0;
  }
  const double inf = __builtin_huge_val();
  const bool miss = had_w && sweep[1] != 0.0;
  if (miss && tid == 0) ws->misses += 1;
  if (had_w && !requested && sweep[0] == 0.0) {
    // no newcomer (the usual pass).  A lane that left W anyway cannot be helped: it iterates plainly.
    if (miss && tid == 0) ws->stale = 1;
    return;
  }

  // This thread's items (a contiguous run of `per`), read ONCE: score, feature count and whether the item sits in
  // W already.  Up to eight items per thread (p <= 8192) they live in registers -- the counting sweeps below (up to
  // fifteen of them when a threshold is fitted) and the position pass at the end then touch no memory; item by item
  // every sweep was a chain of dependent round trips (29 us for this kernel on the headline path).
  const int per = (nitems + WS_THREADS - 1) / WS_THREADS;
  const int i0 = tid * per, i1 = min(i0 + per, nitems);
  const bool cached = per <= 8;
  double c_sc[8];
  int c_sz[8];
  bool c_inw[8];
  if (cached) {
    int first[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int it = (i0 + u < i1) ? i0 + u : 0;
      c_sc[u] = (i0 + u < i1) ? w.score[it] : -1.0;  // (below every threshold: an item that is not there is never taken)
      first[u] = singleton ? it : a.gstart[it];
      c_sz[u] = singleton ? 1 : a.gstart[it + 1] - first[u];
    }
    if (!singleton) {
#pragma unroll
      for (int u = 0; u < 8; ++u) first[u] = a.order[first[u]];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) c_inw[u] = w.pos[first[u]] >= 0;
  }

  // feature count of the items with score >= thr (optionally only those not yet in W), and the
  // largest finite score
  auto count_at = [&](double thr, bool only_new, double* n_sel, double* smax) {
    double v[2] = {0.0, 0.0};
    if (cached) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (c_sc[u] >= thr && !(only_new && c_inw[u])) v[0] += (double)c_sz[u];
        if (c_sc[u] < inf) v[1] = fmax(v[1], c_sc[u]);
      }
    } else {
      for (int it = tid; it < nitems; it += WS_THREADS) {
        const double sc = w.score[it];
        const int first = singleton ? it : a.order[a.gstart[it]];
        const int size = singleton ? 1 : a.gstart[it + 1] - a.gstart[it];
        if (sc >= thr && !(only_new && w.pos[first] >= 0)) v[0] += (double)size;
        if (sc < inf) v[1] = fmax(v[1], sc);
      }
    }
    double s[1] = {v[0]};
    block_sum<1>(s, red);
    *n_sel = s[0];
    if (smax == nullptr) return;  // (only the first count of a fitted threshold asks for it: two barriers less for the others)
    double m = v[1];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmax(m, __shfl_xor(m, off, 64));
    __syncthreads();
    if ((tid & 63) == 0) red[1][tid >> 6] = m;
    __syncthreads();
    double mm = 0.0;
#pragma unroll
    for (int k = 0; k < TAIL_WAVES; ++k) mm = fmax(mm, red[1][k]);
    *smax = mm;
  };

  // smallest threshold >= thr0 whose selection has at most `cap` features (scores >= it are kept)
  auto fit_threshold = [&](double thr0, bool only_new, double cap, double* n_out) -> double {
    double n_sel, smax;
    count_at(thr0, only_new, &n_sel, &smax);
    if (n_sel <= cap) {
      *n_out = n_sel;
      return thr0;
    }
    double lo = thr0, hi = fmax(smax, thr0) * (1.0 + 1e-12) + 1e-300;  // count(hi) = non-zeros only
    double n_hi = 0.0;
    for (int k = 0; k < 14; ++k) {
      const double mid = 0.5 * (lo + hi);
      count_at(mid, only_new, &n_sel, nullptr);
      if (n_sel > cap) {
        lo = mid;
      } else {
        hi = mid;
        n_hi = n_sel;
        if (n_sel >= w.fill * cap) break;  // close enough to the cap
      }
    }
    if (n_hi == 0.0) count_at(hi, only_new, &n_hi, nullptr);
    *n_out = n_hi;
    return hi;
  };

  double n_sel, smax, thr = w.theta;
  const int k_old = had_w ? ws->Kreal : 0;
  bool append = had_w;
  if (append) {
    // newcomers: at most w.append_max per pass, the likeliest first (a feature left out that enters
    // anyway shows up as a miss and is appended then)
    // (a lane is stuck: be generous; the first selection of a solve that took over its predecessor's W: as many as a
    //  fresh selection would take -- appended columns cost their own gather and Gram rows only)
    // (how generous, by how far the lanes left W: a path whose supports outgrow W in waves -- dozens of coordinates a pass --
    //  takes the full factor; ONE lane that met a feature or two at the deep end of a sparse path takes a single append_max:
    //  at four it pulled 150 candidates that never entered into every later Gram and model solve, 0.2-0.3 ms per path on
    //  three of nine draws of the headline's law, tools/ab_knobs_draws.py)
    const int miss_mult = miss ? min(w.miss_factor, 1 + (int)sweep[1] / max(w.miss_div, 1)) : 1;
    const double cap = (double)(carried ? max(w.k_init, 4 * w.append_max) : miss_mult * w.append_max);
    if (sweep[0] <= cap) {
      n_sel = sweep[0];
    } else {
      thr = fit_threshold(w.theta, true, cap, &n_sel);
    }
    if (n_sel == 0.0) {  // (requested with a valid W and nothing to add)
      if (tid == 0) ws->request = 0;
      return;
    }
    if ((double)k_old + n_sel > (double)WS_KCAP) {  // does not fit: select afresh
      if (tid == 0) ws->outgrown += 1;
      if (w.keep_full) {
        // the model-Gram rounds serve what W does not cover.  Where the non-zeros of the live lanes alone fill most of
        // the capacity -- a path whose solutions are outgrowing the set for good -- a fresh selection would be outgrown
        // again a pass later (gather + Gram of 500 columns, 2-3 ms a time): W stays as it is.  Where they do not -- W is
        // full of candidates that never entered, the way strongly correlated designs fill it -- the selection goes ahead
        // as it always did: those lanes are best served by W's own model solver and its direct steps.
        double nz_now;
        count_at(inf, false, &nz_now, nullptr);
        if (nz_now > 0.75 * (double)WS_KCAP) {
          if (tid == 0) {
            ws->request = 0;
            ws->stale = 1;
          }
          return;
        }
      }
      append = false;
      thr = w.theta;
    }
  }
  if (!append) {
    count_at(inf, false, &n_sel, nullptr);
    if (n_sel > (double)WS_KCAP) {
      // the non-zero coefficients of the expansion points alone do not fit (dense early iterates of
      // a cold start, a dense solution, or ONE lane whose plain steps went dense): with a W in place it
      // stays -- the lanes it still serves keep refining, the ones that left it iterate plainly (`stale`);
      // without one, try again after the next pass and give up after 50 tries
      if (tid == 0) {
        ws->overflows += 1;
        ws->outgrown += 1;
        if (had_w) {
          ws->request = 0;
          ws->stale = 1;
        } else {
          ws->request = 1;
          ws->valid = 0;
          if (ws->overflows >= 50) ws->disabled = 1;
        }
      }
      return;
    }
    const double nonzeros = n_sel;
    thr = fit_threshold(w.theta, false, fmax((double)w.k_init, nonzeros), &n_sel);
    for (int j = tid; j < p; j += WS_THREADS) w.pos[j] = -1;
  }
  for (int k = k_old * (append ? 1 : 0) + tid; k < WS_KCAP; k += WS_THREADS) {
    w.idx[k] = -1;
    w.gs[k] = k;
    w.gl[k] = 1;
  }
  __syncthreads();

  // ---- positions: items in index order, members of a group contiguous -----------------------------
  // which of this thread's items are taken (decided before pos[] is written below), and their feature count
  unsigned long long pick = 0ull;
  int mine = 0;
  if (cached) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (c_sc[u] >= thr && !(append && c_inw[u])) {
        pick |= 1ull << u;
        mine += c_sz[u];
      }
    }
  } else {
    for (int it = i0; it < i1 && it - i0 < 64; ++it) {
      if (!(w.score[it] >= thr)) continue;
      if (append && w.pos[singleton ? it : a.order[a.gstart[it]]] >= 0) continue;
      pick |= 1ull << (it - i0);
      mine += singleton ? 1 : a.gstart[it + 1] - a.gstart[it];
    }
  }
  int total = 0;
  int at = (append ? k_old : 0) + block_excl_scan(mine, wave_tot, &total);
  __syncthreads();
  for (int it = i0; it < i1 && it - i0 < 64; ++it) {
    if (!((pick >> (it - i0)) & 1ull)) continue;
    if (singleton) {
      w.idx[at] = it;
      w.pos[it] = at;
      at += 1;
    } else {
      const int k0 = a.gstart[it], len = a.gstart[it + 1] - k0;
      for (int m = 0; m < len; ++m) {
        const int j = a.order[k0 + m];
        w.idx[at + m] = j;
        w.pos[j] = at + m;
        w.gs[at + m] = at;
        w.gl[at + m] = len;
      }
      at += len;
    }
  }
  if (tid == 0) {
    const int Kreal = (append ? k_old : 0) + total;
    ws->Kreal = Kreal;
    ws->K = max(16, (Kreal + 15) & ~15);
    ws->k_new = append ? k_old : 0;
    ws->building = 1;
    ws->valid = 0;
    ws->stale = 0;
    ws->counter = 0;
    if (append) {
      ws->appends += 1;
    } else {
      ws->builds += 1;
      for (int l = 0; l < SLM_MAX_LANES; ++l) ws->Lw[l] = 0.0;
    }
  }
}

static __global__ __launch_bounds__(WS_THREADS) void ws_select_kernel(TailArgs a, WsArgs w) { ws_select_body(a, w); }

// ---------------------------------------------------------------------------------------------
// (ii) gather the new columns: XW[i][k] = X[i][idx[k]] for k >= k_new (0 for padding positions).
// Read from the column-major copy XT, where the 32 rows of a tile of a column are contiguous, through a 32 x 32 LDS
// tile, so reads and writes both move 256-byte segments.  (Gathering from the row-major X touches one 64-byte
// sector per element: 0.2 ms for 112 columns, 0.5 ms per 50-alpha path.)  grid (row tiles, 16 column
// tiles of 32 starting at k_new); tiles beyond K return at once.
// ---------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void ws_gather_kernel(WsArgs w) {
  if (!w.ws->building) return;
  const int K = w.ws->K;
  const int k0 = w.ws->k_new + 32 * (int)blockIdx.y;
  if (k0 >= K) return;
  __shared__ double tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int64_t row_tiles = (w.n + 31) / 32;
  // the four columns this thread reads in every row tile: looked up once (inside the loop every load of X waited
  // for the index load before it)
  int jcol[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int k = k0 + ty + 8 * u;
    jcol[u] = k < K ? w.idx[k] : -1;
  }
  for (int64_t rt = blockIdx.x; rt < row_tiles; rt += gridDim.x) {
    const int64_t i0 = rt * 32;
    const int64_t i = i0 + tx;
    double v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {  // read: lanes walk rows i (contiguous in XT)
      const int j = jcol[u];
      // (XT == nullptr: no memory for the column-major copy -- same tile, sector-granular reads from X)
      v[u] = (j >= 0 && i < w.n) ? (w.XT ? w.XT[((rt * w.ld + j) << 5) + tx] : w.X[i * w.ld + j]) : 0.0;
    }
    __syncthreads();  // the tile of the previous round has been written out
#pragma unroll
    for (int u = 0; u < 4; ++u) tile[ty + 8 * u][tx] = v[u];
    __syncthreads();
    for (int ii = ty; ii < 32; ii += 8) {  // write: lanes walk positions k (contiguous in XW)
      const int64_t i = i0 + ii;
      const int k = k0 + tx;
      if (i < w.n && k < K) w.XW[i * WS_KCAP + k] = tile[tx][ii];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// (ii-b) A path that opened on a row sample (solve_core, "sample start"): the gradient at zero the first call worked
// with is an estimate -- good enough to choose W, not to build the model on.  On W the exact one costs one read of
// the gathered columns, not of X: c_k = -X_k^T y / n, and the loss at zero is y^T y / 2n.  Two launches: partial
// sums over row blocks, then the sums go into g and gprev of every lane (all lanes stand at zero), the loss
// into g[ld], the first entry of the objective history and PathCtl::loss_base.  No row weights (the caller checks).
// ---------------------------------------------------------------------------------------------
struct XtyArgs {
  const WsCtl* ws;
  const int32_t* idx;
  const double* XW;
  const double* y;
  double* part;     // [nblk][WS_KCAP + 1]: per block the sums of every position and, last, of y^2
  double* g;        // [lanes][ld + 16]
  double* gprev;    // [lanes][ld]
  double* z;        // [lanes][ld]: the model solves start from zero on W (the candidate of the first call is a step along the estimate)
  PathCtl* ctl;
  const int* done;
  int64_t n, ld;
  double inv_n;
  int n_lanes, nblk;
};
static __global__ __launch_bounds__(512) void ws_xty_partial_kernel(XtyArgs a) {
  if (*a.done) return;
  __shared__ double red[8][WS_KCAP];
  __shared__ double red_yy[8];
  const int K = a.ws->K;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t rows = (a.n + gridDim.x - 1) / gridDim.x;
  const int64_t i0 = (int64_t)blockIdx.x * rows, i1 = i0 + rows < a.n ? i0 + rows : a.n;
  // a wavefront per row (rows i0 + wave, + 8, ...), a lane per position of each chunk of 64: the loads of a row are one
  // contiguous segment, and eight rows are in flight per workgroup
  double acc[WS_KCAP / 64];
#pragma unroll
  for (int c = 0; c < WS_KCAP / 64; ++c) acc[c] = 0.0;
  double yy = 0.0;
  // The number of 64-column chunks is settled ONCE, outside the row loop, and the loads of four rows are issued together (a
  // position past K reads position 0 of its chunk and counts with a factor of zero).  With the chunk test inside the loop the
  // compiler kept one row in flight: 82 us at 384 columns where this takes 48, 22 -> 15 at 96 (tools/probes/xty_probe.hip).
  auto rows4 = [&](auto NC) {
    constexpr int C = decltype(NC)::value;
    for (int64_t i = i0 + wave; i < i1; i += 32) {
      double xv[4][C], yv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool in = i + 8 * r < i1;
        const int64_t ii = in ? i + 8 * r : i;
        yv[r] = in ? a.y[ii] : 0.0;
#pragma unroll
        for (int c = 0; c < C; ++c) {
          const int k = lane + 64 * c;
          xv[r][c] = a.XW[ii * WS_KCAP + (k < K ? k : 64 * c)];
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        yy = __builtin_fma(yv[r], yv[r], yy);
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = __builtin_fma(lane + 64 * c < K ? xv[r][c] : 0.0, yv[r], acc[c]);
      }
    }
  };
  switch ((K + 63) >> 6) {
    case 1: rows4(std::integral_constant<int, 1>{}); break;
    case 2: rows4(std::integral_constant<int, 2>{}); break;
    case 3: rows4(std::integral_constant<int, 3>{}); break;
    case 4: rows4(std::integral_constant<int, 4>{}); break;
    case 5: rows4(std::integral_constant<int, 5>{}); break;
    case 6: rows4(std::integral_constant<int, 6>{}); break;
    case 7: rows4(std::integral_constant<int, 7>{}); break;
    default: rows4(std::integral_constant<int, 8>{}); break;
  }
  static_assert(WS_KCAP / 64 == 8, "eight chunks of 64 positions");
#pragma unroll
  for (int c = 0; c < WS_KCAP / 64; ++c) red[wave][lane + 64 * c] = acc[c];
  if (lane == 0) red_yy[wave] = yy;
  __syncthreads();
  double* out = a.part + (int64_t)blockIdx.x * (WS_KCAP + 1);
  const int k = threadIdx.x;
  if (k < K) {
    double sum = 0.0;
#pragma unroll
    for (int w = 0; w < 8; ++w) sum += red[w][k];
    out[k] = sum;
  }
  if (k == WS_KCAP - 1) {
    double sum = 0.0;
#pragma unroll
    for (int w = 0; w < 8; ++w) sum += red_yy[w];
    out[WS_KCAP] = sum;
  }
}
// grid: WS_KCAP / 128 workgroups of 512 threads -- 128 positions each, four threads per position that take every
// fourth row block (eight loads in flight each: one thread per position walked the 512 blocks one load at a time, 122 us)
static __global__ __launch_bounds__(512) void ws_xty_apply_kernel(XtyArgs a) {
  if (*a.done) return;
  __shared__ double red[4][128];
  __shared__ double ry[512];
  const int K = a.ws->K;
  const int kl = threadIdx.x & 127, q = threadIdx.x >> 7;
  const int k = (int)blockIdx.x * 128 + kl;
  if ((int)blockIdx.x * 128 >= K) return;  // (uniform for the workgroup)
  double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (k < K) {
    int b = q;
    for (; b + 28 < a.nblk; b += 32) {
#pragma unroll
      for (int u = 0; u < 8; ++u) s[u] += a.part[(int64_t)(b + 4 * u) * (WS_KCAP + 1) + k];
    }
    for (; b < a.nblk; b += 4) s[0] += a.part[(int64_t)b * (WS_KCAP + 1) + k];
  }
  red[q][kl] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  if (blockIdx.x == 0) ry[threadIdx.x] = (int)threadIdx.x < a.nblk ? a.part[(int64_t)threadIdx.x * (WS_KCAP + 1) + WS_KCAP] : 0.0;
  __syncthreads();
  if (q == 0 && k < K) {
    const int j = a.idx[k];
    if (j >= 0) {
      const double c = -((red[0][kl] + red[1][kl]) + (red[2][kl] + red[3][kl])) * a.inv_n;
      for (int l = 0; l < a.n_lanes; ++l) {
        a.g[(int64_t)l * (a.ld + 16) + j] = c;
        a.gprev[(int64_t)l * a.ld + j] = c;
        a.z[(int64_t)l * a.ld + j] = 0.0;
      }
    }
  }
  if (blockIdx.x == 0) {  // the loss at zero: the row blocks' sums of y^2, folded in a fixed order (nblk <= 512)
    for (int off = 256; off >= 1; off >>= 1) {
      if ((int)threadIdx.x < off) ry[threadIdx.x] += ry[threadIdx.x + off];
      __syncthreads();
    }
    if (threadIdx.x < (unsigned)a.n_lanes) {
      const int l = threadIdx.x;
      const double loss0 = 0.5 * ry[0] * a.inv_n;
      a.g[(int64_t)l * (a.ld + 16) + a.ld] = loss0;
      a.ctl[l].hist[0] = loss0;  // (the penalty at zero is zero)
      a.ctl[l].loss_base = loss0;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// (iii) partial Grams with v_mfma_f64_16x16x4_f64.  Grid (nblk, n_sets); 8 wavefronts per
// workgroup laid out 4 x 2, each owning a 4 x 4 block of 16x16 output tiles, so a workgroup covers
// 256 x 128 of the 512 x 512 capacity at a time and walks the slices that exist (2 row halves x 4 column
// quarters; see the loop over `slice`).  16 accumulator tiles = 128 registers per lane, which is why
// it is not one 1024-thread workgroup.  Only the tile
// rows that hold new positions (I >= k_new / 16) are computed; the reduce kernel mirrors them into
// the columns.  Operand maps (guide "Fragment layout"): lane l holds A[i = l & 15][k = l >> 4] and
// B[k = l >> 4][j = l & 15]; result register r of lane l is D[row (l >> 4) + 4 r][col l & 15].  Here
// k runs over 4 consecutive rows of XW, A[i][k] = w_k XW[row_k][16 I + i], B[k][j] = XW[row_k][16 J + j].
// ---------------------------------------------------------------------------------------------
// Row masks of a CV grid are ones except on their fold's test rows, and with contiguous folds (KFold as scikit-learn makes
// it) most row blocks of the Gram kernel see ALL ONES in most sets and ALL ZEROS in one: the partial Gram of such a block is
// the same matrix for every all-ones set (computed once, by the first of them) and nothing for an all-zeros set.
// owner[set][b]: the set whose partial the reduce kernel adds for (set, b) -- the set itself where its weights on the block
// are mixed (or it is the first all-ones set there), -1 where they are all zeros.  Two masks in a call: one read of the
// gathered columns instead of two (an eighth of config 4: 0.35 -> 0.15 ms per pass).  One workgroup per row block.
static __global__ __launch_bounds__(256) void ws_block_owner_kernel(const double* rw, int64_t rw_stride, WsArgs w, int32_t* owner) {
  __shared__ int cls[SLM_MAX_LANES];  // 0: all zeros, 1: all ones, 2: anything else
  __shared__ int red[2][4];
  const int tid = threadIdx.x;
  const int64_t b = blockIdx.x;
  const int64_t base = w.n / w.nblk, rem = w.n % w.nblk;
  const int64_t r0 = b * base + (b < rem ? b : rem);
  const int64_t nrows = base + (b < rem ? 1 : 0);
  for (int st = 0; st < w.n_sets; ++st) {
    const double* v = rw + (int64_t)w.set_lane[st] * rw_stride + r0;
    int zeros = 1, ones = 1;
    for (int64_t i = tid; i < nrows; i += 256) {
      const double x = v[i];
      zeros &= (x == 0.0);
      ones &= (x == 1.0);
    }
    zeros = __all(zeros);
    ones = __all(ones);
    if ((tid & 63) == 0) {
      red[0][tid >> 6] = zeros;
      red[1][tid >> 6] = ones;
    }
    __syncthreads();
    if (tid == 0) {
      const int z = red[0][0] & red[0][1] & red[0][2] & red[0][3], o = red[1][0] & red[1][1] & red[1][2] & red[1][3];
      cls[st] = z ? 0 : (o ? 1 : 2);
    }
    __syncthreads();
  }
  if (tid == 0) {
    int first_ones = -1;
    for (int st = 0; st < w.n_sets; ++st) {
      int o = st;
      if (cls[st] == 0) o = -1;
      else if (cls[st] == 1) {
        if (first_ones < 0) first_ones = st;
        o = first_ones;
      }
      owner[(int64_t)st * w.nblk + b] = o;
    }
  }
}

typedef double ws_d4 __attribute__((ext_vector_type(4)));
constexpr int WS_GRAM_THREADS = 512;
constexpr int WS_GRAM_ZCHUNKS = 4;  // slices of a row block: 4 column quarters x this many chunks of tile rows

static __global__ __launch_bounds__(WS_GRAM_THREADS) void ws_gram_kernel(WsArgs w) {
  if (!w.ws->building) return;
  __shared__ ws_d4 comb[2][16][64];  // 64 KiB: row-split appends fold their four parts through here
  const int K = w.ws->K;
  const int tile_lo = w.ws->k_new >> 4;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles = K >> 4;
  // An append touches at most four tile rows.  With the usual mapping only the wavefronts whose tile
  // rows those are would work (2 of 8, 98 dependent steps each: 0.13 ms of latency); instead the four
  // wavefront pairs split the ROWS of the block between them, all on the new tile rows, and fold
  // their accumulators through LDS in a fixed order.
  // Larger appends (a miss quadruples the cap) are cut into chunks of four tile rows, one per value of
  // slice >> 2 below, each handled the same way (up to 16 new tile rows = 256 columns; the generic
  // mapping left most wavefronts idle on them: 0.44 ms for 170 new columns).
  // (a small fresh selection -- the first 112 columns of a path: 7 x 7 tiles -- keeps half of the wavefronts of ONE
  //  workgroup per row block busy under the usual mapping, each walking all rows of the block: 90 us; split by rows
  //  like an append it is eight wavefronts of two workgroups on a quarter of the rows each)
  const bool row_split = (tile_lo > 0 || tiles <= 8) && tiles - tile_lo <= 4 * WS_GRAM_ZCHUNKS;
  const int set = blockIdx.y;
  const int64_t b = blockIdx.x;
  if (w.owner != nullptr && w.owner[(int64_t)set * w.nblk + b] != set) return;  // (all zeros here, or another set's partial serves)
  const int64_t base = w.n / w.nblk, rem = w.n % w.nblk;
  const int64_t r0 = b * base + (b < rem ? b : rem);
  const int64_t nrows = base + (b < rem ? 1 : 0);
  const double* rw = w.rw ? w.rw + (int64_t)w.set_lane[set] * w.rw_stride : nullptr;
  // rows of this wavefront: everything, or the part-th quarter (in steps of four rows)
  const int part = row_split ? (wave >> 1) : 0;
  const int64_t steps = (nrows + 3) >> 2;
  const int64_t s_begin = row_split ? 4 * (steps * part / 4) : 0;
  const int64_t s_end = row_split ? min(nrows, 4 * (steps * (part + 1) / 4)) : nrows;

  // The (column quarter, chunk of tile rows) slices of this row block, one after the other in ONE workgroup.  As
  // grid.z they were sixteen workgroups per row block of which an append on the headline path needs two: the other
  // fourteen still had to be started, each holding the CU's registers and 64 KB of its LDS until it had read the
  // control block and left -- half of the kernel's 67 us.
  for (int slice = 0; slice < 4 * WS_GRAM_ZCHUNKS; ++slice) {
  const int zhi = slice >> 2;
  if (8 * (slice & 3) >= tiles) continue;  // (no column of this quarter exists)
  if (row_split ? (tile_lo + 4 * zhi >= tiles) : (zhi >= 2)) continue;
  const int wj = 2 * (slice & 3) + (wave & 1);
  const int ntj = min(4, max(0, tiles - 4 * wj));
  const int ibase = row_split ? tile_lo + 4 * zhi : 4 * (4 * zhi + (wave >> 1));
  const int ti_lo = row_split ? 0 : max(0, tile_lo - ibase);  // first tile row of this wave to do
  const int nti = min(4, max(0, tiles - ibase));
  const bool active = ti_lo < nti && ntj > 0;
  if (!row_split && !active) continue;  // (per wavefront: this mode has no barrier)
  ws_d4 acc[4][4];
#pragma unroll
  for (int ti = 0; ti < 4; ++ti)
#pragma unroll
    for (int tj = 0; tj < 4; ++tj) acc[ti][tj] = ws_d4{0.0, 0.0, 0.0, 0.0};

  const int kk = lane >> 4, c = lane & 15;
  if (active) {
    // (the row weight is applied in mma(), not at the load, so nothing waits on a load before the MFMAs)
    // (every load of a step is issued whatever the tile pattern: a tile this wavefront does not need is read from
    //  a column block that exists and never used.  With the loads under conditions the compiler could not count
    //  them and waited for ALL of them -- the ones of the steps ahead included -- before the first MFMA of a step)
    const int ta_max = WS_TILES - 1;
    const double* rwp = rw ? rw : w.XW;  // (no row weights: any readable address, the value is replaced by 1)
    const bool has_rw = rw != nullptr;
    auto load = [&](int64_t s, double(&av)[4], double(&bv)[4], double& wgt) {
      const int64_t i = s + kk;
      const bool ok = i < s_end;
      const int64_t row = r0 + (ok ? i : 0);
      const double wv = rwp[has_rw ? row : 0];
      const double* xr = w.XW + row * WS_KCAP + c;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        av[t] = xr[16 * min(ibase + t, ta_max)];
        bv[t] = xr[16 * min(4 * wj + t, ta_max)];
      }
      wgt = ok ? (has_rw ? wv : 1.0) : 0.0;
    };
    auto mma = [&](const double(&av)[4], const double(&bv)[4], double wgt) {
#pragma unroll
      for (int ti = 0; ti < 4; ++ti) {
        if (ti >= ti_lo && ti < nti) {
          const double a = av[ti] * wgt;
#pragma unroll
          for (int tj = 0; tj < 4; ++tj)
            if (tj < ntj) acc[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv[tj], acc[ti][tj], 0, 0, 0);
        }
      }
    };
    // Four register sets, three steps (twelve rows) of loads ahead of the MFMAs of a step, and no conditions around
    // the loads or the products: a step beyond the wavefront's rows reads row 0 of the block with weight zero.  (With
    // `if (s + 4 < s_end) load(...)` the waits became vmcnt(0): a memory round trip per step.)  The products are
    // accumulated in row order whatever the depth.
    double a0[4], b0[4], a1[4], b1[4], a2[4], b2[4], a3[4], b3[4], w0, w1, w2, w3;
    load(s_begin, a0, b0, w0);
    load(s_begin + 4, a1, b1, w1);
    load(s_begin + 8, a2, b2, w2);
    for (int64_t s = s_begin; s < s_end; s += 16) {
      load(s + 12, a3, b3, w3);
      mma(a0, b0, w0);
      load(s + 16, a0, b0, w0);
      mma(a1, b1, w1);
      load(s + 20, a1, b1, w1);
      mma(a2, b2, w2);
      load(s + 24, a2, b2, w2);
      mma(a3, b3, w3);
    }
  }
  if (row_split) {  // parts 0..3 in order: store, add + store, add + store, add (and write below)
    for (int round = 0; round < 4; ++round) {
      if (part == round && active) {
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
#pragma unroll
          for (int tj = 0; tj < 4; ++tj) {
            if (ti < nti && tj < ntj) {
              if (round > 0) acc[ti][tj] += comb[wave & 1][ti * 4 + tj][lane];
              if (round < 3) comb[wave & 1][ti * 4 + tj][lane] = acc[ti][tj];
            }
          }
      }
      __syncthreads();
    }
  }
  if (!row_split || (part == 3 && active)) {
  // partials are stored tile by tile, the row blocks of one tile next to each other:
  // part[set][tile (I, J)][b][16 x 16] -- the reduce kernel then walks 2 KiB strides, not 2 MiB ones
#pragma unroll
  for (int ti = 0; ti < 4; ++ti) {
    if (ti < ti_lo || ti >= nti) continue;
#pragma unroll
    for (int tj = 0; tj < 4; ++tj) {
      if (tj >= ntj) continue;
      const int64_t tile = (int64_t)(ibase + ti) * WS_TILES + (4 * wj + tj);
      double* out = w.part + (((int64_t)set * (WS_TILES * WS_TILES) + tile) * w.nblk + b) * 256;
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(kk + 4 * r) * 16 + c] = acc[ti][tj][r];
    }
  }
  }
  if (row_split) __syncthreads();  // (the fold buffer is used again by the next slice)
  }
}

// fixed-order sum of the partial Grams, scaled by 1/n_set: the new tile rows and, mirrored, the
// matching columns of the old part.  One workgroup per 16x16 tile (grid (WS_TILES^2, n_sets)); the
// last workgroup publishes the Gram.
static __global__ __launch_bounds__(256) void ws_gram_reduce_kernel(WsArgs w) {
  WsCtl* ws = w.ws;
  if (!ws->building) return;
  const int K = ws->K;
  const int row_lo = (ws->k_new >> 4) << 4;
  const int set = blockIdx.y;
  const int tile = blockIdx.x, I = tile / WS_TILES, J = tile % WS_TILES;
  const int tiles = K >> 4, I_lo = row_lo >> 4;
  if (I < I_lo || I >= tiles || J >= tiles) return;  // (only the working tiles take part in the count below)
  const int i = 16 * I + (threadIdx.x >> 4), j = 16 * J + (threadIdx.x & 15);
  {
    const double* src = w.part + (((int64_t)set * (WS_TILES * WS_TILES) + tile) * w.nblk) * 256 + threadIdx.x;
    // four interleaved chains, 32 loads in flight per thread (the loop is latency-bound); the order
    // of additions is fixed, so the result is reproducible
    double s4[4] = {0.0, 0.0, 0.0, 0.0};
    int b = 0;
    if (w.owner != nullptr) {
      // row blocks whose partial another set computed (ws_block_owner_kernel), or nobody (all rows weigh zero): the same
      // sum in the same order, every term from where it lies.  Offsets of the blocks' partials relative to this set's.
      __shared__ int64_t off_s[512];
      static_assert(sizeof(off_s) >= 8 * 512, "nblk <= 512");
      for (int bb = threadIdx.x; bb < w.nblk; bb += 256) {
        const int o = w.owner[(int64_t)set * w.nblk + bb];
        off_s[bb] = o < 0 ? -1 : (int64_t)(o - set) * (WS_TILES * WS_TILES) * w.nblk * 256;
      }
      __syncthreads();
      for (; b + 32 <= w.nblk; b += 32) {
        double v[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) {
          const int64_t o = off_s[b + u];
          v[u] = o == -1 ? 0.0 : src[(int64_t)(b + u) * 256 + o];
        }
#pragma unroll
        for (int u = 0; u < 32; ++u) s4[u & 3] += v[u];
      }
      for (; b < w.nblk; ++b) {
        const int64_t o = off_s[b];
        s4[b & 3] += o == -1 ? 0.0 : src[(int64_t)b * 256 + o];
      }
    }
    for (; b + 32 <= w.nblk; b += 32) {
      double v[32];
#pragma unroll
      for (int u = 0; u < 32; ++u) v[u] = src[(int64_t)(b + u) * 256];
#pragma unroll
      for (int u = 0; u < 32; ++u) s4[u & 3] += v[u];
    }
    for (; b < w.nblk; ++b) s4[b & 3] += src[(int64_t)b * 256];
    double s = ((s4[0] + s4[1]) + (s4[2] + s4[3])) * w.inv_n[set];
    double* Gs = (w.Gx ? w.Gx : w.Gm) + (int64_t)set * (WS_KCAP * WS_KCAP);
    Gs[i * WS_KCAP + j] = s;
    if (j < row_lo) Gs[j * WS_KCAP + i] = s;
  }
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) {
    const int total = (tiles - I_lo) * tiles * (int)gridDim.y;
    if (atomicAdd(&ws->counter, 1) + 1 == total) {
      ws->counter = 0;
      if (w.Gx) {
        ws->staged = 1;  // this rank's rows only: all-reduce, then ws_publish_kernel
      } else {
        ws->request = 0;
        ws->valid = 1;
        __threadfence();
        ws->building = 0;
      }
    }
  }
}

// Row-sharded mode: after the all-reduce of the staging matrix, move the new rows / columns into the
// Gram and publish it.  grid (WS_PUBLISH_BLOCKS, n_sets): a workgroup walks its share of the K x K corner (as 1 024
// workgroups over the whole 512 x 512 matrix, each with its turn at the one counter, the kernel took 62 us).
constexpr int WS_PUBLISH_BLOCKS = 64;
static __global__ __launch_bounds__(256) void ws_publish_kernel(WsArgs w) {
  WsCtl* ws = w.ws;
  if (!ws->building || !ws->staged) return;
  const int K = ws->K;
  const int row_lo = (ws->k_new >> 4) << 4;
  const int set = blockIdx.y;
  // rows of the corner, WS_KCAP doubles apart; a wavefront moves 64 consecutive columns of one row
  for (int i = blockIdx.x; i < K; i += gridDim.x)
    for (int j = threadIdx.x; j < K; j += 256)
      if (i >= row_lo || j >= row_lo) {
        const int64_t at = (int64_t)set * (WS_KCAP * WS_KCAP) + (int64_t)i * WS_KCAP + j;
        w.Gm[at] = w.Gx[at];
      }
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) {
    const int total = gridDim.x * gridDim.y;
    if (atomicAdd(&ws->counter, 1) + 1 == total) {
      ws->counter = 0;
      ws->staged = 0;
      ws->request = 0;
      ws->valid = 1;
      __threadfence();
      ws->building = 0;
    }
  }
}

// Workgroup sums of the model solver: only the threads of the first `nwc` wavefronts (4 or 8: the ones with q == 0, a
// position each) bring a value, every thread gets bit-identical totals.  block_sum makes all sixteen wavefronts scan
// their zeros and fold sixteen partial sums each: 1.8 us per iteration for seven values, issue-bound on the fp64 DPP adds
// of four wavefronts per SIMD (in-kernel clock marks).  Here the scan runs where the values are, and a lane reads ONE
// partial sum per value and folds it with its quad (or half-row) by commutative pairings.
template <int NV>
__device__ __forceinline__ void ws_sum(double (&v)[NV], double (*lds)[TAIL_WAVES], int nwc) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave < nwc) {
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = wave_sum_lane63(v[k]);
  }
  __syncthreads();  // protect lds from the previous use
  if (wave < nwc && lane == 63) {
#pragma unroll
    for (int k = 0; k < NV; ++k) lds[k][wave] = v[k];
  }
  __syncthreads();
  if (nwc == 4) {
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = group_sum_all<4>(lds[k][lane & 3]);
  } else {
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = group_sum_all<8>(lds[k][lane & 7]);
  }
}

// ---------------------------------------------------------------------------------------------
// (iv) refinement: one workgroup per lane minimises the penalised quadratic model over W.
// Threads q KP + k work on working-set position k (q splits the matrix-vector product; TPC = 4 parts, or 2
// beyond 256 positions).  Up to
// WS_KLDS columns the Gram is copied into LDS first, so an inner iteration never leaves the CU.
// ---------------------------------------------------------------------------------------------
// GROUPED: the dataset has real groups (compiled apart from the per-feature variant: each instance carries one
// direct step, and the registers of the other's never weigh on its iteration loop).
// DIRECT = false: the iteration alone.  A lane whose solve would take a direct step is left, untouched, to the
// DIRECT instance launched right behind (WsCtl::want_full; returns false).  Most solves never take one, and the
// kernel without the factorisation is a fifth of the code, keeps its registers (the full one spills 250 of them
// at 128 per thread) and leaves no scratch lines for the end of the kernel to write back.
// LDS of the model solver: one block per workgroup, handed to ws_refine_lane by the kernel
struct WsSolveLds {
  double delta[WS_KCAP];
  double uim[WS_KCAP];
  double part[WS_THREADS];
  double Gl[WS_KLDS * WS_KLDS];
  // direct step (newton_kernels.hpp)
  NtShared nts;
  double nv[WS_KCAP];   // right-hand side / solution, indexed by rank in the face
  double xsl[WS_KCAP];  // direct step with group norms: the base point,
  double rgl[WS_KCAP];  // the norm of each position's group there,
  double pbl[WS_KCAP];  // and its group weight
  double scale2;        // the length scale of the problem on W (see the iteration)
  int nz[WS_KCAP];
  int act[WS_KCAP];      // position of the ii-th face coordinate
  int rank_of[WS_KCAP];  // rank of a position in the face, or -1
  int nnz, m;
};

template <bool GROUPED, bool DIRECT>
__device__ __forceinline__ bool ws_refine_lane(TailArgs a, const WsArgs& w, double (*red)[TAIL_WAVES], WsSolveLds& sh) {
  double (&delta)[WS_KCAP] = sh.delta;
  double (&uim)[WS_KCAP] = sh.uim;
  int (&nz)[WS_KCAP] = sh.nz;
  int& nnz_s = sh.nnz;
  double (&part)[WS_THREADS] = sh.part;
  double (&Gl)[WS_KLDS * WS_KLDS] = sh.Gl;
  NtShared& nts = sh.nts;
  double (&nv)[WS_KCAP] = sh.nv;
  int (&act)[WS_KCAP] = sh.act;
  int (&rank_of)[WS_KCAP] = sh.rank_of;
  int& m_s = sh.m;
  double (&xsl)[WS_KCAP] = sh.xsl;
  double (&rgl)[WS_KCAP] = sh.rgl;
  double (&pbl)[WS_KCAP] = sh.pbl;
  double& scale2_s = sh.scale2;
  const int lane_id = blockIdx.x;
  PathCtl* ctl = a.ctl + lane_id;
  WsCtl* ws = w.ws;
  if (!ws->valid || ws->building || ws->disabled) return true;
  const int tid = threadIdx.x;
  const int p = a.p;
  const int K = __builtin_amdgcn_readfirstlane(ws->K);
  const int set = w.set_of[lane_id];
  const double* Gm = w.Gm + (int64_t)set * (WS_KCAP * WS_KCAP);
  {
    const int64_t off = (int64_t)lane_id * a.ld;
    a.beta += off; a.z += off; a.zprev += off; a.gprev += off;
    a.a0 += off; a.b0 += off; a.d0 += off;
    a.pts += ctl->pt_off;
  }
  // A lane whose plain step left W (and W could not be extended) is not refined: resetting those
  // coordinates below would undo its progress.  Neither is a lane that keeps being sent back to the
  // same path point (the model solve did not reach the tolerance, e.g. a near-singular Gram): it
  // finishes the point with plain steps.
  if (ws->stale) {
    double out[1] = {0.0};
    for (int j = tid; j < p; j += WS_THREADS)
      if (w.pos[j] < 0 && a.z[j] != a.zprev[j]) out[0] += 1.0;
    block_sum<1>(out, red);
    if (out[0] != 0.0) return true;
  }
  const int point_now = ctl->point + ctl->pt_off;
  // (a point whose refinements keep being sent back because W had to grow -- strongly correlated designs
  // discover their support in waves -- is a different matter from one the model cannot settle)
  const int reps = (ws->last_point[lane_id] == point_now && ws->last_cols[lane_id] == ws->Kreal) ? ws->repeats[lane_id] : 0;
  if (reps >= WS_MAX_REPEATS) return true;
  const int hard_now = w.hard_call ? ws->hard : ws->hard_lane[lane_id];
  if (!DIRECT && w.nt != nullptr && hard_now != 0) {  // a refinement of this lane needed direct steps: so may this one
    if (tid == 0) ws->want_full[lane_id] = 1;
    return false;
  }

  unsigned long long tk_s = wall_clock64();
  auto mark = [&](int slot) {
    if (tid == 0 && lane_id == 0) {
      const unsigned long long now = wall_clock64();
      ws->solve_ticks[slot] += now - tk_s;
      tk_s = now;
    }
  };
  const slm_path_point pt = a.pts[ctl->point];
  const int mode = ctl->mode;
  const double tol = ctl->tol;
  const bool group_pen = (pt.sb != 0.0) || (pt.sd != 0.0);
  const bool g_lds = K <= WS_KLDS;
  if (g_lds) {
    for (int e = tid; e < K * K; e += WS_THREADS) {
      const int r = e / K, c = e - r * K;
      Gl[e] = Gm[r * WS_KCAP + c];
    }
  }

  // 4 threads per position up to 256 positions, 2 beyond (1024 threads, WS_KCAP = 512)
  // Thread q KP + k works on position k, KP = WS_THREADS / TPC: the lanes of a wavefront hold CONSECUTIVE
  // positions and one q, so a Gram row segment is one coalesced 512-byte load.  (With the TPC threads of a
  // position next to each other the lanes alternated between TPC rows 4 KiB apart and every lane became
  // its own memory request: 28 us per product at K = 272, in-kernel clock marks.)
  const int tsh = K <= 256 ? 2 : 1;
  const int TPC = 1 << tsh;
  const int KP = WS_THREADS >> tsh;
  const int k = tid & (KP - 1), q = tid >> (10 - tsh);
  static_assert(WS_THREADS == 1024, "q = tid >> (10 - tsh)");
  const int nwc = KP >> 6;  // wavefronts whose threads account for a position (q == 0): 4 or 8 -- ws_sum
  const int j = k < K ? w.idx[k] : -1;
  const bool live = j >= 0;
  const bool mine = live && q == 0;  // the thread that accounts for position k in reductions
  const int jj = live ? j : 0;
  const double z0 = live ? a.zprev[jj] : 0.0;
  const double g0 = live ? a.gprev[jj] : 0.0;
  const double x_start = live ? a.z[jj] : 0.0;
  const double pa = live ? pt.sa * a.a0[jj] : 0.0;
  const int gsk = live ? w.gs[k] : 0, glk = live ? w.gl[k] : 1;
  const int gix = a.singleton ? jj : a.gid[jj];
  const double pb = live ? pt.sb * a.b0[gix] : 0.0;
  const double pd = live ? pt.sd * a.d0[gix] : 0.0;
  __syncthreads();  // Gl complete

  // G (val - z0) for the vector held as `val` at every position.  Every call is followed by a
  // block_sum before the next one, so delta is never overwritten while it is being read.
  auto matvec = [&](double val, bool dense) -> double {
    if (q == 0) delta[k] = (k < K) ? val - z0 : 0.0;
    __syncthreads();
    double acc = 0.0;
    if (g_lds) {
      if (k < K) {
#pragma unroll 4
        for (int c = q; c < K; c += TPC) acc = __builtin_fma(Gl[c * K + k], delta[c], acc);
      }
    } else if (dense) {
      // Gram through L2 (K > WS_KLDS): the loads of a batch are issued together, then consumed in the same
      // order as before (one FMA chain).  Left to the compiler the loop ran one load at a time: 28 us per
      // product at K = 272 (in-kernel clock marks), i.e. 0.3 ms of power iteration per selection.
      // (the last batch is a full one too, its entries past the end read the batch's first row again and count with a
      //  factor of zero: left to a loop of its own the tail ran one load at a time, 0.2 us each -- twelve of them per
      //  product at K = 176, half the power iteration)
      if (k < K) {
        for (int c = __builtin_amdgcn_readfirstlane(q); c < K; c += 16 * TPC) {
          double gv[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) gv[u] = Gm[(c + u * TPC < K ? c + u * TPC : c) * WS_KCAP + k];
#pragma unroll
          for (int u = 0; u < 16; ++u) acc = __builtin_fma(gv[u], c + u * TPC < K ? delta[c + u * TPC] : 0.0, acc);
        }
      }
    } else {
      if (tid < 64) {  // compact list of the non-zero entries (wave 0, ballots)
        int basep = 0;
        for (int c0 = 0; c0 < K; c0 += 64) {
          const int kk = c0 + tid;
          const bool nzq = kk < K && delta[kk] != 0.0;
          const uint64_t m = __ballot(nzq);
          if (nzq) nz[basep + __popcll(m & ((1ull << tid) - 1ull))] = kk;
          basep += __popcll(m);
        }
        if (tid == 0) nnz_s = basep;
      }
      __syncthreads();
      // (q, the list and its length are the same for all lanes of a wavefront: told to the compiler, the sixteen row
      //  numbers and the in-range tests live in scalar registers)
      const int nnz = __builtin_amdgcn_readfirstlane(nnz_s);
      const int qs = __builtin_amdgcn_readfirstlane(q);
      // (full batches to the end, as above: the headline's solves have 10-60 non-zeros, FEWER than the 16 TPC a batch
      //  used to need -- every one of their products ran in the one-load-at-a-time tail, 2.4 us of a 5.6 us iteration.
      //  Twelve per batch: sixteen cost the kernel without direct steps 28 bytes of scratch.)
      if (k < K) {
        for (int m = qs; m < nnz; m += 12 * TPC) {
          int cc[12];
          double gv[12];
#pragma unroll
          for (int u = 0; u < 12; ++u) cc[u] = __builtin_amdgcn_readfirstlane(nz[m + u * TPC < nnz ? m + u * TPC : m]);
#pragma unroll
          for (int u = 0; u < 12; ++u) gv[u] = Gm[cc[u] * WS_KCAP + k];
#pragma unroll
          for (int u = 0; u < 12; ++u) acc = __builtin_fma(gv[u], m + u * TPC < nnz ? delta[cc[u]] : 0.0, acc);
        }
      }
    }
    // the TPC parts of a position sit in different wavefronts: fold them through LDS, in a fixed order
    part[tid] = acc;
    __syncthreads();
    double tot = part[k];
    for (int qq = 1; qq < TPC; ++qq) tot += part[qq * KP + k];
    return tot;
  };
  // prox of the lane's penalty at the current path point, step s, on the W coordinates
  auto prox_w = [&](double v, double s) -> double {
    double u = live ? soft(v, s * pa) : 0.0;
    if (group_pen) {
      if (a.singleton) {
        const double nrm = fabs(u);
        const double sc = nrm > 0.0 ? fmax(0.0, 1.0 - s * pb / nrm) : 0.0;
        u *= sc / (1.0 + s * pd);
      } else {
        __syncthreads();
        if (q == 0) uim[k] = u;
        __syncthreads();
        double ss = 0.0;
        for (int m = 0; m < glk; ++m) {
          const double t = uim[gsk + m];
          ss = __builtin_fma(t, t, ss);
        }
        const double nrm = sqrt(ss);
        const double sc = (nrm > 0.0 ? fmax(0.0, 1.0 - s * pb / nrm) : 0.0) / (1.0 + s * pd);
        u *= sc;
      }
    }
    return u;
  };
  // penalty value of the vector held as `val` (thread-partial: counted once per position / group)
  auto pen_part = [&](double val) -> double {
    double pv = 0.0;
    if (mine) {
      pv = pa * fabs(val);
      if (group_pen && a.singleton) pv += pb * fabs(val) + 0.5 * pd * val * val;
    }
    if (group_pen && !a.singleton) {
      __syncthreads();
      if (q == 0) uim[k] = live ? val : 0.0;
      __syncthreads();
      if (mine && gsk == k) {  // first member of the group
        double ss = 0.0;
        for (int m = 0; m < glk; ++m) ss = __builtin_fma(uim[k + m], uim[k + m], ss);
        pv += pb * sqrt(ss) + 0.5 * pd * ss;
      }
    }
    return pv;
  };

  mark(0);
  // ---- lambda_max of this Gram (once per selection): power iteration from a fixed start ---------
  double Lw = ws->Lw[set];
  if (!(Lw > 0.0)) {
    double vec = (k < K) ? 1.0 + 0.37 * (double)(((k * 2654435761u) >> 24) & 0xffu) / 255.0 : 0.0;
    double lam = 0.0;
    for (int itp = 0; itp < w.power_iters; ++itp) {
      const double y = matvec(vec + z0, true);  // matvec works on (val - z0)
      double s[1] = {q == 0 && k < K ? y * y : 0.0};
      ws_sum<1>(s, red, nwc);
      lam = sqrt(s[0]);
      vec = lam > 0.0 ? y / lam : 0.0;
    }
    Lw = lam * 1.1;  // from below; the curvature guard in the loop covers the rest
    if (!(Lw > 0.0)) return true;  // empty / zero Gram: nothing to refine
  }

  // ---- direct step: projected Newton on the free coordinates ---------------------------------------
  // Free set F: the non-zero coordinates of x plus the zero ones whose model gradient exceeds their
  // threshold (they want to leave zero); orthant: sign(x), or the side such a coordinate wants to move to.
  // Inside the orthant the model + penalty is a smooth quadratic: d = H_FF^-1 (pseudo-gradient) by a Cholesky
  // solve, then x - t d projected back onto the orthant (a coordinate that would change sign stops at zero),
  // t = 1, 1/2, ... until the model value falls (two-metric projection: F holds no coordinate that the
  // gradient pins at zero, so the projected arc is a descent arc).  When nothing is projected at t = 1 the
  // result IS the minimiser over that face.  Returns 1 when x moved, 0 when there was nothing to do, -1 when
  // H_FF is not positive definite or no trial lowered the model (the iteration simply carries on).  mu_out:
  // estimate of the smallest eigenvalue of the face Hessian (0 = not computed).  This is the variant for
  // per-feature penalties (and singleton "groups"); real group norms: direct_step_group further down.
  constexpr bool group_face = GROUPED;  // real groups: direct_step_group below (it also serves a lane of such a
                                        // dataset whose current point has no group term: b = 0 adds no curvature)
  const bool newton_capable = w.nt != nullptr;
  double* ntF = newton_capable ? w.nt + (int64_t)lane_id * NT_SCRATCH : nullptr;
  double* ntD = newton_capable ? ntF + (int64_t)NT_TILES * 256 : nullptr;
  auto block_min = [&](double val) -> double {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) val = fmin(val, __shfl_xor(val, off, 64));
    __syncthreads();
    if ((tid & 63) == 0) red[0][tid >> 6] = val;
    __syncthreads();
    double mm = red[0][0];
#pragma unroll
    for (int wv = 1; wv < TAIL_WAVES; ++wv) mm = fmin(mm, red[0][wv]);
    return mm;
  };
  int resolve_cap = WS_NEWTON_RESOLVE;
  auto direct_step = [&](double& x, double Lmax, double* mu_out, bool want_mu) -> int {
   if constexpr (GROUPED) {
    return 0;  // (this instance uses direct_step_group)
   } else {
    *mu_out = 0.0;
    const double thr = pa + pb;  // (singleton groups: b acts as a second l1 weight)
    // On an ill-conditioned face the Newton direction lives on cancellations between near-collinear columns:
    // projecting some of its coordinates away leaves a step that is no descent step at any useful length.
    // So the free set is made consistent first, active-set fashion: a zero coordinate stays in F only if the
    // solve moves it to the side it wants to go, a non-zero one only if the solve does not carry it across
    // zero -- the others are set to / kept at zero (and stay out of F for the rest of this call), the
    // pseudo-gradient is re-evaluated there and the system is solved again (WS_NEWTON_RESOLVE times at most:
    // then the projected arc has to do).
    double xb = x;          // base point of the solve: x with the coordinates dropped so far at zero
    bool banned = false;    // this position was dropped: it stays at zero and out of F
    double m_old = 0.0;     // model value at x (relative to the expansion point)
    double d_first = 0.0, t_first = 2.0;  // first solve: direction and the step to the first sign change from x
    bool ok_first = true, tiny_first = false;
    double pg_first = 0.0;
    int m = 0, T = 0, mp = 0, my_rank = -1;
    double dk = 0.0, xi = 1.0;
    unsigned long long tk = wall_clock64();
    auto lap = [&](int slot) {
      if (tid == 0) {
        const unsigned long long now = wall_clock64();
        ws->nt_ticks[lane_id][slot] += now - tk;
        tk = now;
      }
    };
    for (int resolve = 0; resolve < resolve_cap; ++resolve) {
      lap(4);
      const double gdb = matvec(xb, false);  // G (xb - z0)
      const double gx = g0 + gdb;            // model gradient at xb
      if (resolve == 0) {
        double sv[2] = {mine ? (x - z0) * (g0 + 0.5 * gdb) : 0.0, 0.0};
        sv[1] = pen_part(x);
        block_sum<2>(sv, red);
        m_old = sv[0] + sv[1];
      }
      double pg;  // pseudo-gradient of this position at xb; xi: the orthant it may move in
      if (xb != 0.0) {
        xi = xb > 0.0 ? 1.0 : -1.0;
        pg = gx + thr * xi + pd * xb;
      } else {
        pg = fabs(gx) > thr * (1.0 + 1e-12) ? gx - copysign(thr, gx) : 0.0;
        xi = pg > 0.0 ? -1.0 : 1.0;
      }
      // of the zero coordinates that want to leave zero only the strongest enter now (within WS_NEWTON_ENTER of
      // the largest violation): on correlated designs most violators stop violating once a few of them have
      // moved, and a free set full of them solves for a direction that means nothing
      const double viol = (live && !banned && xb == 0.0) ? fabs(pg) : 0.0;
      const double viol_max = -block_min(-viol);
      const bool is_free = live && !banned && (xb != 0.0 || (viol > 0.0 && viol >= WS_NEWTON_ENTER * viol_max));
      // free positions in position order
      __syncthreads();
      if (q == 0 && k < WS_KCAP) {
        nv[k] = is_free ? 1.0 : 0.0;
        rank_of[k] = -1;
      }
      __syncthreads();
      if (tid < 64) {
        int basep = 0;
        for (int c0 = 0; c0 < K; c0 += 64) {
          const int kk = c0 + tid;
          const bool on = kk < K && nv[kk] != 0.0;
          const uint64_t mk = __ballot(on);
          if (on) {
            const int ii = basep + __popcll(mk & ((1ull << tid) - 1ull));
            act[ii] = kk;
            rank_of[kk] = ii;
          }
          basep += __popcll(mk);
        }
        if (tid == 0) m_s = basep;
      }
      __syncthreads();
      m = m_s;
      if (m == 0) {
        if (resolve == 0) return 0;
        dk = 0.0;
        my_rank = -1;
        break;  // everything was dropped: the base point itself is the candidate
      }
      T = (m + 15) >> 4;
      mp = 16 * T;
      my_rank = (k < K) ? rank_of[k] : -1;
      __syncthreads();
      if (tid < mp) nv[tid] = 0.0;
      if (q == 0 && k < WS_KCAP) delta[k] = pd;  // (matvec is done with delta: it now carries the ridge diagonal)
      __syncthreads();
      if (q == 0 && my_rank >= 0) nv[my_rank] = pg;
      lap(0);
      // H_FF, tile by tile (G is symmetric: read along rows)
      const int ntl = T * (T + 1) / 2;
      for (int e = tid; e < ntl * 256; e += WS_THREADS) {
        const int tl = e >> 8, wi = e & 255;
        int I = (int)((sqrtf(8.0f * (float)tl + 1.0f) - 1.0f) * 0.5f);
        while (I * (I + 1) / 2 > tl) --I;
        while ((I + 1) * (I + 2) / 2 <= tl) ++I;
        const int J = tl - I * (I + 1) / 2;
        const int l6 = wi & 63, st = wi >> 6;
        const int ii = 16 * I + (l6 & 15), jj = 16 * J + (l6 >> 4) + 4 * st;
        double hv;
        if (ii < m && jj < m) {
          const int pi = act[ii], pj = act[jj];
          hv = Gm[pj * WS_KCAP + pi];
          if (ii == jj) hv += delta[pi];
        } else {
          hv = ii == jj ? 1.0 : 0.0;
        }
        ntF[e] = hv;
      }
      __syncthreads();
      lap(1);
      if (!nt_factor(ntF, ntD, T, 1e-12 * Lmax, nts)) {
        if (tid == 0) atomicAdd(&ws->newton_nopd, 1);
        return -1;
      }
      lap(2);
      nt_solve(ntF, ntD, T, nv);  // nv = H_FF^-1 pg
      lap(3);
      if (tid == 0) {
        atomicAdd(&ws->newton_factors, 1);
        ws->nt_factors[lane_id] += 1;
        atomicAdd(&ws->newton_unknowns, m);
      }
      dk = my_rank >= 0 ? nv[my_rank] : 0.0;
      if (resolve == 0) {
        ok_first = my_rank >= 0 && (x != 0.0 || dk * pg > 0.0);
        d_first = ok_first ? dk : 0.0;  // (a zero coordinate the first solve sent the wrong way stays where it is)
        pg_first = pg;
        // a coordinate the solve carries across zero ends the straight segment -- unless it sits so close to
        // zero (the dust a prox-gradient step leaves on every violator) that the segment would have no
        // length: those go to zero outright and take no part in the direction
        tiny_first = my_rank >= 0 && x != 0.0 && x * dk > 0.0 && fabs(x) < 1e-6 * fabs(dk);
        if (tiny_first) d_first = 0.0;
        else if (my_rank >= 0 && x != 0.0 && x * dk > 0.0 && fabs(dk) >= fabs(x)) t_first = x / dk;
      }
      // zero coordinates the solve would move to the wrong side (or not at all), non-zero ones it would carry
      // across zero
      const bool wrong = my_rank >= 0 && (xb == 0.0 ? !(dk * pg > 0.0) : (xb - dk) * xi <= 0.0);
      double cnt[1] = {mine && wrong ? 1.0 : 0.0};
      block_sum<1>(cnt, red);
      if (cnt[0] == 0.0 || resolve == resolve_cap - 1) break;
      if (wrong) {
        banned = true;
        xb = 0.0;
      }
    }
    t_first = fmin(1.0, block_min(t_first));
    {
      // the model along x - t d_first, 0 <= t <= t_first (no coordinate changes sign there), is the parabola
      // m_old - t <pg, d> + t^2/2 <d, H d>: with the wrong-way coordinates held back d is not the Newton direction
      // of what moves, so the full segment need not descend -- its minimiser does
      const double gd = matvec(z0 + d_first, false);  // G d_first
      double sv[2] = {0.0, 0.0};
      if (mine) {
        sv[0] = pg_first * d_first;
        sv[1] = d_first * (gd + pd * d_first);
      }
      block_sum<2>(sv, red);
      if (tid == 0) {
        if (!(t_first > 1e-14)) atomicAdd(&ws->newton_ref[0], 1);
        else if (!(sv[0] > 0.0)) atomicAdd(&ws->newton_ref[1], 1);
        else if (!(sv[1] > 0.0)) atomicAdd(&ws->newton_ref[2], 1);
      }
      t_first = (sv[0] > 0.0 && sv[1] > 0.0) ? fmin(t_first, sv[0] / sv[1]) : 0.0;
    }
    // trial points: the base point minus the (projected) step at t = 1, 1/2, 1/4, then -- from x itself, along
    // the first solve -- the straight segment up to the first sign change (a guaranteed descent step: nothing
    // is projected on it)
    double xn = x;
    bool moved = false, projected = false, full = false;
    for (int trial = 0; trial < 4 && !moved; ++trial) {
      if (trial == 3 && !(t_first > 1e-14)) break;
      double xc, cut = 0.0;
      if (trial < 3) {
        const double tt = trial == 0 ? 1.0 : (trial == 1 ? 0.5 : 0.25);
        xc = xb;
        if (my_rank >= 0) {
          xc = xb - tt * dk;
          if (xc * xi <= 0.0) {  // left the orthant (or landed on its boundary): stops at zero
            if (xb != 0.0 || xc != 0.0) cut = 1.0;
            xc = 0.0;
          }
        }
      } else {
        xc = x - t_first * d_first;
        if (tiny_first || (x != 0.0 && xc * x <= 0.0)) xc = 0.0;  // the coordinate that reaches zero there
        cut = 1.0;
      }
      const double gdn = matvec(xc, false);
      double sv[4] = {0.0, 0.0, 0.0, 0.0};
      if (mine) {
        sv[0] = (xc - z0) * (g0 + 0.5 * gdn);
        sv[2] = cut;
        if (!isfinite(xc)) sv[3] = 1.0;
      }
      sv[1] = pen_part(xc);
      block_sum<4>(sv, red);
      if (sv[3] == 0.0 && sv[0] + sv[1] < m_old) {
        moved = true;
        projected = sv[2] > 0.0;
        full = trial == 0;
        xn = xc;
        if (tid == 0) atomicAdd(&ws->newton_trial[trial], 1);
        // dropping every inconsistent coordinate at once did not settle the free set and the step fell back
        // to the segment: on this face (strong cancellations) the following steps go there directly
        if (trial == 3) resolve_cap = 1;
      }
    }
    lap(4);
    if (!moved && tid == 0 && t_first > 1e-14) atomicAdd(&ws->newton_ref[3], 1);
    if (!moved) return -1;
    if (want_mu && full && !projected && m > 0) {  // H_FF is the Hessian on the face of the new point: its
      __syncthreads();                                      // smallest eigenvalue, two inverse-iteration steps
      *mu_out = nt_lambda_min(ntF, ntD, T, mp, nv, red, 2); // from the step itself
      lap(5);
    }
    x = xn;
    return 1;
   }
  };

  // ---- direct step with group norms (GroupLasso, SparseGroupLasso, ridged; round 2) ----------------------
  // On the face of the iterate -- its active groups, inside them the non-zero coordinates with their signs when
  // there is an l1 term -- the objective is smooth but no longer quadratic: the norm of an active group adds the
  // curvature (b_g / r_g)(I - u u^T), r_g = ||x_g||, u = x_g / r_g.  One call is one damped Newton step there:
  // H = G_FF + d + those blocks, d = H^-1 (gradient of the smooth face objective), trial points x - t d for
  // t = 1, 1/2, 1/4, 1/8 with the same projections as above (a coordinate with an l1 kink stops at zero; a group the
  // step would carry through the origin goes to zero as a whole), the first that lowers model + penalty wins.
  // Groups that are zero but want in (||soft(g_g, a)|| > b_g, the strongest violators only) first receive their
  // prox-gradient value -- the penalty has no gradient at a zero group -- and join the face.  The free set is made
  // consistent by dropping what the solve sends the wrong way, as in the l1 case.
  auto group_sum = [&](double val) -> double {  // sum of `val` over the members of this position's group
    __syncthreads();
    if (q == 0 && k < WS_KCAP) uim[k] = (k < K && live) ? val : 0.0;
    __syncthreads();
    double ss = 0.0;
    if (live)
      for (int m2 = 0; m2 < glk; ++m2) ss += uim[gsk + m2];
    return ss;
  };
  auto direct_step_group = [&](double& x, double Lmax, double* mu_out, bool want_mu) -> int {
   if constexpr (!GROUPED) {
    return 0;  // (this instance uses direct_step)
   } else {
    *mu_out = 0.0;
    const bool kink = pa > 0.0;  // this coordinate has an l1 term: its sign is part of the face
    double xb = x;
    bool banned = false;
    double m_old = 0.0;
    int m = 0, T = 0, mp = 0, my_rank = -1;
    double dk = 0.0, xi = 0.0;
    for (int resolve = 0; resolve < resolve_cap; ++resolve) {
      double gx = g0 + matvec(xb, false);
      if (resolve == 0) {
        double sv[2] = {mine ? (x - z0) * (g0 + 0.5 * (gx - g0)) : 0.0, 0.0};
        sv[1] = pen_part(x);
        block_sum<2>(sv, red);
        m_old = sv[0] + sv[1];
      }
      double r2 = group_sum(xb * xb);
      bool g_active = r2 > 0.0;
      const double sk = (live && !banned && !g_active) ? soft(gx, pa) : 0.0;
      const double S = sqrt(group_sum(sk * sk));
      double viol = 0.0;
      if (live && !banned) {
        if (!g_active) viol = fmax(0.0, S - pb);
        else if (xb == 0.0 && kink) viol = fmax(0.0, fabs(gx) - pa);
      }
      const double viol_max = -block_min(-viol);
      if (resolve == 0 && viol_max > 0.0) {
        const bool enter = live && !banned && !g_active && viol > 0.0 && viol >= WS_NEWTON_ENTER * viol_max;
        double cnt[1] = {mine && enter ? 1.0 : 0.0};
        block_sum<1>(cnt, red);
        if (cnt[0] > 0.0) {  // (uniform: every thread takes the same branch)
          if (enter) {
            const double st = 1.0 / Lmax;
            xb = -st * sk * (1.0 - pb / S) / (1.0 + st * pd);
          }
          gx = g0 + matvec(xb, false);
          r2 = group_sum(xb * xb);
          g_active = r2 > 0.0;
        }
      }
      const double rg = sqrt(r2);
      double pg = 0.0;
      bool is_free = false;
      xi = 0.0;
      if (live && !banned && g_active) {
        if (xb != 0.0) {
          xi = kink ? (xb > 0.0 ? 1.0 : -1.0) : 0.0;
          pg = gx + pa * (xb > 0.0 ? 1.0 : -1.0) + (pb / rg + pd) * xb;
          is_free = true;
        } else if (!kink) {
          pg = gx;  // no l1 term: the objective is smooth in this coordinate at zero
          is_free = gx != 0.0;
        } else {
          const double e = fabs(gx) - pa;
          if (e > 0.0 && e >= WS_NEWTON_ENTER * viol_max) {
            pg = gx - copysign(pa, gx);
            xi = pg > 0.0 ? -1.0 : 1.0;
            is_free = true;
          }
        }
      }
      // free positions in position order
      __syncthreads();
      if (q == 0 && k < WS_KCAP) {
        nv[k] = is_free ? 1.0 : 0.0;
        rank_of[k] = -1;
      }
      __syncthreads();
      if (tid < 64) {
        int basep = 0;
        for (int c0 = 0; c0 < K; c0 += 64) {
          const int kk = c0 + tid;
          const bool on = kk < K && nv[kk] != 0.0;
          const uint64_t mk = __ballot(on);
          if (on) {
            const int ii = basep + __popcll(mk & ((1ull << tid) - 1ull));
            act[ii] = kk;
            rank_of[kk] = ii;
          }
          basep += __popcll(mk);
        }
        if (tid == 0) m_s = basep;
      }
      __syncthreads();
      m = m_s;
      if (m == 0) {
        if (resolve == 0) return 0;
        dk = 0.0;
        my_rank = -1;
        break;
      }
      T = (m + 15) >> 4;
      mp = 16 * T;
      my_rank = (k < K) ? rank_of[k] : -1;
      __syncthreads();
      if (tid < mp) nv[tid] = 0.0;
      if (q == 0 && k < WS_KCAP) {
        delta[k] = pd;
        xsl[k] = (k < K && live) ? xb : 0.0;
        rgl[k] = rg;
        pbl[k] = pb;
        nz[k] = gsk;  // (group id; the mat-vec rebuilds its own list when it next runs)
      }
      __syncthreads();
      if (q == 0 && my_rank >= 0) nv[my_rank] = pg;
      const int ntl = T * (T + 1) / 2;
      for (int e = tid; e < ntl * 256; e += WS_THREADS) {
        const int tl = e >> 8, wi = e & 255;
        int I = (int)((sqrtf(8.0f * (float)tl + 1.0f) - 1.0f) * 0.5f);
        while (I * (I + 1) / 2 > tl) --I;
        while ((I + 1) * (I + 2) / 2 <= tl) ++I;
        const int J = tl - I * (I + 1) / 2;
        const int l6 = wi & 63, st = wi >> 6;
        const int ii = 16 * I + (l6 & 15), jj2 = 16 * J + (l6 >> 4) + 4 * st;
        double hv;
        if (ii < m && jj2 < m) {
          const int pi = act[ii], pj = act[jj2];
          hv = Gm[pj * WS_KCAP + pi];
          if (ii == jj2) hv += delta[pi];
          if (nz[pi] == nz[pj]) {  // same group: curvature of its norm
            const double rr = rgl[pi];
            hv += (pbl[pi] / rr) * ((ii == jj2 ? 1.0 : 0.0) - xsl[pi] * xsl[pj] / (rr * rr));
          }
        } else {
          hv = ii == jj2 ? 1.0 : 0.0;
        }
        ntF[e] = hv;
      }
      __syncthreads();
      if (!nt_factor(ntF, ntD, T, 1e-12 * Lmax, nts)) {
        if (tid == 0) atomicAdd(&ws->newton_nopd, 1);
        return -1;
      }
      nt_solve(ntF, ntD, T, nv);
      if (tid == 0) {
        atomicAdd(&ws->newton_factors, 1);
        atomicAdd(&ws->newton_unknowns, m);
        ws->nt_factors[lane_id] += 1;
      }
      dk = my_rank >= 0 ? nv[my_rank] : 0.0;
      // what the solve sends the wrong way: a kinked coordinate across (or to the wrong side of) zero, a whole
      // group through the origin
      bool wrong = false;
      if (my_rank >= 0 && kink) wrong = xb == 0.0 ? !(dk * pg > 0.0) : (xb - dk) * xi <= 0.0;
      const double radial = group_sum(live && !banned ? (xb - dk) * xb : 0.0);
      const bool g_wrong = live && !banned && g_active && !(radial > 0.0);
      double cnt[1] = {mine && (wrong || g_wrong) ? 1.0 : 0.0};
      block_sum<1>(cnt, red);
      if (cnt[0] == 0.0 || resolve == resolve_cap - 1) break;
      if (wrong || g_wrong) {
        banned = true;
        xb = 0.0;
      }
    }
    double xn = x;
    bool moved = false, projected = false, full = false;
    double tt = 1.0;
    for (int trial = 0; trial < 4 && !moved; ++trial, tt *= 0.5) {
      double xc = xb, cut = 0.0;
      if (my_rank >= 0) {
        xc = xb - tt * dk;
        if (kink && xc * xi <= 0.0) {
          if (xb != 0.0 || xc != 0.0) cut = 1.0;
          xc = 0.0;
        }
      }
      const double radial = group_sum(live ? xc * xb : 0.0);
      const double r2b = group_sum(xb * xb);
      if (live && r2b > 0.0 && !(radial > 0.0)) {  // the group would pass through the origin: it goes to zero
        if (xc != 0.0) cut = 1.0;
        xc = 0.0;
      }
      const double gdn = matvec(xc, false);
      double sv[4] = {0.0, 0.0, 0.0, 0.0};
      if (mine) {
        sv[0] = (xc - z0) * (g0 + 0.5 * gdn);
        sv[2] = cut;
        if (!isfinite(xc)) sv[3] = 1.0;
      }
      sv[1] = pen_part(xc);
      block_sum<4>(sv, red);
      if (sv[3] == 0.0 && sv[0] + sv[1] < m_old) {
        moved = true;
        projected = sv[2] > 0.0;
        full = trial == 0;
        xn = xc;
        if (tid == 0) atomicAdd(&ws->newton_trial[trial < 3 ? trial : 2], 1);
      }
    }
    if (!moved) return -1;
    if (want_mu && full && !projected && m > 0) {
      __syncthreads();
      *mu_out = nt_lambda_min(ntF, ntD, T, mp, nv, red, 2);
    }
    x = xn;
    return 1;
   }
  };

  mark(1);
  // ---- FISTA on the model ------------------------------------------------------------------------
  double L = Lw;
  double Ls = L;  // curvature the next step is taken with (L, or less while the steps are spectral)
  double x = x_start, v = x_start, t = 1.0;
  double v_prev = 0.0, gv_prev = 0.0;
  bool have_prev = false;
  mark(2);
  bool ok = true;
  bool settled = false;  // the iteration met its own tolerance: its point minimises the model
  int n_inner = 0;
  // smallest Rayleigh quotient <dv, G dv> / <dv, dv> along the moves of the iteration: an upper estimate
  // of the smallest eigenvalue on the face that closes in as the slow modes come to dominate the moves
  double rq_min = 0.0;
  int rq_n = 0;
  double mu_face = 0.0;       // from the factor of a direct step (0: none taken)
  int since_direct = 0, n_direct = 0, n_direct_bad = 0;
  bool direct_on = newton_capable;
  // direct mode: the iterate only moves by direct steps; the prox-gradient step of every round is just the
  // convergence test (taken when it passes).  Taking it regardless would wreck the next direct step: from a
  // face minimiser one prox-gradient step gives EVERY violator a tiny non-zero value, and a free set full of
  // those solves for a direction that means nothing.  A solve starts in this mode when an earlier refinement of ITS LANE
  // of this call needed direct steps (WsCtl::hard_lane).
  bool direct_mode = direct_on && hard_now != 0;
  // the length scale of the problem on W: a gradient step from the expansion point, ||g0_W|| / L (what "rounding level" is
  // measured against where the iterate itself is zero or dust)
  // (kept in LDS, not in a register across the loop: the kernel sits at the 128 registers of a 1 024-thread workgroup, and a
  //  value more across the iteration was 12 bytes of scratch per thread)
  {
    double sg[1] = {mine ? g0 * g0 : 0.0};
    ws_sum<1>(sg, red, nwc);
    if (tid == 0) scale2_s = sg[0] / (Lw * Lw);
    __syncthreads();
  }
#define scale2 scale2_s
  // The first WS_BB_ITERS steps carry no momentum and take their length from the curvature along the move
  // before: on the well-conditioned faces of an easy path that is there in half the steps of the accelerated
  // iteration, which takes over if it is not.
  bool spectral = !direct_mode && w.bb_steps != 0;
  int it_end = WS_INNER_MAX;
  for (int it = 0; it < it_end; ++it) {
    ++n_inner;
    const double gv = g0 + matvec(v, false);
    const double u = prox_w(v - gv / Ls, 1.0 / Ls);
    //  s[0] = ||u - v||^2  s[1] = ||u||^2  s[2] = (v - u).(u - x)  s[3] = #non-finite
    //  s[4] = ||v - v_prev||^2  s[5] = ||gv - gv_prev||^2   (curvature along the last move of v)
    //  s[6] = <v - v_prev, gv - gv_prev>
    double s[7] = {0, 0, 0, 0, 0, 0, 0};
    if (mine) {
      const double r = u - v;
      s[0] = r * r;
      s[1] = u * u;
      s[2] = -r * (u - x);
      if (!isfinite(u)) s[3] = 1.0;
      if (have_prev) {
        const double dv = v - v_prev, dg = gv - gv_prev;
        s[4] = dv * dv;
        s[5] = dg * dg;
        s[6] = dv * dg;
      }
    }
    ws_sum<7>(s, red, nwc);
    if (s[3] > 0.0 || !isfinite(s[0])) {
      ok = false;
      break;
    }
    v_prev = v;
    gv_prev = gv;
    have_prev = true;
    if (s[4] > 1e-20 * fmax(s[1], scale2) && s[4] > 0.0) {  // (a move at the rounding level measures nothing)
      const double rq = s[6] / s[4];
      if (rq > 0.0 && (rq_n == 0 || rq < rq_min)) rq_min = rq;
      rq_n += 1;
    }
    // (a move at the rounding level measures no curvature either -- and "rounding level" has to be told on the scale of
    //  the PROBLEM, not of the iterate: the point at alpha_max solves to rounding dust (|g_j| - alpha = 1e-16 for the
    //  first feature), its moves are dust against dust, ||dg|| / ||dv|| of one of them sent L from 1.7 to 44 -- through
    //  WsCtl::Lw for every later refinement of the call, whose Rayleigh quotients then all "said" ill-conditioned: 170
    //  direct steps of 200 unknowns on an iid design, 26 ms for an 8-pass path, soak seed 29)
    const bool real_move = s[4] > 1e-20 * fmax(s[1], scale2) && s[4] > 0.0;
    if (real_move && sqrt(s[5] / s[4]) > L) {  // the bound was too low
      L = 1.05 * sqrt(s[5] / s[4]);
      if (!spectral) {  // an accelerated step of 1/L was too long: redo it from x (a spectral step claims nothing of L)
        Ls = L;
        v = x;
        t = 1.0;
        continue;
      }
    }
    if (spectral) {
      // the next step is as long as the curvature along this move allows (Barzilai-Borwein, first form),
      // never longer than WS_BB_MAX_STEP steps of 1/L
      const double rq = s[4] > 1e-20 * s[1] ? s[6] / s[4] : L;
      Ls = fmin(L, fmax(rq, L / WS_BB_MAX_STEP));
      if (it + 1 >= WS_BB_ITERS) {  // not there in the steps such a face takes: momentum from here
        spectral = false;
        Ls = L;
      }
    }
    // (a spectral step is longer than 1/L and moves at least as far from the same point: the test is the stricter for it)
    // (... or the step is rounding noise on the problem's scale, the floor of fista_tail_kernel's stopping rule: a solution
    //  that IS dust -- the path's first point -- has converged, it does not iterate thirty times and take a direct step)
    const bool inner_conv = sqrt(s[0]) <= fmax(WS_INNER_TOL * tol * sqrt(s[1]), kRoundFloor * (sqrt(scale2) + sqrt(s[1])));
    if (DIRECT && direct_mode && !inner_conv) {
      bool stepped = false;
      if (n_direct < WS_NEWTON_MAX) {
        double mu_new = 0.0;
        const int rc = group_face ? direct_step_group(x, L, &mu_new, mu_face == 0.0) : direct_step(x, L, &mu_new, mu_face == 0.0);
        n_direct += 1;
        if (rc > 0) {
          if (mu_new > 0.0) mu_face = mu_new;
          stepped = true;
        } else if (rc < 0) {
          n_direct_bad += 1;
        }
      }
      if (stepped) {
        v = x;
        t = 1.0;
        continue;
      }
      // no usable step from this point (e.g. the coordinates it had to hold back carried the descent): one
      // prox-gradient step moves the iterate somewhere else and the next round tries again; after
      // WS_NEWTON_REFUSALS of those, or at the cap, the iteration finishes the job
      if (n_direct_bad >= WS_NEWTON_REFUSALS || n_direct >= WS_NEWTON_MAX) {
        direct_mode = false;
        direct_on = false;
        since_direct = 0;
        it_end = min(it_end, it + WS_AFTER_DIRECT);
      }
    }
    const bool restart = s[2] > 0.0;
    const double t_use = restart ? 1.0 : t;
    const double t_new = 0.5 * (1.0 + sqrt(1.0 + 4.0 * t_use * t_use));
    const double mom = spectral ? 0.0 : (t_use - 1.0) / t_new;
    v = u + mom * (u - x);
    x = u;
    t = t_new;
    if (inner_conv) {
      settled = true;
      break;
    }
    since_direct += 1;
    if (direct_on && n_direct < WS_NEWTON_MAX &&
        (since_direct >= WS_NEWTON_AFTER || (rq_n >= 5 && rq_min < WS_NEWTON_RQ * L))) {
      if (!DIRECT) {  // (nothing of this solve has been written yet)
        if (tid == 0) ws->want_full[lane_id] = 1;
        return false;
      }
      double mu_new = 0.0;
      const int rc = group_face ? direct_step_group(x, L, &mu_new, mu_face == 0.0) : direct_step(x, L, &mu_new, mu_face == 0.0);
      since_direct = 0;
      n_direct += 1;
      if (rc > 0) {
        if (mu_new > 0.0) mu_face = mu_new;
        v = x;
        t = 1.0;
        direct_mode = true;
      } else if (rc < 0) {
        n_direct_bad += 1;
        if (n_direct_bad >= WS_NEWTON_REFUSALS) {  // singular face / useless steps: the iteration finishes the job
          direct_on = false;
          if (n_direct > n_direct_bad) it_end = min(it_end, it + WS_AFTER_DIRECT);
        }
      }
    }
  }
#undef scale2
  mark(3);
  if (!ok) return true;
  // An iteration that ran out of steps is accepted only if the model says its point is no worse than the start (two
  // products with the Gram, 14 us per call).  Not spent on a point that met the tolerance -- a minimiser of the model is
  // no worse than anything -- unless the START already met it (one iteration): that point is one proximal step from
  // where the lane stood, now and then a hair worse, and taking it resets the lane's step history for nothing.  On
  // paths whose ends outgrow the working set such solves are common (a fifth to a half of all) and accepting them
  // unseen cost 4-6 passes of 24-58 (tools/headline_soak.py); the headline path has none.
  if (!settled || n_inner == 1) {
    double m_start;  // model values relative to the expansion point
    {
      const double gd = matvec(x_start, false);
      const double d = x_start - z0;
      double s[2] = {mine ? d * (g0 + 0.5 * gd) : 0.0, 0.0};
      s[1] = pen_part(x_start);
      ws_sum<2>(s, red, nwc);
      m_start = s[0] + s[1];
    }
    const double gd = matvec(x, false);
    const double d = x - z0;
    double s[3] = {mine ? d * (g0 + 0.5 * gd) : 0.0, 0.0, mine && !isfinite(x) ? 1.0 : 0.0};
    s[1] = pen_part(x);
    ws_sum<3>(s, red, nwc);
    const double m_end = s[0] + s[1];
    if (s[2] > 0.0 || !(m_end <= m_start)) return true;
  }
  // the refined point: model minimiser on W, the expansion point elsewhere (eight features per round: their loads
  // go out together -- one feature at a time every load waited for the one before it, 11 us per call)
  for (int f0 = tid; f0 < p; f0 += 8 * WS_THREADS) {
    int ps[8];
    double zo[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int f = f0 + u * WS_THREADS;
      const int ff = f < p ? f : 0;
      ps[u] = f < p ? w.pos[ff] : 0;
      zo[u] = a.zprev[ff];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int f = f0 + u * WS_THREADS;
      if (f < p && ps[u] < 0) {
        a.z[f] = zo[u];
        if (mode == 0) a.beta[f] = zo[u];
      }
    }
  }
  if (mine) {
    a.z[jj] = x;
    if (mode == 0) a.beta[jj] = x;
  }
  if (tid == 0) {
    if (mode == 1) ctl->have_base = 0;  // the refined point becomes the base of the spectral scheme
    else ctl->t = 1.0;
    ctl->zzero = 0;
    ws->last_point[lane_id] = point_now;
    ws->served[lane_id] = 1;
    ws->repeats[lane_id] = reps + 1;
    ws->last_cols[lane_id] = ws->Kreal;
    // (the lanes of a set end within microseconds of each other: read-compare-write let the smaller of two bounds
    // land last now and then, and the next pass started from a different L -- positive doubles order as integers)
    if (L > 0.0) atomicMax(reinterpret_cast<unsigned long long*>(&ws->Lw[set]), (unsigned long long)__double_as_longlong(L));
    atomicAdd(&ws->refined, 1);
    atomicAdd(&ws->inner_iters, n_inner);
    atomicAdd(&ws->iters_hist[n_inner < 31 ? n_inner : 31], 1);
    if (n_direct) atomicAdd(&ws->newton_steps, n_direct - n_direct_bad);
    if (n_direct > n_direct_bad) {
      ws->hard_next = 1;  // (same value from every lane: the order of the stores is immaterial)
      ws->hard_lane[lane_id] = 1;
    }
    if (n_direct_bad) atomicAdd(&ws->newton_fails, n_direct_bad);
    // strong convexity on the face of the refined point, for the stopping rule of the pass that verifies it
    // (fista_tail_kernel): from the factor when a direct step stood, else from the iteration's own moves
    // (halved: both are estimates from above); 0 = unknown.
    ctl->mu = mu_face > 0.0 ? 0.5 * mu_face : (rq_n >= 3 ? 0.5 * rq_min : 0.0);
  }
  mark(4);
  return true;
}

// MODE 0: the iteration alone (lanes that would take a direct step are left, untouched, with WsCtl::want_full set);
// MODE 1: the solver with direct steps, for the lanes MODE 0 left -- a launch of its own behind it.  (Round 6 tried both in
// one launch: the idle second launch is 4.8 us of the chain between two passes, but the light instance compiled into one
// kernel with the factorisation was no faster for per-feature penalties and 0.35 ms per pass slower for grouped ones --
// its registers went to scratch memory; tools/ab_knobs.py, profiles/r06_fusion_ab.txt.)
template <bool GROUPED, int MODE>
__global__ __launch_bounds__(WS_THREADS) void ws_solve_kernel(TailArgs a, WsArgs w) {
  __shared__ double red[8][TAIL_WAVES];
  __shared__ WsSolveLds sh;
  const int lane_id = blockIdx.x;
  PathCtl* ctl = a.ctl + lane_id;
  if (ctl->done != 0 || ctl->idle != 0 || a.gdone[0] != 0) return;
  const unsigned long long tk_in = wall_clock64();
  if (MODE == 1) {  // only the lanes the light kernel left
    const int mine = w.ws->want_full[lane_id] | w.one_solver;
    __syncthreads();
    if (!mine) return;
    if (threadIdx.x == 0) w.ws->want_full[lane_id] = 0;
  }
  // (every return inside is taken by the whole workgroup)
  if (MODE == 0) {
    if (!ws_refine_lane<GROUPED, false>(a, w, red, sh)) return;
  } else {
    (void)ws_refine_lane<GROUPED, true>(a, w, red, sh);
  }
  __syncthreads();
  // Is the point the next pass evaluates zero outside W?  Then its residual needs only the gathered
  // columns (resid_ws_kernel) and the pass over X is the accumulate-only xtr_ring_kernel.
  const WsCtl* ws = w.ws;
  const bool w_ok = ws->valid && !ws->building && !ws->disabled;
  double out[1] = {0.0};
  if (w_ok) {
    const double* z = a.z + (int64_t)lane_id * a.ld;
    for (int j0 = threadIdx.x; j0 < a.p; j0 += 8 * WS_THREADS) {
      int ps[8];
      double zj[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int j = j0 + u * WS_THREADS;
        const int jj = j < a.p ? j : 0;
        ps[u] = j < a.p ? w.pos[jj] : 0;
        zj[u] = z[jj];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (ps[u] < 0 && zj[u] != 0.0) out[0] += 1.0;
    }
  }
  block_sum<1>(out, red);
  if (threadIdx.x == 0) {
    ctl->zsup = (w_ok && out[0] == 0.0) ? 1 : 0;
    w.ws->lane_ticks[lane_id] += wall_clock64() - tk_in;
  }
}

// ---------------------------------------------------------------------------------------------
// Hold-out scoring of sparse coefficient vectors (grid searches): gather the union of their supports
// once (<= WS_KCAP columns, same tiled gather as above, explicit arguments), then one thread per row
// computes up to SSE_M residuals from K contiguous doubles.  A dense evaluation costs a pass over X
// per four vectors; this costs n x K doubles per sixteen.
// ---------------------------------------------------------------------------------------------
constexpr int SSE_M = 16;

struct GatherArgs {
  const double* X;
  const double* XT;  // nullptr: read the row-major X
  int64_t n, ld;
  const int32_t* idx;  // [K] columns (all >= 0)
  int K;               // multiple of 16 is not required
  double* XW;          // [n][WS_KCAP]
};

static __global__ __launch_bounds__(256) void gather_cols_kernel(GatherArgs w) {
  const int k0 = 32 * (int)blockIdx.y;
  if (k0 >= w.K) return;
  __shared__ double tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int64_t row_tiles = (w.n + 31) / 32;
  for (int64_t rt = blockIdx.x; rt < row_tiles; rt += gridDim.x) {
    const int64_t i0 = rt * 32;
    __syncthreads();
    for (int kk = ty; kk < 32; kk += 8) {
      const int k = k0 + kk;
      const int j = k < w.K ? w.idx[k] : -1;
      const int64_t i = i0 + tx;
      tile[kk][tx] = (j >= 0 && i < w.n) ? (w.XT ? w.XT[((rt * w.ld + j) << 5) + tx] : w.X[i * w.ld + j]) : 0.0;
    }
    __syncthreads();
    for (int ii = ty; ii < 32; ii += 8) {
      const int64_t i = i0 + ii;
      const int k = k0 + tx;
      if (i < w.n && k < w.K) w.XW[i * WS_KCAP + k] = tile[tx][ii];
    }
  }
}

struct SseArgs {
  const double* XW;  // [n][WS_KCAP]
  const double* y;
  const double* rw;  // nullptr: ones
  const double* Zs;  // [m][K] coefficients on the gathered columns
  double* partial;   // [gridDim.x][SSE_M]
  int64_t n;
  int K, m;          // m <= SSE_M vectors in this launch
};

static __global__ __launch_bounds__(256) void sse_sparse_kernel(SseArgs a) {
  extern __shared__ double zs[];  // [K][SSE_M]
  __shared__ double wsum[4][SSE_M];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int e = tid; e < a.K * SSE_M; e += 256) {
    const int k = e / SSE_M, v = e - k * SSE_M;
    zs[e] = v < a.m ? a.Zs[(int64_t)v * a.K + k] : 0.0;
  }
  __syncthreads();
  double sse[SSE_M];
#pragma unroll
  for (int v = 0; v < SSE_M; ++v) sse[v] = 0.0;
  for (int64_t row = (int64_t)blockIdx.x * 256 + tid; row < a.n; row += (int64_t)gridDim.x * 256) {
    const double w = a.rw ? a.rw[row] : 1.0;
    if (w == 0.0) continue;  // (a fold mask: four fifths of the rows)
    const double* xr = a.XW + row * WS_KCAP;
    double dot[SSE_M];
#pragma unroll
    for (int v = 0; v < SSE_M; ++v) dot[v] = 0.0;
    for (int k = 0; k < a.K; ++k) {
      const double x = xr[k];
#pragma unroll
      for (int v = 0; v < SSE_M; ++v) dot[v] = __builtin_fma(x, zs[k * SSE_M + v], dot[v]);
    }
    const double yi = a.y[row];
#pragma unroll
    for (int v = 0; v < SSE_M; ++v) {
      const double e = dot[v] - yi;
      sse[v] = __builtin_fma(w * e, e, sse[v]);
    }
  }
#pragma unroll
  for (int v = 0; v < SSE_M; ++v) {
    double t = sse[v];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) t += __shfl_xor(t, off, 64);
    if (lane == 0) wsum[wave][v] = t;
  }
  __syncthreads();
  if (tid < SSE_M) a.partial[(int64_t)blockIdx.x * SSE_M + tid] = wsum[0][tid] + wsum[1][tid] + wsum[2][tid] + wsum[3][tid];
}

}  // namespace slm

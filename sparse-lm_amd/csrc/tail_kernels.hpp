// O(p) kernels that close one FISTA iteration on the device: proximal step for the sparse-lm
// penalty family, momentum/restart, convergence test and the regularisation-path state machine.
//
// The penalty family (reference src/sparselm/model/_lasso.py:99-107, 267-275,
// 627-639, 795-811 and model/_adaptive_lasso.py:167-175, 354-362, 670-684):
//     sum_j a_j |b_j| + sum_g b_g ||b_g||_2 + 1/2 sum_g d_g ||b_g||_2^2
// prox with step s:  u = soft(v, s a);  per group  u_g * max(0, 1 - s b_g/||u_g||) / (1 + s d_g).
//
// One workgroup of 1024 threads runs the whole O(p) tail (p is a few thousand: 40 KB vectors that
// live in L2), so every reduction is a fixed-order tree and results are bit-reproducible.  Each
// thread keeps its E = ceil(p/1024) features in registers from the first load to the last store.
// Per-group l2 norms use sub-wavefront "teams" of TW lanes (TW = power of two <= 64 chosen from the
// largest group): a team strides over one group's members, gathering them from an LDS image of the
// thresholded vector through the group-sorted permutation (arbitrary non-contiguous labels cost
// nothing), and finishes with a TW-wide xor-butterfly.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/slm_engine.h"
#include "grad_kernel.hpp"  // DPP reductions

namespace slm {

constexpr int TAIL_THREADS = 1024;
constexpr int TAIL_WAVES = TAIL_THREADS / 64;

// Device-resident control block of a path solve.  The host only ever reads copies of it.
struct PathCtl {
  int32_t point;       // current path point
  int32_t n_points;    // end (exclusive) of the range of points this lane is walking
  int32_t iter;        // iterations spent on the current point
  int32_t max_iter;
  int32_t done;        // 1 => every later kernel of the queue returns immediately
  int32_t nonfinite;   // 1 => a non-finite iterate was seen (path aborted)
  int32_t l_bumps;     // times the curvature guard raised L
  int32_t restarts;
  int64_t total_iter;  // gradient evaluations consumed so far (all points)
  double t;            // FISTA momentum scalar
  double L;            // Lipschitz constant in use
  double tol;
  uint32_t flags;
  int32_t pt_off;      // index of this lane's first point in the concatenated point/output arrays
  // spectral (Barzilai-Borwein) mode, see fista_tail_kernel
  int32_t mode;        // 1 = spectral steps with non-monotone acceptance, 0 = FISTA
  int32_t have_base;   // the gradient at the base point beta is known (gprev)
  int32_t rejects;     // candidates rejected so far in this solve (fallback trigger)
  int32_t n_hist;      // valid entries of hist[]
  double ak;           // current inverse step (curvature estimate along the last step)
  double Lhat;         // largest curvature ||dg||/||dz|| seen (lower bound of lambda_max)
  double pen_z;        // penalty value at the candidate z
  double hist[5];      // last accepted objective values (non-monotone reference)
  int32_t pt_lo;       // first point of the range being walked (secant starts need two solved points in it)
  int32_t steals;      // ranges this lane took over from busier lanes
  int32_t idle;        // shared-path mode: finished its range, waiting for steal_kernel to hand out work
  int32_t zzero;       // z has not moved from the all-zero start (split pass: residual = -y, no read of X)
  int32_t stride;      // points between this lane's consecutive path points: 1 (a contiguous range), or the
                       // number of lanes when the lanes of a shared path take its points in turn
  int32_t zsup;        // z is zero outside the working set: its residual can come from the gathered
                       // columns (split_kernels.hpp); maintained by ws_solve_kernel
  int32_t stop_seen;   // row-sharded mode: the stop word of this lane as the last all-reduce delivered it
  double mu;           // strong-convexity estimate on the face of the point under verification, set by the
                       // working-set model solver (0 = unknown); reset when the lane moves to the next point
  double mu_rq;        // smallest Barzilai-Borwein curvature <dz, dg> / <dz, dz> accepted on this point (0 = none yet)
  int32_t tail_pt;     // interleaved lanes: one more point after the lane's regular walk, or -1 (the points beyond
                       // the last full band of a shared path go to the lanes that have just solved their neighbours)
  int32_t pad2_;
  double loss_base;    // smooth loss at zprev, the point whose gradient gprev holds: with them a later solve that starts
                       // where this one ended needs no pass over the data for its first step (solve_core, "carried start")
  // re-weighted rounds on chip (slm_solve_lanes_reweighted, small_kernels.hpp): the lane's points are the rounds of an
  // Adaptive* estimator; after each the kernel renews the penalty weights from the solution
  double rw_coef;      // a_j <- rw_coef * (rw_numer / (|beta_j| + rw_eps)) for j < rw_ncoef   (rw_on & 1)
  double rw_numer;     // b_g <- gscale[g] * (rw_numer / (||beta_g|| + rw_eps)) for g < rw_ngroup   (rw_on & 2)
  double rw_eps;
  double rw_tol;       // the rounds end when the weights moved by no more than this (2-norm)
  int32_t rw_ncoef, rw_ngroup;
  int32_t rw_on;
  int32_t rounds;      // out: rounds run
};

constexpr int BB_HIST = 5;
constexpr int BB_REJECT_LIMIT = 3;    // rejected candidates before a lane falls back to FISTA
constexpr int BB_POINT_LIMIT = 60;    // spectral iterations on one point before falling back
constexpr double BB_SIGMA = 1e-4;
// (round 2: the floor also counts the noise of the gradient itself, kRoundFloor * curvature * ||beta|| -- near a
// minimiser ||g|| is of the size of the penalty while the rounding error of X^T (X beta - y) / n scales with
// lambda_max ||beta||; on weakly convex faces (p > n, tol 1e-12) the rule otherwise asks for a residual below that
// noise and is met, or not, by the luck of the summation order)
// Stopping rule floor: ||prox step|| <= kRoundFloor * ||g|| / curvature is the rounding noise of the
// step itself (16 ulp of the gradient); below it `tol * ||beta||` cannot be met in fp64 when the
// minimiser is itself a rounding-level number (alpha ~ alpha_max).
constexpr double kRoundFloor = 16.0 * 2.220446049250313e-16;
// Strong-convexity estimates below kMuFloor * lambda_max are not trusted: p > n problems, duplicated
// columns -- the objective is then flat along some direction of the face (mu = 0: the minimiser is not
// unique and no residual bounds the distance to "it"); the rule then bounds the residual itself.
constexpr double kMuFloor = 1e-6;

// One workgroup per lane (blockIdx.x): vectors of lane l start at l * ld (g: l * (ld + 16)).
struct TailArgs {
  PathCtl* ctl;               // [n_lanes]
  int* gdone;                 // [0] = every lane finished (or abort): every later kernel returns at once,
                              // [1] = lanes finished so far, [2] = most passes spent on one point so far,
                              // [3] = row-sharded mode: THIS rank has finished (see done_slot),
                              // [4] = row-sharded mode: the ranks' states differ (stop_apply_kernel)
  int done_slot;              // where this rank records "finished": 0 -- it takes effect at once -- or, in
                              // row-sharded mode, 3: [0] is then only set by stop_apply_kernel, from a word
                              // the ranks have all-reduced, so that every rank stops after the same pass
                              // (and enters the same number of collectives) whatever its own state says
  int n_lanes;
  int provisional;            // 1 => the gradient of this call is an estimate (the first call of a path that opens on a row
                              // sample, solve_core "sample start"): it steers the step and the working set, and no point
                              // is accepted on it
  int steal;                  // 1 => all lanes walk ONE path: an idle lane takes over the upper half of
                              //      the points the busiest lane has not reached yet (cold start)
  const slm_path_point* pts;  // concatenated over lanes
  int p;
  int G;
  int singleton;     // 1 => every feature its own group (gidx == identity)
  int team;          // lanes per group team (power of two, 1..64)
  double* beta;      // [ld] current iterate x_k
  double* z;         // [ld] extrapolated point y_k (gradient is evaluated here)
  const double* g;   // [ld+16] gradient at z; g[ld] = loss at z
  int64_t ld;
  double* zprev;     // [ld]
  double* gprev;     // [ld]
  double* gscale;    // [G] scratch
  double* uscratch;  // [ld] scratch per lane (only used when p > 16384)
  const double* a0;  // [p]
  const double* b0;  // [G]
  const double* d0;  // [G]
  const int* order;  // [p] feature index of the k-th element in group-sorted order
  const int* gid;    // [p] group of feature j
  const int* gstart; // [G+1]
  double* betas_out; // [total points][p]
  double* gn_out;    // [total points][G] or nullptr
  slm_point_info* infos;  // [total points]
};

__device__ __forceinline__ double soft(double v, double thr) {
  const double m = fabs(v) - thr;
  return m <= 0.0 ? 0.0 : copysign(m, v);  // NaN propagates (NaN <= 0 is false)
}

// Sum NV values over the 1024-thread workgroup; every thread gets bit-identical totals.
// Stage 1: DPP scan per wavefront (wave_sum_lane63: no LDS round trips).  Stage 2: the 16 wavefront
// totals go through LDS, every wavefront scans them in its first row of 16 lanes with the same DPP
// pattern and broadcasts lane 15 (v_readlane), so all threads hold the same bits.  (The first version used 64- and 16-wide xor butterflies of
// ds_bpermute: ten dependent LDS round trips per value; this one took the tail kernel from 24 us to
// the figure in DESIGN.md.)
template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double (*lds)[TAIL_WAVES]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] = wave_sum_lane63(v[k]);
  __syncthreads();  // protect lds from the previous use
  if (lane == 63) {
#pragma unroll
    for (int k = 0; k < NV; ++k) lds[k][wave] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    double t = lds[k][lane & (TAIL_WAVES - 1)];
    t = dpp_add<0x111, 0xf>(t);  // inclusive scan inside the row of 16 lanes: lane 15 gets the total
    t = dpp_add<0x112, 0xf>(t);
    t = dpp_add<0x114, 0xf>(t);
    t = dpp_add<0x118, 0xf>(t);
    const int lo = __builtin_amdgcn_readlane(__double2loint(t), 15);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(t), 15);
    v[k] = __hiloint2double(hi, lo);
  }
}

// Per-group sum of squares by teams of `team` lanes: group g owns the features order[gstart[g] ..
// gstart[g+1]) (group-sorted permutation); src is indexed by FEATURE (an LDS image).
// fn(g, sumsq) is called by lane 0 of the team.
template <typename F>
__device__ __forceinline__ void for_each_group_sumsq(const double* src, const int* order,
                                                     const int* gstart, int G, int team, F fn) {
  const int tid = threadIdx.x;
  const int nteams = TAIL_THREADS / team;
  const int my_team = tid / team, tl = tid % team;
  // Four rounds of groups at a time, each step of a round for all four before the next step: a round is a chain of
  // three dependent loads (group bounds -> feature index -> value), and one round after the other the 500 groups of
  // BASELINE config 3 / 4 were eight such chains per sweep, 25 us of a 40 us call.
  constexpr int R = 4;
  for (int g0 = 0; g0 < G; g0 += R * nteams) {  // trip count is uniform across the workgroup
    int g[R], k[R], k1[R], j[R];
    double s[R];
#pragma unroll
    for (int u = 0; u < R; ++u) {
      g[u] = g0 + u * nteams + my_team;
      const bool in = g[u] < G;
      k[u] = in ? gstart[g[u]] + tl : 0;
      k1[u] = in ? gstart[g[u] + 1] : 0;
    }
#pragma unroll
    for (int u = 0; u < R; ++u) j[u] = k[u] < k1[u] ? order[k[u]] : -1;
#pragma unroll
    for (int u = 0; u < R; ++u) {
      const double x = j[u] >= 0 ? src[j[u]] : 0.0;
      s[u] = x * x;
      for (int kk = k[u] + team; kk < k1[u]; kk += team) {  // (groups larger than a team)
        const double y = src[order[kk]];
        s[u] = __builtin_fma(y, y, s[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < R; ++u) {
      for (int off = team >> 1; off >= 1; off >>= 1) s[u] += __shfl_xor(s[u], off, 64);
      if (g[u] < G && tl == 0) fn(g[u], s[u]);
    }
  }
}

// E = elements per thread (p <= 1024 * E).  Everything a thread needs of its E features is loaded
// once into registers (independent loads: one L2 round trip), the group phase goes through LDS, the
// only other global round trips are the control block and -- for real group penalties -- the group
// tables.
//
// Two iteration schemes share this kernel (per lane, chosen by ctl->mode):
//
//  * mode 1, spectral proximal gradient (SpaRSA: Barzilai-Borwein step + non-monotone acceptance).
//    State: base point beta with its gradient (gprev), candidate z = prox(beta - gprev/ak).  Each
//    call receives grad/loss at z: the candidate is accepted when F(z) <= max(last 5 F) -
//    sigma/2 ak ||z-beta||^2 (then ak <- <dz,dg>/<dz,dz>, the curvature along the step) or rejected
//    (ak <- 2 ak, same base); a new candidate is produced either way.  On well-conditioned sparse
//    problems (the BASELINE shapes) the local curvature is far below lambda_max and this needs about
//    half the gradient evaluations of FISTA.
//  * mode 0, FISTA with gradient-scheme restart and the curvature guard on L.  A lane falls back to
//    it for good after BB_REJECT_LIMIT rejections or BB_POINT_LIMIT spectral iterations on one point
//    (ill-conditioned / p > n problems, where FISTA's worst-case rate wins).
//
// Stopping rule (both modes).  With G_s(z) = (z - prox_s(z - s grad f(z))) / s the prox-gradient mapping --
// the KKT residual of z: it vanishes exactly at the minimiser and its norm is at most that of the smallest
// subgradient of the objective at z -- a point is accepted when
//       ||G_s(z)||_2  <=  tol * mu * ||beta||_2 ,
// mu an estimate of the strong convexity of the objective on the face of z: for a mu-strongly convex
// objective ||z - z*|| <= (1 + L s) ||G_s(z)|| / mu, so `tol` bounds the RELATIVE DISTANCE TO THE MINIMISER
// (up to that factor <= 2), whatever the conditioning.  mu comes from the working-set model solver when it
// refined this point (PathCtl::mu: the smallest eigenvalue of the face Hessian from its Cholesky factor, or
// the smallest Rayleigh quotient along its moves), capped by the curvatures this kernel measures itself:
// the smallest Barzilai-Borwein quotient accepted on this point and Lhat (spectral mode), L (FISTA mode).
// Without any of these the rule is the classical ||prox step|| <= tol ||beta|| with a step 1/L.
// (the body, for lane `lane_id`; fista_tail_kernel runs it for lane blockIdx.x)
template <int E>
__device__ __forceinline__ void fista_tail_body(TailArgs a, const int lane_id) {
  __shared__ double red[9][TAIL_WAVES];
  // image of the thresholded vector for the group gathers: LDS up to 16K features, the per-lane
  // global scratch beyond (long-row fallback; the tail is negligible next to a two-pass gradient)
  constexpr bool US_IN_LDS = E <= 16;
  __shared__ double us_lds[US_IN_LDS ? E * TAIL_THREADS : 1];
  PathCtl* ctl = a.ctl + lane_id;
  if (ctl->done != 0 || ctl->idle != 0 || a.gdone[0] != 0) return;
  const int tid = threadIdx.x;
  const int p = a.p, G = a.G;
  {  // rebase every per-lane pointer
    const int64_t off = (int64_t)lane_id * a.ld;
    a.beta += off; a.z += off; a.zprev += off; a.gprev += off;
    a.a0 += off; a.b0 += off; a.d0 += off;
    a.g += (int64_t)lane_id * (a.ld + 16);
    a.gscale += (int64_t)lane_id * G;
    a.uscratch += off;
    const int64_t po = ctl->pt_off;
    a.pts += po;
    a.betas_out += po * p;
    a.infos += po;
    if (a.gn_out != nullptr) a.gn_out += po * G;
  }

  double* us = US_IN_LDS ? us_lds : a.uscratch;

  // ---- phase 0: per-feature loads (independent of the control block) ---------------------------
  // (kept to the minimum that must live across the reductions: 1024 threads => 128 VGPRs.  Up to six features per
  //  thread the base point and its gradient stay in registers too; beyond -- p > 6144, BASELINE config 5's 10 000 --
  //  they are read where they are used, once more from the L2 in the branches that need them again: with all four
  //  vectors and the two results in registers a thread of E = 10 needs 120 of its 128 for them alone, and what did
  //  not fit went to scratch memory, 20 ... 544 bytes per thread, whose dirty lines are written back at the end of
  //  every call of this kernel)
  constexpr bool KEEP = E <= 6;
  double zj[E], gj[E], bo[KEEP ? E : 1], gpv[KEEP ? E : 1];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int j = tid + e * TAIL_THREADS;
    const int jj = j < p ? j : 0;
    zj[e] = a.z[jj];
    gj[e] = a.g[jj];
    if (KEEP) {
      bo[KEEP ? e : 0] = a.beta[jj];
      gpv[KEEP ? e : 0] = a.gprev[jj];
    }
  }
  // base point / its gradient of feature slot e (jj: the slot's feature, clamped)
  auto BO = [&](int e, int jj) -> double { return KEEP ? bo[KEEP ? e : 0] : a.beta[jj]; };
  auto GPV = [&](int e, int jj) -> double { return KEEP ? gpv[KEEP ? e : 0] : a.gprev[jj]; };

  // uniform snapshot of the control block (read-only until the final single-thread update)
  const int point = ctl->point;
  const int iter = ctl->iter;
  const double L = ctl->L;
  const double t_old = ctl->t;
  const double tol = ctl->tol;
  const uint32_t flags = ctl->flags;
  const int64_t total_iter = ctl->total_iter;
  const int n_points = ctl->n_points;
  const int pt_lo = ctl->pt_lo;
  const int max_iter = ctl->max_iter;
  const int mode = ctl->mode;
  const int have_base = ctl->have_base;
  const int rejects = ctl->rejects;
  const int n_hist = ctl->n_hist;
  const double ak_old = ctl->ak;
  const double Lhat_old = ctl->Lhat;
  const double pen_z = ctl->pen_z;
  const double mu_ws = ctl->mu;
  const double mu_rq_old = ctl->mu_rq;
  double hist[BB_HIST];
#pragma unroll
  for (int k = 0; k < BB_HIST; ++k) hist[k] = ctl->hist[k];
  const slm_path_point pt = a.pts[point];
  const double loss_z = a.g[a.ld];
  const bool group_pen = (pt.sb != 0.0) || (pt.sd != 0.0);
  const bool cold = (flags & SLM_FLAG_COLD_START) != 0;
  const bool hit_max = (iter + 1 >= max_iter);
  // ||beta|| in the stopping rule never drops below 1e-10 of the scale the data give a coefficient
  // vector (rms residual / sqrt(L)): at alpha ~ alpha_max the minimiser is a rounding-level number
  // (~1e-16) and "tol relative to it" would ask for more digits than fp64 has.
  const double bnorm_floor = 1e-10 * sqrt(2.0 * fmax(loss_z, 0.0) / fmax(L, Lhat_old));

  // prox_{step * penalty} of the per-thread vector v[] (in place), optionally accumulating the
  // penalty value of the result into pen (thread-partial; the caller block-sums it).
  auto prox_inplace = [&](double (&v)[E], double step, double* pen) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int j = tid + e * TAIL_THREADS;
      if (j < p) {
        double uu = soft(v[e], step * pt.sa * a.a0[j]);
        if (group_pen && a.singleton) {
          const double nrm = fabs(uu);
          const double sc = nrm > 0.0 ? fmax(0.0, 1.0 - step * pt.sb * a.b0[j] / nrm) : 0.0;
          uu *= sc / (1.0 + step * pt.sd * a.d0[j]);
          if (pen) *pen += pt.sb * a.b0[j] * fabs(uu) + 0.5 * pt.sd * a.d0[j] * uu * uu;
        }
        v[e] = uu;
      } else {
        v[e] = 0.0;
      }
    }
    if (group_pen && !a.singleton) {
      __syncthreads();  // us may still be read by an earlier phase
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int j = tid + e * TAIL_THREADS;
        if (j < p) us[j] = v[e];
      }
      __syncthreads();
      for_each_group_sumsq(us, a.order, a.gstart, G, a.team, [&](int g, double ss) {
        const double nrm = sqrt(ss);
        const double sc = (nrm > 0.0 ? fmax(0.0, 1.0 - step * pt.sb * a.b0[g] / nrm) : 0.0) /
                          (1.0 + step * pt.sd * a.d0[g]);
        a.gscale[g] = sc;
        if (pen) {
          const double nc = nrm * sc;
          *pen += pt.sb * a.b0[g] * nc + 0.5 * pt.sd * a.d0[g] * nc * nc;
        }
      });
      __syncthreads();
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int j = tid + e * TAIL_THREADS;
        if (j < p) v[e] *= a.gscale[a.gid[j]];
      }
    }
    if (pen && pt.sa != 0.0) {
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int j = tid + e * TAIL_THREADS;
        if (j < p) *pen += pt.sa * a.a0[j] * fabs(v[e]);
      }
    }
  };

  // outcome of this call, filled by either scheme
  double u[E];   // next point z (candidate / extrapolated point), or the reported solution when finalize
  double(&nb)[E] = zj;  // new beta when !finalize: takes over the registers of z once the sums over z are done
  bool finalize = false, conv = false, nonfinite = false;
  double resid = 0.0, bnorm = 0.0;
  double kkt = 0.0, mu_eff = 0.0, new_mu_rq = mu_rq_old;
  // control-block updates
  int new_mode = mode, new_have_base = have_base, new_rejects = rejects, new_n_hist = n_hist;
  double new_loss_base = ctl->loss_base;
  double new_t = t_old, new_L = L, new_ak = ak_old, new_Lhat = Lhat_old, new_pen_z = pen_z;
  bool did_restart = false, l_bad = false;

  if (mode == 1) {
    // ================= spectral (BB) scheme =====================================================
    //  s[0] = ||z - beta||^2   s[1] = <z - beta, g - gbase>   s[2] = ||g - gbase||^2   s[3] = #non-finite g
    //  s[4] = penalty value at z (only computed at the start of a path point, when no candidate
    //         carried it over)
    //  s[5] = ||g||^2 (rounding floor of the stopping rule)
    double s[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int j = tid + e * TAIL_THREADS;
      if (j < p) {
        const double dz = zj[e] - BO(e, j), dg = gj[e] - GPV(e, j);
        s[0] = __builtin_fma(dz, dz, s[0]);
        s[1] = __builtin_fma(dz, dg, s[1]);
        s[2] = __builtin_fma(dg, dg, s[2]);
        if (!isfinite(gj[e])) s[3] += 1.0;
        s[5] = __builtin_fma(gj[e], gj[e], s[5]);
      }
    }
    if (!have_base) {
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int j = tid + e * TAIL_THREADS;
        if (j < p) {
          const double az = fabs(zj[e]);
          s[4] += pt.sa * a.a0[j] * az;
          if (group_pen && a.singleton) s[4] += pt.sb * a.b0[j] * az + 0.5 * pt.sd * a.d0[j] * az * az;
        }
      }
      if (group_pen && !a.singleton) {
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const int j = tid + e * TAIL_THREADS;
          if (j < p) us[j] = zj[e];
        }
        __syncthreads();
        for_each_group_sumsq(us, a.order, a.gstart, G, a.team, [&](int g, double ss) {
          s[4] += pt.sb * a.b0[g] * sqrt(ss) + 0.5 * pt.sd * a.d0[g] * ss;
        });
      }
    }
    block_sum<6>(s, red);
    const double Fz = loss_z + (have_base ? pen_z : s[4]);
    nonfinite = s[3] > 0.0 || !isfinite(Fz);
    bool accept;
    if (!have_base) {
      accept = true;  // z is the start point of this path point: it becomes the base
      new_n_hist = 0;
    } else {
      double fmax_hist = hist[0];
#pragma unroll
      for (int k = 1; k < BB_HIST; ++k)
        if (k < n_hist) fmax_hist = fmax(fmax_hist, hist[k]);
      accept = Fz <= fmax_hist - 0.5 * BB_SIGMA * ak_old * s[0];
      if (accept) {
        if (s[0] > 0.0) {
          new_Lhat = fmax(Lhat_old, sqrt(s[2] / s[0]));
          new_ak = s[1] > 0.0 ? s[1] / s[0] : new_Lhat;
        }
        new_ak = fmin(fmax(new_ak, 1e-6 * new_Lhat), 1e6 * new_Lhat);
        // (a step at the rounding level of the iterate or of the gradient measures nothing)
        if (s[1] > 0.0 && s[0] * new_Lhat * new_Lhat > 1e-20 * s[5] && s[2] > 1e-20 * s[5])
          new_mu_rq = mu_rq_old > 0.0 ? fmin(mu_rq_old, new_ak) : new_ak;
      } else {
        new_ak = fmin(2.0 * ak_old, 1e6 * Lhat_old);
        new_rejects = rejects + 1;
      }
    }
    if (accept) {  // push F(z) into the ring of the last BB_HIST accepted values
      if (new_n_hist < BB_HIST) {
#pragma unroll
        for (int k = 0; k < BB_HIST; ++k)
          if (k == new_n_hist) hist[k] = Fz;
        new_n_hist += 1;
      } else {
#pragma unroll
        for (int k = 0; k + 1 < BB_HIST; ++k) hist[k] = hist[k + 1];
        hist[BB_HIST - 1] = Fz;
      }
      new_have_base = 1;
      new_loss_base = loss_z;
    }
    // base point and its gradient after the decision; (zprev, gprev) = (base, its gradient) stays a
    // consistent pair for the FISTA curvature guard should this lane fall back
    const double step = 1.0 / new_ak;
    // (two loops under one test rather than `accept ? zj[e] : bo[e]` in one: the compiler turns that into a choice
    // between the ADDRESSES of the arrays, which puts them in scratch memory -- sixteen stores, thirty loads and their
    // lines to write back at the end of every call.  nb IS zj from here on: an accepted candidate is the new base
    // as it stands, a rejected one is overwritten by the old base)
    if (accept) {
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int j = tid + e * TAIL_THREADS;
        if (j < p) {
          a.gprev[j] = gj[e];
          a.zprev[j] = nb[e];
        }
        u[e] = nb[e] - step * gj[e];
      }
    } else {
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int j = tid + e * TAIL_THREADS;
        const int jj = j < p ? j : 0;
        nb[e] = BO(e, jj);
        if (j < p) a.zprev[j] = nb[e];
        u[e] = nb[e] - step * GPV(e, jj);
      }
    }
    const bool fallback = !nonfinite && (new_rejects >= BB_REJECT_LIMIT || iter + 1 > BB_POINT_LIMIT);
    if (fallback) {
      new_mode = 0;
      new_t = 1.0;
      new_L = fmax(L, new_Lhat);
#pragma unroll
      for (int e = 0; e < E; ++e) u[e] = nb[e];  // FISTA restarts from the base point
      finalize = hit_max;
      if (finalize) {
        double q[1] = {0.0};
#pragma unroll
        for (int e = 0; e < E; ++e)
          if (tid + e * TAIL_THREADS < p) q[0] = __builtin_fma(u[e], u[e], q[0]);
        block_sum<1>(q, red);
        bnorm = sqrt(q[0]);
        resid = sqrt(s[0]);
      }
    } else {
      double pen_c = 0.0;
      prox_inplace(u, step, &pen_c);
      //  q[0] = ||c - base||^2   q[1] = ||c||^2   q[2] = pen(c)   q[3] = #non-finite
      double q[4] = {0, 0, pen_c, 0};
#pragma unroll
      for (int e = 0; e < E; ++e) {
        if (tid + e * TAIL_THREADS < p) {
          const double dc = u[e] - nb[e];
          q[0] = __builtin_fma(dc, dc, q[0]);
          q[1] = __builtin_fma(u[e], u[e], q[1]);
          if (!isfinite(u[e])) q[3] += 1.0;
        }
      }
      block_sum<4>(q, red);
      nonfinite = nonfinite || q[3] > 0.0 || !isfinite(q[0]) || !isfinite(q[1]);
      new_pen_z = q[2];
      resid = sqrt(q[0]) * fmax(1.0, new_ak / new_Lhat);
      bnorm = sqrt(q[1]);
      kkt = sqrt(q[0]) * new_ak;  // ||G_s(base)||, s = 1 / ak
      mu_eff = fmin(new_ak, new_Lhat);
      if (new_mu_rq > 0.0) mu_eff = fmin(mu_eff, new_mu_rq);
      if (mu_ws > 0.0) mu_eff = fmin(mu_eff, mu_ws);
      mu_eff = fmax(mu_eff, kMuFloor * new_Lhat);
      // (second term: a prox step at the rounding level of the gradient itself cannot be improved)
      conv = !a.provisional && kkt <= fmax(tol * fmax(bnorm, bnorm_floor) * mu_eff, kRoundFloor * (sqrt(s[5]) + new_Lhat * bnorm));
      finalize = nonfinite || conv || hit_max;
    }
  } else {
    // ================= FISTA scheme =================================================================
    //  s[0] = ||b+ - z||^2   s[1] = ||b+||^2   s[2] = (z - b+).(b+ - b)   s[3] = ||g - gprev||^2
    //  s[4] = ||z - zprev||^2   s[5] = ||z||^2   s[6] = #non-finite
    //  s[7] = ||g||^2   s[8] = <g - gprev, z - zprev>
    double s[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const double step = 1.0 / L;
    new_loss_base = loss_z;  // (zprev = z below)
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int j = tid + e * TAIL_THREADS;
      if (j < p) {
        const double dg = gj[e] - GPV(e, j), dzz = zj[e] - a.zprev[j];
        s[3] = __builtin_fma(dg, dg, s[3]);
        s[4] = __builtin_fma(dzz, dzz, s[4]);
        s[8] = __builtin_fma(dg, dzz, s[8]);
        s[5] = __builtin_fma(zj[e], zj[e], s[5]);
        s[7] = __builtin_fma(gj[e], gj[e], s[7]);
        a.gprev[j] = gj[e];
        a.zprev[j] = zj[e];
      }
      u[e] = zj[e] - step * gj[e];
    }
    prox_inplace(u, step, nullptr);
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int j = tid + e * TAIL_THREADS;
      if (j < p) {
        const double bn = u[e];
        const double dz = bn - zj[e];
        s[0] = __builtin_fma(dz, dz, s[0]);
        s[1] = __builtin_fma(bn, bn, s[1]);
        s[2] = __builtin_fma(-dz, bn - BO(e, j), s[2]);
        if (!isfinite(bn)) s[6] += 1.0;
      }
    }
    block_sum<9>(s, red);
    nonfinite = s[6] > 0.0 || !isfinite(s[0]) || !isfinite(s[1]) || !isfinite(loss_z);
    // Curvature guard: ||A dz|| / ||dz|| is a lower bound on lambda_max(A), A = X^T W X / n.  If it
    // exceeds L the step 1/L was too long: raise L, discard the step and restart from beta.
    if (total_iter > 0 && s[4] > 1e-12 * s[5] && s[4] > 0.0) {
      const double curv = sqrt(s[3] / s[4]);
      if (curv > L * (1.0 + 1e-9)) {
        l_bad = true;
        new_L = 1.02 * curv;
      }
      // curvature along the move of the extrapolated point: a Rayleigh quotient of X^T W X / n, i.e. an upper
      // estimate of the strong convexity on the face the iteration is on
      if (s[8] > 0.0 && s[3] > 1e-20 * s[7]) new_mu_rq = mu_rq_old > 0.0 ? fmin(mu_rq_old, s[8] / s[4]) : s[8] / s[4];
    }
    did_restart = !(flags & SLM_FLAG_NO_RESTART) && s[2] > 0.0;
    const double t_use = did_restart ? 1.0 : t_old;
    const double t_new = 0.5 * (1.0 + sqrt(1.0 + 4.0 * t_use * t_use));
    const double mom = (t_use - 1.0) / t_new;
    resid = sqrt(s[0]);
    bnorm = sqrt(s[1]);
    kkt = resid * L;  // ||G_s(z)||, s = 1 / L
    mu_eff = mu_ws > 0.0 ? fmin(mu_ws, L) : L;
    if (new_mu_rq > 0.0) mu_eff = fmin(mu_eff, new_mu_rq);
    mu_eff = fmax(mu_eff, kMuFloor * L);
    conv = !a.provisional && !l_bad && (kkt <= fmax(tol * fmax(bnorm, bnorm_floor) * mu_eff, kRoundFloor * (sqrt(s[7]) + L * bnorm)));
    finalize = nonfinite || conv || hit_max;
    new_t = l_bad ? 1.0 : t_new;
#pragma unroll
    for (int e = 0; e < E; ++e) {  // (z is spent: nb takes its registers)
      const int j = tid + e * TAIL_THREADS;
      const double bn = u[e], bold = BO(e, j < p ? j : 0);
      if (l_bad) {
        nb[e] = bold;  // step rejected: beta unchanged, momentum dropped
        u[e] = bold;
      } else {
        nb[e] = bn;
        if (!finalize) u[e] = bn + mom * (bn - bold);  // next extrapolated point
      }
    }
  }

  // ---- state update --------------------------------------------------------------------------
  // secant prediction of the next point's start from the last two solutions (see slm_path_point)
  double extrap = 0.0;
  const int stride = ctl->stride > 1 ? ctl->stride : 1;
  // (interleaved lanes: the neighbouring points belong to other lanes and finish in this same launch,
  //  so there is no secant through them -- the next point starts from this lane's last solution)
  if (finalize && !cold && !nonfinite && stride == 1 && point - pt_lo >= 1 && point + 1 < n_points)
    extrap = a.pts[point + 1].extrap;
  // End of this lane's range: in shared-path mode the lane goes idle and steal_kernel (launched
  // right after this kernel, when every lane's state is at rest) hands it new work or retires it.
  const int tail_pt = ctl->tail_pt;
  const bool walk_end = point + stride >= n_points;  // (the tail point itself lies beyond n_points)
  const bool range_end = finalize && !nonfinite && walk_end && (tail_pt < 0 || point == tail_pt);
  const bool goes_idle = range_end && a.steal;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int j = tid + e * TAIL_THREADS;
    if (j < p) {
      if (finalize) {
        const double out = u[e];
        a.betas_out[(int64_t)point * p + j] = out;
        double nxt = (cold || goes_idle) ? 0.0 : out;  // a taken-over range starts cold
        if (extrap != 0.0) nxt = out + extrap * (out - a.betas_out[(int64_t)(point - 1) * p + j]);
        a.beta[j] = nxt;
        a.z[j] = nxt;
      } else {
        a.beta[j] = nb[e];
        a.z[j] = u[e];
      }
    }
  }
  if (finalize && a.gn_out != nullptr) {
    // group norms of the reported solution (the reference's auxiliaries.group_norms.value,
    // model/_lasso.py:239-255, consumed by the adaptive re-weighting at _adaptive_lasso.py:364-374)
    __syncthreads();
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int j = tid + e * TAIL_THREADS;
      if (j < p) us[j] = u[e];
    }
    __syncthreads();
    double* gn = a.gn_out + (int64_t)point * G;
    for_each_group_sumsq(us, a.order, a.gstart, G, a.team, [&](int g, double ss) { gn[g] = sqrt(ss); });
  }

  // ---- control block ----------------------------------------------------------------------------
  if (tid == 0) {
    ctl->zzero = 0;  // (z was just rewritten)
    ctl->total_iter = total_iter + 1;
    ctl->L = new_L;
    ctl->mode = new_mode;
    ctl->rejects = new_rejects;
    ctl->ak = new_ak;
    ctl->Lhat = new_Lhat;
    ctl->loss_base = new_loss_base;
    if (l_bad) ctl->l_bumps += 1;
    if (did_restart) ctl->restarts += 1;
    if (finalize) {
      slm_point_info info;
      info.n_iter = iter + 1;
      info.status = (conv && !nonfinite) ? SLM_OK : (nonfinite ? SLM_ERR_NON_FINITE : SLM_ERR_NOT_CONVERGED);
      info.resid = resid;
      info.beta_norm = bnorm;
      info.loss = loss_z;
      info.L = new_mode == 1 ? new_ak : new_L;
      info.mode = new_mode;
      info.rejects = new_rejects;
      info.kkt = kkt;
      info.mu = mu_eff;
      a.infos[point] = info;
      ctl->mu = 0.0;     // (the next point has its own face)
      ctl->mu_rq = 0.0;
      ctl->iter = 0;
      ctl->t = 1.0;
      ctl->have_base = 0;  // the next point's objective differs: start its history afresh
      ctl->n_hist = 0;
      ctl->pen_z = 0.0;
      ctl->point = (walk_end && tail_pt >= 0 && point != tail_pt) ? tail_pt : point + stride;
      if (nonfinite) {
        ctl->nonfinite = 1;
        ctl->done = 1;
        a.gdone[a.done_slot] = 1;  // abort every lane
      } else if (goes_idle) {
        ctl->idle = 1;
      } else if (range_end) {
        ctl->done = 1;
        if (atomicAdd(&a.gdone[1], 1) + 1 == a.n_lanes) a.gdone[a.done_slot] = 1;
      }
    } else {
      ctl->iter = iter + 1;
      atomicMax(&a.gdone[2], iter + 1);  // GlobalCtl::hard: lets the host give a hard problem the working set
      ctl->t = new_t;
      // (a call on an estimated gradient leaves no base behind: the first true gradient starts the history, and a
      //  rejection can never fall back on the estimate)
      ctl->have_base = a.provisional ? 0 : new_have_base;
      ctl->n_hist = a.provisional ? 0 : new_n_hist;
      ctl->pen_z = new_pen_z;
      ctl->mu_rq = new_mu_rq;
#pragma unroll
      for (int k = 0; k < BB_HIST; ++k) ctl->hist[k] = hist[k];
    }
  }
}

template <int E>
__global__ __launch_bounds__(TAIL_THREADS) void fista_tail_kernel(TailArgs a) {
  fista_tail_body<E>(a, (int)blockIdx.x);
}

// ---------------------------------------------------------------------------------------------
// The same kernel for rows of more than 6 144 columns (BASELINE config 5: p = 10 000), STREAMING: nothing per feature
// lives in registers across a workgroup sum.  A thread of fista_tail_kernel<E> carries 6 E doubles (z, g, base, its
// gradient, the new point, the new base) through three sums; from E = 7 on that is more than the 128 registers 1 024
// threads leave each other, and what did not fit went to scratch memory -- 20 ... 544 bytes per thread from E = 7 to 10,
// kilobytes beyond -- whose dirty lines the next kernel boundary has to write back ("Kernel
// boundaries").  Here every phase walks the features (j = tid, tid + 1024, ...: the order, and therefore every sum, is
// that of the register kernel), reads what it needs from the L2-resident vectors and leaves its result in the feature
// image `us` (LDS up to 16 384 features, the per-lane global scratch beyond); the image is the candidate point, the
// base point is zprev (which the decision phase writes anyway).  Same arithmetic, same state machine, any p.
// ---------------------------------------------------------------------------------------------
// E > 0: p <= 1024 E, every walk over the features is E unrolled steps of straight-line code -- loads from a clamped index,
// sums and stores predicated -- so that the loads of a phase are all in flight at once (as runtime loops a phase was ten
// dependent round trips: 100 us per call at p = 10 000 against 25); E = 0: runtime loops, any p.
template <int E, typename F>
__device__ __forceinline__ void tail_for(int tid, int p, F f) {
  if constexpr (E > 0) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int j = tid + e * TAIL_THREADS;
      f(j < p ? j : 0, j < p);
    }
  } else {
    for (int j = tid; j < p; j += TAIL_THREADS) f(j, true);
  }
}

template <int E>
__global__ __launch_bounds__(TAIL_THREADS) void fista_tail_stream_kernel(TailArgs a) {
  __shared__ double red[9][TAIL_WAVES];
  constexpr int US_LDS = 16 * TAIL_THREADS;
  __shared__ double us_lds[US_LDS];
  const int lane_id = blockIdx.x;
  PathCtl* ctl = a.ctl + lane_id;
  if (ctl->done != 0 || ctl->idle != 0 || a.gdone[0] != 0) return;
  const int tid = threadIdx.x;
  const int p = a.p, G = a.G;
  {  // rebase every per-lane pointer
    const int64_t off = (int64_t)lane_id * a.ld;
    a.beta += off; a.z += off; a.zprev += off; a.gprev += off;
    a.a0 += off; a.b0 += off; a.d0 += off;
    a.g += (int64_t)lane_id * (a.ld + 16);
    a.gscale += (int64_t)lane_id * G;
    a.uscratch += off;
    const int64_t po = ctl->pt_off;
    a.pts += po;
    a.betas_out += po * p;
    a.infos += po;
    if (a.gn_out != nullptr) a.gn_out += po * G;
  }
  double* us = p <= US_LDS ? us_lds : a.uscratch;

  const int point = ctl->point;
  const int iter = ctl->iter;
  const double L = ctl->L;
  const double t_old = ctl->t;
  const double tol = ctl->tol;
  const uint32_t flags = ctl->flags;
  const int64_t total_iter = ctl->total_iter;
  const int n_points = ctl->n_points;
  const int pt_lo = ctl->pt_lo;
  const int max_iter = ctl->max_iter;
  const int mode = ctl->mode;
  const int have_base = ctl->have_base;
  const int rejects = ctl->rejects;
  const int n_hist = ctl->n_hist;
  const double ak_old = ctl->ak;
  const double Lhat_old = ctl->Lhat;
  const double pen_z = ctl->pen_z;
  const double mu_ws = ctl->mu;
  const double mu_rq_old = ctl->mu_rq;
  double hist[BB_HIST];
#pragma unroll
  for (int k = 0; k < BB_HIST; ++k) hist[k] = ctl->hist[k];
  const slm_path_point pt = a.pts[point];
  const double loss_z = a.g[a.ld];
  const bool group_pen = (pt.sb != 0.0) || (pt.sd != 0.0);
  const bool cold = (flags & SLM_FLAG_COLD_START) != 0;
  const bool hit_max = (iter + 1 >= max_iter);
  const double bnorm_floor = 1e-10 * sqrt(2.0 * fmax(loss_z, 0.0) / fmax(L, Lhat_old));

  // the image us[] holds v; on return it holds prox_{step * penalty}(v).  pen (nullable): thread-partial penalty value
  // of the result, accumulated in the order of fista_tail_kernel's prox_inplace
  auto prox_image = [&](double step, double* pen) {
    tail_for<E>(tid, p, [&](int j, bool ok) {
      double uu = soft(us[j], step * pt.sa * a.a0[j]);
      if (group_pen && a.singleton) {
        const double nrm = fabs(uu);
        const double sc = nrm > 0.0 ? fmax(0.0, 1.0 - step * pt.sb * a.b0[j] / nrm) : 0.0;
        uu *= sc / (1.0 + step * pt.sd * a.d0[j]);
        if (pen && ok) *pen += pt.sb * a.b0[j] * fabs(uu) + 0.5 * pt.sd * a.d0[j] * uu * uu;
      }
      if (ok) us[j] = uu;
    });
    if (group_pen && !a.singleton) {
      __syncthreads();
      for_each_group_sumsq(us, a.order, a.gstart, G, a.team, [&](int g, double ss) {
        const double nrm = sqrt(ss);
        const double sc = (nrm > 0.0 ? fmax(0.0, 1.0 - step * pt.sb * a.b0[g] / nrm) : 0.0) /
                          (1.0 + step * pt.sd * a.d0[g]);
        a.gscale[g] = sc;
        if (pen) {
          const double nc = nrm * sc;
          *pen += pt.sb * a.b0[g] * nc + 0.5 * pt.sd * a.d0[g] * nc * nc;
        }
      });
      __syncthreads();
      tail_for<E>(tid, p, [&](int j, bool ok) {
        const double v = us[j] * a.gscale[a.gid[j]];
        if (ok) us[j] = v;
      });
    }
    if (pen && pt.sa != 0.0) {
      tail_for<E>(tid, p, [&](int j, bool ok) {
        if (ok) *pen += pt.sa * a.a0[j] * fabs(us[j]);
      });
    }
  };

  bool finalize = false, conv = false, nonfinite = false;
  double resid = 0.0, bnorm = 0.0;
  double kkt = 0.0, mu_eff = 0.0, new_mu_rq = mu_rq_old;
  int new_mode = mode, new_have_base = have_base, new_rejects = rejects, new_n_hist = n_hist;
  double new_loss_base = ctl->loss_base;
  double new_t = t_old, new_L = L, new_ak = ak_old, new_Lhat = Lhat_old, new_pen_z = pen_z;
  bool did_restart = false, l_bad = false;
  // how the last phase finds the new base point and the next point of feature j:
  //   mode 1: base = zprev[j], next = us[j] (the candidate; the base itself on the switch to FISTA: us holds it then)
  //   mode 0: base / next from us[j] (the proximal point) and beta[j], see below
  double mom = 0.0;

  if (mode == 1) {
    double s[6] = {0, 0, 0, 0, 0, 0};
    tail_for<E>(tid, p, [&](int j, bool ok) {
      const double z = a.z[j], g = a.g[j];
      const double dz = z - a.beta[j], dg = g - a.gprev[j];
      if (ok) {
        s[0] = __builtin_fma(dz, dz, s[0]);
        s[1] = __builtin_fma(dz, dg, s[1]);
        s[2] = __builtin_fma(dg, dg, s[2]);
        if (!isfinite(g)) s[3] += 1.0;
        s[5] = __builtin_fma(g, g, s[5]);
      }
    });
    if (!have_base) {
      tail_for<E>(tid, p, [&](int j, bool ok) {
        const double az = fabs(a.z[j]);
        if (ok) {
          s[4] += pt.sa * a.a0[j] * az;
          if (group_pen && a.singleton) s[4] += pt.sb * a.b0[j] * az + 0.5 * pt.sd * a.d0[j] * az * az;
        }
      });
      if (group_pen && !a.singleton) {
        tail_for<E>(tid, p, [&](int j, bool ok) {
          const double z = a.z[j];
          if (ok) us[j] = z;
        });
        __syncthreads();
        for_each_group_sumsq(us, a.order, a.gstart, G, a.team, [&](int g, double ss) {
          s[4] += pt.sb * a.b0[g] * sqrt(ss) + 0.5 * pt.sd * a.d0[g] * ss;
        });
      }
    }
    block_sum<6>(s, red);
    const double Fz = loss_z + (have_base ? pen_z : s[4]);
    nonfinite = s[3] > 0.0 || !isfinite(Fz);
    bool accept;
    if (!have_base) {
      accept = true;
      new_n_hist = 0;
    } else {
      double fmax_hist = hist[0];
#pragma unroll
      for (int k = 1; k < BB_HIST; ++k)
        if (k < n_hist) fmax_hist = fmax(fmax_hist, hist[k]);
      accept = Fz <= fmax_hist - 0.5 * BB_SIGMA * ak_old * s[0];
      if (accept) {
        if (s[0] > 0.0) {
          new_Lhat = fmax(Lhat_old, sqrt(s[2] / s[0]));
          new_ak = s[1] > 0.0 ? s[1] / s[0] : new_Lhat;
        }
        new_ak = fmin(fmax(new_ak, 1e-6 * new_Lhat), 1e6 * new_Lhat);
        if (s[1] > 0.0 && s[0] * new_Lhat * new_Lhat > 1e-20 * s[5] && s[2] > 1e-20 * s[5])
          new_mu_rq = mu_rq_old > 0.0 ? fmin(mu_rq_old, new_ak) : new_ak;
      } else {
        new_ak = fmin(2.0 * ak_old, 1e6 * Lhat_old);
        new_rejects = rejects + 1;
      }
    }
    if (accept) {
      if (new_n_hist < BB_HIST) {
#pragma unroll
        for (int k = 0; k < BB_HIST; ++k)
          if (k == new_n_hist) hist[k] = Fz;
        new_n_hist += 1;
      } else {
#pragma unroll
        for (int k = 0; k + 1 < BB_HIST; ++k) hist[k] = hist[k + 1];
        hist[BB_HIST - 1] = Fz;
      }
      new_have_base = 1;
      new_loss_base = loss_z;
    }
    const double step = 1.0 / new_ak;
    const bool fallback = !nonfinite && (new_rejects >= BB_REJECT_LIMIT || iter + 1 > BB_POINT_LIMIT);
    __syncthreads();  // (the image may still be read by the group sums above)
    // base point (-> zprev) and its gradient (-> gprev) after the decision; the image gets base - step * gradient,
    // or, on the switch to FISTA, the base itself
    if (accept) {
      tail_for<E>(tid, p, [&](int j, bool ok) {
        const double z = a.z[j], g = a.g[j];
        if (ok) {
          a.gprev[j] = g;
          a.zprev[j] = z;
          us[j] = fallback ? z : z - step * g;
        }
      });
    } else {
      tail_for<E>(tid, p, [&](int j, bool ok) {
        const double b = a.beta[j], gp = a.gprev[j];
        if (ok) {
          a.zprev[j] = b;
          us[j] = fallback ? b : b - step * gp;
        }
      });
    }
    if (fallback) {
      new_mode = 0;
      new_t = 1.0;
      new_L = fmax(L, new_Lhat);
      finalize = hit_max;
      if (finalize) {
        double q[1] = {0.0};
        tail_for<E>(tid, p, [&](int j, bool ok) {
          if (ok) q[0] = __builtin_fma(us[j], us[j], q[0]);
        });
        block_sum<1>(q, red);
        bnorm = sqrt(q[0]);
        resid = sqrt(s[0]);
      }
    } else {
      double pen_c = 0.0;
      prox_image(step, &pen_c);
      double q[4] = {0, 0, pen_c, 0};
      tail_for<E>(tid, p, [&](int j, bool ok) {
        const double c = us[j];
        const double dc = c - a.zprev[j];
        if (ok) {
          q[0] = __builtin_fma(dc, dc, q[0]);
          q[1] = __builtin_fma(c, c, q[1]);
          if (!isfinite(c)) q[3] += 1.0;
        }
      });
      block_sum<4>(q, red);
      nonfinite = nonfinite || q[3] > 0.0 || !isfinite(q[0]) || !isfinite(q[1]);
      new_pen_z = q[2];
      resid = sqrt(q[0]) * fmax(1.0, new_ak / new_Lhat);
      bnorm = sqrt(q[1]);
      kkt = sqrt(q[0]) * new_ak;
      mu_eff = fmin(new_ak, new_Lhat);
      if (new_mu_rq > 0.0) mu_eff = fmin(mu_eff, new_mu_rq);
      if (mu_ws > 0.0) mu_eff = fmin(mu_eff, mu_ws);
      mu_eff = fmax(mu_eff, kMuFloor * new_Lhat);
      conv = !a.provisional && kkt <= fmax(tol * fmax(bnorm, bnorm_floor) * mu_eff, kRoundFloor * (sqrt(s[5]) + new_Lhat * bnorm));
      finalize = nonfinite || conv || hit_max;
    }
  } else {
    // ================= FISTA scheme =================================================================
    double s[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const double step = 1.0 / L;
    new_loss_base = loss_z;  // (zprev = z below)
    tail_for<E>(tid, p, [&](int j, bool ok) {
      const double z = a.z[j], g = a.g[j];
      const double dg = g - a.gprev[j], dzz = z - a.zprev[j];
      if (ok) {
        s[3] = __builtin_fma(dg, dg, s[3]);
        s[4] = __builtin_fma(dzz, dzz, s[4]);
        s[8] = __builtin_fma(dg, dzz, s[8]);
        s[5] = __builtin_fma(z, z, s[5]);
        s[7] = __builtin_fma(g, g, s[7]);
        a.gprev[j] = g;
        a.zprev[j] = z;
        us[j] = z - step * g;
      }
    });
    prox_image(step, nullptr);
    tail_for<E>(tid, p, [&](int j, bool ok) {
      const double bn = us[j];
      const double dz = bn - a.z[j], db = bn - a.beta[j];
      if (ok) {
        s[0] = __builtin_fma(dz, dz, s[0]);
        s[1] = __builtin_fma(bn, bn, s[1]);
        s[2] = __builtin_fma(-dz, db, s[2]);
        if (!isfinite(bn)) s[6] += 1.0;
      }
    });
    block_sum<9>(s, red);
    nonfinite = s[6] > 0.0 || !isfinite(s[0]) || !isfinite(s[1]) || !isfinite(loss_z);
    if (total_iter > 0 && s[4] > 1e-12 * s[5] && s[4] > 0.0) {
      const double curv = sqrt(s[3] / s[4]);
      if (curv > L * (1.0 + 1e-9)) {
        l_bad = true;
        new_L = 1.02 * curv;
      }
      if (s[8] > 0.0 && s[3] > 1e-20 * s[7]) new_mu_rq = mu_rq_old > 0.0 ? fmin(mu_rq_old, s[8] / s[4]) : s[8] / s[4];
    }
    did_restart = !(flags & SLM_FLAG_NO_RESTART) && s[2] > 0.0;
    const double t_use = did_restart ? 1.0 : t_old;
    const double t_new = 0.5 * (1.0 + sqrt(1.0 + 4.0 * t_use * t_use));
    mom = (t_use - 1.0) / t_new;
    resid = sqrt(s[0]);
    bnorm = sqrt(s[1]);
    kkt = resid * L;
    mu_eff = mu_ws > 0.0 ? fmin(mu_ws, L) : L;
    if (new_mu_rq > 0.0) mu_eff = fmin(mu_eff, new_mu_rq);
    mu_eff = fmax(mu_eff, kMuFloor * L);
    conv = !a.provisional && !l_bad && (kkt <= fmax(tol * fmax(bnorm, bnorm_floor) * mu_eff, kRoundFloor * (sqrt(s[7]) + L * bnorm)));
    finalize = nonfinite || conv || hit_max;
    new_t = l_bad ? 1.0 : t_new;
  }

  // ---- state update --------------------------------------------------------------------------
  double extrap = 0.0;
  const int stride = ctl->stride > 1 ? ctl->stride : 1;
  if (finalize && !cold && !nonfinite && stride == 1 && point - pt_lo >= 1 && point + 1 < n_points)
    extrap = a.pts[point + 1].extrap;
  const int tail_pt = ctl->tail_pt;
  const bool walk_end = point + stride >= n_points;
  const bool range_end = finalize && !nonfinite && walk_end && (tail_pt < 0 || point == tail_pt);
  const bool goes_idle = range_end && a.steal;
  tail_for<E>(tid, p, [&](int j, bool ok) {
    double nbv, uu;  // new base, next point (or the reported solution)
    const double image = us[j];
    if (mode == 1) {
      nbv = a.zprev[j];
      uu = image;
    } else {
      const double bold = a.beta[j];
      if (l_bad) {
        nbv = bold;
        uu = bold;
      } else {
        nbv = image;
        uu = finalize ? image : image + mom * (image - bold);
      }
    }
    if (!ok) return;
    if (finalize) {
      a.betas_out[(int64_t)point * p + j] = uu;
      double nxt = (cold || goes_idle) ? 0.0 : uu;
      if (extrap != 0.0) nxt = uu + extrap * (uu - a.betas_out[(int64_t)(point - 1) * p + j]);
      a.beta[j] = nxt;
      a.z[j] = nxt;
      us[j] = uu;  // (the group norms below read the reported solution)
    } else {
      a.beta[j] = nbv;
      a.z[j] = uu;
    }
  });
  if (finalize && a.gn_out != nullptr) {
    __syncthreads();
    double* gn = a.gn_out + (int64_t)point * G;
    for_each_group_sumsq(us, a.order, a.gstart, G, a.team, [&](int g, double ss) { gn[g] = sqrt(ss); });
  }

  // ---- control block (as in fista_tail_kernel) ---------------------------------------------------
  if (tid == 0) {
    ctl->zzero = 0;
    ctl->total_iter = total_iter + 1;
    ctl->L = new_L;
    ctl->mode = new_mode;
    ctl->rejects = new_rejects;
    ctl->ak = new_ak;
    ctl->Lhat = new_Lhat;
    ctl->loss_base = new_loss_base;
    if (l_bad) ctl->l_bumps += 1;
    if (did_restart) ctl->restarts += 1;
    if (finalize) {
      slm_point_info info;
      info.n_iter = iter + 1;
      info.status = (conv && !nonfinite) ? SLM_OK : (nonfinite ? SLM_ERR_NON_FINITE : SLM_ERR_NOT_CONVERGED);
      info.resid = resid;
      info.beta_norm = bnorm;
      info.loss = loss_z;
      info.L = new_mode == 1 ? new_ak : new_L;
      info.mode = new_mode;
      info.rejects = new_rejects;
      info.kkt = kkt;
      info.mu = mu_eff;
      a.infos[point] = info;
      ctl->mu = 0.0;
      ctl->mu_rq = 0.0;
      ctl->iter = 0;
      ctl->t = 1.0;
      ctl->have_base = 0;
      ctl->n_hist = 0;
      ctl->pen_z = 0.0;
      ctl->point = (walk_end && tail_pt >= 0 && point != tail_pt) ? tail_pt : point + stride;
      if (nonfinite) {
        ctl->nonfinite = 1;
        ctl->done = 1;
        a.gdone[a.done_slot] = 1;
      } else if (goes_idle) {
        ctl->idle = 1;
      } else if (range_end) {
        ctl->done = 1;
        if (atomicAdd(&a.gdone[1], 1) + 1 == a.n_lanes) a.gdone[a.done_slot] = 1;
      }
    } else {
      ctl->iter = iter + 1;
      atomicMax(&a.gdone[2], iter + 1);
      ctl->t = new_t;
      // (a call on an estimated gradient leaves no base behind: the first true gradient starts the history, and a
      //  rejection can never fall back on the estimate)
      ctl->have_base = a.provisional ? 0 : new_have_base;
      ctl->n_hist = a.provisional ? 0 : new_n_hist;
      ctl->pen_z = new_pen_z;
      ctl->mu_rq = new_mu_rq;
#pragma unroll
      for (int k = 0; k < BB_HIST; ++k) ctl->hist[k] = hist[k];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Shared-path work distribution.  Runs as ONE small workgroup after the tail kernel, when no lane is
// touching its control block, so the hand-out is sequential and reproducible: idle lanes are served
// in index order; each takes the upper half of the points the busiest lane has not started
// (keeping at least one for the victim) and starts cold; with nothing left to take it retires.
// ---------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void steal_kernel(TailArgs a) {
  if (a.gdone[0] != 0) return;
  // the words the hand-out looks at, fetched by a thread per lane in ONE round trip: thread 0 walking the control blocks by
  // itself -- lanes x lanes dependent loads -- was 90 us behind the LAST pass of a 25-lane group path, where every lane is
  // idle (3 % of config 3's path), 7 us elsewhere
  __shared__ int s_idle[SLM_MAX_LANES], s_done[SLM_MAX_LANES], s_end[SLM_MAX_LANES], s_pt[SLM_MAX_LANES];
  const int t = threadIdx.x;
  if (t < a.n_lanes) {
    const PathCtl* c = a.ctl + t;
    s_idle[t] = c->idle;
    s_done[t] = c->done;
    s_end[t] = c->n_points;
    s_pt[t] = c->point;
  }
  __syncthreads();
  // nobody is busy -- behind the last pass of a path every lane is idle -- : nothing to hand out, all idle lanes retire, each by
  // its own thread (the sequential walk below, lanes x lanes reads of the table, was 50 us of that case)
  const bool mine_idle = t < a.n_lanes && s_idle[t] && !s_done[t];
  const bool mine_busy = t < a.n_lanes && !s_idle[t] && !s_done[t];
  static_assert(SLM_MAX_LANES <= 64, "one wavefront sees every lane");
  if (t < 64) {
    const unsigned long long busy = __ballot(mine_busy), idle = __ballot(mine_idle);
    if (busy == 0ull) {
      if (mine_idle) {
        a.ctl[t].idle = 0;
        a.ctl[t].done = 1;
      }
      const int n_ret = __popcll(idle);
      if (t == 0 && n_ret > 0 && atomicAdd(&a.gdone[1], n_ret) + n_ret == a.n_lanes) a.gdone[a.done_slot] = 1;
      return;
    }
  } else {
    return;
  }
  if (t != 0) return;
  int retired = 0;
  for (int l = 0; l < a.n_lanes; ++l) {
    if (!s_idle[l] || s_done[l]) continue;
    PathCtl* me = a.ctl + l;
    int best = -1, best_rem = 0;
    for (int v = 0; v < a.n_lanes; ++v) {
      if (v == l || s_done[v] || s_idle[v]) continue;
      const int rem = s_end[v] - (s_pt[v] + 1);  // points the victim has not started
      if (rem > best_rem) {
        best = v;
        best_rem = rem;
      }
    }
    if (best >= 0 && best_rem >= 1) {
      const int hi = s_end[best];
      const int mid = hi - (best_rem + 1) / 2;
      a.ctl[best].n_points = mid;
      s_end[best] = mid;
      me->point = mid;
      me->pt_lo = mid;
      me->n_points = hi;
      me->steals += 1;
      me->idle = 0;
      me->zzero = 1;  // cold start: the tail kernel zeroed z (and beta) when the lane went idle
      s_pt[l] = mid;
      s_end[l] = hi;
      s_idle[l] = 0;
    } else {
      me->idle = 0;
      me->done = 1;
      s_idle[l] = 0;
      s_done[l] = 1;
      retired += 1;
    }
  }
  // (one addition for all the lanes that retire here: an atomic that returns its value is a round trip, twenty-five of them
  //  behind the last pass of a path were most of what was left of this kernel)
  if (retired > 0 && atomicAdd(&a.gdone[1], retired) + retired == a.n_lanes) a.gdone[a.done_slot] = 1;
}

// ---------------------------------------------------------------------------------------------
// Interleaved lanes, dense end of a path (rounds on the model Gram: a point takes two or three passes there, and the lanes
// that hold the deepest points of a band fall a pass or two behind the others).  The points beyond the last full band --
// the tail points -- belong to the LAST lanes (host_logic.hpp: interleaved_walk), which are exactly those: the first lanes
// finish their walk and sit idle while the last points of the path wait for their owners (two of ten passes of a
// noise-fitting 50-point path served one or two lanes each).  So a lane that has finished takes over the tail point of a
// lane that has not started it: the shallowest such point first, finished lanes in index order, ONE thread -- the
// hand-over is reproducible.  The taker starts from its own last solution (what every lane does at its next point), with
// its residual taken from X (zsup = 0: the working set may have changed since the lane last looked).  Launched behind the
// tail kernel of every pass once the rounds are on (solve_core); before that the lanes finish their points together and
// the static assignment -- short steps from the neighbours' solutions -- is the one that avoids misses.
// ---------------------------------------------------------------------------------------------
static __global__ void tail_handover_kernel(TailArgs a) {
  if (threadIdx.x != 0 || blockIdx.x != 0 || a.gdone[0] != 0) return;
  for (int l = 0; l < a.n_lanes; ++l) {
    PathCtl* me = a.ctl + l;
    if (!me->done || me->nonfinite) continue;
    int best = -1;
    for (int v = 0; v < a.n_lanes; ++v) {
      const PathCtl* cv = a.ctl + v;
      if (v == l || cv->done || cv->idle || cv->tail_pt < 0 || cv->point == cv->tail_pt) continue;
      if (best < 0 || cv->tail_pt < a.ctl[best].tail_pt) best = v;
    }
    if (best < 0) return;  // (no tail point is waiting: none will be for the lanes after this one either)
    PathCtl* cv = a.ctl + best;
    me->point = cv->tail_pt;
    me->tail_pt = cv->tail_pt;  // (the lane's range ends with it: fista_tail_kernel, range_end)
    cv->tail_pt = -1;
    me->zsup = 0;
    me->steals += 1;
    me->done = 0;
    atomicSub(&a.gdone[1], 1);
  }
}

// The same hand-over at the SPARSE end, for a lane that has fallen behind: a lane whose verification missed early in the path
// repeats that point beside the next band and arrives at its tail point a pass after everybody else -- a whole pass over X
// (and its chain: 0.9 ms of a 3 ms path) for one or two points, on one dataset in six of the headline's shape.  A lane is BEHIND
// when the points it still has to have verified -- the one it stands on, its regular ones, its tail point -- outnumber the
// passes left before the expected end of the path (`passes_left`: the host queues the passes and knows); a finished lane
// then takes its tail point, as above.  The taker keeps `zsup` -- its solution lies on the working set as long as the set has
// only been appended to since (one build, not stale): its residual comes from the gathered columns like everybody's, not
// from a read of X of its own.  Lanes on schedule keep their points: short steps from the neighbours' solutions miss least.
// `ws_builds`, `ws_stale`: words of the working set's control block (nullptr: no working set -- zsup is cleared).
static __global__ __launch_bounds__(64) void lag_handover_kernel(TailArgs a, const int32_t* ws_builds, const int32_t* ws_stale, int passes_left) {
  // one wavefront: lane l of it fetches the control words of path lane l (one round trip for all of them -- a single thread
  // walking the blocks was 10 us a pass), the first decides from the copies; no finished lane, the usual case: nothing to do
  __shared__ int s_done[SLM_MAX_LANES], s_want[SLM_MAX_LANES], s_tail[SLM_MAX_LANES];
  const int l = threadIdx.x;
  const int stop = a.gdone[0];
  int done = 0, want = 0, tail = -1;
  if (l < a.n_lanes) {
    const PathCtl* c = a.ctl + l;
    const int c_done = c->done, c_bad = c->nonfinite, c_idle = c->idle, c_tail = c->tail_pt, c_pt = c->point, c_end = c->n_points,
              c_stride = c->stride;
    done = c_done && !c_bad;
    tail = c_tail;
    if (!c_done && !c_idle && c_tail >= 0 && c_pt != c_tail) {
      const int regular_left = c_pt < c_end ? (c_end - 1 - c_pt) / max(c_stride, 1) : 0;
      want = 1 + regular_left + 1 > passes_left;  // (behind: more points to have verified than passes left)
    }
  }
  // (all of them or none: with fewer finished lanes than tail points waiting, the pass the hand-over is meant to save is made
  //  anyway, and the takers' long steps -- from their own last points, not from a neighbour's -- only risk misses of their own:
  //  soak seed 13, four lanes behind and two free: the same four passes, a miss more, 0.3 ms)
  const int n_done = __popcll(__ballot(done)), n_want = __popcll(__ballot(want));
  if (stop != 0 || n_want == 0 || n_done < n_want) return;
  if (l < SLM_MAX_LANES) {
    s_done[l] = done;
    s_want[l] = want;
    s_tail[l] = tail;
  }
  __syncthreads();
  if (l != 0) return;
  const bool appended_only = ws_builds != nullptr && *ws_builds == 1 && *ws_stale == 0;
  for (int t = 0; t < a.n_lanes; ++t) {
    if (!s_done[t]) continue;
    int best = -1;
    for (int v = 0; v < a.n_lanes; ++v)
      if (s_want[v] && (best < 0 || s_tail[v] < s_tail[best])) best = v;
    if (best < 0) return;
    PathCtl* me = a.ctl + t;
    PathCtl* cv = a.ctl + best;
    me->point = s_tail[best];
    me->tail_pt = s_tail[best];
    cv->tail_pt = -1;
    s_want[best] = 0;
    if (!appended_only) me->zsup = 0;
    me->steals += 1;
    me->done = 0;
    atomicSub(&a.gdone[1], 1);
  }
}

// ---------------------------------------------------------------------------------------------
// Row-sharded mode: the stop decision is taken from reduced data.  After the tail (and hand-out) kernels of a
// pass every rank packs "I have finished" into a small vector, the vector is summed over the ranks, and
// gdone[0] -- the flag every kernel and the host loop obey -- is set only when ALL ranks reported it.  The ranks
// run the same state machine on the same all-reduced gradients, so they normally finish in the same pass;
// if their states ever differ (a collective that is not bit-identical on every rank, a rank-dependent
// rounding) the late ranks simply keep the early ones in the loop: the collective counts stay equal.
// ---------------------------------------------------------------------------------------------
constexpr int STOP_WORDS = 16;
// words: [0] ranks that have finished, [1] ranks taking part, [2] c, [3] c^2 with c a small-integer digest of
// this rank's control blocks (path point, passes on it, mode, done of every lane): if the ranks' states are the
// same, n sum(c^2) == (sum c)^2 exactly (all terms are integers below 2^53); if not, the solve is over -- its
// ranks no longer iterate on one problem -- and every rank reports it (gdone[4]) after the same pass.
static __global__ void stop_pack_kernel(const int* gdone, const PathCtl* ctl, int n_lanes, double* words) {
  if (threadIdx.x < STOP_WORDS) words[threadIdx.x] = 0.0;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long c = 0ull;
    for (int l = 0; l < n_lanes; ++l) {
      const PathCtl& q = ctl[l];
      const unsigned long long v = (unsigned long long)(unsigned)q.point * 1031ull + (unsigned long long)(unsigned)q.iter * 7ull +
                                   (unsigned long long)(q.done ? 3 : 0) + (unsigned long long)(q.mode ? 1 : 0) +
                                   (unsigned long long)(q.idle ? 5 : 0);
      c = (c * 31ull + v * (unsigned long long)(l + 1)) & 0xfffffull;  // < 2^20: c^2 and the sums stay exact
    }
    words[0] = gdone[3] != 0 ? 1.0 : 0.0;
    words[1] = 1.0;
    words[2] = (double)c;
    words[3] = (double)c * (double)c;
  }
}
static __global__ void stop_apply_kernel(int* gdone, const double* words) {
  if (threadIdx.x != 0) return;
  const double n = words[1];
  if (!(n >= 1.0)) return;
  if (n * words[3] != words[2] * words[2]) {
    gdone[4] = 1;
    gdone[0] = 1;
  } else if (words[0] >= n) {
    gdone[0] = 1;
  }
}

// ---------------------------------------------------------------------------------------------
// Power iteration step for the Lipschitz constant:  v <- A v / ||A v||,  lambda <- ||A v||
// (g = A v comes from the fused gradient kernel run with y = 0).
// ---------------------------------------------------------------------------------------------
struct PowerArgs {
  const double* g;  // [n_lanes][ld + 16]
  double* v;        // [n_lanes][ld]
  double* lambda;   // [n_lanes]
  int p;
  int64_t ld;
};

// one workgroup per lane
static __global__ __launch_bounds__(TAIL_THREADS) void power_step_kernel(PowerArgs a) {
  __shared__ double red[1][TAIL_WAVES];
  a.g += (int64_t)blockIdx.x * (a.ld + 16);
  a.v += (int64_t)blockIdx.x * a.ld;
  a.lambda += blockIdx.x;
  double s[1] = {0.0};
  for (int j = threadIdx.x; j < a.p; j += TAIL_THREADS) s[0] = __builtin_fma(a.g[j], a.g[j], s[0]);
  block_sum<1>(s, red);
  const double nrm = sqrt(s[0]);
  const double inv = nrm > 0.0 ? 1.0 / nrm : 0.0;
  for (int j = threadIdx.x; j < a.p; j += TAIL_THREADS) a.v[j] = a.g[j] * inv;
  if (threadIdx.x == 0) a.lambda[0] = nrm;
}

// Step-size seed of a solve straight from the power iteration's result (one estimate for all lanes): L with the
// safety margin, a slightly short first inverse step, and half of L as the first curvature floor -- what the
// host writes into the control blocks when it has the number (solve_core).  A non-finite estimate is left in:
// the first tail kernel then reports the non-finite iterate.
struct SeedArgs {
  PathCtl* ctl;
  const double* lambda;
  int n_lanes;
  double margin;
  // lane l's operator is bounded by factor[l] times the one the estimate was taken on (lanes with their own row
  // weights and 1/n scaling: max weight x n / n_l, see solve_core); 1 for lanes that share it
  double factor[SLM_MAX_LANES];
};
static __global__ void seed_step_kernel(SeedArgs a) {
  const int l = threadIdx.x;
  if (l >= a.n_lanes) return;
  double L0 = a.lambda[0] * a.margin;
  if (L0 <= 0.0) L0 = 1.0;  // X == 0
  const double L = L0 * a.factor[l];
  a.ctl[l].L = L;
  a.ctl[l].ak = 1.25 * L;
  a.ctl[l].Lhat = 0.5 * L0;
}

__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
  x ^= x >> 27; x *= 0x94d049bb133111ebull;
  x ^= x >> 31;
  return x;
}

// Deterministic start vector for the power iteration: unit-norm hash noise.
static __global__ __launch_bounds__(TAIL_THREADS) void power_init_kernel(double* v, int p, int64_t ld) {
  __shared__ double red[1][TAIL_WAVES];
  v += (int64_t)blockIdx.x * ld;  // one workgroup per lane, same start vector
  double s[1] = {0.0};
  for (int j = threadIdx.x; j < p; j += TAIL_THREADS) {
    const uint64_t h = mix64(0x9e3779b97f4a7c15ull * (uint64_t)(j + 1));
    const double x = (double)(h >> 11) * (1.0 / 9007199254740992.0) - 0.5;
    v[j] = x;
    s[0] = __builtin_fma(x, x, s[0]);
  }
  block_sum<1>(s, red);
  const double inv = 1.0 / sqrt(s[0]);
  for (int j = threadIdx.x; j < p; j += TAIL_THREADS) v[j] *= inv;
  for (int64_t j = p + threadIdx.x; j < ld; j += TAIL_THREADS) v[j] = 0.0;
}

}  // namespace slm

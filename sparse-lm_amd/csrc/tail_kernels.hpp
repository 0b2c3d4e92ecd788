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

namespace slm {

constexpr int TAIL_THREADS = 1024;
constexpr int TAIL_WAVES = TAIL_THREADS / 64;

// Device-resident control block of a path solve.  The host only ever reads copies of it.
struct PathCtl {
  int32_t point;       // current path point
  int32_t n_points;
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
};

// One workgroup per lane (blockIdx.x): vectors of lane l start at l * ld (g: l * (ld + 16)).
struct TailArgs {
  PathCtl* ctl;               // [n_lanes]
  int* gdone;                 // [0] = every lane finished (or abort), [1] = lanes finished so far
  int n_lanes;
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
// Stage 1: 64-wide xor butterfly per wavefront.  Stage 2: the 16 wavefront partials go through LDS
// and every 16-lane group folds them with a 16-wide xor butterfly (commutative pairing => the same
// bits in every lane).
template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double (*lds)[TAIL_WAVES]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v[k] += __shfl_xor(v[k], off, 64);
  }
  __syncthreads();  // protect lds from the previous use
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < NV; ++k) lds[k][wave] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    double t = lds[k][lane & (TAIL_WAVES - 1)];
#pragma unroll
    for (int off = TAIL_WAVES / 2; off >= 1; off >>= 1) t += __shfl_xor(t, off, 64);
    v[k] = t;
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
  for (int g0 = 0; g0 < G; g0 += nteams) {  // trip count is uniform across the workgroup
    const int g = g0 + my_team;
    double s = 0.0;
    if (g < G) {
      const int k1 = gstart[g + 1];
      for (int k = gstart[g] + tl; k < k1; k += team) {
        const double x = src[order[k]];
        s = __builtin_fma(x, x, s);
      }
    }
    for (int off = team >> 1; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    if (g < G && tl == 0) fn(g, s);
  }
}

// E = elements per thread (p <= 1024 * E).  Everything a thread needs of its E features is loaded
// once into registers (independent loads: one L2 round trip), the group phase goes through LDS, the
// only other global round trips are the control block and -- for real group penalties -- the group
// tables.
template <int E>
__global__ __launch_bounds__(TAIL_THREADS) void fista_tail_kernel(TailArgs a) {
  __shared__ double red[8][TAIL_WAVES];
  __shared__ double us[E * TAIL_THREADS];
  const int lane_id = blockIdx.x;
  PathCtl* ctl = a.ctl + lane_id;
  if (ctl->done != 0 || a.gdone[0] != 0) return;
  const int tid = threadIdx.x;
  const int p = a.p, G = a.G;
  {  // rebase every per-lane pointer
    const int64_t off = (int64_t)lane_id * a.ld;
    a.beta += off; a.z += off; a.zprev += off; a.gprev += off;
    a.a0 += off; a.b0 += off; a.d0 += off;
    a.g += (int64_t)lane_id * (a.ld + 16);
    a.gscale += (int64_t)lane_id * G;
    const int64_t po = ctl->pt_off;
    a.pts += po;
    a.betas_out += po * p;
    a.infos += po;
    if (a.gn_out != nullptr) a.gn_out += po * G;
  }

  // ---- phase 0: per-feature loads (independent of the control block) ---------------------------
  double zj[E], gj[E], bo[E], a0j[E], gpv[E], zpv[E];
  int gi[E];
  bool ok[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int j = tid + e * TAIL_THREADS;
    ok[e] = j < p;
    const int jj = ok[e] ? j : 0;
    zj[e] = a.z[jj];
    gj[e] = a.g[jj];
    bo[e] = a.beta[jj];
    a0j[e] = a.a0[jj];
    gpv[e] = a.gprev[jj];
    zpv[e] = a.zprev[jj];
    gi[e] = a.singleton ? jj : a.gid[jj];
  }

  // uniform snapshot of the control block (read-only until the final single-thread update)
  const int point = ctl->point;
  const int iter = ctl->iter;
  const double L = ctl->L;
  const double t_old = ctl->t;
  const double tol = ctl->tol;
  const uint32_t flags = ctl->flags;
  const int64_t total_iter = ctl->total_iter;
  const int n_points = ctl->n_points;
  const int max_iter = ctl->max_iter;
  const slm_path_point pt = a.pts[point];
  const double loss_z = a.g[a.ld];
  const double step = 1.0 / L;
  const bool group_pen = (pt.sb != 0.0) || (pt.sd != 0.0);

  // ---- phase 1: gradient step, soft threshold, curvature-guard sums -----------------------------
  //  s[0] = ||b+ - z||^2   s[1] = ||b+||^2   s[2] = (z - b+).(b+ - b)   s[3] = ||g - gprev||^2
  //  s[4] = ||z - zprev||^2   s[5] = ||z||^2   s[6] = #non-finite
  double s[7] = {0, 0, 0, 0, 0, 0, 0};
  double u[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int j = tid + e * TAIL_THREADS;
    u[e] = 0.0;
    if (ok[e]) {
      const double dg = gj[e] - gpv[e], dzz = zj[e] - zpv[e];
      s[3] = __builtin_fma(dg, dg, s[3]);
      s[4] = __builtin_fma(dzz, dzz, s[4]);
      s[5] = __builtin_fma(zj[e], zj[e], s[5]);
      a.gprev[j] = gj[e];
      a.zprev[j] = zj[e];
      double uu = soft(zj[e] - step * gj[e], step * pt.sa * a0j[e]);
      if (group_pen && a.singleton) {
        const double nrm = fabs(uu);
        const double sc = nrm > 0.0 ? fmax(0.0, 1.0 - step * pt.sb * a.b0[j] / nrm) : 0.0;
        uu *= sc / (1.0 + step * pt.sd * a.d0[j]);
      }
      u[e] = uu;
    }
  }
  // ---- phase 2: block soft threshold + ridge shrink per group (teams gather through LDS) --------
  if (group_pen && !a.singleton) {
#pragma unroll
    for (int e = 0; e < E; ++e)
      if (ok[e]) us[tid + e * TAIL_THREADS] = u[e];
    __syncthreads();
    for_each_group_sumsq(us, a.order, a.gstart, G, a.team, [&](int g, double ss) {
      const double nrm = sqrt(ss);
      const double sc = nrm > 0.0 ? fmax(0.0, 1.0 - step * pt.sb * a.b0[g] / nrm) : 0.0;
      a.gscale[g] = sc / (1.0 + step * pt.sd * a.d0[g]);
    });
    __syncthreads();
#pragma unroll
    for (int e = 0; e < E; ++e)
      if (ok[e]) u[e] *= a.gscale[gi[e]];
  }

  // ---- phase 3: reductions --------------------------------------------------------------------
#pragma unroll
  for (int e = 0; e < E; ++e) {
    if (ok[e]) {
      const double bn = u[e];
      const double dz = bn - zj[e];
      s[0] = __builtin_fma(dz, dz, s[0]);
      s[1] = __builtin_fma(bn, bn, s[1]);
      s[2] = __builtin_fma(-dz, bn - bo[e], s[2]);
      if (!isfinite(bn)) s[6] += 1.0;
    }
  }
  block_sum<7>(s, red);

  // ---- phase 4: uniform decisions -------------------------------------------------------------
  const bool nonfinite = s[6] > 0.0 || !isfinite(s[0]) || !isfinite(s[1]) || !isfinite(loss_z);
  // Curvature guard: ||A dz|| / ||dz|| is a lower bound on lambda_max(A), A = X^T W X / n.  If it
  // exceeds L the step 1/L was too long: raise L, discard the step and restart from beta.
  bool l_bad = false;
  double L_new = L;
  if (total_iter > 0 && s[4] > 1e-12 * s[5] && s[4] > 0.0) {
    const double curv = sqrt(s[3] / s[4]);
    if (curv > L * (1.0 + 1e-9)) {
      l_bad = true;
      L_new = 1.02 * curv;
    }
  }
  const bool restart = !(flags & SLM_FLAG_NO_RESTART) && s[2] > 0.0;
  const double t_use = restart ? 1.0 : t_old;
  const double t_new = 0.5 * (1.0 + sqrt(1.0 + 4.0 * t_use * t_use));
  const double mom = (t_use - 1.0) / t_new;
  const double resid = sqrt(s[0]), bnorm = sqrt(s[1]);
  const bool conv = !l_bad && (resid <= tol * bnorm);
  const bool hit_max = (iter + 1 >= max_iter);
  const bool finalize = nonfinite || conv || hit_max;
  const bool cold = (flags & SLM_FLAG_COLD_START) != 0;
  // secant prediction of the next point's start from the last two solutions (see slm_path_point)
  double extrap = 0.0;
  if (finalize && !cold && !nonfinite && point >= 1 && point + 1 < n_points)
    extrap = a.pts[point + 1].extrap;

  // ---- phase 5: state update ------------------------------------------------------------------
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int j = tid + e * TAIL_THREADS;
    if (ok[e]) {
      const double bn = u[e];
      if (l_bad && !finalize) {
        a.z[j] = bo[e];  // step rejected: beta unchanged, momentum dropped
      } else if (finalize) {
        const double out = l_bad ? bo[e] : bn;
        u[e] = out;
        a.betas_out[(int64_t)point * p + j] = out;
        double nxt = cold ? 0.0 : out;
        if (extrap != 0.0) nxt = out + extrap * (out - a.betas_out[(int64_t)(point - 1) * p + j]);
        a.beta[j] = nxt;
        a.z[j] = nxt;
      } else {
        a.z[j] = bn + mom * (bn - bo[e]);
        a.beta[j] = bn;
      }
    }
  }
  if (finalize && a.gn_out != nullptr) {
    // group norms of the reported solution (the reference's auxiliaries.group_norms.value,
    // model/_lasso.py:239-255, consumed by the adaptive re-weighting at _adaptive_lasso.py:364-374)
    __syncthreads();
#pragma unroll
    for (int e = 0; e < E; ++e)
      if (ok[e]) us[tid + e * TAIL_THREADS] = u[e];
    __syncthreads();
    double* gn = a.gn_out + (int64_t)point * G;
    for_each_group_sumsq(us, a.order, a.gstart, G, a.team, [&](int g, double ss) { gn[g] = sqrt(ss); });
  }

  // ---- phase 6: control block ------------------------------------------------------------------
  if (tid == 0) {
    ctl->total_iter = total_iter + 1;
    ctl->L = L_new;
    if (l_bad) ctl->l_bumps += 1;
    if (restart) ctl->restarts += 1;
    if (finalize) {
      slm_point_info info;
      info.n_iter = iter + 1;
      info.status = (conv && !nonfinite) ? SLM_OK : (nonfinite ? SLM_ERR_NON_FINITE : SLM_ERR_NOT_CONVERGED);
      info.resid = resid;
      info.beta_norm = bnorm;
      info.loss = loss_z;
      info.L = L_new;
      a.infos[point] = info;
      ctl->iter = 0;
      ctl->t = 1.0;
      ctl->point = point + 1;
      if (nonfinite) {
        ctl->nonfinite = 1;
        ctl->done = 1;
        a.gdone[0] = 1;  // abort every lane
      } else if (point + 1 >= n_points) {
        ctl->done = 1;
        if (atomicAdd(&a.gdone[1], 1) + 1 == a.n_lanes) a.gdone[0] = 1;
      }
    } else {
      ctl->iter = iter + 1;
      ctl->t = l_bad ? 1.0 : t_new;
    }
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
__global__ __launch_bounds__(TAIL_THREADS) void power_step_kernel(PowerArgs a) {
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

__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
  x ^= x >> 27; x *= 0x94d049bb133111ebull;
  x ^= x >> 31;
  return x;
}

// Deterministic start vector for the power iteration: unit-norm hash noise.
__global__ __launch_bounds__(TAIL_THREADS) void power_init_kernel(double* v, int p, int64_t ld) {
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

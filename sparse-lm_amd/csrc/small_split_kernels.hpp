// SparseGroupLasso(standardize=True) on chip: the operator splitting of sparselm_amd/model/_split.py -- the reference's
// lambda1 ||b||_1 + lambda2 sum_g w_g ||X_g b_g||_2 (src/sparselm/model/_lasso.py:616-639 with the standardised group
// norms of :249-252) -- with ALL of its sweeps in one launch, for the problem sizes of small_kernels.hpp.
//
//     b      <- argmin 1/(2n)||X b - y||^2 + sum_j a_j |b_j| + rho/2 sum_g ||M_g b_g - gamma_g + u_g||^2
//     gamma  <- group soft-threshold of (M b + u) at b_g / rho
//     u      <- u + M b - gamma                                   (over-relaxed; M_g^T M_g = X_g^T X_g)
//
// On the host every sweep is an upload of new targets, an engine solve and a numpy step: 0.3 ms a sweep, 88 sweeps for
// a 100 x 80 fit.  Here the Gram matrix G = X^T X / n is built once in LDS; M_g is the (transposed) Cholesky factor of
// n G_gg -- any square root of X_g^T X_g gives the same norms; a pivot at rounding level drops its direction, as the
// host's SVD truncation does -- so the b-step is a weighted Lasso with matrix G + rho n blockdiag(G_gg) and linear term
// c + rho L (gamma - u): the matrix-vector product of small_kernels.hpp plus a few terms per lane, solved by the same
// accelerated proximal steps + conjugate gradients on the face, warm-started from the sweep before.  The gamma / u
// steps, the residuals, the re-balancing of rho and the stopping rule are those of _split.py, on registers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "small_kernels.hpp"

namespace slm {

struct SplitSglArgs {
  const double* X;     // [n][ld]
  const double* y;
  const double* rw;    // row weights of the dataset (nullptr: ones)
  int64_t n, ld;
  int p, G, singleton;
  const int* order;    // group-sorted order, group of a feature, first position of a group
  const int* gid;
  const int* gstart;
  const double* a;     // [p] l1 weights
  const double* b;     // [G] group weights
  const double* beta0; // [p] warm start (nullptr: zero)
  double* beta_out;    // [p]
  double* gn_out;      // [G] ||X_g b_g||_2 (nullptr: not wanted)
  slm_point_info* info;  // [1]: n_iter = sweeps, rejects = matrix-vector products, resid = max(primal, dual residual)
  double* state;       // [2 ld + 4]: gamma, u (group-sorted order), rho, valid, direct b-steps, factorisations -- kept with the dataset
  int warm;            // continue from `state` (the re-weighting loop of the adaptive estimator)
  double tol, tol_inner, inv_n;
  int max_sweeps, max_iters, gmax, stage_doubles;
};

constexpr double SS_RELAX = 1.6;

static __global__ __launch_bounds__(SM_THREADS) void small_stdsgl_kernel(SplitSglArgs a) {
  extern __shared__ double sm_lds[];  // G [p][p], c [p], vz [p], vu [p], L [p][gmax], then the stage
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = a.p, G = a.G, gm = a.gmax;
  double* Gs = sm_lds;
  double* cs = Gs + p * p;
  double* vz = cs + p;
  double* vu = vz + p;
  double* Ll = vu + p;          // row r of a group's Cholesky factor L (L L^T = n G_gg) at position gs + r: [p][gm]
  double* stage = Ll + p * gm;  // the stage of the build; afterwards the partial products of wavefronts 1..3
  __shared__ double yy_s, sm_bd;
  __shared__ int sm_cmd, sm_m;
  const bool built = sm_build_gram(a.X, a.y, a.rw, a.n, a.ld, p, a.order, a.inv_n, a.stage_doubles, Gs, cs, stage, &yy_s);
  double* pp = stage;
  // direct solves of the b-step on its face (sm_face_factor): group of every position, then the face's head and factor
  int* gpos = reinterpret_cast<int*>(pp + 3 * p);
  int* fidx = gpos + SM_PMAX;
  double* fadd = pp + 3 * p + SM_PMAX;
  double* fdia = fadd + SM_PMAX;
  double* invd = fdia + SM_PMAX;
  double* Ff = invd + SM_PMAX;
  const int face_cap = sm_face_cap(a.stage_doubles - 3 * p - 64);
  const int mchunk = (((p + 3) >> 2) + 7) & ~7;
  if (wave != 0) {
    const int s0w = lane, s1w = lane + 64;
    const bool on0w = s0w < p, on1w = s1w < p;
    const int m_lo = wave * mchunk < p ? wave * mchunk : p, m_hi = (wave + 1) * mchunk < p ? (wave + 1) * mchunk : p;
    for (;;) {
      __syncthreads();
      const int cmd = sm_cmd;
      if (cmd == 0) break;
      if (cmd == 2) {
        sm_face_factor(Gs, p, fidx, fadd, sm_m, Ff, fdia, invd, true, gpos, sm_bd);
        continue;
      }
      double y0, y1;
      sm_partial(Gs, vz, p, m_lo, m_hi, on0w ? s0w : 0, on1w ? s1w : 0, p > 64, y0, y1);
      if (on0w) pp[(wave - 1) * p + s0w] = y0;
      if (on1w) pp[(wave - 1) * p + s1w] = y1;
      __syncthreads();
    }
    return;
  }
  const int s0 = lane, s1 = lane + 64;
  const bool on0 = s0 < p, on1 = s1 < p, wide = p > 64;
  const int sc0 = on0 ? s0 : 0, sc1 = on1 ? s1 : 0;
  const int j0 = on0 ? a.order[s0] : 0, j1 = on1 ? a.order[s1] : 0;
  const int g0 = a.singleton ? j0 : a.gid[j0], g1 = a.singleton ? j1 : a.gid[j1];
  int gs0 = s0, gn0 = 1, gs1 = s1, gn1 = 1;
  if (!a.singleton) {
    if (on0) { gs0 = a.gstart[g0]; gn0 = a.gstart[g0 + 1] - gs0; }
    if (on1) { gs1 = a.gstart[g1]; gn1 = a.gstart[g1 + 1] - gs1; }
  }
  const int r0 = s0 - gs0, r1 = s1 - gs1;  // row of the position inside its group
  const double c0 = on0 ? cs[s0] : 0.0, c1 = on1 ? cs[s1] : 0.0;
  const double thr0 = on0 ? a.a[j0] : 0.0, thr1 = on1 ? a.a[j1] : 0.0;
  const double bg0 = on0 ? a.b[g0] : 0.0, bg1 = on1 ? a.b[g1] : 0.0;
  const double nrows = 1.0 / a.inv_n;
  double rho_n = 0.0;  // rho * n: the weight of blockdiag(G_gg) in the b-step's matrix

  // y = (G + rho_n blockdiag(G_gg)) v
  auto matvec = [&](double v0, double v1, double& y0, double& y1) {
    if (on0) vz[s0] = v0;
    if (on1) vz[s1] = v1;
    if (lane == 0) sm_cmd = 1;
    __syncthreads();
    sm_partial(Gs, vz, p, 0, mchunk < p ? mchunk : p, sc0, sc1, wide, y0, y1);
    __syncthreads();
    y0 += (pp[sc0] + pp[p + sc0]) + pp[2 * p + sc0];
    if (wide) y1 += (pp[sc1] + pp[p + sc1]) + pp[2 * p + sc1];
    if (rho_n != 0.0) {
      if (on0) {
        double t = 0.0;
        for (int m = 0; m < gn0; ++m) t = __builtin_fma(Gs[(gs0 + m) * p + s0], vz[gs0 + m], t);
        y0 = __builtin_fma(rho_n, t, y0);
      }
      if (on1) {
        double t = 0.0;
        for (int m = 0; m < gn1; ++m) t = __builtin_fma(Gs[(gs1 + m) * p + s1], vz[gs1 + m], t);
        y1 = __builtin_fma(rho_n, t, y1);
      }
    }
  };
  auto release_helpers = [&]() {
    if (lane == 0) sm_cmd = 0;
    __syncthreads();
  };
  // products with the groups' factors: (L^T v)_k = sum_{m >= k} L[m][k] v_m and (L w)_m = sum_{k <= m} L[m][k] w_k
  auto mul_Lt = [&](double v0, double v1, double& y0, double& y1) {  // M v
    if (on0) vu[s0] = v0;
    if (on1) vu[s1] = v1;
    sm_lds_sync();
    y0 = y1 = 0.0;
    if (on0)
      for (int m = r0; m < gn0; ++m) y0 = __builtin_fma(Ll[(gs0 + m) * gm + r0], vu[gs0 + m], y0);
    if (on1)
      for (int m = r1; m < gn1; ++m) y1 = __builtin_fma(Ll[(gs1 + m) * gm + r1], vu[gs1 + m], y1);
    __builtin_amdgcn_wave_barrier();
  };
  auto mul_L = [&](double w0, double w1, double& y0, double& y1) {  // M^T w
    if (on0) vu[s0] = w0;
    if (on1) vu[s1] = w1;
    sm_lds_sync();
    y0 = y1 = 0.0;
    if (on0)
      for (int k = 0; k <= r0; ++k) y0 = __builtin_fma(Ll[s0 * gm + k], vu[gs0 + k], y0);
    if (on1)
      for (int k = 0; k <= r1; ++k) y1 = __builtin_fma(Ll[s1 * gm + k], vu[gs1 + k], y1);
    __builtin_amdgcn_wave_barrier();
  };
  auto group_norm = [&](double v0, double v1, double& n0, double& n1) {  // norm of every position's group
    if (on0) vu[s0] = v0;
    if (on1) vu[s1] = v1;
    sm_lds_sync();
    n0 = n1 = 0.0;
    if (on0) {
      double ss = 0.0;
      for (int m = 0; m < gn0; ++m) ss = __builtin_fma(vu[gs0 + m], vu[gs0 + m], ss);
      n0 = sqrt(ss);
    }
    if (on1) {
      double ss = 0.0;
      for (int m = 0; m < gn1; ++m) ss = __builtin_fma(vu[gs1 + m], vu[gs1 + m], ss);
      n1 = sqrt(ss);
    }
    __builtin_amdgcn_wave_barrier();
  };

  if (a.gn_out != nullptr)
    for (int g = lane; g < G; g += 64) a.gn_out[g] = 0.0;  // (groups without columns)
  slm_point_info info;
  memset(&info, 0, sizeof(info));
  info.mode = 2;
  info.status = SLM_ERR_NOT_CONVERGED;
  if (!built) {
    if (lane == 0) a.info[0] = info;
    release_helpers();
    return;
  }

  // ---- the groups' Cholesky factors, all groups in step over the column index -------------------------------------
  for (int e = lane; e < p * gm; e += 64) Ll[e] = 0.0;
  sm_lds_sync();
  for (int j = 0; j < gm; ++j) {
    double va = 0.0, vb = 0.0;
    const bool act0 = on0 && j < gn0 && r0 >= j, act1 = on1 && j < gn1 && r1 >= j;
    if (act0) {
      va = nrows * Gs[s0 * p + gs0 + j];
      for (int k = 0; k < j; ++k) va = __builtin_fma(-Ll[s0 * gm + k], Ll[(gs0 + j) * gm + k], va);
    }
    if (act1) {
      vb = nrows * Gs[s1 * p + gs1 + j];
      for (int k = 0; k < j; ++k) vb = __builtin_fma(-Ll[s1 * gm + k], Ll[(gs1 + j) * gm + k], vb);
    }
    // (a pivot at rounding level: the column is linearly dependent on the ones before -- its direction is dropped)
    if (act0 && r0 == j) Ll[s0 * gm + j] = va > 1e-12 * nrows * Gs[s0 * p + s0] ? sqrt(va) : 0.0;
    if (act1 && r1 == j) Ll[s1 * gm + j] = vb > 1e-12 * nrows * Gs[s1 * p + s1] ? sqrt(vb) : 0.0;
    sm_lds_sync();
    if (act0 && r0 > j) {
      const double pv = Ll[(gs0 + j) * gm + j];
      Ll[s0 * gm + j] = pv > 0.0 ? va / pv : 0.0;
    }
    if (act1 && r1 > j) {
      const double pv = Ll[(gs1 + j) * gm + j];
      Ll[s1 * gm + j] = pv > 0.0 ? vb / pv : 0.0;
    }
    sm_lds_sync();
  }

  // lambda_max(G): twelve power steps (rho_n = 0 here)
  double L;
  {
    double v0 = on0 ? 1.0 + 0.37 * (double)(((unsigned)(s0 * 2654435761u) >> 24) & 0xffu) / 255.0 : 0.0;
    double v1 = on1 ? 1.0 + 0.37 * (double)(((unsigned)(s1 * 2654435761u) >> 24) & 0xffu) / 255.0 : 0.0;
    double lam = 0.0;
    for (int it = 0; it < 12; ++it) {
      double y0, y1;
      matvec(v0, v1, y0, y1);
      if (!on0) y0 = 0.0;
      if (!on1) y1 = 0.0;
      lam = sqrt(sm_sum(y0 * y0 + y1 * y1));
      const double inv = lam > 0.0 ? 1.0 / lam : 0.0;
      v0 = y0 * inv;
      v1 = y1 * inv;
    }
    L = lam * 1.05;
    if (!(L > 0.0)) L = 1.0;
  }

  // ---- state ------------------------------------------------------------------------------------------------------
  double x0 = (on0 && a.beta0) ? a.beta0[j0] : 0.0, x1 = (on1 && a.beta0) ? a.beta0[j1] : 0.0;
  double gam0 = 0.0, gam1 = 0.0, u0 = 0.0, u1 = 0.0, rho = a.inv_n;
  const bool resume = a.warm && a.state != nullptr && a.state[2 * a.ld + 1] == 1.0;
  if (resume) {
    if (on0) { gam0 = a.state[s0]; u0 = a.state[a.ld + s0]; }
    if (on1) { gam1 = a.state[s1]; u1 = a.state[a.ld + s1]; }
    rho = a.state[2 * a.ld];
  } else if (a.beta0 != nullptr) {
    mul_Lt(x0, x1, gam0, gam1);
  }
  long long products = 0;
  bool bad = false;

  // the b-step: weighted Lasso with matrix G + rho_n blockdiag(G_gg) and linear term (ce0, ce1), from (x0, x1)
  auto inner = [&](double ce0, double ce1, double Lp) {
    double z0 = x0, z1 = x1, tk = 1.0, qz0, qz1, zp0 = 0.0, zp1 = 0.0, qp0 = 0.0, qp1 = 0.0;
    bool have_prev = false;
    double mu_rq = 0.0, gnorm = 0.0;
    uint64_t pat_p = ~0ull, pat_n = ~0ull, pat_p1 = ~0ull, pat_n1 = ~0ull;
    int still = 0, it = 0, cg_runs = 0;
    const double tol = a.tol_inner;
    auto prox = [&](double v0, double v1, double t, double& w0, double& w1) {
      w0 = on0 ? soft(v0, t * thr0) : 0.0;
      w1 = on1 ? soft(v1, t * thr1) : 0.0;
    };
    // (plain steps that confirm an accepted point: see small_solve_kernel)
    auto confirm = [&](double v0, double v1, double rn_start, double t) {
      double rn_prev = rn_start, rhoc = 0.0, rn = rn_start, bn = 0.0;
      for (int v = 0; v < 4 && rn > 0.0; ++v) {
        double qv0, qv1, h0, h1;
        matvec(v0, v1, qv0, qv1);
        ++it;
        qv0 = on0 ? qv0 - ce0 : 0.0;
        qv1 = on1 ? qv1 - ce1 : 0.0;
        prox(v0 - t * qv0, v1 - t * qv1, t, h0, h1);
        const double e0 = h0 - v0, e1 = h1 - v1;
        rn = sqrt(sm_sum(e0 * e0 + e1 * e1));
        bn = sqrt(sm_sum(h0 * h0 + h1 * h1));
        if (v > 0) rhoc = fmax(rhoc, rn_prev > 0.0 ? rn / rn_prev : 0.0);
        rn_prev = rn;
        v0 = h0;
        v1 = h1;
      }
      x0 = v0;
      x1 = v1;
      const double err = rhoc < 1.0 ? rhoc / (1.0 - rhoc) * rn : 1e300;
      if (err <= tol * bn || rn * Lp <= kRoundFloor * (gnorm + Lp * bn)) return true;
      if (rhoc > 0.0 && rhoc < 1.0) mu_rq = mu_rq > 0.0 ? fmin(mu_rq, Lp * (1.0 - rhoc)) : Lp * (1.0 - rhoc);
      return false;
    };
    bool conv = false;
    while (it < a.max_iters && !conv) {
      matvec(z0, z1, qz0, qz1);
      ++it;
      qz0 = on0 ? qz0 - ce0 : 0.0;
      qz1 = on1 ? qz1 - ce1 : 0.0;
      if (have_prev) {
        const double dz0 = z0 - zp0, dz1 = z1 - zp1;
        const double dd = sm_sum(dz0 * dz0 + dz1 * dz1);
        if (dd > 0.0) {
          const double rq = sm_sum(dz0 * (qz0 - qp0) + dz1 * (qz1 - qp1)) / dd;
          if (rq > Lp) Lp = 1.05 * rq;
          if (rq > 0.0) mu_rq = mu_rq > 0.0 ? fmin(mu_rq, rq) : rq;
        }
      }
      const double t = 1.0 / Lp;
      double w0, w1;
      prox(z0 - t * qz0, z1 - t * qz1, t, w0, w1);
      const double e0 = z0 - w0, e1 = z1 - w1;
      const double s_kkt = sm_sum(e0 * e0 + e1 * e1), s_b = sm_sum(w0 * w0 + w1 * w1);
      if ((it & 7) == 1) gnorm = sqrt(sm_sum(qz0 * qz0 + qz1 * qz1));
      const double s_rs = sm_sum(e0 * (w0 - x0) + e1 * (w1 - x1));
      if (!(s_kkt == s_kkt) || !(s_b < 1e300)) {
        bad = true;
        break;
      }
      const double kkt = sqrt(s_kkt) * Lp, bnorm = sqrt(s_b);
      double mu_eff = mu_rq > 0.0 ? fmin(mu_rq, Lp) : Lp;
      mu_eff = fmax(mu_eff, kMuFloor * Lp);
      if (kkt <= fmax(tol * bnorm * mu_eff, kRoundFloor * (gnorm + Lp * bnorm))) {
        if (confirm(w0, w1, sqrt(s_kkt), t)) {
          conv = true;
          break;
        }
        z0 = x0; z1 = x1;
        tk = 1.0;
        have_prev = false;
        still = 0;
        pat_p = pat_n = pat_p1 = pat_n1 = ~0ull;
        continue;
      }
      const bool restart = s_rs > 0.0;
      const double tk_new = restart ? 1.0 : 0.5 * (1.0 + sqrt(1.0 + 4.0 * tk * tk));
      const double mom = restart ? 0.0 : (tk - 1.0) / tk_new;
      zp0 = z0; zp1 = z1; qp0 = qz0; qp1 = qz1;
      have_prev = true;
      z0 = w0 + mom * (w0 - x0);
      z1 = w1 + mom * (w1 - x1);
      x0 = w0;
      x1 = w1;
      tk = tk_new;
      const uint64_t np0 = __ballot(on0 && x0 > 0.0), nn0 = __ballot(on0 && x0 < 0.0);
      const uint64_t np1 = __ballot(on1 && x1 > 0.0), nn1 = __ballot(on1 && x1 < 0.0);
      still = (np0 == pat_p && nn0 == pat_n && np1 == pat_p1 && nn1 == pat_n1) ? still + 1 : 0;
      pat_p = np0; pat_n = nn0; pat_p1 = np1; pat_n1 = nn1;
      // conjugate gradients on the face (see small_solve_kernel)
      if (still >= SM_STILL && cg_runs < 6 && (np0 | nn0 | np1 | nn1) != 0ull) {
        ++cg_runs;
        still = 0;
        bool f0 = on0 && x0 != 0.0, f1 = on1 && x1 != 0.0;
        double q0, q1;
        matvec(x0, x1, q0, q1);
        ++it;
        q0 -= ce0;
        q1 -= ce1;
        int hits = 0;
        const int face0 = __popcll(np0 | nn0) + __popcll(np1 | nn1);
        const int cg_cap = 2 * face0 + 10;
        double rr0 = f0 ? -(q0 + copysign(thr0, x0)) : 0.0, rr1 = f1 ? -(q1 + copysign(thr1, x1)) : 0.0;
        double d0v = rr0, d1v = rr1;
        double rr = sm_sum(rr0 * rr0 + rr1 * rr1);
        const double rr_start = rr;
        for (int k = 0; k < cg_cap && it < a.max_iters && rr > 0.0; ++k) {
          double h0, h1;
          matvec(d0v, d1v, h0, h1);
          ++it;
          h0 = f0 ? h0 : 0.0;
          h1 = f1 ? h1 : 0.0;
          const double dHd = sm_sum(d0v * h0 + d1v * h1), dd = sm_sum(d0v * d0v + d1v * d1v);
          if (!(dd > 0.0)) break;
          if (dHd > 0.0) mu_rq = mu_rq > 0.0 ? fmin(mu_rq, dHd / dd) : dHd / dd;
          double alpha = dHd > 1e-14 * Lp * dd ? rr / dHd : 1e300;
          const double lim0 = (f0 && d0v * x0 < 0.0) ? -x0 / d0v : 1e300;
          const double lim1 = (f1 && d1v * x1 < 0.0) ? -x1 / d1v : 1e300;
          const double amax = sm_min(fmin(lim0, lim1));
          const bool hit = alpha >= amax;
          if (hit) alpha = amax;
          if (!(alpha < 1e299)) break;
          x0 = f0 ? __builtin_fma(alpha, d0v, x0) : x0;
          x1 = f1 ? __builtin_fma(alpha, d1v, x1) : x1;
          if (hit) {
            if (f0 && lim0 <= amax) { x0 = 0.0; f0 = false; }
            if (f1 && lim1 <= amax) { x1 = 0.0; f1 = false; }
            matvec(x0, x1, q0, q1);
            ++it;
            q0 -= ce0;
            q1 -= ce1;
            rr0 = f0 ? -(q0 + copysign(thr0, x0)) : 0.0;
            rr1 = f1 ? -(q1 + copysign(thr1, x1)) : 0.0;
            d0v = rr0;
            d1v = rr1;
            rr = sm_sum(rr0 * rr0 + rr1 * rr1);
            if (++hits > face0) break;
            continue;
          }
          rr0 = f0 ? __builtin_fma(-alpha, h0, rr0) : 0.0;
          rr1 = f1 ? __builtin_fma(-alpha, h1, rr1) : 0.0;
          const double rr_new = sm_sum(rr0 * rr0 + rr1 * rr1);
          if (!(rr_new == rr_new)) {
            bad = true;
            break;
          }
          const double xn = sqrt(sm_sum(x0 * x0 + x1 * x1));
          double mu2 = mu_rq > 0.0 ? fmin(mu_rq, Lp) : Lp;
          mu2 = fmax(mu2, kMuFloor * Lp);
          if (sqrt(rr_new) <= 0.1 * fmax(tol * xn * mu2, kRoundFloor * (gnorm + Lp * xn)) || rr_new <= 1e-30 * rr_start) break;
          const double bt = rr_new / rr;
          d0v = __builtin_fma(bt, d0v, rr0);
          d1v = __builtin_fma(bt, d1v, rr1);
          rr = rr_new;
        }
        if (bad) break;
        z0 = x0; z1 = x1;
        tk = 1.0;
        have_prev = false;
        pat_p = pat_n = pat_p1 = pat_n1 = ~0ull;
      }
    }
    products += it;
    return conv;
  };

  // ---- the b-step by ONE direct solve where the face of the sweep before still holds ---------------------------------
  // Between sweeps only the linear term moves; once the splitting has found the support, the b-step's minimiser keeps its
  // face and its signs, and (G~_AA) t_A = c~_A - thr_A s_A gives it exactly: L D L^T of the face (sm_face_factor, kept as
  // long as face and rho stand), two triangular solves, one product for the optimality conditions -- the gradient on the
  // face below what the iteration's stopping rule asks (with the smallest pivot for the curvature), |q_j| <= thr_j off it,
  // every sign kept.  Anything else leaves the sweep to the iteration above.
  if (on0) gpos[s0] = g0;
  if (on1) gpos[s1] = g1;
  uint64_t fm0 = 0ull, fm1 = 0ull;  // the face the factor in LDS belongs to
  double f_rho = -1.0;
  int direct_hits = 0, face_factors = 0;
  auto direct_b = [&](double ce0, double ce1) {
    const bool f0 = on0 && x0 != 0.0, f1 = on1 && x1 != 0.0;
    const uint64_t m0 = __ballot(f0), m1 = __ballot(f1);
    const int n0 = __popcll(m0), m = n0 + __popcll(m1);
    if (m == 0 || m > face_cap) return false;
    const uint64_t below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    const int rk0 = __popcll(m0 & below), rk1 = n0 + __popcll(m1 & below);
    if (!(m0 == fm0 && m1 == fm1 && f_rho == rho_n)) {
      if (f0) { fidx[rk0] = s0; fadd[rk0] = 0.0; }
      if (f1) { fidx[rk1] = s1; fadd[rk1] = 0.0; }
      if (lane == 0) { sm_m = m; sm_bd = rho_n; sm_cmd = 2; }
      __syncthreads();
      sm_face_factor(Gs, p, fidx, fadd, m, Ff, fdia, invd, false, gpos, rho_n);
      fm0 = m0; fm1 = m1; f_rho = rho_n;
      ++face_factors;
    }
    const int i0 = lane, i1 = lane + 64;
    const bool h0 = i0 < m, h1 = i1 < m;
    const double il0 = h0 ? invd[i0] : 1.0, il1 = h1 ? invd[i1] : 1.0;
    if (__ballot((h0 && il0 == 0.0) || (h1 && il1 == 0.0)) != 0ull) return false;  // (a dropped pivot: a singular face)
    const double mu_est = 1.0 / sm_max(fmax(h0 ? il0 : 0.0, h1 ? il1 : 0.0));      // the smallest pivot
    double t0 = 0.0, t1 = 0.0, q0 = 0.0, q1 = 0.0;
    double r0 = f0 ? ce0 - copysign(thr0, x0) : 0.0, r1 = f1 ? ce1 - copysign(thr1, x1) : 0.0;  // right-hand side, then residual
    bool ok = false;
    for (int pass = 0; pass < 2 && !ok; ++pass) {  // (the second pass: one step of iterative refinement)
      __builtin_amdgcn_wave_barrier();
      if (f0) vu[rk0] = r0;
      if (f1) vu[rk1] = r1;
      sm_lds_sync();
      double w0 = h0 ? vu[i0] : 0.0, w1 = h1 ? vu[i1] : 0.0;
      sm_face_solve(Ff, invd, m, lane, w0, w1);
      __builtin_amdgcn_wave_barrier();
      if (h0) vu[i0] = w0;
      if (h1) vu[i1] = w1;
      sm_lds_sync();
      t0 += f0 ? vu[rk0] : 0.0;
      t1 += f1 ? vu[rk1] : 0.0;
      if (__ballot((f0 && !(t0 * x0 > 0.0)) || (f1 && !(t1 * x1 > 0.0))) != 0ull) return false;  // a sign would change (or NaN)
      matvec(t0, t1, q0, q1);
      ++products;
      q0 = on0 ? q0 - ce0 : 0.0;
      q1 = on1 ? q1 - ce1 : 0.0;
      r0 = f0 ? -(q0 + copysign(thr0, x0)) : 0.0;
      r1 = f1 ? -(q1 + copysign(thr1, x1)) : 0.0;
      const double rn = sqrt(sm_sum(r0 * r0 + r1 * r1)), tn = sqrt(sm_sum(t0 * t0 + t1 * t1));
      const double gn = sqrt(sm_sum(q0 * q0 + q1 * q1));
      const double Lt = L * (1.0 + rho_n);
      const double allow = fmax(0.1 * a.tol_inner * tn * fmax(mu_est, kMuFloor * Lt), kRoundFloor * (gn + Lt * tn));
      const bool off0 = on0 && !f0 && fabs(q0) > thr0 + allow, off1 = on1 && !f1 && fabs(q1) > thr1 + allow;
      if (__ballot(off0 || off1) != 0ull) return false;  // a coordinate off the face wants in
      ok = rn <= allow;
    }
    if (!ok) return false;
    x0 = t0;
    x1 = t1;
    ++direct_hits;
    return true;
  };

  // ---- the sweeps --------------------------------------------------------------------------------------------------
  int sweeps = 0;
  bool converged = false;
  double rp = 0.0, rd = 0.0;
  for (sweeps = 1; sweeps <= a.max_sweeps && !bad; ++sweeps) {
    rho_n = rho * nrows;
    double t0, t1;
    mul_L(gam0 - u0, gam1 - u1, t0, t1);
    const double ce0 = __builtin_fma(rho, t0, c0), ce1 = __builtin_fma(rho, t1, c1);
    if (!direct_b(ce0, ce1)) (void)inner(ce0, ce1, L * (1.0 + rho_n));  // (short of its tolerance: absorbed by the sweeps)
    if (bad) break;
    double v0, v1;
    mul_Lt(x0, x1, v0, v1);
    const double vh0 = SS_RELAX * v0 + (1.0 - SS_RELAX) * gam0, vh1 = SS_RELAX * v1 + (1.0 - SS_RELAX) * gam1;
    double nr0, nr1;
    group_norm(vh0 + u0, vh1 + u1, nr0, nr1);
    const double sh0 = on0 ? (nr0 * rho > bg0 ? 1.0 - bg0 / (rho * nr0) : 0.0) : 0.0;
    const double sh1 = on1 ? (nr1 * rho > bg1 ? 1.0 - bg1 / (rho * nr1) : 0.0) : 0.0;
    const double gn0v = (vh0 + u0) * sh0, gn1v = (vh1 + u1) * sh1;
    u0 += vh0 - gn0v;
    u1 += vh1 - gn1v;
    double dl0, dl1, lu0, lu1;
    mul_L(gn0v - gam0, gn1v - gam1, dl0, dl1);
    mul_L(u0, u1, lu0, lu1);
    rp = sqrt(sm_sum((v0 - gn0v) * (v0 - gn0v) + (v1 - gn1v) * (v1 - gn1v)));
    rd = rho * sqrt(sm_sum(dl0 * dl0 + dl1 * dl1));
    gam0 = gn0v;
    gam1 = gn1v;
    const double ep = fmax(fmax(sqrt(sm_sum(v0 * v0 + v1 * v1)), sqrt(sm_sum(gam0 * gam0 + gam1 * gam1))), 1e-300);
    const double ed = fmax(rho * sqrt(sm_sum(lu0 * lu0 + lu1 * lu1)), 1e-300);
    if (!(rp == rp) || !(rd == rd)) {
      bad = true;
      break;
    }
    if (rp <= a.tol * ep && rd <= a.tol * ed) {
      converged = true;
      break;
    }
    if (sweeps == 5 || sweeps == 10 || sweeps == 20 || sweeps == 40 || sweeps == 80 || sweeps == 160 || sweeps == 320) {
      const double ratio = (rp / ep) / fmax(rd / ed, 1e-300);
      if (ratio > 5.0 || ratio < 0.2) {
        const double factor = fmin(10.0, fmax(0.1, sqrt(ratio)));
        u0 /= factor;  // (u is the multiplier divided by rho)
        u1 /= factor;
        rho *= factor;
      }
    }
  }
  if (sweeps > a.max_sweeps) sweeps = a.max_sweeps;

  // gamma is exactly group-sparse, M b only to the residual: a group whose gamma_g vanished is out
  {
    double ng0, ng1;
    group_norm(gam0, gam1, ng0, ng1);
    if (ng0 == 0.0) x0 = 0.0;
    if (ng1 == 0.0) x1 = 0.0;
  }
  // record: coefficients, ||X_g b_g||, loss with the matrix of the data alone
  rho_n = 0.0;
  double q0, q1, v0, v1, nv0, nv1;
  matvec(x0, x1, q0, q1);
  q0 -= c0;
  q1 -= c1;
  mul_Lt(x0, x1, v0, v1);
  group_norm(v0, v1, nv0, nv1);
  if (on0) a.beta_out[j0] = x0;
  if (on1) a.beta_out[j1] = x1;
  if (a.gn_out != nullptr) {
    if (on0 && r0 == 0) a.gn_out[g0] = nv0;
    if (on1 && r1 == 0) a.gn_out[g1] = nv1;
  }
  if (a.state != nullptr) {
    if (on0) { a.state[s0] = gam0; a.state[a.ld + s0] = u0; }
    if (on1) { a.state[s1] = gam1; a.state[a.ld + s1] = u1; }
    if (lane == 0) {
      a.state[2 * a.ld] = rho;
      a.state[2 * a.ld + 1] = bad ? 0.0 : 1.0;
      a.state[2 * a.ld + 2] = (double)direct_hits;
      a.state[2 * a.ld + 3] = (double)face_factors;
    }
  }
  const double loss = 0.5 * sm_sum((on0 ? x0 * (q0 - c0) : 0.0) + (on1 ? x1 * (q1 - c1) : 0.0)) + 0.5 * yy_s;
  const double bn = sqrt(sm_sum(x0 * x0 + x1 * x1));
  if (lane == 0) {
    info.n_iter = sweeps;
    info.status = bad ? SLM_ERR_NON_FINITE : (converged ? SLM_OK : SLM_ERR_NOT_CONVERGED);
    info.resid = fmax(rp, rd);
    info.beta_norm = bn;
    info.loss = loss;
    info.L = rho;
    info.rejects = (int32_t)(products > 2000000000ll ? 2000000000ll : products);
    info.kkt = rp;
    info.mu = rd;
    a.info[0] = info;
  }
  release_helpers();
}

}  // namespace slm

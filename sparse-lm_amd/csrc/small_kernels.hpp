// On-chip solver for the problem sizes sparse-lm's own users run (the reference's tests: 25 x 20 and 25 x 30,
// /root/reference/tests/conftest.py:17-19; its README example: 100 x 80, README.md:42-55): ONE launch per call,
// one workgroup per lane, no pass over X after the first.
//
// The state machine of engine.hip spends a dozen launches per pass; below a few thousand matrix entries every one of
// them is a 4-8 us kernel boundary around microseconds of work, and a fit costs 0.35-2.3 ms however little it
// computes.  Here a lane's workgroup forms its Gram matrix G = X^T W X / n and c = X^T W y / n once (p <= 128: G
// lives in LDS), and wavefront 0 then walks the lane's whole path on
//
//     1/2 b^T G b - c^T b + sum_j a_j |b_j| + sum_g b_g ||b_g||_2 + 1/2 sum_g d_g ||b_g||^2
//
// -- the reference's objective (model/_lasso.py:109-121 with the penalties of :99-107, :267-275, :627-639,
// :795-811) up to the constant 1/(2n) y^T W y -- with every vector in registers (lane l holds positions l and l + 64
// of the group-sorted order) and the matrix-vector product as the only loop: p LDS reads and fused multiply-adds per
// lane, all lanes at once -- half a microsecond at p = 80.  (Coordinate descent, the first version, is a chain through
// every coordinate: 300 cycles each on one wavefront, 10 us per sweep at p = 80 -- twenty products' worth.)
//   * accelerated proximal gradient steps with the gradient-scheme restart (the iteration of SURVEY Appendix C) find the
//     face of the minimiser;
//   * once the sign pattern has stood still for a few steps and the point has no group norms, conjugate gradients on the
//     face -- (G_AA + D) x = c_A - thr_A sign(x_A), warm-started, steps cut at the first sign change (that coordinate
//     leaves the face) -- finish what the proximal steps would crawl along on an ill-conditioned face; the proximal
//     steps that follow confirm the point or extend the face;
//   * faces those do not settle in 24 steps get a direct solve: L D L^T of the face's matrix by wavefronts 1-3
//     (sm_face_factor), two triangular solves on wavefront 0 (sm_face_solve), a projected line search;
//   * points with group norms mix their proximal steps (Anderson acceleration) and take Newton steps on the face of the
//     iterate with the same factorisation (the group norms' curvature added to the Gram block).
// A point is accepted by the rule of the general path (fista_tail_kernel): KKT residual of the gradient point -- the
// proximal-gradient mapping -- below tol * mu * ||b||, mu the smallest curvature <dq, dz> / <dz, dz> measured along the
// steps (not trusted below 1e-6 lambda_max), with the same rounding floor.  A point that has not got there after
// `max_iters` products is reported as SLM_ERR_NOT_CONVERGED and solve_core hands the call to the general path (whose
// model solver has Newton steps): this kernel only has to be fast where it converges.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "grad_kernel.hpp"
#include "tail_kernels.hpp"

namespace slm {

constexpr int SM_PMAX = 128;      // features: two positions per lane of the iterating wavefront
constexpr int SM_THREADS = 256;   // four wavefronts build the Gram; wavefront 0 iterates
constexpr int SM_FACE_ROUNDS = 1; // active-set rounds of a direct solve (more than one did not pay: see the direct block)
constexpr int SM_STILL = 4;       // proximal steps with an unchanged sign pattern before conjugate gradients take over
constexpr int SM_LDS_BYTES = 156 * 1024;  // of the 160 KiB of a CU

struct SmallArgs {
  TailArgs t;          // control blocks, path points, penalties, group structure, outputs -- as fista_tail_kernel uses them
  const double* X;     // [n][ld]
  const double* y;
  const double* rw;    // row weights (nullptr: ones); lane l reads rw + l * rw_stride
  int64_t rw_stride;
  int64_t n;
  double inv_n[SLM_MAX_CELLS];
  int max_iters;       // matrix-vector products per path point before the point is given up
  int cold;            // SLM_FLAG_COLD_START: every point starts from zero
  int stage_doubles;   // LDS left beside the Gram matrix and the vectors: the stage of the rows while G is built
};

__device__ __forceinline__ double sm_sum(double v) { return read_lane63(wave_sum_lane63(v)); }  // DPP scan, no LDS
// Minimum / maximum over the wavefront with DPP moves, the pattern of wave_sum_lane63 (no LDS round trips: the xor
// butterfly of __shfl_xor is six dependent ds_bpermute pairs, 0.25 us of a 1.7 us conjugate-gradient step).  A lane a move
// does not reach (row_mask / out of range) keeps its own value, which is neutral for both.
template <int CTRL, int ROW_MASK, bool IS_MIN>
__device__ __forceinline__ double sm_dpp_fold(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const int lo2 = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
  const int hi2 = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
  const double o = __hiloint2double(hi2, lo2);
  return IS_MIN ? fmin(v, o) : fmax(v, o);
}
template <bool IS_MIN>
__device__ __forceinline__ double sm_extreme(double v) {
  v = sm_dpp_fold<0x111, 0xf, IS_MIN>(v);  // row_shr:1
  v = sm_dpp_fold<0x112, 0xf, IS_MIN>(v);  // row_shr:2
  v = sm_dpp_fold<0x114, 0xf, IS_MIN>(v);  // row_shr:4
  v = sm_dpp_fold<0x118, 0xf, IS_MIN>(v);  // row_shr:8
  v = sm_dpp_fold<0x142, 0xa, IS_MIN>(v);  // row_bcast:15 into rows 1 and 3
  v = sm_dpp_fold<0x143, 0xc, IS_MIN>(v);  // row_bcast:31 into rows 2 and 3
  return read_lane63(v);
}
__device__ __forceinline__ double sm_min(double v) { return sm_extreme<true>(v); }
__device__ __forceinline__ double sm_max(double v) { return sm_extreme<false>(v); }
__device__ __forceinline__ void sm_lds_sync() {  // one wavefront: LDS writes before the reads that follow
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// Gram matrix of [X, y] (rows scaled by sqrt(w)) in the group-sorted order, by all SM_THREADS threads of the workgroup:
// Gs [p][p] = X^T W X / n, cs [p] = X^T W y / n, *yy_out = y^T W y / n (LDS).  `stage`: LDS scratch of stage_doubles.
// Ends with a barrier.  Returns false when the stage holds not even one row.
__device__ __forceinline__ bool sm_build_gram(const double* X, const double* y, const double* rw, int64_t n, int64_t ld, int p,
                                              const int* order, double inv_n, int stage_doubles, double* Gs, double* cs,
                                              double* stage, double* yy_out) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  bool ok = true;
  // ---- Gram matrix of [X, y] (rows scaled by sqrt(w)) in the group-sorted order ---------------------------------
  // G = X^T W X / n, c = X^T W y / n and y^T W y / n are the blocks of one symmetric matrix of p + 1 columns.  The rows
  // are staged through LDS in chunks (what the Gram matrix leaves of it): all four
  // wavefronts load a chunk coalesced, sqrt(w_i) applied once, then every thread accumulates its 4 x 4 blocks of the
  // upper triangle from LDS.  (Straight from global memory every row cost each thread a dependent round trip: 0.25 ms
  // for 400 x 100.)
  {
    const int P1 = p + 1;                 // the y column is position p
    const int nb = (P1 + 3) >> 2;
    const int ps = 4 * nb;                // row stride of the stage: whole blocks, zero beyond the y column
    const int nblocks = nb * (nb + 1) / 2;
    constexpr int MAXB = 3;  // blocks per thread (p + 1 <= 129: 33 x 34 / 2 = 561 blocks over 256 threads)
    double acc[MAXB][4][4];
    int bis[MAXB], bjs[MAXB];
#pragma unroll
    for (int r = 0; r < MAXB; ++r) {
      const int blk = tid + r * SM_THREADS;
      int bi = 0, rest = blk < nblocks ? blk : 0;  // row bi of the triangle holds nb - bi blocks
      while (rest >= nb - bi) {
        rest -= nb - bi;
        ++bi;
      }
      bis[r] = blk < nblocks ? bi : -1;
      bjs[r] = bi + rest;
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[r][u][v] = 0.0;
    }
    const int rows_per_chunk = stage_doubles / ps;
    if (rows_per_chunk > 0) {
      for (int64_t i0 = 0; i0 < n; i0 += rows_per_chunk) {
        const int rows = (int)(n - i0 < rows_per_chunk ? n - i0 : rows_per_chunk);
        __syncthreads();  // (the previous chunk has been consumed)
        for (int r = wave; r < rows; r += SM_THREADS / 64) {  // a wavefront per row: lanes walk the positions
          const int64_t i = i0 + r;
          const double sw = rw ? sqrt(rw[i]) : 1.0;
          const double* row = X + i * ld;
          for (int sidx = lane; sidx < ps; sidx += 64) {
            double v = 0.0;
            if (sidx < p) v = row[order[sidx]] * sw;
            else if (sidx == p) v = y[i] * sw;
            stage[r * ps + sidx] = v;
          }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < MAXB; ++r) {
          if (bis[r] < 0) continue;
          const double* xs_p = stage + 4 * bis[r];
          const double* xt_p = stage + 4 * bjs[r];
          for (int rr = 0; rr < rows; ++rr) {
            double xs[4], xt[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              xs[u] = xs_p[rr * ps + u];
              xt[u] = xt_p[rr * ps + u];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
              for (int v = 0; v < 4; ++v) acc[r][u][v] = __builtin_fma(xs[u], xt[v], acc[r][u][v]);
          }
        }
      }
    } else {  // (no room for a stage beside the Gram matrix: cannot happen for p <= SM_PMAX)
      ok = false;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < MAXB; ++r) {
      if (bis[r] < 0) continue;
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int sp = 4 * bis[r] + u, tp = 4 * bjs[r] + v;
          if (sp > tp || tp > p) continue;  // (upper triangle of the diagonal blocks; positions beyond the y column)
          const double val = acc[r][u][v] * inv_n;
          if (tp < p) {
            Gs[sp * p + tp] = val;
            Gs[tp * p + sp] = val;
          } else if (sp < p) {
            cs[sp] = val;
          } else {
            *yy_out = val;
          }
        }
    }
  }
  __syncthreads();
  return ok;
}

// this lane's part of y = G v over the rows m_lo..m_hi-1 of the (symmetric) matrix: eight terms at a time, their sixteen
// or twenty-four loads asked for together, four chains per half (two terms at a time the product spent two thirds of its
// time waiting for LDS round trips)
__device__ __forceinline__ void sm_partial(const double* Gs, const double* vz, int p, int m_lo, int m_hi, int sc0, int sc1,
                                           bool wide, double& y0, double& y1) {
  double e[4] = {0.0, 0.0, 0.0, 0.0}, f[4] = {0.0, 0.0, 0.0, 0.0};
  int m = m_lo;
  for (; m + 8 <= m_hi; m += 8) {
    double zv[8], ga[8], gb[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      zv[u] = vz[m + u];
      ga[u] = Gs[(m + u) * p + sc0];
      gb[u] = wide ? Gs[(m + u) * p + sc1] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      e[u & 3] = __builtin_fma(ga[u], zv[u], e[u & 3]);
      if (wide) f[u & 3] = __builtin_fma(gb[u], zv[u], f[u & 3]);
    }
  }
  for (; m < m_hi; ++m) {
    const double za = vz[m];
    e[m & 3] = __builtin_fma(Gs[m * p + sc0], za, e[m & 3]);
    if (wide) f[m & 3] = __builtin_fma(Gs[m * p + sc1], za, f[m & 3]);
  }
  y0 = (e[0] + e[2]) + (e[1] + e[3]);
  y1 = (f[0] + f[2]) + (f[1] + f[3]);
}

// value of lane k (uniform) of a wavefront's double
__device__ __forceinline__ double sm_lane(double v, int k) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), k), hi = __builtin_amdgcn_readlane(__double2hiint(v), k);
  return __hiloint2double(hi, lo);
}

// LDS beside the Gram matrix that the direct solves on a face use: the positions of the face, the ridge terms of its
// diagonal, the diagonal itself as gathered, the inverted pivots, and the factor (lower triangle, packed by rows: (i, j)
// at i (i + 1) / 2 + j)
constexpr int SM_FACE_HEAD = 128 + 4 * SM_PMAX;  // doubles in front of the factor: idx, group of a position (ints), add, dia, invd, gv
__device__ __forceinline__ int sm_face_cap(int free_doubles) {  // largest face whose factor fits
  int m = 0;
  while (m < SM_PMAX && (m + 1) * (m + 2) / 2 <= free_doubles - SM_FACE_HEAD) ++m;
  return m;
}

__device__ __forceinline__ double sm_recip(double d) {  // 1 / d: the hardware's estimate and two Newton steps
  double y = __builtin_amdgcn_rcp(d);
  y = __builtin_fma(y, __builtin_fma(-d, y, 1.0), y);
  y = __builtin_fma(y, __builtin_fma(-d, y, 1.0), y);
  return y;
}

// H = G_AA + diag(add) of the face A = idx[0..m) factored as L D L^T by the workgroup, right-looking in blocks of FOUR
// columns, two barriers per block: F[i][c] (i > c) keeps u_ic = L[i][c] d_c, the diagonal keeps d_c, invd[c] = 1 / d_c.
//   panel:  every thread factors the 4 x 4 diagonal block for itself (ten loads, four pivots); the thread of a row below
//           it finishes that row's entries: u_i1 = a_i1 - L_i0 u_10, u_i2 = a_i2 - L_i0 u_20 - L_i1 u_21, ...
//   update: F[i][j] -= sum_c u_ic invd_c u_jc behind the panel, a thread's columns' four terms kept in registers, all
//           loads of a row ahead of its stores.  (The finished diagonal block is written here: nobody reads it any more.)
// (One column per barrier, the first version, was a chain of dependent LDS round trips per column: 80 us for 80 unknowns.)
// A pivot at rounding level (a column that depends on the ones before it: p > n, duplicated features) is dropped:
// invd = 0, its unknown stays where it is.  `worker`: wavefront 0 -- which holds the iteration's state in its registers --
// only keeps the barriers; the other three do the arithmetic.  Starts and ends with a barrier.
// `fgrp` / `bd` (the splitting of small_split_kernels.hpp): entries of G inside a group -- fgrp[position] equal -- count
// (1 + bd) times: the matrix is G + bd blockdiag(G_gg).  `fgv` (Newton steps on faces with group norms): inside a group
// the entry (i, j) also loses fgv[i] fgv[j] -- the rank-one part of the Hessian of b ||x_g||, with fgv = x sqrt(b / ||x_g||^3)
// and b / ||x_g|| in the diagonal term `fadd`.
__device__ __forceinline__ void sm_face_factor(const double* Gs, int p, const int* fidx, const double* fadd, int m, double* F,
                                               double* fdia, double* invd, bool worker, const int* fgrp = nullptr,
                                               double bd = 0.0, const double* fgv = nullptr) {
  constexpr int NB = SM_PMAX / 16;          // column batches of a thread in the update
  constexpr int TW = SM_THREADS - 64;       // working threads
  const int t = (int)threadIdx.x - 64;      // 0 .. TW-1 for the workers
  const int ty = t >> 4, tx = t & 15;       // 12 x 16
  if (worker) {
    for (int i = ty; i < m; i += TW / 16) {
      const int si = fidx[i], base = i * (i + 1) / 2;
      for (int j = tx; j <= i; j += 16) {
        const int sj = fidx[j];
        double v = Gs[si * p + sj];
        if (fgrp != nullptr && fgrp[si] == fgrp[sj]) {
          v = __builtin_fma(bd, v, v);
          if (fgv != nullptr) v = __builtin_fma(-fgv[i], fgv[j], v);
        }
        if (i == j) v += fadd[i];
        F[base + j] = v;
        if (i == j) fdia[i] = v;
      }
    }
  }
  for (int k = 0; k < m; k += 4) {
    __syncthreads();
    double n0 = 0.0, n1 = 0.0, n2 = 0.0, n3 = 0.0, d1 = 0.0, d2 = 0.0, d3 = 0.0, u21 = 0.0, u31 = 0.0, u32 = 0.0;
    const bool v1 = k + 1 < m, v2 = k + 2 < m, v3 = k + 3 < m;
    const int r1 = v1 ? k + 1 : k, r2 = v2 ? k + 2 : k, r3 = v3 ? k + 3 : k;
    const int bk1 = r1 * (r1 + 1) / 2, bk2 = r2 * (r2 + 1) / 2, bk3 = r3 * (r3 + 1) / 2;
    if (worker) {
      // ---- the 4 x 4 diagonal block (rows beyond m count as absent) ----
      const int bk0 = k * (k + 1) / 2;
      const double a00 = F[bk0 + k], a10 = F[bk1 + k], a11 = F[bk1 + r1], a20 = F[bk2 + k], a21 = F[bk2 + r1], a22 = F[bk2 + r2];
      const double a30 = F[bk3 + k], a31 = F[bk3 + r1], a32 = F[bk3 + r2], a33 = F[bk3 + r3];
      const double g0 = fdia[k], g1 = fdia[r1], g2 = fdia[r2], g3 = fdia[r3];
      // this thread's row below the block
      const int i = k + 4 + t;
      const int bi = i * (i + 1) / 2;
      double w0 = 0.0, x1 = 0.0, x2 = 0.0, x3 = 0.0;
      if (i < m) {
        w0 = F[bi + k];
        x1 = F[bi + k + 1];
        x2 = F[bi + k + 2];
        x3 = F[bi + k + 3];
      }
      n0 = a00 > 1e-13 * g0 ? sm_recip(a00) : 0.0;
      const double u10 = v1 ? a10 : 0.0, u20 = v2 ? a20 : 0.0, u30 = v3 ? a30 : 0.0;
      const double l10 = u10 * n0, l20 = u20 * n0, l30 = u30 * n0;
      d1 = __builtin_fma(-l10, u10, a11);
      n1 = (v1 && d1 > 1e-13 * g1) ? sm_recip(d1) : 0.0;
      u21 = v2 ? __builtin_fma(-l20, u10, a21) : 0.0;
      u31 = v3 ? __builtin_fma(-l30, u10, a31) : 0.0;
      const double l21 = u21 * n1, l31 = u31 * n1;
      d2 = __builtin_fma(-l21, u21, __builtin_fma(-l20, u20, a22));
      n2 = (v2 && d2 > 1e-13 * g2) ? sm_recip(d2) : 0.0;
      u32 = v3 ? __builtin_fma(-l31, u21, __builtin_fma(-l30, u20, a32)) : 0.0;
      const double l32 = u32 * n2;
      d3 = __builtin_fma(-l32, u32, __builtin_fma(-l31, u31, __builtin_fma(-l30, u30, a33)));
      n3 = (v3 && d3 > 1e-13 * g3) ? sm_recip(d3) : 0.0;
      if (i < m) {
        const double m0 = w0 * n0;
        const double w1 = __builtin_fma(-m0, u10, x1);
        const double m1 = w1 * n1;
        const double w2 = __builtin_fma(-m1, u21, __builtin_fma(-m0, u20, x2));
        const double m2 = w2 * n2;
        const double w3 = __builtin_fma(-m2, u32, __builtin_fma(-m1, u31, __builtin_fma(-m0, u30, x3)));
        F[bi + k + 1] = w1;  // (its own row: nobody else reads it before the barrier)
        F[bi + k + 2] = w2;
        F[bi + k + 3] = w3;
      }
    }
    __syncthreads();
    if (worker) {
      if (t == 0) {  // the finished block and its pivots
        invd[k] = n0;
        if (v1) { invd[r1] = n1; F[bk1 + r1] = d1; }
        if (v2) { invd[r2] = n2; F[bk2 + r1] = u21; F[bk2 + r2] = d2; }
        if (v3) { invd[r3] = n3; F[bk3 + r1] = u31; F[bk3 + r2] = u32; F[bk3 + r3] = d3; }
      }
      // ---- update behind the panel: rows i = k + 4 + ty + 12 a, columns j = k + 4 + tx + 16 b <= i ----
      const int jb = k + 4 + tx;
      double cj[NB][4];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int j = jb + 16 * b;
        const int bj = j * (j + 1) / 2 + k;
        const bool on = j < m;
        cj[b][0] = on ? F[bj] * n0 : 0.0;
        cj[b][1] = on ? F[bj + 1] * n1 : 0.0;
        cj[b][2] = on ? F[bj + 2] * n2 : 0.0;
        cj[b][3] = on ? F[bj + 3] * n3 : 0.0;
      }
      for (int i = k + 4 + ty; i < m; i += TW / 16) {
        const int base = i * (i + 1) / 2;
        const double e0 = F[base + k], e1 = F[base + k + 1], e2 = F[base + k + 2], e3 = F[base + k + 3];
        double f[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const int j = jb + 16 * b;
          f[b] = j <= i ? F[base + j] : 0.0;
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const int j = jb + 16 * b;
          if (j <= i)
            F[base + j] = __builtin_fma(-e3, cj[b][3], __builtin_fma(-e2, cj[b][2], __builtin_fma(-e1, cj[b][1], __builtin_fma(-e0, cj[b][0], f[b]))));
        }
      }
    }
  }
  __syncthreads();
}

// the two triangular solves with the factor above, on ONE wavefront: lane l holds unknowns l and l + 64 (w0, w1: the
// right-hand side on entry, the solution on return).  The entries of L and the pivots a step needs do not depend on the
// running vector: eight steps' worth are asked for together, the chain is a broadcast and two products per step.
__device__ __forceinline__ void sm_face_solve(const double* F, const double* invd, int m, int lane, double& w0, double& w1) {
  const int i0 = lane, i1 = lane + 64;
  const bool h0 = i0 < m, h1 = i1 < m;
  const double il0 = h0 ? invd[i0] : 0.0, il1 = h1 ? invd[i1] : 0.0;
  const int b0 = i0 * (i0 + 1) / 2, b1 = i1 * (i1 + 1) / 2;
  for (int k0 = 0; k0 < m; k0 += 8) {  // L w = rhs  (L[i][k] = F[i][k] invd[k])
    double fa[8], fb[8], iv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = k0 + u;
      iv[u] = k < m ? invd[k] : 0.0;
      fa[u] = (h0 && i0 > k) ? F[b0 + k] : 0.0;
      fb[u] = (h1 && i1 > k) ? F[b1 + k] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = k0 + u;
      if (k < m) {
        const double ck = (k < 64 ? sm_lane(w0, k) : sm_lane(w1, k - 64)) * iv[u];
        w0 = __builtin_fma(-fa[u], ck, w0);
        w1 = __builtin_fma(-fb[u], ck, w1);
      }
    }
  }
  w0 *= il0;  // D
  w1 *= il1;
  for (int k0 = m - 1; k0 > 0; k0 -= 8) {  // L^T x = w
    double fa[8], fb[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = k0 - u;
      const int bk = k * (k + 1) / 2;
      fa[u] = (k > 0 && i0 < k) ? F[bk + i0] * il0 : 0.0;
      fb[u] = (k > 0 && h1 && i1 < k) ? F[bk + i1] * il1 : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = k0 - u;
      if (k > 0) {
        const double xk = k < 64 ? sm_lane(w0, k) : sm_lane(w1, k - 64);
        w0 = __builtin_fma(-fa[u], xk, w0);
        w1 = __builtin_fma(-fb[u], xk, w1);
      }
    }
  }
}

static __global__ __launch_bounds__(SM_THREADS) void small_solve_kernel(SmallArgs a) {
  extern __shared__ double sm_lds[];  // G [p][p], c [p], vz [p], vu [p], then the stage
  const int lane_id = blockIdx.x;
  PathCtl* ctl = a.t.ctl + lane_id;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = a.t.p, G = a.t.G;
  double* Gs = sm_lds;
  double* cs = Gs + p * p;
  double* vz = cs + p;   // operand of the matrix-vector product
  double* vu = vz + p;   // group norms of the proximal step
  const int64_t voff = (int64_t)lane_id * a.t.ld;
  const double* a0 = a.t.a0 + voff;
  const double* b0 = a.t.b0 + voff;
  const double* d0 = a.t.d0 + voff;
  const double* rw = a.rw ? a.rw + (int64_t)lane_id * a.rw_stride : nullptr;
  const double inv_n = a.inv_n[lane_id];
  const int64_t n = a.n, ld = a.t.ld;
  bool bad_setup = false;
  const unsigned long long tk0 = wall_clock64();

  __shared__ double yy_s;
  if (!sm_build_gram(a.X, a.y, rw, n, ld, p, a.t.order, inv_n, a.stage_doubles, Gs, cs, vu + p, &yy_s)) bad_setup = true;
  // Wavefront 0 iterates; the other three serve its matrix-vector products, a quarter of the rows of G each: they wait
  // at the barrier, multiply when the command word says so, leave when it says zero.
  __shared__ int sm_cmd, sm_m, sm_grp;
  double* pp = vu + p;  // [3][p]: the partial products of wavefronts 1..3 (the stage of the build is free now)
  int* fidx = reinterpret_cast<int*>(pp + 3 * p);  // direct solves on a face (sm_face_factor)
  int* gpos = fidx + SM_PMAX;                      // group of every position (Newton steps on faces with group norms)
  double* fadd = pp + 3 * p + 128;
  double* fdia = fadd + SM_PMAX;
  double* invd = fdia + SM_PMAX;
  double* fgv = invd + SM_PMAX;
  double* Ff = fgv + SM_PMAX;
  const int face_cap = sm_face_cap(a.stage_doubles - 3 * p);
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
        sm_face_factor(Gs, p, fidx, fadd, sm_m, Ff, fdia, invd, true, sm_grp ? gpos : nullptr, 0.0, sm_grp ? fgv : nullptr);
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
  const double yy = yy_s;  // 1/n y^T W y: the constant of the loss
  const unsigned long long tk1 = wall_clock64();
  unsigned long long tk_it = 0, tk_cg = 0, tk_rec = 0;

  // ---- per-position constants: lane l holds positions l and l + 64 ------------------------------------------
  const int s0 = lane, s1 = lane + 64;
  const bool on0 = s0 < p, on1 = s1 < p;
  const bool wide = p > 64;
  const int sc0 = on0 ? s0 : 0, sc1 = on1 ? s1 : 0;  // (lanes beyond p read column 0; what they accumulate is never used)
  const int j0 = on0 ? a.t.order[s0] : 0, j1 = on1 ? a.t.order[s1] : 0;
  const int g0 = a.t.singleton ? j0 : a.t.gid[j0], g1 = a.t.singleton ? j1 : a.t.gid[j1];
  if (on0) gpos[s0] = g0;
  if (on1) gpos[s1] = g1;
  const double c0 = on0 ? cs[s0] : 0.0, c1 = on1 ? cs[s1] : 0.0;
  // (not const: the rounds of a re-weighted lane renew them, see the end of the point loop)
  double A0 = on0 ? a0[j0] : 0.0, A1 = on1 ? a0[j1] : 0.0;
  double B0 = on0 ? b0[g0] : 0.0, B1 = on1 ? b0[g1] : 0.0;
  const double D0 = on0 ? d0[g0] : 0.0, D1 = on1 ? d0[g1] : 0.0;
  // first position / size of the group of each position (group-sorted order: groups are contiguous)
  int gs0 = s0, gn0 = 1, gs1 = s1, gn1 = 1;
  if (!a.t.singleton) {
    if (on0) { gs0 = a.t.gstart[g0]; gn0 = a.t.gstart[g0 + 1] - gs0; }
    if (on1) { gs1 = a.t.gstart[g1]; gn1 = a.t.gstart[g1 + 1] - gs1; }
  }

  // y = G v for the vector held as (v0, v1): v goes through LDS (every lane needs all of it), the rows of G are read
  // along the lanes (symmetric: row m holds column m); the four wavefronts take a quarter of the rows each
  // Up to 32 positions the product stays in THIS wavefront: lane (h, k) = (lane >> 5, lane & 31) sums the rows 16 h .. 16 h + 15
  // of column k -- sixteen matrix entries and sixteen operand entries, all asked for at once -- and the halves meet in one
  // cross-half move: no barrier, no helper wavefront (they keep waiting for a factorisation).  With the helpers a product of
  // the reference's 25 x 30 fixture cost 0.78 us, two barriers around eight rows each, half of the kernel's time.
  const bool one_wave = p <= 32;
  auto matvec = [&](double v0, double v1, double& y0, double& y1) {
    if (one_wave) {
      if (on0) vz[s0] = v0;
      sm_lds_sync();
      const int hh = lane >> 5, kk = lane & 31, kc = kk < p ? kk : 0;
      double gv[16], zv[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int c = 16 * hh + i, cc = c < p ? c : 0;
        gv[i] = Gs[cc * p + kc];
        zv[i] = vz[cc];
      }
      double e[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int i = 0; i < 16; ++i) e[i & 3] = __builtin_fma(gv[i], 16 * hh + i < p ? zv[i] : 0.0, e[i & 3]);
      const double part = (e[0] + e[2]) + (e[1] + e[3]);
      y0 = part + __shfl_xor(part, 32, 64);
      y1 = 0.0;
      __builtin_amdgcn_wave_barrier();  // (vz is written again by the next product: keep the reads above it)
      return;
    }
    if (on0) vz[s0] = v0;
    if (on1) vz[s1] = v1;
    if (lane == 0) sm_cmd = 1;
    __syncthreads();
    sm_partial(Gs, vz, p, 0, mchunk < p ? mchunk : p, sc0, sc1, wide, y0, y1);
    __syncthreads();
    y0 += (pp[sc0] + pp[p + sc0]) + pp[2 * p + sc0];
    if (wide) y1 += (pp[sc1] + pp[p + sc1]) + pp[2 * p + sc1];
  };
  auto release_helpers = [&]() {  // (every way out of the iteration passes here: the other wavefronts wait at a barrier)
    if (lane == 0) sm_cmd = 0;
    __syncthreads();
  };

  // lambda_max(G): twelve power steps from a fixed start, 5 % margin (the curvature test below repairs an under-estimate)
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
    if (!(L > 0.0)) L = 1.0;  // X == 0 (on the rows that count)
  }

  // ---- the lane's path ---------------------------------------------------------------------------------------
  const int64_t po = ctl->pt_off;  // (0 when the lanes are ranges of one path: their point indices are global)
  const slm_path_point* pts = a.t.pts + po;
  double* betas_out = a.t.betas_out + po * p;
  double* gn_out = a.t.gn_out ? a.t.gn_out + po * G : nullptr;
  slm_point_info* infos = a.t.infos + po;
  const int first = ctl->point, last = ctl->n_points;
  const double tol = ctl->tol;
  const unsigned long long tk2 = wall_clock64();
  bool bad = false;
  long long iters_all = 0;
  int face_solves = 0;  // direct solves on a face (SLM_TRACE=2)
  // re-weighted rounds (PathCtl::rw_*): the points of the lane are the solves of an Adaptive* estimator's loop
  const int rw_on = ctl->rw_on;
  const double rw_coef = ctl->rw_coef, rw_numer = ctl->rw_numer, rw_eps = ctl->rw_eps, rw_tol = ctl->rw_tol;
  const int rw_ncoef = ctl->rw_ncoef, rw_ngroup = ctl->rw_ngroup;
  const double* rw_gscale = a.t.gscale + (int64_t)lane_id * G;
  int rounds = 0;
  if (bad_setup) {  // (nothing was built: every point goes to the general path)
    for (int point = first + lane; point < last; point += 64) {
      slm_point_info info;
      memset(&info, 0, sizeof(info));
      info.status = SLM_ERR_NOT_CONVERGED;
      info.mode = 2;
      infos[point] = info;
    }
    if (lane == 0) ctl->done = 1;
    release_helpers();
    return;
  }
  // warm start of the first point (solve_setup_kernel left it in beta: zero, or the caller's beta0)
  double x0 = on0 ? a.t.beta[voff + j0] : 0.0, x1 = on1 ? a.t.beta[voff + j1] : 0.0;
  for (int point = first; point < last && !bad; ++point) {
    const slm_path_point pt = pts[point];
    const double pa0 = pt.sa * A0, pa1 = pt.sa * A1;
    const double pb0 = pt.sb * B0, pb1 = pt.sb * B1;
    const double pd0 = pt.sd * D0, pd1 = pt.sd * D1;
    if (a.cold && point > first) x0 = x1 = 0.0;
    // positions of real groups with a group term take part in a group norm; elsewhere b is a second l1 weight
    const bool real0 = on0 && gn0 > 1 && pb0 != 0.0, real1 = on1 && gn1 > 1 && pb1 != 0.0;
    const bool lasso_type = __ballot(real0 || real1) == 0ull;
    const double thr0 = pa0 + (gn0 == 1 ? pb0 : 0.0), thr1 = pa1 + (gn1 == 1 ? pb1 : 0.0);
    double Lp = L;  // the step bound of this point (raised by the curvature test)
    // u = prox_t(v) of the point's penalty
    auto prox = [&](double v0, double v1, double t, double& u0, double& u1) {
      u0 = on0 ? soft(v0, t * thr0) : 0.0;
      u1 = on1 ? soft(v1, t * thr1) : 0.0;
      if (!lasso_type) {
        if (on0) vu[s0] = u0;
        if (on1) vu[s1] = u1;
        sm_lds_sync();
        if (real0) {
          double ss = 0.0;
          for (int m = 0; m < gn0; ++m) ss = __builtin_fma(vu[gs0 + m], vu[gs0 + m], ss);
          const double nrm = sqrt(ss);
          u0 *= nrm > 0.0 ? fmax(0.0, 1.0 - t * pb0 / nrm) : 0.0;
        }
        if (real1) {
          double ss = 0.0;
          for (int m = 0; m < gn1; ++m) ss = __builtin_fma(vu[gs1 + m], vu[gs1 + m], ss);
          const double nrm = sqrt(ss);
          u1 *= nrm > 0.0 ? fmax(0.0, 1.0 - t * pb1 / nrm) : 0.0;
        }
        __builtin_amdgcn_wave_barrier();
      }
      u0 /= 1.0 + t * pd0;
      u1 /= 1.0 + t * pd1;
    };

    double z0 = x0, z1 = x1, tk = 1.0;
    double qz0 = 0.0, qz1 = 0.0;            // G z - c at the last gradient point
    double zp0 = 0.0, zp1 = 0.0, qp0 = 0.0, qp1 = 0.0;  // ... and the one before (curvature along the step)
    bool have_prev = false;
    double mu_rq = 0.0, kkt = 0.0, bnorm = 0.0, gnorm = 0.0, resid = 0.0;
    uint64_t pat_p = ~0ull, pat_n = ~0ull, pat_p1 = ~0ull, pat_n1 = ~0ull;
    int still = 0, it = 0, cg_runs = 0;
    bool conv = false;
    // A point that passes the stopping rule is CONFIRMED by plain proximal steps.  The curvatures measured along
    // accelerated moves (momentum, conjugate gradients, Anderson mixing) need not see the flattest direction of the face, and
    // a residual this small is judged by them; the residuals of plain steps contract by 1 - mu / L per step along the
    // slowest direction still present, and rho / (1 - rho) ||r|| estimates the distance to the minimiser from that
    // contraction itself -- four products.  Returns true with the point in (x0, x1); false with the last gradient point
    // and its gradient in (yl, ql) and mu_rq tightened to what the contraction says.
    auto confirm = [&](double v0, double v1, double rn_start, double t, double& yl0, double& yl1, double& ql0, double& ql1) {
      double rn_prev = rn_start, rho = 0.0, rn = rn_start, bn = bnorm;
      for (int v = 0; v < 4 && rn > 0.0; ++v) {
        double qv0, qv1, h0, h1;
        matvec(v0, v1, qv0, qv1);
        ++it;
        qv0 = on0 ? qv0 - c0 : 0.0;
        qv1 = on1 ? qv1 - c1 : 0.0;
        prox(v0 - t * qv0, v1 - t * qv1, t, h0, h1);
        const double e0 = h0 - v0, e1 = h1 - v1;
        rn = sqrt(sm_sum(e0 * e0 + e1 * e1));
        bn = sqrt(sm_sum(h0 * h0 + h1 * h1));
        if (v > 0) rho = fmax(rho, rn_prev > 0.0 ? rn / rn_prev : 0.0);  // (the first ratio straddles the accelerated move)
        rn_prev = rn;
        yl0 = v0; yl1 = v1; ql0 = qv0; ql1 = qv1;
        v0 = h0;
        v1 = h1;
      }
      const double err = rho < 1.0 ? rho / (1.0 - rho) * rn : 1e300;  // (no contraction seen: nothing is confirmed)
      x0 = v0;
      x1 = v1;
      bnorm = bn;
      resid = rn;
      kkt = rn * Lp;
      if (err <= tol * bn || rn * Lp <= kRoundFloor * (gnorm + Lp * bn)) {
        if (rn > 0.0 && err > 0.0) mu_rq = kkt / err;  // (what the record reports: kkt / mu = the distance estimate)
        return true;
      }
      if (rho > 0.0 && rho < 1.0) mu_rq = mu_rq > 0.0 ? fmin(mu_rq, Lp * (1.0 - rho)) : Lp * (1.0 - rho);
      return false;
    };

    // (points with group norms have no conjugate gradients on their faces: where the Anderson mixing does not get there in
    //  a few hundred products -- p > n with groups -- the general path's Newton steps do, in a handful of passes)
    const int it_cap = lasso_type ? a.max_iters : (a.max_iters < 320 ? a.max_iters : 320);
    if (!lasso_type) {
      // ---- points with group norms: proximal gradient steps with Anderson acceleration ------------------------------
      // T(y) = prox_t(y - t (G y - c)) is the fixed-point map of the minimiser; on the face of the minimiser it is smooth,
      // and the combination of the last SM_AA images T(y_i) whose residuals T(y_i) - y_i cancel best (weights summing to
      // one: a four-by-four system every lane solves for itself) converges like a Krylov method where the plain steps
      // crawl (p > n, correlated groups: the reference's own fixtures).  Guarded: a candidate is taken only if the
      // objective falls at least as far below F(y) as a proximal step guarantees, ||T(y) - y||^2 / (2 t); otherwise the
      // step itself is taken.  One product per accepted candidate, two per refusal.
      constexpr int SM_AA = 4;
      auto penalty = [&](double v0, double v1) {
        double pv = (on0 ? thr0 * fabs(v0) + 0.5 * pd0 * v0 * v0 : 0.0) + (on1 ? thr1 * fabs(v1) + 0.5 * pd1 * v1 * v1 : 0.0);
        if (on0) vu[s0] = v0;
        if (on1) vu[s1] = v1;
        sm_lds_sync();
        if (real0 && s0 == gs0) {  // (once per group: its first member)
          double ss = 0.0;
          for (int m = 0; m < gn0; ++m) ss = __builtin_fma(vu[gs0 + m], vu[gs0 + m], ss);
          pv += pb0 * sqrt(ss);
        }
        if (real1 && s1 == gs1) {
          double ss = 0.0;
          for (int m = 0; m < gn1; ++m) ss = __builtin_fma(vu[gs1 + m], vu[gs1 + m], ss);
          pv += pb1 * sqrt(ss);
        }
        __builtin_amdgcn_wave_barrier();
        return sm_sum(pv);
      };
      double y0 = x0, y1 = x1, qy0, qy1;
      matvec(y0, y1, qy0, qy1);
      ++it;
      qy0 = on0 ? qy0 - c0 : 0.0;
      qy1 = on1 ? qy1 - c1 : 0.0;
      double Fy = 0.5 * sm_sum(y0 * (qy0 - c0) + y1 * (qy1 - c1)) + penalty(y0, y1);
      double gh0[SM_AA], gh1[SM_AA], rh0[SM_AA], rh1[SM_AA], M[SM_AA][SM_AA];
      int nh = 0;
#pragma unroll
      for (int i = 0; i < SM_AA; ++i) {
        gh0[i] = gh1[i] = rh0[i] = rh1[i] = 0.0;
#pragma unroll
        for (int k = 0; k < SM_AA; ++k) M[i][k] = 0.0;
      }
      // Newton steps on the face of the iterate (its non-zero coordinates; groups at zero stay out): on the face the
      // objective is smooth -- gradient q + thr s + d x + b x / ||x_g||, Hessian G_AA + diag(d + b / ||x_g||) minus the
      // rank-one b x_g x_g^T / ||x_g||^3 of every group -- and one L D L^T of that Hessian (sm_face_factor) gives the step the
      // mixing of proximal steps needs tens of products for, or never finds (p > n with groups: the reference's own
      // fixture spent its 320 products and went to the general path).  Taken along the projected segment (coordinates
      // that would change sign held at zero) at the first of 1, 1/2, 1/4, ... that lowers the objective; the proximal
      // steps between two Newton steps move groups in and out of the face.
      int newton_left = 8, next_newton = it + 24;
      auto newton = [&]() {
        const bool f0 = on0 && y0 != 0.0, f1 = on1 && y1 != 0.0;
        const uint64_t m0 = __ballot(f0), m1 = __ballot(f1);
        const int n0 = __popcll(m0), m = n0 + __popcll(m1);
        if (m == 0 || m > face_cap) return false;
        // norms of the groups of y
        if (on0) vu[s0] = y0;
        if (on1) vu[s1] = y1;
        sm_lds_sync();
        double nr0 = 0.0, nr1 = 0.0;
        if (real0 && f0) {
          double ss = 0.0;
          for (int mm = 0; mm < gn0; ++mm) ss = __builtin_fma(vu[gs0 + mm], vu[gs0 + mm], ss);
          nr0 = sqrt(ss);
        }
        if (real1 && f1) {
          double ss = 0.0;
          for (int mm = 0; mm < gn1; ++mm) ss = __builtin_fma(vu[gs1 + mm], vu[gs1 + mm], ss);
          nr1 = sqrt(ss);
        }
        __builtin_amdgcn_wave_barrier();
        const double bn0 = (real0 && nr0 > 0.0) ? pb0 / nr0 : 0.0, bn1 = (real1 && nr1 > 0.0) ? pb1 / nr1 : 0.0;
        const double gr0 = f0 ? qy0 + copysign(thr0, y0) + (pd0 + bn0) * y0 : 0.0;
        const double gr1 = f1 ? qy1 + copysign(thr1, y1) + (pd1 + bn1) * y1 : 0.0;
        const uint64_t below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
        const int rk0 = __popcll(m0 & below), rk1 = n0 + __popcll(m1 & below);
        if (f0) { fidx[rk0] = s0; fadd[rk0] = pd0 + bn0; fgv[rk0] = nr0 > 0.0 ? y0 * sqrt(bn0) / nr0 : 0.0; vu[rk0] = -gr0; }
        if (f1) { fidx[rk1] = s1; fadd[rk1] = pd1 + bn1; fgv[rk1] = nr1 > 0.0 ? y1 * sqrt(bn1) / nr1 : 0.0; vu[rk1] = -gr1; }
        if (lane == 0) { sm_m = m; sm_grp = 1; sm_cmd = 2; }
        __syncthreads();
        sm_face_factor(Gs, p, fidx, fadd, m, Ff, fdia, invd, false, gpos, 0.0, fgv);
        ++face_solves;
        const int i0 = lane, i1 = lane + 64;
        const bool h0 = i0 < m, h1 = i1 < m;
        double w0 = h0 ? vu[i0] : 0.0, w1 = h1 ? vu[i1] : 0.0;
        sm_face_solve(Ff, invd, m, lane, w0, w1);
        __builtin_amdgcn_wave_barrier();
        if (h0) vu[i0] = w0;
        if (h1) vu[i1] = w1;
        sm_lds_sync();
        const double d0 = f0 ? vu[rk0] : 0.0, d1 = f1 ? vu[rk1] : 0.0;
        __builtin_amdgcn_wave_barrier();
        const double dn = sm_sum(d0 * d0 + d1 * d1);
        if (!(dn < 1e300) || !(dn > 0.0)) return false;
        const double g_old = sm_sum(gr0 * gr0 + gr1 * gr1);
        double alpha = 1.0;
        for (int tr = 0; tr < 6; ++tr, alpha *= 0.5) {
          double v0 = f0 ? __builtin_fma(alpha, d0, y0) : y0, v1 = f1 ? __builtin_fma(alpha, d1, y1) : y1;
          const bool k0 = f0 && v0 * y0 <= 0.0, k1 = f1 && v1 * y1 <= 0.0;
          if (k0) v0 = 0.0;
          if (k1) v1 = 0.0;
          double qv0, qv1;
          matvec(v0, v1, qv0, qv1);
          ++it;
          qv0 = on0 ? qv0 - c0 : 0.0;
          qv1 = on1 ? qv1 - c1 : 0.0;
          const double Fv = 0.5 * sm_sum(v0 * (qv0 - c0) + v1 * (qv1 - c1)) + penalty(v0, v1);
          // Close to the minimiser a step that still halves the gradient changes the objective by less than its rounding
          // noise (the error enters F squared): such a step -- full, nothing clipped -- is judged by the face's gradient
          bool by_gradient = false;
          if (!(Fv < Fy) && tr == 0 && __ballot(k0 || k1) == 0ull && Fv <= Fy + 1e-12 * fabs(Fy)) {
            // (penalty() left the image of v in vu)
            double nv0 = 0.0, nv1 = 0.0;
            if (real0 && f0) {
              double ss = 0.0;
              for (int mm = 0; mm < gn0; ++mm) ss = __builtin_fma(vu[gs0 + mm], vu[gs0 + mm], ss);
              nv0 = sqrt(ss);
            }
            if (real1 && f1) {
              double ss = 0.0;
              for (int mm = 0; mm < gn1; ++mm) ss = __builtin_fma(vu[gs1 + mm], vu[gs1 + mm], ss);
              nv1 = sqrt(ss);
            }
            __builtin_amdgcn_wave_barrier();
            const double e0 = f0 ? qv0 + copysign(thr0, v0) + (pd0 + ((real0 && nv0 > 0.0) ? pb0 / nv0 : 0.0)) * v0 : 0.0;
            const double e1 = f1 ? qv1 + copysign(thr1, v1) + (pd1 + ((real1 && nv1 > 0.0) ? pb1 / nv1 : 0.0)) * v1 : 0.0;
            by_gradient = sm_sum(e0 * e0 + e1 * e1) < 0.25 * g_old;
          }
          if (Fv < Fy || by_gradient) {
            y0 = v0; y1 = v1; qy0 = qv0; qy1 = qv1; Fy = Fv;
            x0 = y0; x1 = y1;
            return true;
          }
        }
        return false;
      };
      while (it < it_cap && !conv) {
        const unsigned long long tka = wall_clock64();
        // (the Newton steps are spent and the point still has not settled -- a group sliding towards zero, where the
        //  curvature b / ||x_g|| of its norm blows up and the steps shrink: nearly unpenalised p > n fits of the adaptive
        //  estimators' later rounds -- the general path's model solver is the better place for it)
        if (newton_left == 0 && it >= next_newton) break;
        if (newton_left > 0 && it >= next_newton) {
          --newton_left;
          const bool moved = newton();
          next_newton = it + (moved ? 6 : 40);
          tk_cg += wall_clock64() - tka;
          if (moved) nh = 0;  // (the mixing's history belongs to the point left behind)
        }
        const double t = 1.0 / Lp;
        double g0v, g1v;
        prox(y0 - t * qy0, y1 - t * qy1, t, g0v, g1v);
        const double r0 = g0v - y0, r1 = g1v - y1;
        const double s_r = sm_sum(r0 * r0 + r1 * r1), s_b = sm_sum(g0v * g0v + g1v * g1v);
        if ((it & 7) == 1) gnorm = sqrt(sm_sum(qy0 * qy0 + qy1 * qy1));
        kkt = sqrt(s_r) * Lp;
        resid = sqrt(s_r);
        bnorm = sqrt(s_b);
        if (!(s_r == s_r) || !(s_b < 1e300)) {
          bad = true;
          break;
        }
        double mu_eff = mu_rq > 0.0 ? fmin(mu_rq, Lp) : Lp;
        mu_eff = fmax(mu_eff, kMuFloor * Lp);
        if (kkt <= fmax(tol * bnorm * mu_eff, kRoundFloor * (gnorm + Lp * bnorm))) {
          // (confirmed by plain proximal steps: see `confirm`)
          if (confirm(g0v, g1v, sqrt(s_r), t, y0, y1, qy0, qy1)) {
            conv = true;
            break;
          }
          // not there yet: the criterion has been tightened to what the contraction says; the acceleration starts afresh
          Fy = 0.5 * sm_sum(y0 * (qy0 - c0) + y1 * (qy1 - c1)) + penalty(y0, y1);
          nh = 0;
          continue;
        }
        // history: newest first
#pragma unroll
        for (int i = SM_AA - 1; i > 0; --i) {
          gh0[i] = gh0[i - 1]; gh1[i] = gh1[i - 1]; rh0[i] = rh0[i - 1]; rh1[i] = rh1[i - 1];
#pragma unroll
          for (int k = SM_AA - 1; k > 0; --k) M[i][k] = M[i - 1][k - 1];
        }
        gh0[0] = g0v; gh1[0] = g1v; rh0[0] = r0; rh1[0] = r1;
        if (nh < SM_AA) ++nh;
#pragma unroll
        for (int k = 0; k < SM_AA; ++k) {
          const double dot = k < nh ? sm_sum(r0 * rh0[k] + r1 * rh1[k]) : 0.0;
          M[0][k] = dot;
          M[k][0] = dot;
        }
        // weights: (M + eps I) w = 1, alpha = w / sum(w)  (Gaussian elimination without pivoting on the regularised,
        // symmetric positive definite matrix; every lane holds the same numbers)
        double cand0 = g0v, cand1 = g1v;
        bool have_cand = false;
        if (nh >= 2) {
          double A[SM_AA][SM_AA], w[SM_AA];
          double top = 0.0;
#pragma unroll
          for (int i = 0; i < SM_AA; ++i) top = fmax(top, i < nh ? M[i][i] : 0.0);
#pragma unroll
          for (int i = 0; i < SM_AA; ++i) {
#pragma unroll
            for (int k = 0; k < SM_AA; ++k) A[i][k] = (i < nh && k < nh) ? M[i][k] : (i == k ? 1.0 : 0.0);
            A[i][i] += i < nh ? 1e-10 * top : 0.0;
            w[i] = i < nh ? 1.0 : 0.0;
          }
#pragma unroll
          for (int k = 0; k < SM_AA; ++k) {
            const double piv = A[k][k];
            const double ip = piv != 0.0 ? 1.0 / piv : 0.0;
#pragma unroll
            for (int i = k + 1; i < SM_AA; ++i) {
              const double fct = A[i][k] * ip;
#pragma unroll
              for (int c2 = k; c2 < SM_AA; ++c2) A[i][c2] -= fct * A[k][c2];
              w[i] -= fct * w[k];
            }
          }
#pragma unroll
          for (int k = SM_AA - 1; k >= 0; --k) {
            double acc = w[k];
#pragma unroll
            for (int c2 = k + 1; c2 < SM_AA; ++c2) acc -= A[k][c2] * w[c2];
            w[k] = A[k][k] != 0.0 ? acc / A[k][k] : 0.0;
          }
          double sw = 0.0;
#pragma unroll
          for (int i = 0; i < SM_AA; ++i) sw += w[i];
          if (sw == sw && fabs(sw) > 1e-300) {
            cand0 = cand1 = 0.0;
#pragma unroll
            for (int i = 0; i < SM_AA; ++i) {
              const double al = w[i] / sw;
              cand0 = __builtin_fma(al, gh0[i], cand0);
              cand1 = __builtin_fma(al, gh1[i], cand1);
            }
            have_cand = (cand0 == cand0) && (cand1 == cand1);
            have_cand = __ballot(!have_cand) == 0ull;
          }
        }
        const double zold0 = y0, zold1 = y1, qold0 = qy0, qold1 = qy1;
        bool taken = false;
        if (have_cand) {
          double qc0, qc1;
          matvec(cand0, cand1, qc0, qc1);
          ++it;
          qc0 = on0 ? qc0 - c0 : 0.0;
          qc1 = on1 ? qc1 - c1 : 0.0;
          const double Fc = 0.5 * sm_sum(cand0 * (qc0 - c0) + cand1 * (qc1 - c1)) + penalty(cand0, cand1);
          if (Fc <= Fy - 0.5 * Lp * s_r) {
            y0 = cand0; y1 = cand1; qy0 = qc0; qy1 = qc1; Fy = Fc;
            taken = true;
          }
        }
        if (!taken) {
          matvec(g0v, g1v, qy0, qy1);
          ++it;
          qy0 = on0 ? qy0 - c0 : 0.0;
          qy1 = on1 ? qy1 - c1 : 0.0;
          y0 = g0v; y1 = g1v;
          const double Fg = 0.5 * sm_sum(y0 * (qy0 - c0) + y1 * (qy1 - c1)) + penalty(y0, y1);
          // (the step bound was too short for this step: the objective ROSE, by more than its own rounding noise -- near the
          //  minimiser differences of F are noise, and a test that noise can trip would double the bound pass after pass)
          if (Fg - Fy > 1e-12 * (fabs(Fy) + fabs(Fg)) && Lp < 1e3 * L) Lp *= 2.0;
          Fy = Fg;
          if (have_cand) nh = 1;  // (a refused candidate: the history that produced it goes, but for the newest pair)
        }
        // curvature along the move of the gradient point
        {
          const double dz0 = y0 - zold0, dz1 = y1 - zold1;
          const double dd = sm_sum(dz0 * dz0 + dz1 * dz1);
          if (dd > 0.0) {
            const double rq = sm_sum(dz0 * (qy0 - qold0) + dz1 * (qy1 - qold1)) / dd;
            if (rq > Lp) Lp = 1.05 * rq;
            if (rq > 0.0) mu_rq = mu_rq > 0.0 ? fmin(mu_rq, rq) : rq;
          }
        }
        x0 = y0;
        x1 = y1;
        tk_it += wall_clock64() - tka;
      }
    }
    while (lasso_type && it < it_cap && !conv) {
      const unsigned long long tka = wall_clock64();
      matvec(z0, z1, qz0, qz1);
      ++it;
      qz0 -= c0;
      qz1 -= c1;
      if (!on0) qz0 = 0.0;
      if (!on1) qz1 = 0.0;
      // curvature along the last move of the gradient point: a lower bound of the step bound, an estimate of mu
      if (have_prev) {
        const double dz0 = z0 - zp0, dz1 = z1 - zp1;
        const double dd = sm_sum(dz0 * dz0 + dz1 * dz1);
        if (dd > 0.0) {
          const double dq = sm_sum(dz0 * (qz0 - qp0) + dz1 * (qz1 - qp1));
          const double rq = dq / dd;
          if (rq > Lp) Lp = 1.05 * rq;  // (the power steps under-estimated lambda_max)
          if (rq > 0.0) mu_rq = mu_rq > 0.0 ? fmin(mu_rq, rq) : rq;
        }
      }
      const double t = 1.0 / Lp;
      double u0, u1;
      prox(z0 - t * qz0, z1 - t * qz1, t, u0, u1);
      // the proximal-gradient mapping at z: the KKT residual the point is accepted on
      const double w0 = z0 - u0, w1 = z1 - u1;
      const double s_kkt = sm_sum(w0 * w0 + w1 * w1), s_b = sm_sum(u0 * u0 + u1 * u1);
      if ((it & 7) == 1) gnorm = sqrt(sm_sum(qz0 * qz0 + qz1 * qz1));  // (only the rounding floor uses it)
      const double s_rs = sm_sum(w0 * (u0 - x0) + w1 * (u1 - x1));
      kkt = sqrt(s_kkt) * Lp;
      bnorm = sqrt(s_b);
      resid = sqrt(s_kkt);
      if (!(s_kkt == s_kkt) || !(s_b < 1e300)) {  // NaN / overflow
        bad = true;
        break;
      }
      double mu_eff = mu_rq > 0.0 ? fmin(mu_rq, Lp) : Lp;
      mu_eff = fmax(mu_eff, kMuFloor * Lp);
      if (kkt <= fmax(tol * bnorm * mu_eff, kRoundFloor * (gnorm + Lp * bnorm))) {
        double yl0, yl1, ql0, ql1;
        if (confirm(u0, u1, sqrt(s_kkt), t, yl0, yl1, ql0, ql1)) {
          conv = true;
          break;
        }
        // not there yet: the iteration goes on from the confirmed point's neighbourhood, momentum and history afresh
        z0 = x0; z1 = x1;
        tk = 1.0;
        have_prev = false;
        still = 0;
        pat_p = pat_n = pat_p1 = pat_n1 = ~0ull;
        continue;
      }
      // momentum with the gradient-scheme restart
      const bool restart = s_rs > 0.0;
      const double tk_new = restart ? 1.0 : 0.5 * (1.0 + sqrt(1.0 + 4.0 * tk * tk));
      const double mom = restart ? 0.0 : (tk - 1.0) / tk_new;
      zp0 = z0; zp1 = z1; qp0 = qz0; qp1 = qz1;
      have_prev = true;
      z0 = u0 + mom * (u0 - x0);
      z1 = u1 + mom * (u1 - x1);
      x0 = u0;
      x1 = u1;
      tk = tk_new;
      // sign pattern of the iterate: unchanged for SM_STILL steps => the face is (probably) the minimiser's
      const uint64_t np0 = __ballot(on0 && x0 > 0.0), nn0 = __ballot(on0 && x0 < 0.0);
      const uint64_t np1 = __ballot(on1 && x1 > 0.0), nn1 = __ballot(on1 && x1 < 0.0);
      still = (np0 == pat_p && nn0 == pat_n && np1 == pat_p1 && nn1 == pat_n1) ? still + 1 : 0;
      pat_p = np0; pat_n = nn0; pat_p1 = np1; pat_n1 = nn1;
      const unsigned long long tkb = wall_clock64();
      tk_it += tkb - tka;

      // ---- conjugate gradients on the face of the iterate (points without group norms) -------------------------
      // (G_AA + D) x_A = c_A - thr_A sign(x_A): warm start x, residual r = -(q + thr sign(x) + d x) on A.  A step that
      // would carry a coordinate across zero stops there; the coordinate leaves A and the iteration starts over on the
      // smaller face.  Off A every vector is zero, so the products with the full G are products with G_AA.
      const int face_now = __popcll(np0 | nn0) + __popcll(np1 | nn1);
      const bool face_fits = face_now <= face_cap && face_now <= (int)n;  // (more unknowns than rows: a singular face)
      bool want_direct = false;
      if (lasso_type && !conv && still >= SM_STILL && cg_runs < (face_fits ? 12 : 6) && face_now != 0) {
        ++cg_runs;
        still = 0;
        bool f0 = on0 && x0 != 0.0, f1 = on1 && x1 != 0.0;  // the face
        double q0, q1;
        matvec(x0, x1, q0, q1);
        ++it;
        q0 -= c0;
        q1 -= c1;
        int hits = 0;
        const int face0 = __popcll(np0 | nn0) + __popcll(np1 | nn1);
        // (a face whose factor fits beside G gets a direct solve when these steps do not settle it soon)
        const int cg_cap = face_fits ? (2 * face0 + 10 < 24 ? 2 * face0 + 10 : 24) : 2 * face0 + 10;
        bool cg_done = false;
        double r0 = f0 ? -(q0 + copysign(thr0, x0) + pd0 * x0) : 0.0, r1 = f1 ? -(q1 + copysign(thr1, x1) + pd1 * x1) : 0.0;
        double d0v = r0, d1v = r1;
        double rr = sm_sum(r0 * r0 + r1 * r1);
        const double rr_start = rr;
        for (int k = 0; k < cg_cap && it < it_cap && rr > 0.0; ++k) {
          double h0, h1;
          matvec(d0v, d1v, h0, h1);
          ++it;
          h0 = f0 ? h0 + pd0 * d0v : 0.0;
          h1 = f1 ? h1 + pd1 * d1v : 0.0;
          const double dHd = sm_sum(d0v * h0 + d1v * h1), dd = sm_sum(d0v * d0v + d1v * d1v);
          if (!(dd > 0.0)) break;
          if (dHd > 0.0) mu_rq = mu_rq > 0.0 ? fmin(mu_rq, dHd / dd) : dHd / dd;
          // the step to the minimiser along d (none: the face is flat along d -- go as far as the face allows)
          double alpha = dHd > 1e-14 * Lp * dd ? rr / dHd : 1e300;
          // ... and the step to the first sign change
          const double lim0 = (f0 && d0v * x0 < 0.0) ? -x0 / d0v : 1e300;
          const double lim1 = (f1 && d1v * x1 < 0.0) ? -x1 / d1v : 1e300;
          const double amax = sm_min(fmin(lim0, lim1));
          const bool hit = alpha >= amax;
          if (hit) alpha = amax;
          if (!(alpha < 1e299)) break;  // (flat and unbounded along d: nothing to gain)
          x0 = f0 ? __builtin_fma(alpha, d0v, x0) : x0;
          x1 = f1 ? __builtin_fma(alpha, d1v, x1) : x1;
          q0 = __builtin_fma(alpha, h0 - (f0 ? pd0 * d0v : 0.0), q0);  // q follows x with the product just made: G d
          q1 = __builtin_fma(alpha, h1 - (f1 ? pd1 * d1v : 0.0), q1);
          if (hit) {
            // the blocking coordinates land on zero and leave the face; conjugacy is lost: steepest descent restarts it
            if (f0 && lim0 <= amax) { x0 = 0.0; f0 = false; }
            if (f1 && lim1 <= amax) { x1 = 0.0; f1 = false; }
            // (q was advanced with the full G d, whose rows off the face are garbage-free: d is zero there)
            matvec(x0, x1, q0, q1);
            ++it;
            q0 -= c0;
            q1 -= c1;
            r0 = f0 ? -(q0 + copysign(thr0, x0) + pd0 * x0) : 0.0;
            r1 = f1 ? -(q1 + copysign(thr1, x1) + pd1 * x1) : 0.0;
            d0v = r0;
            d1v = r1;
            rr = sm_sum(r0 * r0 + r1 * r1);
            if (++hits > face0) break;
            continue;
          }
          r0 = f0 ? __builtin_fma(-alpha, h0, r0) : 0.0;
          r1 = f1 ? __builtin_fma(-alpha, h1, r1) : 0.0;
          const double rr_new = sm_sum(r0 * r0 + r1 * r1);
          if (!(rr_new == rr_new)) {
            bad = true;
            break;
          }
          const double xn = sqrt(sm_sum(x0 * x0 + x1 * x1));
          double mu_eff2 = mu_rq > 0.0 ? fmin(mu_rq, Lp) : Lp;
          mu_eff2 = fmax(mu_eff2, kMuFloor * Lp);
          // (a tenth of what the point is accepted on: the proximal steps that follow confirm it)
          if (sqrt(rr_new) <= 0.1 * fmax(tol * xn * mu_eff2, kRoundFloor * (gnorm + Lp * xn)) || rr_new <= 1e-30 * rr_start) {
            rr = rr_new;
            cg_done = true;
            break;
          }
          const double bt = rr_new / rr;
          d0v = __builtin_fma(bt, d0v, r0);
          d1v = __builtin_fma(bt, d1v, r1);
          rr = rr_new;
        }
        if (bad) break;
        want_direct = face_fits && !cg_done && rr > 0.0;
        z0 = x0; z1 = x1;   // the proximal steps go on from the face's minimiser, momentum and history afresh
        tk = 1.0;
        have_prev = false;
        pat_p = pat_n = pat_p1 = pat_n1 = ~0ull;
        tk_cg += wall_clock64() - tkb;
      }
      if (want_direct) {
        const unsigned long long tkd = wall_clock64();
        // ---- direct solve on the face: (G_AA + D) x_A = c_A - thr_A s_A by an L D L^T factorisation the other three
        // wavefronts work on (sm_face_factor), two triangular solves on this one -- for faces conjugate gradients have not
        // settled in their steps (ill-conditioned G_AA: the nearly unpenalised small alphas of the reference's README grid).
        // The iterate moves towards the face's minimiser t along the PROJECTED segment -- x + alpha (t - x) with every
        // coordinate that would leave its sign held at zero -- for the first alpha of 1, 1/2, 1/4, ... at which the
        // objective does not rise (one product each): many coordinates leave the face at once.  The proximal steps that
        // follow confirm the point or extend the face.
        // (SM_FACE_ROUNDS > 1 continues as an active-set method: coordinates at zero whose bound the gradient violates
        //  join with the sign of -q_j and the larger face is solved again.  On the README's noise-free folds -- seventy of
        //  eighty coefficients at rounding level, thresholds of 1e-8 against entries of 5e-5 in the inverse -- the rounds
        //  flip subsets back and forth: 18-35 factorisations per point against 4-5 this way.  Stopping at the first sign
        //  change and factoring again: a factorisation per coordinate.)
        bool f0 = on0 && x0 != 0.0, f1 = on1 && x1 != 0.0;
        double sg0 = f0 ? copysign(1.0, x0) : 0.0, sg1 = f1 ? copysign(1.0, x1) : 0.0;
        // objective at v, with q = G v - c
        auto objective = [&](double v0, double v1, double& g0, double& g1) {
          matvec(v0, v1, g0, g1);
          ++it;
          const double Fv = sm_sum((on0 ? v0 * (0.5 * g0 - c0) + thr0 * fabs(v0) + 0.5 * pd0 * v0 * v0 : 0.0) +
                                   (on1 ? v1 * (0.5 * g1 - c1) + thr1 * fabs(v1) + 0.5 * pd1 * v1 * v1 : 0.0));
          g0 = on0 ? g0 - c0 : 0.0;
          g1 = on1 ? g1 - c1 : 0.0;
          return Fv;
        };
        double qx0, qx1;
        double F_cur = objective(x0, x1, qx0, qx1);
        bool any_move = false;
        for (int round = 0; round < SM_FACE_ROUNDS; ++round) {
          const uint64_t m0 = __ballot(f0), m1 = __ballot(f1);
          const int n0 = __popcll(m0), m = n0 + __popcll(m1);
          if (m == 0 || m > face_cap) break;
          const uint64_t below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
          const int rk0 = __popcll(m0 & below), rk1 = n0 + __popcll(m1 & below);
          if (f0) { fidx[rk0] = s0; fadd[rk0] = pd0; vu[rk0] = c0 - thr0 * sg0; }
          if (f1) { fidx[rk1] = s1; fadd[rk1] = pd1; vu[rk1] = c1 - thr1 * sg1; }
          if (lane == 0) { sm_m = m; sm_grp = 0; sm_cmd = 2; }
          __syncthreads();
          sm_face_factor(Gs, p, fidx, fadd, m, Ff, fdia, invd, false);
          ++face_solves;
          // rank space: lane l holds unknowns l and l + 64 of the face
          const int i0 = lane, i1 = lane + 64;
          const bool h0 = i0 < m, h1 = i1 < m;
          double w0 = h0 ? vu[i0] : 0.0, w1 = h1 ? vu[i1] : 0.0;
          const double il0 = h0 ? invd[i0] : 0.0, il1 = h1 ? invd[i1] : 0.0;
          sm_face_solve(Ff, invd, m, lane, w0, w1);
          // (a dropped pivot left its unknown without an equation: it stays where it is)
          __builtin_amdgcn_wave_barrier();
          if (h0) vu[i0] = il0 != 0.0 ? w0 : 1e300;
          if (h1) vu[i1] = il1 != 0.0 ? w1 : 1e300;
          sm_lds_sync();
          double t0 = f0 ? vu[rk0] : 0.0, t1 = f1 ? vu[rk1] : 0.0;
          if (t0 == 1e300) t0 = x0;
          if (t1 == 1e300) t1 = x1;
          __builtin_amdgcn_wave_barrier();
          if (!(sm_sum(t0 * t0 + t1 * t1) < 1e300)) break;  // (the factor went wrong: the proximal steps carry on)
          bool moved = false, clipped = false;
          double alpha = 1.0;
          for (int tr = 0; tr < 6 && !moved; ++tr, alpha *= 0.5) {
            double v0 = f0 ? __builtin_fma(alpha, t0 - x0, x0) : x0, v1 = f1 ? __builtin_fma(alpha, t1 - x1, x1) : x1;
            const bool k0 = f0 && v0 * sg0 <= 0.0, k1 = f1 && v1 * sg1 <= 0.0;
            if (k0) v0 = 0.0;
            if (k1) v1 = 0.0;
            double g0, g1;
            const double Fv = objective(v0, v1, g0, g1);
            if (Fv <= F_cur) {
              x0 = v0; x1 = v1;
              qx0 = g0; qx1 = g1;
              F_cur = Fv;
              moved = true;
              clipped = __ballot(k0 || k1) != 0ull || tr > 0;
            }
          }
          if (!moved) break;
          any_move = true;
          // the next face: what is non-zero, and the coordinates at zero whose bound the gradient violates
          const double slack = 1e-9;
          const bool a0 = on0 && x0 == 0.0 && fabs(qx0) > thr0 * (1.0 + slack) + 1e-14 * Lp * bnorm;
          const bool a1 = on1 && x1 == 0.0 && fabs(qx1) > thr1 * (1.0 + slack) + 1e-14 * Lp * bnorm;
          f0 = on0 && (x0 != 0.0 || a0);
          f1 = on1 && (x1 != 0.0 || a1);
          sg0 = x0 != 0.0 ? copysign(1.0, x0) : (a0 ? -copysign(1.0, qx0) : 0.0);
          sg1 = x1 != 0.0 ? copysign(1.0, x1) : (a1 ? -copysign(1.0, qx1) : 0.0);
          if (!clipped && __ballot(a0 || a1) == 0ull) break;  // the minimiser, as far as this arithmetic can tell
        }
        if (!any_move) cg_runs += 3;  // (no descent along the projected segment: leave the rest to the proximal steps)
        z0 = x0; z1 = x1;
        tk = 1.0;
        have_prev = false;
        pat_p = pat_n = pat_p1 = pat_n1 = ~0ull;
        tk_cg += wall_clock64() - tkd;
      }
    }
    iters_all += it;
    const unsigned long long tkc = wall_clock64();
    // the reported point: the result of the last proximal step, with the gradient taken at it for the record
    double q0, q1;
    matvec(x0, x1, q0, q1);
    q0 -= c0;
    q1 -= c1;
    const double be0 = x0, be1 = x1;
    const double b2 = bnorm * bnorm, d2 = resid * resid;
    const int sweep = it;
    const double err_est = 0.0;
    double mu_rep = mu_rq > 0.0 ? fmin(mu_rq, Lp) : Lp;
    mu_rep = fmax(mu_rep, kMuFloor * Lp);
    const double Lall = Lp;

    // ---- the point's record: minimal-norm subgradient (the KKT residual), loss, group norms, coefficients --------
    double kkt2 = 0.0;
    double nr0 = fabs(be0), nr1 = fabs(be1);
    {
      // norm of every position's group (a group of one: the absolute value), and for the zero groups the norm of
      // soft(q_g, a_g), which has to stay below b_g
      double tn0 = 0.0, tn1 = 0.0;
      if (!a.t.singleton && (!lasso_type || gn_out != nullptr || (rw_on & 2))) {
        if (on0) vu[s0] = be0;
        if (on1) vu[s1] = be1;
        sm_lds_sync();
        if (on0 && gn0 > 1) {
          double ss = 0.0;
          for (int m = 0; m < gn0; ++m) ss = __builtin_fma(vu[gs0 + m], vu[gs0 + m], ss);
          nr0 = sqrt(ss);
        }
        if (on1 && gn1 > 1) {
          double ss = 0.0;
          for (int m = 0; m < gn1; ++m) ss = __builtin_fma(vu[gs1 + m], vu[gs1 + m], ss);
          nr1 = sqrt(ss);
        }
        __builtin_amdgcn_wave_barrier();
      }
      if (!lasso_type) {
        if (on0) vu[s0] = soft(q0, pa0);
        if (on1) vu[s1] = soft(q1, pa1);
        sm_lds_sync();
        if (real0 && nr0 == 0.0) {
          double ss = 0.0;
          for (int m = 0; m < gn0; ++m) ss = __builtin_fma(vu[gs0 + m], vu[gs0 + m], ss);
          tn0 = sqrt(ss);
        }
        if (real1 && nr1 == 0.0) {
          double ss = 0.0;
          for (int m = 0; m < gn1; ++m) ss = __builtin_fma(vu[gs1 + m], vu[gs1 + m], ss);
          tn1 = sqrt(ss);
        }
        __builtin_amdgcn_wave_barrier();
      }
      auto sub = [&](bool on, bool real, double q, double be, double thr, double pa, double pb, double pd, double nr, double tn,
                     bool first_member) {
        if (!on) return 0.0;
        if (!real) {  // on its own: a group term of a group of one is part of the threshold
          const double sm = q + pd * be;
          return be != 0.0 ? sm + copysign(thr, be) : soft(sm, thr);
        }
        if (nr > 0.0) {  // smooth in the group term: gradient + b u / ||u|| + d u, l1 part by its minimal-norm element
          const double sm = q + pd * be + pb * be / nr;
          return be != 0.0 ? sm + copysign(pa, be) : soft(sm, pa);
        }
        return first_member ? fmax(0.0, tn - pb) : 0.0;  // zero group: counted once
      };
      const double r0 = sub(on0, real0, q0, be0, thr0, pa0, pb0, pd0, nr0, tn0, s0 == gs0);
      const double r1 = sub(on1, real1, q1, be1, thr1, pa1, pb1, pd1, nr1, tn1, s1 == gs1);
      kkt2 = sm_sum(r0 * r0 + r1 * r1);
      if (gn_out != nullptr) {
        if (on0 && s0 == gs0) gn_out[(int64_t)point * G + g0] = nr0;
        if (on1 && s1 == gs1) gn_out[(int64_t)point * G + g1] = nr1;
      }
    }
    const double loss = 0.5 * sm_sum(be0 * (q0 - c0) + be1 * (q1 - c1)) + 0.5 * yy;
    if (on0) betas_out[(int64_t)point * p + j0] = be0;
    if (on1) betas_out[(int64_t)point * p + j1] = be1;
    if (lane == 0) {
      slm_point_info info;
      info.n_iter = sweep;
      info.status = bad ? SLM_ERR_NON_FINITE : (conv ? SLM_OK : SLM_ERR_NOT_CONVERGED);
      info.resid = sqrt(d2);
      info.beta_norm = sqrt(b2);
      info.loss = loss;
      info.L = Lall;
      info.mode = 2;  // on chip: proximal gradient steps and conjugate gradients on the Gram matrix
      info.rejects = 0;
      info.kkt = sqrt(kkt2);
      info.mu = mu_rep;
      infos[point] = info;
    }
    // ---- a re-weighted lane: the next round's weights from this round's solution ---------------------------------
    // (the loop of AdaptiveLasso._solve, reference model/_adaptive_lasso.py:206-232: solve, renew the weights from the
    //  solution -- :364-374 for the group norms --, stop when they no longer move; here without leaving the kernel.  The
    //  expressions are the host loop's own, operation for operation: scale * (numerator / (|x| + eps)))
    if (rw_on && !bad) {
      ++rounds;
      double nA0 = A0, nA1 = A1, nB0 = B0, nB1 = B1;
      if (rw_on & 1) {
        if (on0 && j0 < rw_ncoef) nA0 = rw_coef * (rw_numer / (fabs(be0) + rw_eps));
        if (on1 && j1 < rw_ncoef) nA1 = rw_coef * (rw_numer / (fabs(be1) + rw_eps));
      }
      if (rw_on & 2) {
        if (on0 && g0 < rw_ngroup) nB0 = rw_gscale[g0] * (rw_numer / (nr0 + rw_eps));
        if (on1 && g1 < rw_ngroup) nB1 = rw_gscale[g1] * (rw_numer / (nr1 + rw_eps));
      }
      // (a group's weight counts once: at its first member)
      const double dA0 = nA0 - A0, dA1 = nA1 - A1, dB0 = (on0 && s0 == gs0) ? nB0 - B0 : 0.0, dB1 = (on1 && s1 == gs1) ? nB1 - B1 : 0.0;
      const double moved = sqrt(sm_sum((dA0 * dA0 + dA1 * dA1) + (dB0 * dB0 + dB1 * dB1)));
      A0 = nA0; A1 = nA1; B0 = nB0; B1 = nB1;
      if (!conv || moved <= rw_tol) {  // settled -- or a round that did not: the caller's own loop takes the lane (solve_core)
        for (int k = point + 1 + lane; k < last; k += 64) {
          slm_point_info skipped;
          memset(&skipped, 0, sizeof(skipped));
          skipped.mode = 3;  // a round that was not run
          infos[k] = skipped;
        }
        tk_rec += wall_clock64() - tkc;
        break;
      }
    }
    tk_rec += wall_clock64() - tkc;
  }
  release_helpers();
  if (lane == 0) {
    ctl->rounds = rounds;
    ctl->total_iter = 1;  // X was read once
    ctl->iter = (int32_t)(iters_all > 2000000000ll ? 2000000000ll : iters_all);
    ctl->nonfinite = bad ? 1 : 0;
    ctl->done = 1;
    // (SLM_TRACE=2: where the kernel's time went, in ms -- the 100 MHz wall clock)
    ctl->n_hist = face_solves;
    ctl->hist[0] = (double)(tk1 - tk0) * 1e-5;
    ctl->hist[1] = (double)(tk2 - tk1) * 1e-5;
    ctl->hist[2] = (double)tk_it * 1e-5;
    ctl->hist[3] = (double)tk_cg * 1e-5;
    ctl->hist[4] = (double)tk_rec * 1e-5;
  }
}

}  // namespace slm

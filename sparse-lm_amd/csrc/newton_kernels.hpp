// Dense symmetric positive definite solves inside ONE workgroup: the direct step of the working-set model
// solver (ws_kernels.hpp).
//
// When the penalised quadratic model restricted to the face of the current iterate (its non-zero
// coordinates with their signs) is ill-conditioned, proximal-gradient iterations on it crawl (rate
// 1 - 1/sqrt(kappa)); the minimiser over that face is the solution of  H d = r  with H = G_AA (+ the
// curvature of the group / ridge terms), m = |A| <= 512 unknowns.  This file factors H = L L^T and solves
// with it, 1024 threads, the factor in global scratch (L2-resident: <= 1.1 MB per lane).
//
// Counterpart in the reference: the KKT-system factorisations inside the interior-point solver cvxpy
// dispatches to (src/sparselm/model/_base.py:512-519); nothing of it is in the reference tree.
//
// Layout.  Blocked by 16 (the v_mfma_f64_16x16x4_f64 tile).  Only the lower triangle of tiles is kept,
// tile (I, J), I >= J, at (I (I + 1) / 2 + J) * 256.  INSIDE a tile, element (r, c) sits at
//     slot(r, c) = (c >> 2) * 64 + (c & 3) * 16 + r,
// the operand layout of the MFMA: lane l of a wavefront finds, at [step * 64 + l], the element
// (r = l & 15, c = (l >> 4) + 4 step) it must supply as A[i = l & 15][k = l >> 4] (or, for the transposed
// role, B[k = l >> 4][j = l & 15]) in MFMA number `step` of a 16 x 16 x 16 product.  The accumulator of
// the TRANSPOSED product, register R of lane l = D[(l >> 4) + 4 R][l & 15], is then element (l & 15,
// (l >> 4) + 4 R) of the untransposed result: again [R * 64 + l].  So both kinds of access are four
// coalesced 512-byte rows per tile, and panel / trailing updates are
//     L_IJ^T   = Linv_JJ  A_IJ^T                 (A operand: Linv_JJ,  B operand: A_IJ)
//     A_IK^T  -= L_KJ     L_IJ^T                 (A operand: -L_KJ,    B operand: L_IJ,  C: A_IK)
// (tools/newton_tile_model.py runs exactly this indexing in numpy against numpy.linalg).
// The 16 x 16 diagonal blocks are factored and inverted by one wavefront in registers (a row per lane,
// pivots and multipliers broadcast with v_readlane); their inverses are kept, so the triangular solves
// are sequences of 16 x 16 mat-vecs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tail_kernels.hpp"

namespace slm {

constexpr int NT_B = 16;                                  // tile edge
constexpr int NT_MAXT = 32;                               // 512 unknowns
constexpr int NT_TILES = NT_MAXT * (NT_MAXT + 1) / 2;     // lower-triangular tiles
constexpr int64_t NT_SCRATCH = (int64_t)(NT_TILES + NT_MAXT) * 256;  // doubles per lane: factor + inverse diagonal blocks
typedef double nt_d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int nt_tile_off(int I, int J) { return (I * (I + 1) / 2 + J) * 256; }
__device__ __forceinline__ int nt_slot(int r, int c) { return (c >> 2) * 64 + (c & 3) * 16 + r; }

__device__ __forceinline__ double nt_readlane(double v, int src) {  // src must be wave-uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

// LDS the routines below need (the caller owns it; nothing else may live in it during a call)
struct NtShared {
  double tile[NT_B][NT_B + 1];   // natural-layout staging of a diagonal block
  double tinv[NT_B][NT_B + 1];   // ... and of its inverse
  double dinv[256];              // inverse of the current diagonal block, operand layout
  int fail;
};

// Cholesky factorisation in place: F holds H (tiles as above, T tile rows; the caller pads the last block
// with an identity), Dinv receives the inverses of the diagonal blocks of L (operand layout, [T][256]).
// Every thread of the 1024-thread workgroup must call it.  Returns false (to every thread) when a pivot is
// not safely positive -- H numerically singular or indefinite (p > n faces, duplicated columns): the caller
// falls back to the iteration.  `piv_floor`: smallest acceptable pivot (relative to the largest diagonal
// entry, times a rounding margin).
__device__ __forceinline__ bool nt_factor(double* F, double* Dinv, int T, double piv_floor, NtShared& sh) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid == 0) sh.fail = 0;
  __syncthreads();
  for (int J = 0; J < T; ++J) {
    // ---- diagonal block: one wavefront, a row per lane (lanes 16..63 mirror lanes 0..15) -----------
    if (wave == 0) {
      double* tp = F + nt_tile_off(J, J);
#pragma unroll
      for (int s = 0; s < 4; ++s) sh.tile[lane & 15][(lane >> 4) + 4 * s] = tp[s * 64 + lane];
      __builtin_amdgcn_wave_barrier();
      double a[NT_B];
#pragma unroll
      for (int c = 0; c < NT_B; ++c) a[c] = sh.tile[lane & 15][c];
      bool bad = false;
#pragma unroll
      for (int k = 0; k < NT_B; ++k) {
        double piv = nt_readlane(a[k], k);
        if (!(piv > piv_floor)) {
          bad = true;
          piv = 1.0;
        }
        const double dinv = 1.0 / sqrt(piv);
        a[k] *= dinv;  // lane i: l_ik (lane k: l_kk = sqrt(piv))
#pragma unroll
        for (int j = k + 1; j < NT_B; ++j) a[j] = __builtin_fma(-a[k], nt_readlane(a[k], j), a[j]);
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int c = 0; c < NT_B; ++c) sh.tile[lane & 15][c] = c <= (lane & 15) ? a[c] : 0.0;  // row (lane & 15) of L
      __builtin_amdgcn_wave_barrier();
      // inverse of L: lane j solves L x = e_j; the entries of L come back from LDS as broadcast reads (the
      // rows in registers are dead by now: 16 live doubles instead of 32)
      {
        double x[NT_B];
        const int jc = lane & 15;
#pragma unroll
        for (int i = 0; i < NT_B; ++i) {
          double s = (i == jc) ? 1.0 : 0.0;
#pragma unroll
          for (int k = 0; k < i; ++k) s = __builtin_fma(-sh.tile[i][k], x[k], s);
          x[i] = s / sh.tile[i][i];
        }
#pragma unroll
        for (int c = 0; c < NT_B; ++c) sh.tinv[c][jc] = x[c];  // column jc of L^-1
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const double lv = sh.tile[lane & 15][(lane >> 4) + 4 * s];
        const double iv = sh.tinv[lane & 15][(lane >> 4) + 4 * s];
        tp[s * 64 + lane] = lv;
        sh.dinv[s * 64 + lane] = iv;
        Dinv[J * 256 + s * 64 + lane] = iv;
      }
      if (bad && lane == 0) sh.fail = 1;
    }
    __syncthreads();
    if (sh.fail) return false;
    // ---- panel: L_IJ^T = Linv_JJ A_IJ^T for the tiles below the diagonal block ------------------------
    for (int I = J + 1 + wave; I < T; I += TAIL_WAVES) {
      double* ap = F + nt_tile_off(I, J);
      double bo[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) bo[s] = ap[s * 64 + lane];
      nt_d4 acc = nt_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sh.dinv[s * 64 + lane], bo[s], acc, 0, 0, 0);
#pragma unroll
      for (int R = 0; R < 4; ++R) ap[R * 64 + lane] = acc[R];
    }
    __syncthreads();
    // ---- trailing update: A_IK^T -= L_KJ L_IJ^T for J < K <= I -----------------------------------------
    const int rem = T - J - 1;
    const int npairs = rem * (rem + 1) / 2;
    for (int q = wave; q < npairs; q += TAIL_WAVES) {
      int a_ = (int)((sqrtf(8.0f * (float)q + 1.0f) - 1.0f) * 0.5f);  // q = a (a + 1) / 2 + b, 0 <= b <= a
      while (a_ * (a_ + 1) / 2 > q) --a_;
      while ((a_ + 1) * (a_ + 2) / 2 <= q) ++a_;
      const int b_ = q - a_ * (a_ + 1) / 2;
      const int I = J + 1 + a_, Kp = J + 1 + b_;
      const double* li = F + nt_tile_off(I, J);
      const double* lk = F + nt_tile_off(Kp, J);
      double* cp = F + nt_tile_off(I, Kp);
      double ao[4], bo[4];
      nt_d4 acc;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        ao[s] = -lk[s * 64 + lane];
        bo[s] = li[s * 64 + lane];
        acc[s] = cp[s * 64 + lane];
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ao[s], bo[s], acc, 0, 0, 0);
#pragma unroll
      for (int R = 0; R < 4; ++R) cp[R * 64 + lane] = acc[R];
    }
    __syncthreads();
  }
  return true;
}

// (tile or its transpose) times the 16-vector v (LDS), by one wavefront: lane (r = l & 15, q = l >> 4) sums
// its four columns, two xor shuffles add the quarters; every lane returns the r-th entry of the product.
__device__ __forceinline__ double nt_tile_matvec(const double* op, const double* v, bool transposed, int lane) {
  const int r = lane & 15, q = lane >> 4;
  double part = 0.0;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int c = q + 4 * s;
    const double e = transposed ? op[nt_slot(c, r)] : op[s * 64 + lane];
    part = __builtin_fma(e, v[c], part);
  }
  part += __shfl_xor(part, 16, 64);
  part += __shfl_xor(part, 32, 64);
  return part;
}

// v <- H^-1 v with the factor from nt_factor; v: LDS, 16 T entries.  Every thread must call it.
__device__ __forceinline__ void nt_solve(const double* F, const double* Dinv, int T, double* v) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  __syncthreads();
  for (int J = 0; J < T; ++J) {  // L y = v
    if (wave == 0) {
      const double y = nt_tile_matvec(Dinv + J * 256, v + 16 * J, false, lane);
      __builtin_amdgcn_wave_barrier();
      if (lane < 16) v[16 * J + lane] = y;
    }
    __syncthreads();
    for (int I = J + 1 + wave; I < T; I += TAIL_WAVES) {
      const double t = nt_tile_matvec(F + nt_tile_off(I, J), v + 16 * J, false, lane);
      if (lane < 16) v[16 * I + lane] -= t;
    }
    __syncthreads();
  }
  for (int J = T - 1; J >= 0; --J) {  // L^T x = y
    if (wave == 0) {
      const double x = nt_tile_matvec(Dinv + J * 256, v + 16 * J, true, lane);
      __builtin_amdgcn_wave_barrier();
      if (lane < 16) v[16 * J + lane] = x;
    }
    __syncthreads();
    for (int I = wave; I < J; I += TAIL_WAVES) {
      const double t = nt_tile_matvec(F + nt_tile_off(J, I), v + 16 * J, true, lane);
      if (lane < 16) v[16 * I + lane] -= t;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// Stand-alone entry (slm_dense_spd_solve; tests): x = H^-1 rhs for one dense m x m matrix given in
// natural row-major order, plus the reciprocal of ||H^-1 u|| after two inverse-iteration steps from the
// solution direction -- the estimate of lambda_min(H) the model solver hands to the stopping rule.
// ---------------------------------------------------------------------------------------------
struct DenseSolveArgs {
  const double* H;    // [m][m]
  const double* rhs;  // [m]
  double* x;          // [m]
  double* mu;         // [1]
  int* status;        // [1] 0 = ok, 1 = not positive definite
  double* scratch;    // NT_SCRATCH doubles
  int m;
};

// smallest eigenvalue estimate of the factored matrix: inverse iteration from `v` (LDS, overwritten)
__device__ __forceinline__ double nt_lambda_min(const double* F, const double* Dinv, int T, int mp, double* v,
                                                double (*red)[TAIL_WAVES], int iters) {
  const int tid = threadIdx.x;
  double lam = 0.0;
  for (int it = 0; it < iters; ++it) {
    double s[1] = {tid < mp ? v[tid] * v[tid] : 0.0};
    block_sum<1>(s, red);
    const double nrm = sqrt(s[0]);
    if (!(nrm > 0.0)) return 0.0;
    __syncthreads();
    if (tid < mp) v[tid] /= nrm;
    nt_solve(F, Dinv, T, v);
    double s2[1] = {tid < mp ? v[tid] * v[tid] : 0.0};
    block_sum<1>(s2, red);
    lam = 1.0 / sqrt(s2[0]);  // ||H^-1 u|| <= 1 / lambda_min: from above, closing in with every step
  }
  return lam;
}

static __global__ __launch_bounds__(TAIL_THREADS) void dense_spd_solve_kernel(DenseSolveArgs a) {
  __shared__ NtShared sh;
  __shared__ double v[NT_MAXT * NT_B];
  __shared__ double red[1][TAIL_WAVES];
  const int tid = threadIdx.x;
  const int m = a.m, T = (m + 15) >> 4, mp = 16 * T;
  double* F = a.scratch;
  double* Dinv = a.scratch + (int64_t)NT_TILES * 256;
  const int ntl = T * (T + 1) / 2;
  double dmax = 0.0;
  for (int i = 0; i < m; ++i) dmax = fmax(dmax, a.H[(int64_t)i * m + i]);
  for (int e = tid; e < ntl * 256; e += TAIL_THREADS) {
    const int t = e >> 8, w = e & 255;
    int I = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
    while (I * (I + 1) / 2 > t) --I;
    while ((I + 1) * (I + 2) / 2 <= t) ++I;
    const int J = t - I * (I + 1) / 2;
    const int l = w & 63, s = w >> 6;
    const int ii = 16 * I + (l & 15), jj = 16 * J + (l >> 4) + 4 * s;
    F[e] = (ii < m && jj < m) ? a.H[(int64_t)jj * m + ii] : (ii == jj ? 1.0 : 0.0);
  }
  if (tid < mp) v[tid] = tid < m ? a.rhs[tid] : 0.0;
  __syncthreads();
  const bool ok = nt_factor(F, Dinv, T, 1e-13 * dmax, sh);
  if (!ok) {
    if (tid == 0) {
      a.status[0] = 1;
      a.mu[0] = 0.0;
    }
    return;
  }
  nt_solve(F, Dinv, T, v);
  if (tid < m) a.x[tid] = v[tid];
  const double lam = nt_lambda_min(F, Dinv, T, mp, v, red, 2);
  if (tid == 0) {
    a.status[0] = 0;
    a.mu[0] = lam;
  }
}

}  // namespace slm

// Certified partial pass ("light pass", round 6): the re-verification of a few lanes after a miss without a pass over X.
//
// A working-set path verifies every point with the true gradient of a pass over X.  When a verification MISSES -- the
// plain step of the tail kernel leaves W: a feature the set did not hold wants in -- the columns are appended, the model
// is solved again and the new point b needs its own verification: one more pass over X (0.58 ms at 100k x 5k) for what
// is typically ONE lane, the deepest point of a path, where noise features enter that no earlier gradient could have
// told (five of nine draws of the headline's law and the literal make_regression dataset: a fourth pass for point 49).
//
// But b differs from the lane's base point z -- the last point whose gradient it holds, g(z) -- only on W, so
//     g(b) = g(z) + X^T dR / n,     dR = X_W (b - z)_W      (an n-vector from the gathered columns, no read of X),
// and for any column j, |X_j^T dR| / n <= (||X_j|| / sqrt n) (||dR|| / sqrt n) = c_j D (Cauchy-Schwarz; c_j the column norms,
// kept with the dataset beside its column-major copy).  So:
//   * on W the new gradient is exact from the Gram:            g_W(b) = g_W(z) + G_WW (b - z)_W;
//   * a column outside W with |g_j(z)| + c_j (S + D) < threshold_j CANNOT enter at b whatever its exact gradient is (S: the
//     slack of g(z) itself, zero when it came from a pass over X): the prox step keeps it at zero either way;
//   * the BORDERLINE columns -- the others, a few dozen to a few hundred -- get their exact gradient from their own rows of
//     the column-major copy: X_j^T dR, 0.8 MB per column instead of 4 GB for all.
// The tail kernel then runs on this hybrid gradient -- exact on W and on the borderline set, the base point's elsewhere,
// where it provably cannot matter -- under its unchanged acceptance test and stopping rule: the proximal-gradient mapping
// it evaluates is the one the true gradient gives, coordinate for coordinate.  A point accepted this way is a certified
// minimiser to the same tolerance; a lane that misses again appends and tries again, with the slack carried along.
// If a condition fails (a live lane off W, more than LT_LANES live lanes, more than LT_CAP borderline columns) the attempt
// stands down on the device (LightCtl::ok = 0) and the pass over X queued right behind it runs as always; when it stands,
// those kernels return at once (SplitArgs::skip).  Per-feature and group penalties (a group is certified as a whole, see
// light_select_kernel), shared paths over the dataset's own unweighted rows, one device (solve loop: PathCall::light_eligible).  Reference counterpart: none -- /root/reference/src/sparselm/model/_base.py:512-519 hands the
// problem to cvxpy once; this is a property of the verification scheme of the engine.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "split_kernels.hpp"
#include "tail_kernels.hpp"
#include "ws_kernels.hpp"

namespace slm {

constexpr int LT_CAP = 1024;  // borderline columns one attempt may read (0.8 MB each at n = 100k)
constexpr int LT_LANES = 8;   // live lanes one attempt serves: their points and their moves share sixteen MFMA columns
constexpr int LT_WAVES = 8;   // wavefronts of light_resid_kernel

struct LightCtl {
  int32_t ok;        // the attempt of this pass stands: the kernels of the pass over X behind it return at once
  int32_t n_cols;    // borderline columns listed (lt_idx)
  int32_t n_live;    // live lanes of the attempt ...
  int32_t lane_of[LT_LANES];  // ... and which they are
  int32_t attempts;  // over the solve
  int32_t used;      // ... of which stood (passes over X saved)
  int32_t cols_total;  // borderline columns read by all of them
  int32_t why;       // why the last attempt stood down: 1 too many live lanes, 2 a live lane off W or without the set,
                     // 3 too many borderline columns, 4 W holds a column a lane's hybrid gradient is not exact on (SLM_TRACE=3)
  int32_t id;        // number of the attempt under way (1, 2, ...: `attempts` as light_prepare_kernel counted it)
  int32_t epoch[SLM_MAX_LANES];  // per LANE: the attempt its base gradient g(z) comes from, 0: from a pass over X.  A hybrid
                                 // gradient is exact on the columns that attempt stamped (LightArgs::stamp) -- W and the
                                 // borderline set of its time -- and the working set's model reads g(z) on ALL of W: a lane
                                 // whose W has since taken in a column outside that set goes back to a pass over X (why 4)
  double D[LT_LANES];            // ||X_W (b - z)|| / sqrt(n) of live lane s
  double slack[SLM_MAX_LANES];   // per LANE: what its base gradient g(z) may be off by, in units of c_j, outside the columns
                                 // it is exact on: 0 after a pass over X, + D after every light pass
};

struct LightArgs {
  LightCtl* lt;
  const PathCtl* ctl;
  const int* done;
  const slm_path_point* pts;
  const WsCtl* ws;
  const int32_t* idx;  // [WS_KCAP]
  const int32_t* pos;  // [ld]
  const double* XW;    // [n][WS_KCAP]
  const double* XT;    // column-major copy, tiles of 32 rows
  const double* Gm;    // [WS_KCAP][WS_KCAP] (one row set)
  const double* y;
  const double* colnorm;  // [ld] ||X_j|| / sqrt(n)
  const double *z, *zprev, *gprev, *a0, *b0;  // per lane, stride ld
  double* g;             // [lanes][ld + 16]
  double* loss_partial;  // [nblk][slots]
  double* dR;            // [n][LT_LANES] the moves' residual changes
  double* d2_part;       // [nblk][LT_LANES]
  int32_t* cols;         // [LT_CAP] borderline columns
  int32_t* stamp;        // [ld] the attempt that last made column j's gradient exact for its live lanes (0: none this solve)
  double* part;          // [nblk][LT_CAP][LT_LANES] partial column products
  int64_t n, ld, rows_base, rows_rem;
  int p, n_lanes, slots, nblk;
  double inv_n;
  // group structure (singleton != 0: every feature its own group)
  const int* order;   // [p] feature of the k-th element in group-sorted order
  const int* gstart;  // [G + 1]
  int G, singleton;
};

// ---------------------------------------------------------------------------------------------
// c_j = ||X_j||_2 / sqrt(n) from the column-major copy: a workgroup per eight adjacent columns (2 KiB contiguous per row
// tile), a thread per (column, row of the tile), fixed order.  Built once per dataset beside the copy.
// ---------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void colnorm_kernel(const double* __restrict__ XT, int64_t n, int64_t ld, double inv_n,
                                                             double* __restrict__ out) {
  const int c = threadIdx.x >> 5, r = threadIdx.x & 31;
  const int64_t j = (int64_t)blockIdx.x * 8 + c;
  const int64_t tiles = (n + 31) >> 5;
  double s[4] = {0.0, 0.0, 0.0, 0.0};
  if (j < ld) {
    const double* src = XT + (j << 5) + r;
    const int64_t stride = ld << 5;
    int64_t t = 0;
    for (; t + 8 <= tiles; t += 8) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[(t + u) * stride];
#pragma unroll
      for (int u = 0; u < 8; ++u) s[u & 3] = __builtin_fma(v[u], v[u], s[u & 3]);
    }
    for (; t < tiles; ++t) {
      const double v = src[t * stride];
      s[0] = __builtin_fma(v, v, s[0]);
    }
  }
  double tot = (s[0] + s[1]) + (s[2] + s[3]);
#pragma unroll
  for (int off = 16; off >= 1; off >>= 1) tot += __shfl_xor(tot, off, 64);  // (the 32 rows of a column sit in one half of a wavefront)
  if (r == 0 && j < ld) out[j] = sqrt(tot * inv_n);
}

// ---------------------------------------------------------------------------------------------
// (1) which lanes are live, and may this pass be a light one at all?  One workgroup.
// ---------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(1024) void light_prepare_kernel(LightArgs a) {
  __shared__ double red[1][TAIL_WAVES];
  __shared__ int live_s[SLM_MAX_LANES];
  LightCtl* lt = a.lt;
  const int tid = threadIdx.x;
  if (*a.done != 0) {
    if (tid == 0) lt->ok = 0;
    return;
  }
  if (tid < SLM_MAX_LANES) live_s[tid] = (tid < a.n_lanes && !a.ctl[tid].done && !a.ctl[tid].idle) ? 1 : 0;
  __syncthreads();
  int n_live = 0, why = 0;
  for (int l = 0; l < a.n_lanes; ++l) n_live += live_s[l];
  const bool w_ok = a.ws->valid && !a.ws->building && !a.ws->disabled;
  if (n_live > LT_LANES || n_live == 0) why = 1;
  if (!w_ok) why = 2;
  for (int l = 0; l < a.n_lanes && why == 0; ++l) {
    if (!live_s[l]) continue;
    if (!a.ctl[l].zsup || a.ctl[l].mode != 1) why = 2;  // (spectral lanes whose next point lies on W)
  }
  if (why == 0) {
    // the base points must lie on W as well (a base that left W -- plain steps after the set was outgrown -- has a move the
    // gathered columns cannot express)
    double off[1] = {0.0};
    for (int l = 0; l < a.n_lanes; ++l) {
      if (!live_s[l]) continue;
      const double* zp = a.zprev + (int64_t)l * a.ld;
      for (int j = tid; j < a.p; j += 1024)
        if (a.pos[j] < 0 && zp[j] != 0.0) off[0] += 1.0;
    }
    block_sum<1>(off, red);
    if (off[0] != 0.0) why = 2;
  }
  if (why == 0) {
    // a live lane whose base gradient is a hybrid one: every column of W has to be one that attempt made exact
    const int K = a.ws->K;
    double bad[1] = {0.0};
    for (int l = 0; l < a.n_lanes; ++l) {
      if (!live_s[l] || lt->epoch[l] == 0) continue;
      for (int k = tid; k < K; k += 1024) {
        const int j = a.idx[k];
        if (j >= 0 && a.stamp[j] != lt->epoch[l]) bad[0] += 1.0;
      }
    }
    block_sum<1>(bad, red);
    if (bad[0] != 0.0) why = 4;
  }
  if (tid == 0) {
    lt->attempts += 1;
    lt->id = lt->attempts;
    lt->why = why;
    lt->ok = why == 0 ? 1 : 0;  // (light_select_kernel has the last word)
    lt->n_cols = 0;
    int s = 0;
    for (int l = 0; l < a.n_lanes && s < LT_LANES; ++l)
      if (live_s[l]) lt->lane_of[s++] = l;
    lt->n_live = why == 0 ? s : 0;
    if (why != 0)  // the pass over X behind this attempt delivers true gradients again
      for (int l = 0; l < SLM_MAX_LANES; ++l) {
        lt->slack[l] = 0.0;
        lt->epoch[l] = 0;
      }
  }
}

// ---------------------------------------------------------------------------------------------
// (2) one read of the gathered columns: the residuals of the live lanes' new points (their losses) and of their MOVES
// (dR = X_W (b - z)_W, and its squared norm), as sixteen columns of one MFMA product -- slot s: point of live lane s,
// slot 8 + s: its move.  The loop of resid_mfma_body (split_kernels.hpp), no row weights.
// ---------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(LT_WAVES * 64) void light_resid_kernel(LightArgs a) {
  if (!a.lt->ok) return;
  __shared__ double zw[WS_KCAP][16];  // 64 KiB
  __shared__ double lsum[LT_WAVES][16];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int K = a.ws->K;
  const int n_live = a.lt->n_live;
  const int64_t b = blockIdx.x;
  const int64_t r0 = b * a.rows_base + (b < a.rows_rem ? b : a.rows_rem);
  const int64_t nrows = a.rows_base + (b < a.rows_rem ? 1 : 0);
  const int64_t rend = r0 + nrows;
  for (int e = tid; e < K * 16; e += LT_WAVES * 64) {
    const int k = e >> 4, sl = e & 15;
    const int s = sl & 7;
    const int j = a.idx[k];
    double v = 0.0;
    if (j >= 0 && s < n_live) {
      const int64_t o = (int64_t)a.lt->lane_of[s] * a.ld + j;
      v = sl < 8 ? a.z[o] : a.z[o] - a.zprev[o];
    }
    zw[k][sl] = v;
  }
  __syncthreads();
  const int i16 = lane & 15, q = lane >> 4;
  const int ngroups = K >> 4;
  const int ntiles = (int)((nrows + 15) >> 4);
  double acc_sq = 0.0;  // slot i16: sum of squared errors (slots < 8) / of squared residual changes (slots >= 8)
  for (int t = wave; t < ntiles; t += LT_WAVES) {
    const int64_t row0 = r0 + 16 * (int64_t)t;
    const int64_t rl = row0 + i16 < rend ? row0 + i16 : rend - 1;  // (rows past the block are computed and dropped)
    const double* xp = a.XW + rl * WS_KCAP + 4 * q;
    slm_d4 acc = slm_d4{0.0, 0.0, 0.0, 0.0};
    double yv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t row = row0 + q + 4 * r;
      yv[r] = a.y[row < rend ? row : rend - 1];
    }
    slm_d4 xa[RM_U], xb[RM_U];
    auto load = [&](slm_d4(&xv)[RM_U], int g0) {
#pragma unroll
      for (int u = 0; u < RM_U; ++u) xv[u] = *reinterpret_cast<const slm_d4*>(xp + 16 * min(g0 + u, ngroups - 1));
    };
    auto compute = [&](const slm_d4(&xv)[RM_U], int g0) {
#pragma unroll
      for (int u = 0; u < RM_U; ++u)
        if (g0 + u < ngroups) {
          const double* zr = &zw[16 * (g0 + u) + 4 * q][i16];
#pragma unroll
          for (int m = 0; m < 4; ++m) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xv[u][m], zr[m * 16], acc, 0, 0, 0);
        }
    };
    load(xa, 0);
    for (int g0 = 0; g0 < ngroups; g0 += 2 * RM_U) {
      load(xb, g0 + RM_U);
      compute(xa, g0);
      load(xa, g0 + 2 * RM_U);
      compute(xb, g0 + RM_U);
    }
    // result register r of this lane: row row0 + q + 4 r, slot i16
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t row = row0 + q + 4 * r;
      if (row < rend) {
        if (i16 < 8) {
          const double err = acc[r] - yv[r];
          acc_sq = __builtin_fma(err, err, acc_sq);
        } else {
          a.dR[row * LT_LANES + (i16 - 8)] = acc[r];
          acc_sq = __builtin_fma(acc[r], acc[r], acc_sq);
        }
      }
    }
  }
  acc_sq += __shfl_xor(acc_sq, 16, 64);
  acc_sq += __shfl_xor(acc_sq, 32, 64);
  if (lane < 16) lsum[wave][lane] = acc_sq;
  __syncthreads();
  if (tid < 16) {
    double t = 0.0;
    for (int w2 = 0; w2 < LT_WAVES; ++w2) t += lsum[w2][tid];
    const int s = tid & 7;
    if (s < n_live) {
      if (tid < 8) a.loss_partial[b * a.slots + a.lt->lane_of[s]] = t;
      else a.d2_part[b * LT_LANES + s] = t;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// (3) the moves' lengths D_s, and the borderline columns: outside W, and for some live lane
//     |g_j(z)| + c_j (slack + D) (1 + 1e-9) >= threshold_j.  One workgroup; columns in index order.
// ---------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(1024) void light_select_kernel(LightArgs a) {
  LightCtl* lt = a.lt;
  if (!lt->ok) return;
  __shared__ double Ds[LT_LANES], sa_s[LT_LANES], sb_s[LT_LANES];
  __shared__ int lane_s[LT_LANES];
  __shared__ int wave_tot[TAIL_WAVES];
  const int tid = threadIdx.x;
  const int n_live = lt->n_live;
  // the moves' lengths: a wavefront per live lane sums the row blocks' parts in a fixed order (four per lane of the wavefront,
  // then the wavefront's tree) -- eight threads walking 256 blocks one after the other were 20 us of this kernel
  __shared__ double d2_s[LT_LANES];
  {
    const int s = tid >> 6, i = tid & 63;
    if (s < LT_LANES) {
      double t = 0.0;
      if (s < n_live)
        for (int b = i; b < a.nblk; b += 64) t += a.d2_part[(int64_t)b * LT_LANES + s];
      t = wave_sum_lane63(t);
      if (i == 63) d2_s[s] = t;
    }
  }
  __syncthreads();
  if (tid < LT_LANES) {
    double t = 0.0;
    if (tid < n_live) {
      t = sqrt(d2_s[tid] * a.inv_n);
      const int l = lt->lane_of[tid];
      const PathCtl* c = a.ctl + l;
      const slm_path_point pt = a.pts[c->pt_off + c->point];
      lane_s[tid] = l;
      sa_s[tid] = pt.sa;
      sb_s[tid] = pt.sb;
      t += lt->slack[l];  // what the base gradient itself may be off by
    }
    Ds[tid] = t;
  }
  __syncthreads();
  if (!a.singleton) {
    // Group penalties: a group outside W stays at zero at the new point iff || soft(g_g, a) ||_2 <= b_g, and every coordinate of
    // the true gradient lies within c_j (S + D) of the base point's: the group is certified when
    //     sqrt( sum_j max(|g_j(z)| + c_j (S + D) - a_j, 0)^2 ) < b_g        (b_g = 0: when that sum is exactly zero)
    // for every live lane; otherwise ALL its members are borderline columns (groups enter the working set whole).
    // A thread per group (tid, tid + 1024, ...), members in group-sorted order: the list is in group order.
    const int gper = (a.G + 1023) / 1024;  // (<= 64)
    unsigned long long gpick = 0ull;
    int gmine = 0;
    for (int u = 0; u < gper; ++u) {
      const int g = tid + 1024 * u;
      if (g >= a.G) break;
      const int k0 = a.gstart[g], k1 = a.gstart[g + 1];
      if (k1 <= k0 || a.pos[a.order[k0]] >= 0) continue;
      bool near = false;
      for (int sl = 0; sl < n_live && !near; ++sl) {
        const int64_t lo = (int64_t)lane_s[sl] * a.ld;
        double u2 = 0.0;
        for (int k = k0; k < k1; ++k) {
          const int j = a.order[k];
          const double e = fabs(a.gprev[lo + j]) + (a.colnorm[j] * (1.0 + 1e-9) + 1e-300) * Ds[sl] - sa_s[sl] * a.a0[lo + j];
          if (!(e <= 0.0)) u2 = __builtin_fma(e, e, u2);  // (NaN counts)
        }
        const double thr = sb_s[sl] * a.b0[lo + g];
        near = thr > 0.0 ? !(sqrt(u2) < thr * (1.0 - 1e-12)) : !(u2 == 0.0);
      }
      if (near) {
        gpick |= 1ull << u;
        gmine += k1 - k0;
      }
    }
    int gtotal = 0;
    int gat = block_excl_scan(gmine, wave_tot, &gtotal);
    __syncthreads();
    const bool gfits = gtotal <= LT_CAP;
    if (gfits)
      for (int u = 0; u < gper; ++u)
        if ((gpick >> u) & 1ull) {
          const int g = tid + 1024 * u;
          for (int k = a.gstart[g]; k < a.gstart[g + 1]; ++k) a.cols[gat++] = a.order[k];
        }
    if (tid == 0) {
      if (gfits) {
        lt->n_cols = gtotal;
        lt->used += 1;
        lt->cols_total += gtotal;
        for (int sl = 0; sl < n_live; ++sl) {
          lt->D[sl] = Ds[sl] - lt->slack[lane_s[sl]];
          lt->slack[lane_s[sl]] = Ds[sl];
          lt->epoch[lane_s[sl]] = lt->id;
        }
      } else {
        lt->ok = 0;
        lt->why = 3;
        for (int l = 0; l < SLM_MAX_LANES; ++l) {
          lt->slack[l] = 0.0;
          lt->epoch[l] = 0;
        }
      }
    }
    return;
  }
  const int per = (a.p + 1023) / 1024;
  const int j0 = tid * per, j1 = min(j0 + per, a.p);
  unsigned long long pick = 0ull;  // (per <= 64: p <= 65 536, the engine's bound)
  int mine = 0;
  if (per <= 8) {
    // this thread's features, everything they need asked for at once (feature by feature the scan was a chain of dependent
    // round trips: 30 us at p = 5 000)
    int ps[8];
    double cn[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = j0 + u < j1 ? j0 + u : 0;
      ps[u] = a.pos[j];
      cn[u] = a.colnorm[j] * (1.0 + 1e-9) + 1e-300;
    }
    bool near[8] = {false, false, false, false, false, false, false, false};
    for (int sl = 0; sl < n_live; ++sl) {
      double gv[8], av[8], bv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t o = (int64_t)lane_s[sl] * a.ld + (j0 + u < j1 ? j0 + u : 0);
        gv[u] = a.gprev[o];
        av[u] = a.a0[o];
        bv[u] = a.b0[o];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const double thr = sa_s[sl] * av[u] + sb_s[sl] * bv[u];
        near[u] = near[u] || !(fabs(gv[u]) + cn[u] * Ds[sl] < thr * (1.0 - 1e-12));  // (NaN counts as near)
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (j0 + u < j1 && ps[u] < 0 && near[u]) {
        pick |= 1ull << u;
        mine += 1;
      }
  } else {
    for (int j = j0; j < j1; ++j) {
      if (a.pos[j] >= 0) continue;
      const double cj = a.colnorm[j] * (1.0 + 1e-9) + 1e-300;
      bool near = false;
      for (int sl = 0; sl < n_live; ++sl) {
        const int64_t o = (int64_t)lane_s[sl] * a.ld + j;
        const double thr = sa_s[sl] * a.a0[o] + sb_s[sl] * a.b0[o];
        near = near || !(fabs(a.gprev[o]) + cj * Ds[sl] < thr * (1.0 - 1e-12));
      }
      if (near) {
        pick |= 1ull << (j - j0);
        mine += 1;
      }
    }
  }
  int total = 0;
  int at = block_excl_scan(mine, wave_tot, &total);
  __syncthreads();
  const bool fits = total <= LT_CAP;
  if (fits)
    for (int j = j0; j < j1; ++j)
      if ((pick >> (j - j0)) & 1ull) a.cols[at++] = j;
  if (tid == 0) {
    if (fits) {
      lt->n_cols = total;
      lt->used += 1;
      lt->cols_total += total;
      for (int s = 0; s < n_live; ++s) {
        lt->D[s] = Ds[s] - lt->slack[lane_s[s]];
        lt->slack[lane_s[s]] = Ds[s];
        lt->epoch[lane_s[s]] = lt->id;
      }
    } else {
      lt->ok = 0;
      lt->why = 3;
      for (int l = 0; l < SLM_MAX_LANES; ++l) {
        lt->slack[l] = 0.0;
        lt->epoch[l] = 0;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// (4) the borderline columns' products with the moves' residual changes, row block by row block: partial[b][c][s] =
// sum over the rows of block b of X[i][cols[c]] dR[i][s].  A wavefront per column (four in turn), its lanes over the rows
// (64 consecutive rows of a column: two 256-byte segments of the copy), dR of the block from an LDS image.
// ---------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void light_coldot_kernel(LightArgs a) {
  if (!a.lt->ok) return;
  extern __shared__ double dr_lds[];  // [rows of the block][LT_LANES]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_cols = a.lt->n_cols, n_live = a.lt->n_live;
  const int64_t b = blockIdx.x;
  const int64_t r0 = b * a.rows_base + (b < a.rows_rem ? b : a.rows_rem);
  const int nrows = (int)(a.rows_base + (b < a.rows_rem ? 1 : 0));
  for (int e = tid; e < nrows * LT_LANES; e += 256) dr_lds[e] = a.dR[r0 * LT_LANES + e];
  __syncthreads();
  for (int c = wave; c < n_cols; c += 4) {
    const int64_t j = a.cols[c];
    double acc[LT_LANES];
#pragma unroll
    for (int s = 0; s < LT_LANES; ++s) acc[s] = 0.0;
    for (int i0 = 0; i0 < nrows; i0 += 256) {  // four loads of 64 rows in flight
      double xv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + 64 * u + lane;
        const int64_t row = r0 + (i < nrows ? i : 0);
        xv[u] = a.XT[(((row >> 5) * a.ld + j) << 5) + (row & 31)];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + 64 * u + lane;
        if (i < nrows) {
#pragma unroll
          for (int s = 0; s < LT_LANES; ++s)
            if (s < n_live) acc[s] = __builtin_fma(xv[u], dr_lds[i * LT_LANES + s], acc[s]);
        }
      }
    }
#pragma unroll
    for (int s = 0; s < LT_LANES; ++s) {
      if (s < n_live) {
        double t = acc[s];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) t += __shfl_xor(t, off, 64);
        if (lane == 0) a.part[((int64_t)b * LT_CAP + c) * LT_LANES + s] = t;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// (5) the hybrid gradient of live lane s = blockIdx.x: the base point's everywhere, + G_WW (b - z)_W on W, + the column
// products on the borderline set; the loss from the residual kernel's block sums.  One workgroup per live lane.
// ---------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(1024) void light_assemble_kernel(LightArgs a) {
  const LightCtl* lt = a.lt;
  if (!lt->ok || (int)blockIdx.x >= lt->n_live) return;
  __shared__ double delta[WS_KCAP];
  __shared__ double red[1][TAIL_WAVES];
  const int tid = threadIdx.x;
  const int s = blockIdx.x;
  const int l = lt->lane_of[s];
  const int K = a.ws->K;
  const double* z = a.z + (int64_t)l * a.ld;
  const double* zp = a.zprev + (int64_t)l * a.ld;
  const double* gp = a.gprev + (int64_t)l * a.ld;
  double* g = a.g + (int64_t)l * (a.ld + 16);
  for (int k = tid; k < WS_KCAP; k += 1024) {
    const int j = k < K ? a.idx[k] : -1;
    delta[k] = j >= 0 ? z[j] - zp[j] : 0.0;
  }
  for (int j = tid; j < a.p; j += 1024) g[j] = gp[j];
  __syncthreads();
  const int id = lt->id;  // (every live lane's workgroup stamps the same columns with the same number)
  // on W: exact from the Gram (columns of the symmetric matrix read along rows: consecutive threads, consecutive entries)
  for (int k = tid; k < K; k += 1024) {
    const int j = a.idx[k];
    if (j < 0) continue;
    double acc = 0.0;
    for (int c = 0; c < K; c += 8) {
      double gv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) gv[u] = a.Gm[(int64_t)(c + u < K ? c + u : c) * WS_KCAP + k];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = __builtin_fma(gv[u], c + u < K ? delta[c + u] : 0.0, acc);
    }
    g[j] = gp[j] + acc;
    a.stamp[j] = id;
  }
  // on the borderline set: the row blocks' sums in block order
  const int n_cols = lt->n_cols;
  for (int c = tid; c < n_cols; c += 1024) {
    const int j = a.cols[c];
    double s4[4] = {0.0, 0.0, 0.0, 0.0};
    for (int b = 0; b < a.nblk; ++b) s4[b & 3] += a.part[((int64_t)b * LT_CAP + c) * LT_LANES + s];
    g[j] = gp[j] + ((s4[0] + s4[1]) + (s4[2] + s4[3])) * a.inv_n;
    a.stamp[j] = id;
  }
  double ls[1] = {0.0};
  for (int b = tid; b < a.nblk; b += 1024) ls[0] += a.loss_partial[(int64_t)b * a.slots + l];
  block_sum<1>(ls, red);
  if (tid == 0) g[a.ld] = ls[0] * 0.5 * a.inv_n;
}

}  // namespace slm

// Covariance passes (SLM_FLAG_COVARIANCE): when many solves share a row set -- the ten l1_ratio rows x fifty alphas of a
// CV fold, the re-weighting rounds of an Adaptive* grid -- the gradient of a pass, X^T W (X z - y) / n, is G z - c with
// G = X^T W X / n and c = X^T W y / n kept per row set: 8 p^2 bytes (200 MB at p = 5 000) instead of the 4 GB of X per
// pass for sixteen lanes.  The product G Z IS the second half of the split pass with G in the place of X and the lanes'
// points in the place of the residuals (xtr_mfma_kernel on (G, Z[p][16])); what is here is the little around it:
// packing Z, the finish g = G z - c with the loss 1/2 z^T G z - c^T z + 1/2 y^T W y / n, and building G: the product
// itself (cov_syrk_kernel, on the matrix cores) and the pieces around it.
// The reference has no counterpart (cvxpy canonicalises X^T X-free conic forms, model/_base.py:414-467); scikit-learn's
// `precompute=True` of lasso_path is the same idea on the host.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "split_kernels.hpp"
#include "ws_kernels.hpp"

namespace slm {

// Z[j][l] = z_l[j] (lane-minor, the B operand of xtr_mfma_kernel); lane slots beyond n_lanes are zero
// (grid.y: the halves of a call of more than sixteen lanes -- a plane of Z each, ld * 16 doubles apart)
static __global__ __launch_bounds__(256) void cov_pack_kernel(const double* z, int64_t ld, int n_lanes, double* Z, const int* done) {
  if (done != nullptr && *done != 0) return;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;  // over ld * 16
  if (e >= ld * SPLIT_RSTRIDE) return;
  const int64_t j = e >> 4;
  const int l = SPLIT_LANES * (int)blockIdx.y + (int)(e & 15);
  Z[(int64_t)blockIdx.y * ld * SPLIT_RSTRIDE + e] = l < n_lanes ? z[(int64_t)l * ld + j] : 0.0;
}

// The product of a covariance pass, partial[by][l][col] = sum_{row in block by} Z[row][l] G[row][col]: xtr_mfma_kernel's loop
// (same operands, same layouts: G in the place of X, the lanes' points Z[ld][16] in the place of the residuals), with one
// more way to run.  A point the model solver produced is zero outside the working set W (PathCtl::zsup, what lets the
// split pass form residuals from the gathered columns), so when EVERY live lane's point is such a point only the K <= 512
// rows of G listed in `idx` can contribute: the workgroups then read those rows (a.xrows_ws of the list each: indices first,
// then all loads of the block at once) -- 8 K ld bytes, 12 MB at K = 300, instead of the 200 MB of G, 8-10 us instead of 37
// per row set and pass.  Plain steps (zsup = 0), a working set under construction, or no working set at all: every row.
// The row sets of a call (the folds its lanes train on) go in ONE launch, a grid slice (blockIdx.z) each: their products are
// independent, and one after the other each was too small to fill the device (2.3 launches of 25 us per pass of config 4's
// grid where one of 30 does; the same for the two kernels that finish a pass).
struct CovBatch {
  const double* G[SLM_MAX_LANES];   // per row set: the Gram [ld][ld] ...
  const double* c[SLM_MAX_LANES];   // ... X^T W y / n [ld] ...
  double yy[SLM_MAX_LANES];         // ... and y^T W y / n
  int32_t set_of[SLM_MAX_LANES];    // row set of lane l
  int64_t part_stride;              // doubles between the partial sums of two row sets
  int64_t half_stride;              // cov_gz_body<2> (the model-Gram rounds of a call of more than sixteen lanes): doubles between
                                    // the two halves' blocks of partial sums
};

// (the loop is a function of its own: mg_gz_kernel, mg_kernels.hpp, runs it on the model Gram -- with H = 2 for both halves of a
//  call of more than sixteen lanes on one read of the Gram: the second half's points in a second plane of Z, a.r_plane doubles on)
// T: the Gram's element type -- double (the folds' Grams of covariance passes), or float (the model Gram, mg_kernels.hpp:
// stored in fp32, widened on load, multiplied and summed in fp64 like the other)
// `need_in`: bit h = half h of the launch has a lane that wants this product (the caller's own test -- mg_gz*_kernel: lanes
// still iterating); the body adds what it can tell itself: a half none of whose lanes belongs to THIS row set (blockIdx.z)
// -- or, with control blocks, none of whose live lanes does -- is not multiplied (a grid's lanes come fold by fold: of
// five row sets and thirty-two lanes most (set, half) pairs are empty), and a workgroup with nothing to do returns before
// it reads a byte.
template <int H = 1, typename T = double>
__device__ __forceinline__ void cov_gz_body(SplitArgs a, const CovBatch& cb, unsigned need_in = ~0u) {
  unsigned need = need_in & ((1u << H) - 1u);
  unsigned setmask[H];
#pragma unroll
  for (int h = 0; h < H; ++h) {
    const int l = (int)(threadIdx.x & 63);
    const int L = SPLIT_LANES * (a.lane0 + h) + l;
    const bool mine = l < SPLIT_LANES && L < a.n_lanes && cb.set_of[L < SLM_MAX_LANES ? L : 0] == (int)blockIdx.z;
    setmask[h] = (unsigned)__ballot(mine);
    if (setmask[h] == 0u) need &= ~(1u << h);
  }
  if (need == 0u) return;
  const T* GX = reinterpret_cast<const T*>(cb.G[blockIdx.z]);
  auto load2 = [&](int64_t at) -> d2 {  // two consecutive entries from element offset `at`
    if constexpr (sizeof(T) == 8) {
      return *reinterpret_cast<const d2*>(GX + at);
    } else {
      const float2 f = *reinterpret_cast<const float2*>(GX + at);
      return d2{(double)f.x, (double)f.y};
    }
  };
  a.partial += (int64_t)blockIdx.z * cb.part_stride;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  int bx = (int)blockIdx.x, by = (int)blockIdx.y;
  {  // XCD-aware tile order, as in xtr_mfma_kernel
    const int total = (int)(gridDim.x * gridDim.y), lin = bx + (int)gridDim.x * by;
    const int xcd = lin & 7, slot = lin >> 3, base = total >> 3, rem = total & 7;
    const int m = xcd * base + (xcd < rem ? xcd : rem) + slot;
    bx = m % (int)gridDim.x;
    by = m / (int)gridDim.x;
  }
  const int col0 = (bx * XTR_WAVES + wave) * XTR_CW;
  const int ld = (int)a.ld;
  if (col0 >= ld) return;  // (no barrier below)
  const int kq = lane >> 4, i16 = lane & 15;
  int coff[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int col = col0 + 32 * c + 2 * i16;
    coff[c] = col < ld - 2 ? col : ld - 2;
  }
  slm_d4 acc[H][8];
#pragma unroll
  for (int h = 0; h < H; ++h)
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[h][t] = slm_d4{0.0, 0.0, 0.0, 0.0};
  bool listed = false;
  if (a.ctl != nullptr && a.ws != nullptr && a.xrows_ws > 0) {
    // (a.lane0: the first half of the call this launch serves -- enqueue_gradient_cov; H = 2: both halves here)
    bool all_on = true;
#pragma unroll
    for (int h = 0; h < H; ++h) {
      unsigned live, on_ws;
      split_masks(a, live, on_ws, a.lane0 + h);
      live &= setmask[h];  // (this row set's lanes: the others' points are multiplied by THEIR Gram, in their grid slice)
      on_ws &= setmask[h];
      if (live == 0u) need &= ~(1u << h);
      all_on = all_on && live == on_ws;
    }
    if (need == 0u) return;  // (lanes that have all finished: nothing to multiply)
    listed = all_on && (int64_t)a.xrows_ws * (int64_t)gridDim.y >= (int64_t)a.ws->K;
  }
  if (listed) {
    const int K = a.ws->K;
    const int k0 = by * a.xrows_ws;
    // the rows of this block: indices first (one round trip), then every load of the block (a.xrows_ws <= 32: eight steps)
    int rows[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int k = k0 + 4 * t + kq;
      rows[t] = (4 * t < a.xrows_ws && k < K) ? a.idx[k] : -1;
    }
    d2 xv[8][4];
    double rv[H][8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int64_t r = rows[t] >= 0 ? rows[t] : 0;  // (a row that is not there multiplies row 0 by zero)
#pragma unroll
      for (int c = 0; c < 4; ++c) xv[t][c] = load2(r * a.ld + coff[c]);
#pragma unroll
      for (int h = 0; h < H; ++h) rv[h][t] = a.R[(int64_t)h * a.r_plane + r * SPLIT_RSTRIDE + i16];
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
#pragma unroll
      for (int h = 0; h < H; ++h) {
        if (!((need >> h) & 1u)) continue;
        const double z = rows[t] >= 0 ? rv[h][t] : 0.0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          acc[h][2 * c] = __builtin_amdgcn_mfma_f64_16x16x4f64(xv[t][c].x, z, acc[h][2 * c], 0, 0, 0);
          acc[h][2 * c + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(xv[t][c].y, z, acc[h][2 * c + 1], 0, 0, 0);
        }
      }
    }
  } else {
    const int64_t r0 = (int64_t)by * a.xrows;
    const int64_t r1 = r0 + a.xrows < a.n ? r0 + a.xrows : a.n;
    const int nb = r1 > r0 ? (int)((r1 - r0) / (4 * XTR_U)) : 0;  // full batches
    const int64_t xp = (r0 + kq) * a.ld;  // (element offset into the Gram)
    const double* rp = a.R + (r0 + kq) * SPLIT_RSTRIDE + i16;
    d2 xa[XTR_U][4], xb[XTR_U][4];
    double ra[H][XTR_U], rb[H][XTR_U];
    auto load = [&](d2(&xv)[XTR_U][4], double(&rv)[H][XTR_U], int b) {
      const int64_t xq = xp + (int64_t)b * (4 * XTR_U) * a.ld;
      const double* rq = rp + (int64_t)b * (4 * XTR_U) * SPLIT_RSTRIDE;
#pragma unroll
      for (int u = 0; u < XTR_U; ++u) {
#pragma unroll
        for (int c = 0; c < 4; ++c) xv[u][c] = load2(xq + (int64_t)u * 4 * a.ld + coff[c]);
#pragma unroll
        for (int h = 0; h < H; ++h) rv[h][u] = rq[(int64_t)h * a.r_plane + u * 4 * SPLIT_RSTRIDE];
      }
    };
    auto compute = [&](d2(&xv)[XTR_U][4], double(&rv)[H][XTR_U]) {
#pragma unroll
      for (int u = 0; u < XTR_U; ++u)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int h = 0; h < H; ++h) {
            if (H > 1 && !((need >> h) & 1u)) continue;  // (uniform for the workgroup)
            acc[h][2 * c] = __builtin_amdgcn_mfma_f64_16x16x4f64(xv[u][c].x, rv[h][u], acc[h][2 * c], 0, 0, 0);
            acc[h][2 * c + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(xv[u][c].y, rv[h][u], acc[h][2 * c + 1], 0, 0, 0);
          }
    };
    if (nb > 0) {
      load(xa, ra, 0);
      const int pairs = (nb - 1) >> 1;
      for (int k = 0; k < pairs; ++k) {
        load(xb, rb, 2 * k + 1);
        compute(xa, ra);
        load(xa, ra, 2 * k + 2);
        compute(xb, rb);
      }
      if ((nb - 1) & 1) {
        load(xb, rb, nb - 1);
        compute(xa, ra);
        compute(xb, rb);
      } else {
        compute(xa, ra);
      }
    }
    for (int64_t row = r0 + (int64_t)nb * (4 * XTR_U); row < r1; row += 4) {
      const bool ok = row + kq < r1;
      const int64_t rr = ok ? row + kq : r1 - 1;
      double rv[H];
#pragma unroll
      for (int h = 0; h < H; ++h) rv[h] = ok ? a.R[(int64_t)h * a.r_plane + rr * SPLIT_RSTRIDE + i16] : 0.0;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const d2 x = load2(rr * a.ld + coff[c]);
#pragma unroll
        for (int h = 0; h < H; ++h) {
          if (H > 1 && !((need >> h) & 1u)) continue;
          acc[h][2 * c] = __builtin_amdgcn_mfma_f64_16x16x4f64(x.x, rv[h], acc[h][2 * c], 0, 0, 0);
          acc[h][2 * c + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(x.y, rv[h], acc[h][2 * c + 1], 0, 0, 0);
        }
      }
    }
  }
  if (i16 < SPLIT_LANES) {  // tile 2c+e holds columns col0 + 32 c + 2 i + e
#pragma unroll
    for (int h = 0; h < H; ++h) {
      if (!((need >> h) & 1u)) continue;  // (nobody reads the sums of a half without lanes of this row set)
      double* out = a.partial + (int64_t)h * cb.half_stride + ((int64_t)by * SPLIT_LANES + i16) * a.ld;
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int col = col0 + 32 * c + 2 * (kq + 4 * r) + e;
            if (col < ld) out[col] = acc[h][2 * c + e][r];
          }
    }
  }
}

static __global__ __launch_bounds__(XTR_WAVES * 64, 2) void cov_gz_mfma_kernel(SplitArgs a, CovBatch cb) {
  if (a.done != nullptr && *a.done != 0) return;
  cov_gz_body<1>(a, cb);
}
// a call of more than sixteen lanes: both halves' points (the planes of Z, a.r_plane doubles apart) against ONE read of
// every Gram; partial sums of the second half cb.half_stride doubles on
static __global__ __launch_bounds__(XTR_WAVES * 64, 1) void cov_gz32_mfma_kernel(SplitArgs a, CovBatch cb) {
  if (a.done != nullptr && *a.done != 0) return;
  cov_gz_body<2>(a, cb);
}

struct CovFinishArgs {
  const double* partial;  // [row sets][nblk][16][ld] of cov_gz_mfma_kernel
  const double* z;        // [lanes][ld]
  double* g;              // [lanes][ld + 16]
  const int* done;
  int nblk;
  int64_t ld;
};

// grid = (ld / 16, lanes): g_l[col] = sum_blk partial[blk][l][col] - c[col]   (fixed order: bit-identical run to run)
static __global__ __launch_bounds__(256) void cov_reduce_kernel(CovFinishArgs a, CovBatch cb) {
  if (a.done != nullptr && *a.done != 0) return;
  const int lane = blockIdx.y;
  const int set = cb.set_of[lane];
  const int half = lane / SPLIT_LANES, l16 = lane % SPLIT_LANES;  // (a block of partial sums per half of the lanes and row set)
  a.partial += (int64_t)half * cb.half_stride + (int64_t)set * cb.part_stride;
  const double* c = cb.c[set];
  __shared__ double lds[16][17];
  const int tid = threadIdx.x, cl = tid & 15, slice = tid >> 4;
  const int64_t col = (int64_t)blockIdx.x * 16 + cl;
  double s = 0.0;
  for (int b = slice; b < a.nblk; b += 16) s += a.partial[((int64_t)b * SPLIT_LANES + l16) * a.ld + col];
  lds[slice][cl] = s;
  __syncthreads();
  if (slice == 0) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += lds[k][cl];
    a.g[(int64_t)lane * (a.ld + 16) + col] = t - c[col];
  }
}

// grid = lanes, 256 threads: loss_l = 1/2 sum_j z_j (g_j - c_j) + 1/2 yy  (g = G z - c already), into g_l[ld]
static __global__ __launch_bounds__(256) void cov_loss_kernel(CovFinishArgs a, CovBatch cb) {
  if (a.done != nullptr && *a.done != 0) return;
  const int lane = blockIdx.x;
  const double* c = cb.c[cb.set_of[lane]];
  const double yy = cb.yy[cb.set_of[lane]];
  __shared__ double red[256];
  const double* z = a.z + (int64_t)lane * a.ld;
  double* g = a.g + (int64_t)lane * (a.ld + 16);
  double s = 0.0;
  for (int64_t j = threadIdx.x; j < a.ld; j += 256) s = __builtin_fma(z[j], g[j] - c[j], s);
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w >= 1; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) g[a.ld] = 0.5 * red[0] + 0.5 * yy;
}

// The working set's Gram under covariance passes: G_WW is a sub-matrix of the row set's Gram -- no gathered columns, no
// product over the rows.  Stands in for ws_gather_kernel + ws_gram_kernel + ws_gram_reduce_kernel with the latter's grid
// ((WS_TILES^2, n_sets)) and protocol: the new tile rows and, mirrored, the matching columns of the old part; the last
// workgroup publishes the Gram.
struct CovSets {
  const double* G[SLM_MAX_LANES];  // per SET of the working set (WsArgs::set_of): the row set's Gram, [ld][ld]
};

static __global__ __launch_bounds__(256) void ws_gram_cov_kernel(WsArgs w, CovSets cs) {
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
    const int fi = w.idx[i], fj = w.idx[j];  // (padding positions: -1)
    const double s = (fi >= 0 && fj >= 0) ? cs.G[set][(int64_t)fi * w.ld + fj] : 0.0;
    double* Gs = w.Gm + (int64_t)set * (WS_KCAP * WS_KCAP);
    Gs[i * WS_KCAP + j] = s;
    if (j < row_lo) Gs[j * WS_KCAP + i] = s;
  }
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) {
    const int total = (tiles - I_lo) * tiles * (int)gridDim.y;
    if (atomicAdd(&ws->counter, 1) + 1 == total) {
      ws->counter = 0;
      ws->request = 0;
      ws->valid = 1;
      __threadfence();
      ws->building = 0;
    }
  }
}

// ---- building a Gram ------------------------------------------------------------------------------------------------
// fingerprint of a row-weight vector: two weighted sums with fixed pseudo-random multipliers, one workgroup per vector
// (grid = vectors: they run side by side), fixed order
struct CovFpArgs {
  const double* w[SLM_MAX_LANES];
};
static __global__ __launch_bounds__(1024) void cov_fingerprint_kernel(CovFpArgs fa, int64_t n, double* out_all) {
  __shared__ double r1[1024], r2[1024];
  const double* w = fa.w[blockIdx.x];
  double* out = out_all + 2 * blockIdx.x;
  double s1 = 0.0, s2 = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) {
    const uint64_t h = (uint64_t)(i + 1) * 0x9E3779B97F4A7C15ull;
    const double m1 = (double)((h >> 40) & 0xffffffull) * (1.0 / 16777216.0) + 0.5;
    const double m2 = (double)((h >> 12) & 0xffffffull) * (1.0 / 16777216.0) + 0.5;
    const double wi = w ? w[i] : 1.0;
    s1 = __builtin_fma(wi, m1, s1);
    s2 = __builtin_fma(wi, m2, s2);
  }
  r1[threadIdx.x] = s1;
  r2[threadIdx.x] = s2;
  __syncthreads();
  for (int k = 512; k >= 1; k >>= 1) {
    if ((int)threadIdx.x < k) {
      r1[threadIdx.x] += r1[threadIdx.x + k];
      r2[threadIdx.x] += r2[threadIdx.x + k];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = r1[0];
    out[1] = r2[0];
  }
}

// T[r][:] = s_r * X[rows[r]][:]  (rows == nullptr: row r itself; scale == nullptr: 1)
static __global__ __launch_bounds__(256) void cov_rows_kernel(const double* X, int64_t ld, const int64_t* rows, const double* scale,
                                                       int64_t n_rows, double* T) {
  const int64_t r = blockIdx.x;
  if (r >= n_rows) return;
  const int64_t src = rows ? rows[r] : r;
  const double s = scale ? sqrt(scale[src]) : 1.0;
  const double2* in = reinterpret_cast<const double2*>(X + src * ld);
  double2* out = reinterpret_cast<double2*>(T + r * ld);
  for (int64_t j = threadIdx.x; j < ld / 2; j += 256) {
    double2 v = in[j];
    v.x *= s;
    v.y *= s;
    out[j] = v;
  }
}

// the same into a block padded with rows of zeros: T[r][:] = X[rows[r]][:] for r < n_rows, 0 for n_rows <= r < gridDim.x
static __global__ __launch_bounds__(256) void cov_rows_pad_kernel(const double* X, int64_t ld, const int64_t* rows, int64_t n_rows, double* T) {
  const int64_t r = blockIdx.x;
  double2* out = reinterpret_cast<double2*>(T + r * ld);
  if (r >= n_rows) {
    for (int64_t j = threadIdx.x; j < ld / 2; j += 256) out[j] = double2{0.0, 0.0};
    return;
  }
  const double2* in = reinterpret_cast<const double2*>(X + rows[r] * ld);
  for (int64_t j = threadIdx.x; j < ld / 2; j += 256) out[j] = in[j];
}

// G = (A - B) * s  (B == nullptr: G = A * s), element-wise over count doubles
static __global__ __launch_bounds__(256) void cov_combine_kernel(const double* A, const double* B, double s, int64_t count, double* G) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256)
    G[i] = (A[i] - (B ? B[i] : 0.0)) * s;
}

// G += A, element-wise over count doubles
static __global__ __launch_bounds__(256) void cov_accumulate_kernel(const double* A, int64_t count, double* G) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) G[i] += A[i];
}

// c = -g, yy = 2 loss from a standard pass at z = 0 (g = -X^T W y / n, loss = y^T W y / (2 n))
static __global__ __launch_bounds__(256) void cov_linear_kernel(const double* g, int64_t ld, double* c, double* yy_out) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j < ld) c[j] = -g[j];
  if (j == 0) yy_out[0] = 2.0 * g[ld];
}

// ---- the folds of a K-fold split, built from gathered blocks of their test rows (engine.hip, cov_folds_begin) ----------
// R[i][0] = y[rows[i]], the other fifteen lane slots zero: the B operand of xtr_mfma_kernel for X_block^T y_block
static __global__ __launch_bounds__(256) void cov_targets_kernel(const double* y, const int64_t* rows, int64_t m, double* R) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;  // over m * 16
  if (e >= m * SPLIT_RSTRIDE) return;
  const int64_t i = e >> 4;
  R[e] = (e & 15) == 0 ? y[rows ? rows[i] : i] : 0.0;
}

// t[col] = sum_blk partial[blk][lane 0][col], fixed order; grid = ld / 256 (+1)
static __global__ __launch_bounds__(256) void cov_xty_kernel(const double* partial, int nblk, int64_t ld, double* t) {
  const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (col >= ld) return;
  double s = 0.0;
  for (int b = 0; b < nblk; ++b) s += partial[(int64_t)b * SPLIT_LANES * ld + col];
  t[col] = s;
}

// out[0] = sum_i y[rows[i]]^2 (one workgroup, fixed order), out[1..15] = 0
static __global__ __launch_bounds__(1024) void cov_yy_kernel(const double* y, const int64_t* rows, int64_t m, double* out) {
  __shared__ double red[1024];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < m; i += 1024) {
    const double v = y[rows ? rows[i] : i];
    s = __builtin_fma(v, v, s);
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 512; k >= 1; k >>= 1) {
    if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x < 16) out[threadIdx.x] = threadIdx.x == 0 ? red[0] : 0.0;
}

// S[i] = sum_f parts[f * stride + i], f in fixed order, over len doubles
static __global__ __launch_bounds__(256) void cov_sum_kernel(const double* parts, int count, int64_t stride, int64_t len, double* S) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < len; i += (int64_t)gridDim.x * 256) {
    double t = parts[i];
    for (int f = 1; f < count; ++f) t += parts[(int64_t)f * stride + i];
    S[i] = t;
  }
}

// ---- the folds' parts as PACKED lower triangles -------------------------------------------------------------------------
// P[i (i + 1) / 2 + j], j <= i < ld: half the bytes of the square in everything that follows the products -- the ranks' sum,
// the folds' sum, (all - part) / n -- until cov_unpack_kernel writes the square a pass reads.
__host__ __device__ inline int64_t cov_tri(int64_t i, int64_t j) { return i * (i + 1) / 2 + j; }

// C = A^T A of up to sixteen row blocks in ONE launch, a 64 x 128 result tile per WAVEFRONT: 32 accumulator tiles = 256
// registers, the whole AGPR half of a one-wave-per-SIMD budget, so a 4-row step is 32 products for six 16-byte loads per lane
// -- 4 rows x 256 contiguous bytes per instruction, the two doubles of a lane feeding the even / odd column tile as in
// xtr_mfma_kernel -- 3 bytes per lane and product where the 48 x 48 sub-tiles of cov_syrk_kernel<3> move 5.3, and it was the
// memory system, not the matrix pipe, that held that kernel at 44-50 TFLOP/s.  (96 x 96 would be 2.7 bytes, but its 288
// accumulator registers do not fit the 256 AGPRs: 568 spills.)  Wavefronts are independent (no LDS, no barrier): grid =
// (tiles / 4, blocks), and the batch is what fills the chip to the last round -- the 1 600 tiles of one fold at ld = 5 008
// are 1.6 rounds of 1 024 wavefronts, the 8 000 of five folds 7.8.  Tile (bi, bj): rows 64 bi.. of the result (first
// operand), columns 128 bj.. (second); kept when it touches the lower triangle, 128 bj <= 64 bi + 63 -- block rows 2 m and
// 2 m + 1 have m + 1 tiles each.  Operand index i of column tile u = 2 c + e stands for matrix column base + 32 c + 2 i + e.
struct SyrkBatch {
  const double* A[SLM_MAX_LANES];  // row-major blocks, leading dimension ld
  int64_t rows[SLM_MAX_LANES];
  double* P[SLM_MAX_LANES];        // packed lower triangles
};
constexpr int SYRK_TI = 64, SYRK_TJ = 128;  // a wavefront's tile: rows x columns of the result
constexpr int SYRK_NU = SYRK_TI / 16, SYRK_NV = SYRK_TJ / 16;
constexpr int SYRK_R = 2;  // 4-row steps per register set (two sets: one multiplied, one in flight)

// rows a block of `rows` rows must hold (zeros behind the data): whole rings of steps, and the ring loaded past the end
static inline int64_t cov_syrk_padded_rows(int64_t rows) {
  const int64_t steps = (rows + 3) / 4, rounds = (steps + 2 * SYRK_R - 1) / (2 * SYRK_R);
  return 4 * (2 * SYRK_R * rounds + SYRK_R);
}

static inline int cov_syrk_tiles(int64_t ld) {
  const int64_t nbi = (ld + SYRK_TI - 1) / SYRK_TI;
  int64_t t = 0;
  for (int64_t bi = 0; bi < nbi; ++bi) t += bi / 2 + 1;
  return (int)t;
}

static __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void cov_syrk_packed_kernel(SyrkBatch b, int64_t ld, int n_tiles) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int tile = (int)blockIdx.x * 4 + wave;
  if (tile >= n_tiles) return;
  int m = 0;  // pair of block rows: m (m + 1) tiles lie before it
  while ((m + 1) * (m + 2) <= tile) ++m;
  const int r = tile - m * (m + 1);
  const int bi = 2 * m + (r >= m + 1 ? 1 : 0), bj = r >= m + 1 ? r - (m + 1) : r;
  const double* A = b.A[blockIdx.y];
  const int64_t rows = b.rows[blockIdx.y];
  double* P = b.P[blockIdx.y];
  const int kq = lane >> 4, i16 = lane & 15;
  const int64_t i0 = (int64_t)bi * SYRK_TI, j0 = (int64_t)bj * SYRK_TJ;
  int64_t ca[SYRK_NU / 2], cb[SYRK_NV / 2];  // (columns past the row re-read its last pair: what they bring is never stored)
#pragma unroll
  for (int c = 0; c < SYRK_NU / 2; ++c) {
    const int64_t a_ = i0 + 32 * c + 2 * i16;
    ca[c] = a_ < ld - 2 ? a_ : ld - 2;
  }
#pragma unroll
  for (int c = 0; c < SYRK_NV / 2; ++c) {
    const int64_t b_ = j0 + 32 * c + 2 * i16;
    cb[c] = b_ < ld - 2 ? b_ : ld - 2;
  }
  slm_d4 acc[SYRK_NU][SYRK_NV];
#pragma unroll
  for (int u = 0; u < SYRK_NU; ++u)
#pragma unroll
    for (int v = 0; v < SYRK_NV; ++v) acc[u][v] = slm_d4{0.0, 0.0, 0.0, 0.0};
  // two register sets of SYRK_R steps each: one is multiplied while the other's loads are in flight (the loads of a set are
  // issued BEFORE the products of the other in program order, so every wait is an exact count -- written as one ring with
  // the load behind each product, the compiler hoisted all loads to the top of the body and consumed them in the same
  // iteration).  No test on the row: the block is padded with rows of zeros (cov_syrk_padded_rows), so every load, the
  // ones past the end included, is a plain load; a select on the loaded value made every load wait for itself.
  d2 xa[SYRK_R][SYRK_NU / 2], ya[SYRK_R][SYRK_NV / 2], xb[SYRK_R][SYRK_NU / 2], yb[SYRK_R][SYRK_NV / 2];
  const int64_t steps = (rows + 3) / 4;
  auto load = [&](d2(&x)[SYRK_R][SYRK_NU / 2], d2(&y)[SYRK_R][SYRK_NV / 2], int64_t step) {
#pragma unroll
    for (int t = 0; t < SYRK_R; ++t) {
      const double* row = A + ((step + t) * 4 + kq) * ld;
#pragma unroll
      for (int c = 0; c < SYRK_NU / 2; ++c) x[t][c] = *reinterpret_cast<const d2*>(row + ca[c]);
#pragma unroll
      for (int c = 0; c < SYRK_NV / 2; ++c) y[t][c] = *reinterpret_cast<const d2*>(row + cb[c]);
    }
  };
  auto compute = [&](d2(&x)[SYRK_R][SYRK_NU / 2], d2(&y)[SYRK_R][SYRK_NV / 2]) {
#pragma unroll
    for (int t = 0; t < SYRK_R; ++t)
#pragma unroll
      for (int u = 0; u < SYRK_NU; ++u) {
        const double au = (u & 1) ? x[t][u >> 1].y : x[t][u >> 1].x;
#pragma unroll
        for (int v = 0; v < SYRK_NV; ++v)
          acc[u][v] = __builtin_amdgcn_mfma_f64_16x16x4f64(au, (v & 1) ? y[t][v >> 1].y : y[t][v >> 1].x, acc[u][v], 0, 0, 0);
      }
  };
  load(xa, ya, 0);
  for (int64_t s = 0; s < steps; s += 2 * SYRK_R) {
    load(xb, yb, s + SYRK_R);
    __builtin_amdgcn_sched_barrier(0);  // (the loads stay ahead of the products they are to hide behind)
    compute(xa, ya);
    __builtin_amdgcn_sched_barrier(0);
    load(xa, ya, s + 2 * SYRK_R);
    __builtin_amdgcn_sched_barrier(0);
    compute(xb, yb);
    __builtin_amdgcn_sched_barrier(0);
  }
  // result register q of lane l of tile (u, v): D[i = (l >> 4) + 4 q][j = l & 15]
#pragma unroll
  for (int u = 0; u < SYRK_NU; ++u)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t gi = i0 + 32 * (u >> 1) + 2 * (kq + 4 * q) + (u & 1);
      if (gi >= ld) continue;
      double* prow = P + gi * (gi + 1) / 2;
#pragma unroll
      for (int v = 0; v < SYRK_NV; ++v) {
        const int64_t gj = j0 + 32 * (v >> 1) + 2 * i16 + (v & 1);
        if (gj <= gi) prow[gj] = acc[u][v][q];  // (gj <= gi < ld)
      }
    }
}

// G = (A - B) * scale from packed triangles to the square, mirrored (B == nullptr: G = A * scale).  One 32 x 32 tile of the
// lower triangle per workgroup: rows of the packed tile are read in 256-byte stretches, the mirror goes through LDS.
static __global__ __launch_bounds__(256) void cov_unpack_kernel(const double* A, const double* B, double scale, int64_t ld, double* G) {
  int bi = 0, rest = (int)blockIdx.x;
  while (rest > bi) {
    rest -= bi + 1;
    ++bi;
  }
  const int bj = rest;
  __shared__ double t[32][33];
  const int c = threadIdx.x & 31, r0 = threadIdx.x >> 5;  // 8 rows per sweep
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + 8 * k;
    const int64_t gi = 32LL * bi + r, gj = 32LL * bj + c;
    double v = 0.0;
    if (gi < ld && gj <= gi) {
      const int64_t at = cov_tri(gi, gj);
      v = (A[at] - (B ? B[at] : 0.0)) * scale;
      G[gi * ld + gj] = v;
    }
    t[r][c] = v;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + 8 * k;  // row of the mirrored tile = column of the original
    const int64_t gi = 32LL * bj + r, gj = 32LL * bi + c;  // element (gi, gj) = original (gj, gi)
    if (gj < ld && gi < gj) G[gi * ld + gj] = t[c][r];  // strictly above the diagonal
  }
}

// ---- the Gram product itself: C = A^T A for the row-major rows x ld block A, on v_mfma_f64_16x16x4_f64 -------------------
// One workgroup per 128 x 128 (or 96 x 96) tile of the LOWER triangle (tile rows bi >= tile columns bj; the mirror is written with it),
// four wavefronts as 2 x 2, each a 64 x 64 sub-tile = 4 x 4 result tiles (64 doubles of accumulators per lane).  The
// contraction runs over the rows of A, four per MFMA: A[i][k] = A_[r + k][i0 + i] and B[k][j] = A_[r + k][j0 + j] are the
// SAME access -- lane l reads element (row r + (l >> 4), column base + (l & 15)): four rows x 128 contiguous bytes per
// load instruction -- so both operands come straight from the matrix, no transposed copy, no LDS.  COV_DEPTH steps of loads
// are in flight ahead of the products (plain register ring, the loop fully unrolled over the ring).  2 rows ld^2 flops on the
// lower triangle: 2.5 TFLOP at 100 000 x 5 000 -- the matrix cores' fp64 rate bounds it (78 TFLOP/s peak: 32 ms).
// (Until this kernel the product was the BLAS library's dgemm, 76 ms warm -- and 8-13 s for its first call in a process on
//  a freshly booted box, while its code objects came off the disk.)
#ifndef SLM_COV_DEPTH
#define SLM_COV_DEPTH 4
#endif
constexpr int COV_DEPTH = SLM_COV_DEPTH;

// NT: result tiles per side of a wavefront's sub-tile (4: 128 x 128 per workgroup, 3: 96 x 96).  The smaller tile costs a
// third more loads per product and wins where it fills the CUs' rounds better: 820 tiles of 128 at ld = 5 008 are four rounds
// of 256 workgroups with the last round almost empty, 1 431 tiles of 96 are six rounds of 56 % the work each (cov_tile_for).
template <int NT>
__global__ __launch_bounds__(256) void cov_syrk_kernel(const double* A, int64_t rows, int64_t ld, double* C) {
  constexpr int COV_TILE = 32 * NT;
  constexpr int SUB = 16 * NT;  // a wavefront's sub-tile
  // tile pair of this workgroup: linear index over the lower triangle, row by row
  int bi = 0, rest = (int)blockIdx.x;
  while (rest > bi) {
    rest -= bi + 1;
    ++bi;
  }
  const int bj = rest;  // bj <= bi
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int wi = wave >> 1, wj = wave & 1;
  const int kq = lane >> 4, i16 = lane & 15;
  const int64_t i0 = (int64_t)bi * COV_TILE + SUB * wi, j0 = (int64_t)bj * COV_TILE + SUB * wj;
  // columns of this lane's operand elements (clamped inside the matrix: what they bring is never stored)
  int64_t ca[NT], cb[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int64_t a_ = i0 + 16 * t + i16, b_ = j0 + 16 * t + i16;
    ca[t] = a_ < ld ? a_ : ld - 1;
    cb[t] = b_ < ld ? b_ : ld - 1;
  }
  slm_d4 acc[NT][NT];
#pragma unroll
  for (int u = 0; u < NT; ++u)
#pragma unroll
    for (int v = 0; v < NT; ++v) acc[u][v] = slm_d4{0.0, 0.0, 0.0, 0.0};
  double ra[COV_DEPTH][NT], rb[COV_DEPTH][NT];
  const int64_t steps = (rows + 3) / 4;
  auto load = [&](int slot, int64_t step) {
    const int64_t r = step * 4 + kq;
    const bool ok = r < rows;
    const double* row = A + (ok ? r : 0) * ld;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      ra[slot][t] = ok ? row[ca[t]] : 0.0;
      rb[slot][t] = ok ? row[cb[t]] : 0.0;
    }
  };
  auto compute = [&](int slot) {
#pragma unroll
    for (int u = 0; u < NT; ++u)
#pragma unroll
      for (int v = 0; v < NT; ++v) acc[u][v] = __builtin_amdgcn_mfma_f64_16x16x4f64(ra[slot][u], rb[slot][v], acc[u][v], 0, 0, 0);
  };
#pragma unroll
  for (int d = 0; d < COV_DEPTH; ++d) load(d, d < steps ? d : steps - 1);  // (steps >= 1)
  int64_t s = 0;
  for (; s + COV_DEPTH <= steps - COV_DEPTH; s += COV_DEPTH) {  // whole rings with a full ring behind them
#pragma unroll
    for (int d = 0; d < COV_DEPTH; ++d) {
      compute(d);
      load(d, s + COV_DEPTH + d);
    }
  }
  for (; s < steps; ++s) {  // the rest: no more loads beyond the end
    const int slot = (int)(s % COV_DEPTH);
    // (the ring's slot index has to be a constant for the arrays to stay in registers)
#pragma unroll
    for (int d = 0; d < COV_DEPTH; ++d)
      if (slot == d) {
        compute(d);
        if (s + COV_DEPTH < steps) load(d, s + COV_DEPTH);
      }
  }
  // result register q of lane l of tile (u, v): D[i = (l >> 4) + 4 q][j = l & 15]
#pragma unroll
  for (int u = 0; u < NT; ++u)
#pragma unroll
    for (int v = 0; v < NT; ++v)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int64_t gi = i0 + 16 * u + kq + 4 * q, gj = j0 + 16 * v + i16;
        if (gi < ld && gj < ld) {
          const double val = acc[u][v][q];
          if (bi != bj || gj <= gi) C[gi * ld + gj] = val;          // the lower triangle (diagonal tiles: their own lower half)
          if (bi != bj || gj < gi) C[gj * ld + gi] = val;           // ... and its mirror
        }
      }
}

// the tile size that costs the fewest rounds x tile area on `cus` compute units (one workgroup per CU at a time)
static inline int cov_tile_for(int64_t ld, int cus) {
  double best = 0.0;
  int pick = 4;
  for (int nt : {4, 3}) {
    const int64_t t = 32 * nt, n = (ld + t - 1) / t, tiles = n * (n + 1) / 2;
    const double cost = (double)((tiles + cus - 1) / cus) * (double)(t * t) * (nt == 3 ? 1.06 : 1.0);  // (more loads per product)
    if (best == 0.0 || cost < best) {
      best = cost;
      pick = nt;
    }
  }
  return pick;
}

}  // namespace slm

// Split gradient pass for working-set solves: residuals first, then ONE stream over X for X^T r.
//
// With the working-set refinement (ws_kernels.hpp) nearly every point a lane asks the gradient at is
// supported on W, the <= 512 columns already gathered into the compact matrix XW.  Its residual
// r = W (X z - y) then needs only XW (n x K, a few tens of MB), and the pass over X reduces to the
// second half of the fused kernel, G = X^T R / n with R the n x 16 matrix of the lanes' residuals: a
// skinny GEMM, run on the matrix cores, so SIXTEEN lanes share one read of X where the fused kernel tops
// out at four, and its row loop has no dot product, no cross-wave exchange and no barrier at all.
//
//   resid_mfma_kernel    R[i][l] = w_l,i (XW_i . zW_l - y_i)      lanes whose z is supported on W
//   rowdot_mfma_kernel   R[i][l] = w_l,i (x_i . z_l - y_i)        the others (reads X; returns at once
//                                                                 when there are none)
//   xtr_mfma_kernel      partial[blk][l][:] = sum_{i in blk} R[i][l] x_i     (reads X once)
//
// The residual kernels use the same contiguous row blocks as the fused kernels; xtr_mfma_kernel has its
// own (column block, row block) grid; reduce_partials_kernel and everything after it are unchanged.  R is
// lane-minor ([n][16]): a row of R is a row of the MFMA B operand.  Reference counterpart: the `X @ beta` inside the cvxpy objective
// (src/sparselm/model/_lasso.py:109-121), as for the fused kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "grad_kernel.hpp"
#include "tail_kernels.hpp"
#include "ws_kernels.hpp"

namespace slm {

constexpr int SPLIT_LANES = 16;    // lane slots of one HALF of the split pass: the 16 columns of an MFMA B operand
constexpr int SPLIT_RSTRIDE = 16;  // doubles per row of a plane of R (two 64-byte scalar loads)
// A call of up to SLM_MAX_LANES = 32 lanes is two halves of sixteen: R has a plane per half ([halves][n][16]), the residual
// kernels run per half (grid.y), and xtr_mfma_kernel multiplies every row of X it has loaded by BOTH planes -- thirty-two
// lanes per read of X (measured: 0.70 ms against 0.57 for sixteen).  partial / loss_partial are [blocks][16 halves][...].
constexpr int SPLIT_HALVES = SLM_MAX_LANES / SPLIT_LANES;
static_assert(SLM_MAX_LANES % SPLIT_LANES == 0 && SPLIT_HALVES >= 1 && SPLIT_HALVES <= 2, "one or two halves of sixteen lanes");
constexpr int ROWDOT_LANES = 5;    // lanes per launch of rowdot_ring_kernel (z + row in registers: 162 VGPRs)

struct SplitArgs {
  const double* X;
  const double* y;
  const double* rw;      // row weights (nullptr = ones); lane l uses rw + l * rw_stride
  int64_t rw_stride;
  const double* z;       // [n_lanes][ld]
  double* R;             // [n][SPLIT_RSTRIDE] weighted residuals (SPLIT_LANES used)
  double* partial;       // [nblk][SPLIT_LANES][ld]
  double* loss_partial;  // [nblk][SPLIT_LANES]   sum_i w e^2 of the block
  const int* done;
  const PathCtl* ctl;    // nullptr => every lane takes its residual from X (slm_gradient, tests)
  const double* XW;      // [n][WS_KCAP]
  const int32_t* idx;    // [WS_KCAP]
  const WsCtl* ws;
  int64_t n, ld, rows_base, rows_rem;
  int p2;
  int n_lanes;
  int lane0;  // rowdot_ring_kernel: first lane of this launch
  int xrows;  // xtr_mfma_kernel: rows per workgroup row (multiple of 8); partial is then [gridDim.y][SPLIT_LANES][ld]
  int xrows_ws;  // cov_gz_mfma_kernel: listed rows per workgroup row when only the working set's rows are read (multiple of 4, <= 32)
  const double* XT;  // rowdot_mfma_kernel: column-major copy of X in tiles of 32 rows (tile_columns_kernel)
  int lane_slots;    // lane slots of partial / loss_partial: 16 x halves of the call (0 = 16)
  int64_t r_plane;   // doubles between the planes of R (the second half's residuals; 0 with one half)
  const int* skip;   // non-null and *skip != 0: this pass over X is not needed -- a certified partial pass has delivered the
                     // lanes' gradients (light_kernels.hpp) -- and every kernel of it returns at once
};
__device__ __forceinline__ bool split_off(const SplitArgs& a) {
  return (a.done != nullptr && *a.done != 0) || (a.skip != nullptr && *a.skip != 0);
}
__device__ __forceinline__ int split_slots(const SplitArgs& a) { return a.lane_slots > 0 ? a.lane_slots : SPLIT_LANES; }

// which lanes take their residual from XW: live, flagged by ws_solve_kernel, and W still published
// Which lane slots a residual kernel serves.  Lane l of every wavefront looks at control block l and the
// wavefront votes: one round of loads.  (Walking the sixteen blocks in a loop with short-circuit tests made
// every flag its own dependent load -- 11 us before rowdot_mfma_kernel found out it had nothing to do.)
// (`half`: the sixteen lanes 16 half ... 16 half + 15 of the call; bit l of a mask is lane 16 half + l)
__device__ __forceinline__ void split_masks(const SplitArgs& a, unsigned& live, unsigned& on_ws, int half = 0) {
  const int l = threadIdx.x & 63;
  const int L = SPLIT_LANES * half + l;
  int lv = 0, sp = 0;
  if (l < SPLIT_LANES && L < a.n_lanes) {
    if (a.ctl == nullptr) {
      lv = 1;
    } else {
      const int dn = a.ctl[L].done, id = a.ctl[L].idle, zs = a.ctl[L].zsup;
      lv = (dn == 0) & (id == 0);
      sp = lv & (zs != 0);
    }
  }
  const bool ws_ok = a.ctl != nullptr && a.ws != nullptr && a.ws->valid && !a.ws->building;
  live = (unsigned)__ballot(lv != 0);
  on_ws = ws_ok ? (unsigned)__ballot(sp != 0) : 0u;
}
__device__ __forceinline__ unsigned split_ws_mask(const SplitArgs& a, int half = 0) {
  unsigned live, on_ws;
  split_masks(a, live, on_ws, half);
  return on_ws;
}
// lane slots whose residual has to come from X
__device__ __forceinline__ unsigned split_x_mask(const SplitArgs& a, int half = 0) {
  unsigned live, on_ws;
  split_masks(a, live, on_ws, half);
  return live & ~on_ws;
}

// ---------------------------------------------------------------------------------------------
// residuals from the gathered columns: one THREAD per row.  A thread walks its row of XW (K
// contiguous doubles) while all threads of the workgroup read the same zW[k][0..7] from an LDS image
// (broadcast reads), so a row costs K/2 16-byte loads + 8K FMAs and no cross-lane reduction.  (One
// wavefront per row with eight DPP reductions per row measured 149 us; the rows of a wavefront are
// 2 KiB apart here, but XW is small and the lines are fully used over the k loop.)
// ---------------------------------------------------------------------------------------------
template <int B>
__global__ __launch_bounds__(256) void resid_ws_kernel(SplitArgs a) {
  static_assert(B == SPLIT_LANES && B % 2 == 0, "one launch serves every lane slot");
  if (split_off(a)) return;
  const unsigned mask = split_ws_mask(a);
  if (mask == 0u) return;
  __shared__ double zw[WS_KCAP][B];  // 64 KiB
  __shared__ double lsum[4][B];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = a.ws->K;  // multiple of 16; positions >= Kreal hold zero columns
  const int64_t b = blockIdx.x;
  const int64_t r0 = b * a.rows_base + (b < a.rows_rem ? b : a.rows_rem);
  const int64_t nrows = a.rows_base + (b < a.rows_rem ? 1 : 0);
  for (int e = tid; e < K * B; e += 256) {
    const int k = e / B, l = e - k * B;
    const int j = a.idx[k];
    zw[k][l] = (j >= 0 && l < a.n_lanes) ? a.z[(int64_t)l * a.ld + j] : 0.0;
  }
  __syncthreads();
  double loss[B];
#pragma unroll
  for (int l = 0; l < B; ++l) loss[l] = 0.0;
  for (int64_t i = tid; i < nrows; i += 256) {
    const int64_t row = r0 + i;
    const d2* xr = reinterpret_cast<const d2*>(a.XW + row * WS_KCAP);
    double dot[B];
#pragma unroll
    for (int l = 0; l < B; ++l) dot[l] = 0.0;
    for (int k = 0; k < K; k += 2) {
      const d2 x = xr[k >> 1];
#pragma unroll
      for (int l = 0; l < B; ++l) {
        dot[l] = __builtin_fma(x.x, zw[k][l], dot[l]);
        dot[l] = __builtin_fma(x.y, zw[k + 1][l], dot[l]);
      }
    }
    const double yi = a.y[row];
    double res[B];
#pragma unroll
    for (int l = 0; l < B; ++l) {
      const double m = a.rw ? a.rw[(int64_t)l * a.rw_stride + row] : 1.0;
      const double e = dot[l] - yi;
      res[l] = e * m;
      loss[l] = __builtin_fma(res[l], e, loss[l]);
    }
    // only the lanes served here own their slot of R (the others belong to rowdot_ring_kernel)
    if (mask == (1u << B) - 1u) {
      d2* out = reinterpret_cast<d2*>(a.R + row * SPLIT_RSTRIDE);
#pragma unroll
      for (int l = 0; l < B; l += 2) out[l >> 1] = d2{res[l], res[l + 1]};
    } else {
#pragma unroll
      for (int l = 0; l < B; ++l)
        if ((mask >> l) & 1u) a.R[row * SPLIT_RSTRIDE + l] = res[l];
    }
  }
#pragma unroll
  for (int l = 0; l < B; ++l) {
    double t = loss[l];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) t += __shfl_xor(t, off, 64);
    if (lane == 0) lsum[wave][l] = t;
  }
  __syncthreads();
  if (tid < B && ((mask >> tid) & 1u))
    a.loss_partial[b * split_slots(a) + tid] = lsum[0][tid] + lsum[1][tid] + lsum[2][tid] + lsum[3][tid];
}

// ---------------------------------------------------------------------------------------------
// residuals from X for the lanes the kernel above does not serve: the first half of grad_ring_kernel
// (same ring, same DPP reductions, one barrier per row for the cross-wave exchange)
// ---------------------------------------------------------------------------------------------
template <int W, int C, int B, int D>
__global__ __launch_bounds__(W * 64) void rowdot_ring_kernel(SplitArgs a) {
  constexpr int T = W * 64;
  constexpr int SLOT = T * C * 16;
  constexpr int RING = (D + 1) * SLOT;
  constexpr int RED = 2 * B * W * 8;
  static_assert(RING + RED <= 160 * 1024, "ring does not fit the 160 KiB LDS");
  static_assert(D * C < 64, "too many DMA loads in flight for vmcnt");
  if (split_off(a)) return;
  // lanes served here: live, not served by resid_ws_kernel, inside this launch's window of B lanes
  const int lane0 = a.lane0 + (int)blockIdx.y * B;  // grid.y = windows of B lanes (one launch for all of them)
  const unsigned mask = (split_x_mask(a) >> lane0) & ((1u << B) - 1u);
  if (mask == 0u) return;

  __shared__ __attribute__((aligned(16))) char smem[RING + RED];
  double* red = reinterpret_cast<double*>(smem + RING);  // [2][B][W]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t b = blockIdx.x;
  const int64_t r0 = b * a.rows_base + (b < a.rows_rem ? b : a.rows_rem);
  const int64_t nrows = a.rows_base + (b < a.rows_rem ? 1 : 0);

  // Cold start: every lane served here still sits at z = 0 (PathCtl::zzero), so e = -y and X is
  // not needed at all -- the first pass of a path costs one read of X, not two.
  bool all_zero = a.ctl != nullptr;
  if (all_zero) {
    for (int l = 0; l < B; ++l)
      if (((mask >> l) & 1u) && !a.ctl[lane0 + l].zzero) all_zero = false;
  }
  if (all_zero) {
    double ls[B];
#pragma unroll
    for (int l = 0; l < B; ++l) ls[l] = 0.0;
    for (int64_t i = tid; i < nrows; i += T) {
      const int64_t row = r0 + i;
      const double e = -a.y[row];
#pragma unroll
      for (int l = 0; l < B; ++l) {
        if ((mask >> l) & 1u) {
          const double m = a.rw ? a.rw[(int64_t)(lane0 + l) * a.rw_stride + row] : 1.0;
          a.R[row * SPLIT_RSTRIDE + lane0 + l] = e * m;
          ls[l] = __builtin_fma(e * m, e, ls[l]);
        }
      }
    }
#pragma unroll
    for (int l = 0; l < B; ++l) {
      double t = ls[l];
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) t += __shfl_xor(t, off, 64);
      if (lane == 0) red[l * W + wave] = t;
    }
    __syncthreads();
    if (tid < B && ((mask >> tid) & 1u)) {
      double t = 0.0;
      for (int w2 = 0; w2 < W; ++w2) t += red[tid * W + w2];
      a.loss_partial[b * split_slots(a) + lane0 + tid] = t;
    }
    return;
  }

  uint32_t coff[C];
  d2 zr[B][C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int ci = c * T + tid;
    const bool valid = ci < a.p2;
    const int cc = valid ? ci : a.p2 - 1;
    coff[c] = (uint32_t)cc * 16u;
#pragma unroll
    for (int l = 0; l < B; ++l) {
      const d2 zz = lane0 + l < a.n_lanes ? reinterpret_cast<const d2*>(a.z + (int64_t)(lane0 + l) * a.ld)[cc]
                                          : d2{0.0, 0.0};
      zr[l][c] = valid ? zz : d2{0.0, 0.0};
    }
  }
  double loss[B];
#pragma unroll
  for (int l = 0; l < B; ++l) loss[l] = 0.0;

  auto issue_row = [&](int64_t i, int slot) {
    const char* rp = reinterpret_cast<const char*>(a.X + (r0 + i) * a.ld);
#pragma unroll
    for (int c = 0; c < C; ++c) {
      char* dst = smem + slot * SLOT + (c * T + wave * 64) * 16;
      __builtin_amdgcn_global_load_lds((gptr_t)(rp + coff[c]), (lptr_t)dst, 16, 0, SLM_NT_LOADS ? 2 : 0);
    }
  };

  if (nrows > 0) {
#pragma unroll
    for (int k = 0; k < D; ++k)
      if (k < nrows) issue_row(k, k);
    int slot = 0, slot_in = D;
    for (int64_t i = 0; i < nrows; ++i) {
      const int64_t left = nrows - 1 - i;
      if (left >= D) {
        issue_row(i + D, slot_in);
        wait_vmcnt<D * C>();
      } else {
        if (D >= 3 && left == 2) wait_vmcnt<(D >= 3 ? 2 : 0) * C>();
        else if (D >= 2 && left == 1) wait_vmcnt<(D >= 2 ? 1 : 0) * C>();
        else wait_vmcnt<0>();
      }
      const int64_t row = r0 + i;
      uint64_t yi_bits = smem_load_u64(a.y + row);
      d2 x[C];
#pragma unroll
      for (int c = 0; c < C; ++c)
        x[c] = *reinterpret_cast<const d2*>(smem + slot * SLOT + (c * T + tid) * 16);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(yi_bits) : : "memory");
      const double yi = __longlong_as_double((long long)yi_bits);
      double dot[B];
#pragma unroll
      for (int l = 0; l < B; ++l) {
        double t = 0.0;
#pragma unroll
        for (int c = 0; c < C; ++c) {
          t = __builtin_fma(x[c].x, zr[l][c].x, t);
          t = __builtin_fma(x[c].y, zr[l][c].y, t);
        }
        dot[l] = wave_sum_lane63(t);
        if constexpr (W == 1) dot[l] = read_lane63(dot[l]);
      }
      if constexpr (W > 1) {
        const int parity = (int)(i & 1);
        if (lane == 63) {
#pragma unroll
          for (int l = 0; l < B; ++l) red[(parity * B + l) * W + wave] = dot[l];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int l = 0; l < B; ++l) dot[l] = group_sum_all<W>(red[(parity * B + l) * W + (lane & (W - 1))]);
      }
      if (tid == 0) {
#pragma unroll
        for (int l = 0; l < B; ++l) {
          if ((mask >> l) & 1u) {
            const double m = a.rw ? a.rw[(int64_t)(lane0 + l) * a.rw_stride + row] : 1.0;
            const double e = dot[l] - yi;
            const double res = e * m;
            a.R[row * SPLIT_RSTRIDE + lane0 + l] = res;
            loss[l] = __builtin_fma(res, e, loss[l]);
          }
        }
      }
      slot = (slot == D) ? 0 : slot + 1;
      slot_in = (slot_in == D) ? 0 : slot_in + 1;
    }
  }
  if (tid == 0) {
#pragma unroll
    for (int l = 0; l < B; ++l)
      if ((mask >> l) & 1u) a.loss_partial[b * split_slots(a) + lane0 + l] = loss[l];
  }
}

// ---------------------------------------------------------------------------------------------
// X^T R on the matrix cores: partial[by][l][col] = sum_{row in block by} R[row][l] X[row][col]
// as v_mfma_f64_16x16x4_f64 products, D[i][j] += sum_k A[i][k] B[k][j] with k = 4 consecutive rows,
// A[i][k] = X[row_k][col_i] (16 columns) and B[k][j] = R[row_k][j] -- R's 16-double rows ARE the B
// operand (lane l of the wavefront holds B[k = l >> 4][j = l & 15]: one coalesced 512-byte load per four
// rows; the slots of lanes a call does not use hold zeros).  Lane l holds A[i = l & 15][k = l >> 4], so ONE 16-byte load per
// lane brings four rows x 32 consecutive columns (256 contiguous bytes per row) and its two doubles feed
// two MFMAs (even and odd columns).  A wavefront owns 128 columns (8 result tiles = 64 registers), a
// workgroup 512, grid.y splits the rows; two 8-row batches are in flight per wavefront (plain register
// double buffering, no LDS, no barrier).  The vector units only move data: 27 TFLOP/s of the 2 n p 16
// products run on the MFMA pipe at a third of its fp64 rate, so the kernel is bound by HBM alone.
// Measured at n = 100k, p = 5k on one box (tools/probes/xtr_mfma.hip): 0.60 ms (6.8 TB/s) against 0.63 ms
// for the vector-FMA kernel it replaced (rows DMA'd through an LDS ring, residuals by scalar loads, ten
// lanes at most: tools/probes/xtr_lanes.hip keeps that loop); 32 residual columns cost 0.70 ms, 48 cost 1.0 ms.
// Result register r of lane l is D[i = (l >> 4) + 4 r][j = l & 15] (guide "Fragment layout").
// ---------------------------------------------------------------------------------------------
typedef double slm_d4 __attribute__((ext_vector_type(4)));
constexpr int XTR_WAVES = 4;                   // wavefronts per workgroup
constexpr int XTR_CW = 128;                    // columns per wavefront
constexpr int XTR_CB = XTR_WAVES * XTR_CW;     // columns per workgroup
constexpr int XTR_U = 2;                       // 4-row steps per batch

// E > 0 (with H = 1): a call of 16 + E lanes, E <= 4 -- the sixteen lanes of plane 0 on the matrix cores as ever, the first E
// slots of plane 1 on the VECTOR units against the same loads of X: ex[e][c] += x * R1[row][e], a thread's own rows (row
// kq of every group of four) and its own eight columns, the four row classes summed by two shuffles at the end.  16 E / 2
// v_fma_f64 per 4-row step beside its 8 MFMAs (fp64 vector and matrix rates are the same on this chip: a sixteenth of the
// matrix work per extra lane), against a whole second set of MFMAs in xtr32_mfma_kernel: eighteen lanes at the price of
// sixteen, which takes a 50-point path from four passes over X to three.
template <int H, int E = 0>
__device__ __forceinline__ void xtr_mfma_body(SplitArgs& a) {
  static_assert(SPLIT_RSTRIDE == 16 && SPLIT_LANES <= 16, "a row of a plane of R is the 16-wide B operand");
  static_assert(H == 1 || H == 2, "one or two planes of R");
  static_assert(E == 0 || (H == 1 && (E == 2 || E == 4)), "extras: pairs of slots of plane 1 beside the sixteen of plane 0");
  constexpr int E2 = E > 0 ? E / 2 : 1;  // pairs of extra slots (16-byte loads of plane 1's rows)
  if (split_off(a)) return;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  // XCD-aware tile order (for speed only): workgroups are handed to the 8 XCDs round-robin by linear id, and
  // the column blocks of one row block all read that block's slice of R -- give each XCD a contiguous run of
  // (row block, column block) tiles so a slice is fetched into one L2 instead of eight (PMC: 103 MB of the
  // 4.109 GB read per launch were R re-reads).
  int bx = (int)blockIdx.x, by = (int)blockIdx.y;
  {
    const int total = (int)(gridDim.x * gridDim.y), lin = bx + (int)gridDim.x * by;
    const int xcd = lin & 7, slot = lin >> 3, base = total >> 3, rem = total & 7;
    const int m = xcd * base + (xcd < rem ? xcd : rem) + slot;
    bx = m % (int)gridDim.x;
    by = m / (int)gridDim.x;
  }
  const int col0 = (bx * XTR_WAVES + wave) * XTR_CW;
  const int ld = (int)a.ld;
  if (col0 >= ld) return;  // (no barrier below)
  const int64_t r0 = (int64_t)by * a.xrows;
  const int64_t r1 = r0 + a.xrows < a.n ? r0 + a.xrows : a.n;
  const int kq = lane >> 4, i16 = lane & 15;
  int coff[4];  // columns past the row (ld is a multiple of 16, not of 32) re-read its last pair; not stored
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
  const int nb = (int)((r1 - r0) / (4 * XTR_U));  // full batches
  const double* xp = a.X + (r0 + kq) * a.ld;
  const double* rp = a.R + (r0 + kq) * SPLIT_RSTRIDE + i16;
  d2 xa[XTR_U][4], xb[XTR_U][4];
  double ra[H][XTR_U], rb[H][XTR_U];
  d2 ea[E2][XTR_U], eb[E2][XTR_U];  // (E > 0) plane 1's first E slots of this thread's rows
  d2 ex[2 * E2][4];                 // (E > 0) extra lane e, column pair c: sums over this thread's rows
#pragma unroll
  for (int e = 0; e < 2 * E2; ++e)
#pragma unroll
    for (int c = 0; c < 4; ++c) ex[e][c] = d2{0.0, 0.0};
  const double* rp1 = a.R + a.r_plane + (r0 + kq) * SPLIT_RSTRIDE;
  auto load = [&](d2(&xv)[XTR_U][4], double(&rv)[H][XTR_U], d2(&ev)[E2][XTR_U], int b) {
    const double* xq = xp + (int64_t)b * (4 * XTR_U) * a.ld;
    const double* rq = rp + (int64_t)b * (4 * XTR_U) * SPLIT_RSTRIDE;
#pragma unroll
    for (int u = 0; u < XTR_U; ++u) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const d2* src = reinterpret_cast<const d2*>(xq + (int64_t)u * 4 * a.ld + coff[c]);
        xv[u][c] = SLM_NT_LOADS ? __builtin_nontemporal_load(src) : *src;
      }
#pragma unroll
      for (int h = 0; h < H; ++h) rv[h][u] = rq[(int64_t)h * a.r_plane + u * 4 * SPLIT_RSTRIDE];
      if constexpr (E > 0) {
        const d2* eq = reinterpret_cast<const d2*>(rp1 + ((int64_t)b * (4 * XTR_U) + u * 4) * SPLIT_RSTRIDE);
#pragma unroll
        for (int q = 0; q < E2; ++q) ev[q][u] = eq[q];
      }
    }
  };
  auto extras = [&](const d2(&x)[4], const d2(&ev)[E2]) {  // (the vector units' share of a 4-row step)
#pragma unroll
    for (int q = 0; q < E2; ++q)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        ex[2 * q][c].x = __builtin_fma(x[c].x, ev[q].x, ex[2 * q][c].x);
        ex[2 * q][c].y = __builtin_fma(x[c].y, ev[q].x, ex[2 * q][c].y);
        ex[2 * q + 1][c].x = __builtin_fma(x[c].x, ev[q].y, ex[2 * q + 1][c].x);
        ex[2 * q + 1][c].y = __builtin_fma(x[c].y, ev[q].y, ex[2 * q + 1][c].y);
      }
  };
  auto compute = [&](d2(&xv)[XTR_U][4], double(&rv)[H][XTR_U], d2(&ev)[E2][XTR_U]) {
#pragma unroll
    for (int u = 0; u < XTR_U; ++u) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int h = 0; h < H; ++h) {
          acc[h][2 * c] = __builtin_amdgcn_mfma_f64_16x16x4f64(xv[u][c].x, rv[h][u], acc[h][2 * c], 0, 0, 0);
          acc[h][2 * c + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(xv[u][c].y, rv[h][u], acc[h][2 * c + 1], 0, 0, 0);
        }
      if constexpr (E > 0) {
        d2 eu[E2];
#pragma unroll
        for (int q = 0; q < E2; ++q) eu[q] = ev[q][u];
        extras(xv[u], eu);
      }
    }
  };
  // (straight-line steady state, the odd batch peeled off: with `if (b + 1 < nb) load(...)` in the loop the compiler
  //  had to wait at the first product of a batch as if the batch behind it had not been asked for -- vmcnt(1) with
  //  ten loads just issued.  The exact counts changed nothing measurable here -- 0.587-0.618 ms either way on one box,
  //  profiles/r03a_xtr_peel_ab.txt: the kernel waits for the memory system, not for its own waits -- but they are what
  //  took rowdot_mfma_kernel, which has the same loop, from 0.69 to 0.60 ms)
  if (nb > 0) {
    load(xa, ra, ea, 0);
    const int pairs = (nb - 1) >> 1;
    for (int k = 0; k < pairs; ++k) {
      load(xb, rb, eb, 2 * k + 1);
      compute(xa, ra, ea);
      load(xa, ra, ea, 2 * k + 2);
      compute(xb, rb, eb);
    }
    if ((nb - 1) & 1) {
      load(xb, rb, eb, nb - 1);
      compute(xa, ra, ea);
      compute(xb, rb, eb);
    } else {
      compute(xa, ra, ea);
    }
  }
  // the last rows of the block (fewer than 8): 4-row steps, rows past the end contribute R = 0
  for (int64_t row = r0 + (int64_t)nb * (4 * XTR_U); row < r1; row += 4) {
    const bool ok = row + kq < r1;
    const int64_t rr = ok ? row + kq : r1 - 1;
    double rv[H];
#pragma unroll
    for (int h = 0; h < H; ++h) rv[h] = ok ? a.R[(int64_t)h * a.r_plane + rr * SPLIT_RSTRIDE + i16] : 0.0;
    d2 x[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      x[c] = *reinterpret_cast<const d2*>(a.X + rr * a.ld + coff[c]);
#pragma unroll
      for (int h = 0; h < H; ++h) {
        acc[h][2 * c] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[c].x, rv[h], acc[h][2 * c], 0, 0, 0);
        acc[h][2 * c + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[c].y, rv[h], acc[h][2 * c + 1], 0, 0, 0);
      }
    }
    if constexpr (E > 0) {
      d2 eu[E2];
#pragma unroll
      for (int q = 0; q < E2; ++q)
        eu[q] = ok ? reinterpret_cast<const d2*>(a.R + a.r_plane + rr * SPLIT_RSTRIDE)[q] : d2{0.0, 0.0};
      extras(x, eu);
    }
  }
  constexpr int SLOTS = E > 0 ? 2 * SPLIT_LANES : SPLIT_LANES * H;  // lane slots of a row block of `partial`
  if constexpr (E > 0) {
    // the four row classes of a column pair sit in lanes i16, i16 + 16, i16 + 32, i16 + 48: summed in that fixed order
    // (the same bits run to run); row class 0 stores
#pragma unroll
    for (int e = 0; e < E; ++e) {
      double* out = a.partial + ((int64_t)by * SLOTS + SPLIT_LANES + e) * a.ld;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        d2 v = ex[e][c];
        v.x += __shfl_xor(v.x, 16, 64);
        v.y += __shfl_xor(v.y, 16, 64);
        v.x += __shfl_xor(v.x, 32, 64);
        v.y += __shfl_xor(v.y, 32, 64);
        const int col = col0 + 32 * c + 2 * i16;
        if (kq == 0 && col < ld) *reinterpret_cast<d2*>(out + col) = v;
      }
    }
  }
  if (i16 < SPLIT_LANES) {  // tile 2c+e holds columns col0 + 32 c + 2 i + e
#pragma unroll
    for (int h = 0; h < H; ++h) {
      double* out = a.partial + ((int64_t)by * SLOTS + SPLIT_LANES * h + i16) * a.ld;
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

static __global__ __launch_bounds__(XTR_WAVES * 64, 2) void xtr_mfma_kernel(SplitArgs a) { xtr_mfma_body<1>(a); }
// thirty-two lanes per read of X: both planes of R against every row loaded (sixteen result tiles = 128 registers)
static __global__ __launch_bounds__(XTR_WAVES * 64, 1) void xtr32_mfma_kernel(SplitArgs a) { xtr_mfma_body<2>(a); }
// the same product over the head of the rows only (solve_core's sample start): a name of its own, so that a profile's
// per-kernel statistics of xtr_mfma_kernel are those of passes over all of X
static __global__ __launch_bounds__(XTR_WAVES * 64, 2) void xtr_sample_kernel(SplitArgs a) { xtr_mfma_body<1>(a); }
static __global__ __launch_bounds__(XTR_WAVES * 64, 1) void xtr32_sample_kernel(SplitArgs a) { xtr_mfma_body<2>(a); }
// seventeen to twenty lanes: sixteen on the matrix cores, the others on the vector units (above); one workgroup per CU is
// the launch's geometry anyway, so the extra sums may take the registers of a second one (256 did not hold them: scratch)
static __global__ __launch_bounds__(XTR_WAVES * 64, 1) void xtr18_mfma_kernel(SplitArgs a) { xtr_mfma_body<1, 2>(a); }
static __global__ __launch_bounds__(XTR_WAVES * 64, 1) void xtr20_mfma_kernel(SplitArgs a) { xtr_mfma_body<1, 4>(a); }
static __global__ __launch_bounds__(XTR_WAVES * 64, 1) void xtr18_sample_kernel(SplitArgs a) { xtr_mfma_body<1, 2>(a); }
static __global__ __launch_bounds__(XTR_WAVES * 64, 1) void xtr20_sample_kernel(SplitArgs a) { xtr_mfma_body<1, 4>(a); }

// ---------------------------------------------------------------------------------------------
// The first half on the matrix cores as well: R[row][l] = w_l,row (x_row . z_l - y_row) for ALL sixteen lane
// slots in one read of X (rowdot_ring_kernel serves five per read).  The contraction runs over columns,
// which MFMA spreads over the four lane groups of a wavefront, so the operand that is contiguous along
// ROWS is needed: the column-major copy XT the working set keeps for its gathers.  D[i][j] += sum_k
// A[i][k] B[k][j] with i = lane slot, j = row, k = column: lane l = (j = l & 15, q = l >> 4) loads
// z[l & 15][c0 + 4q .. 4q + 3] once per 16 columns (32 bytes, rows of z are contiguous) and, for step
// m = 0..3, XT[c0 + 4q + m][rows 2j, 2j + 1] (16 bytes: 4 columns x 256 contiguous bytes per
// instruction); the two doubles feed two MFMAs (even / odd rows of a 32-row tile).  A wavefront carries
// XZ_T adjacent tiles at once so they share the loads of z (128 rows = 1 KiB contiguous per column); 4
// wavefronts cover 512 rows per round of the workgroup's row block (the same blocks as resid_ws_kernel:
// both write loss_partial[block][lane]).  Two 16-column batches in flight per wavefront (register double
// buffering, 256 VGPRs).  Measured at n = 100k, p = 5k: 0.74 ms (8 wavefronts x 2 tiles: 0.80 ms; tiles
// dealt round-robin instead of adjacent: 0.84 ms).
// ---------------------------------------------------------------------------------------------
constexpr int XZ_WAVES = 4;
constexpr int XZ_T = 4;

// One step of rowdot_mfma_kernel for one wavefront: NT adjacent tiles of 32 rows x this wavefront's g_n groups of 16
// columns.  xt0: tile 0, first column of the wavefront, this lane's row pair (tile t: + 32 ld t doubles); zp: this lane's
// four z values of the first group.  The partial products go to part[t][e][r][lane] (`mine` points at this lane).
// Straight-line steady state -- two batches of a group each, the second in flight while the first is multiplied -- with
// the odd batch peeled off, so that every wait is an exact count.
// H: halves of the lanes served by ONE read of the copy (a second A operand -- the second half's z -- against the same loads)
// E > 0 (with H = 1): a call of 16 + E lanes, E <= 4 -- as in xtr18 / xtr20_mfma_kernel the extra lanes ride on the VECTOR units
// against the loads the matrix cores' sixteen use: ex[e][t] += z_e[col] * XT[col][rows 2j, 2j + 1] for this lane's four
// columns of a group, the four column classes (q) summed by two shuffles at the end, the wavefronts' column quarters through
// `pe` (LDS: [e][tile][32 rows] per wavefront).  zpe[e]: this lane's four z values of extra lane e in the first group.
template <int NT, int H, int E = 0>
__device__ __forceinline__ void rowdot_step(const double* xt0, int64_t ld, const double* const (&zp)[H],
                                            const double* const (&zpe)[E > 0 ? E : 1], int g_n, double* mine, double* pe) {
  constexpr int EE = E > 0 ? E : 1;
  slm_d4 acc[H][NT][2];
#pragma unroll
  for (int h = 0; h < H; ++h)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[h][t][0] = acc[h][t][1] = slm_d4{0.0, 0.0, 0.0, 0.0};
  d2 ex[EE][NT];
#pragma unroll
  for (int e = 0; e < EE; ++e)
#pragma unroll
    for (int t = 0; t < NT; ++t) ex[e][t] = d2{0.0, 0.0};
  slm_d4 za[H], zb[H];
  slm_d4 zea[EE], zeb[EE];
  d2 xa[4][NT], xb[4][NT];
  auto load = [&](slm_d4(&zv)[H], slm_d4(&zev)[EE], d2(&xv)[4][NT], int g) {
#pragma unroll
    for (int h = 0; h < H; ++h) zv[h] = *reinterpret_cast<const slm_d4*>(zp[h] + 16 * g);
    if constexpr (E > 0) {
#pragma unroll
      for (int e = 0; e < E; ++e) zev[e] = *reinterpret_cast<const slm_d4*>(zpe[e] + 16 * g);
    }
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const d2* src = reinterpret_cast<const d2*>(xt0 + (((int64_t)t * ld + 16 * g + m) << 5));
        xv[m][t] = SLM_NT_LOADS ? __builtin_nontemporal_load(src) : *src;
      }
  };
  auto compute = [&](const slm_d4(&zv)[H], const slm_d4(&zev)[EE], d2(&xv)[4][NT]) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int h = 0; h < H; ++h) {
          acc[h][t][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(zv[h][m], xv[m][t].x, acc[h][t][0], 0, 0, 0);
          acc[h][t][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(zv[h][m], xv[m][t].y, acc[h][t][1], 0, 0, 0);
        }
        if constexpr (E > 0) {
#pragma unroll
          for (int e = 0; e < E; ++e) {
            ex[e][t].x = __builtin_fma(zev[e][m], xv[m][t].x, ex[e][t].x);
            ex[e][t].y = __builtin_fma(zev[e][m], xv[m][t].y, ex[e][t].y);
          }
        }
      }
  };
  if (g_n > 0) {
    load(za, zea, xa, 0);
    const int pairs = (g_n - 1) >> 1;
    for (int k = 0; k < pairs; ++k) {
      load(zb, zeb, xb, 2 * k + 1);
      compute(za, zea, xa);
      load(za, zea, xa, 2 * k + 2);
      compute(zb, zeb, xb);
    }
    if ((g_n - 1) & 1) {
      load(zb, zeb, xb, g_n - 1);
      compute(za, zea, xa);
      compute(zb, zeb, xb);
    } else {
      compute(za, zea, xa);
    }
  }
#pragma unroll
  for (int h = 0; h < H; ++h)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int r = 0; r < 4; ++r) mine[(((h * XZ_T + t) * 2 + e) * 4 + r) * 64] = acc[h][t][e][r];
  if constexpr (E > 0) {
    // the four column classes of a row pair sit in lanes j, j + 16, j + 32, j + 48: summed in that fixed order; class 0 stores
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int e = 0; e < E; ++e)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        d2 v = ex[e][t];
        v.x += __shfl_xor(v.x, 16, 64);
        v.y += __shfl_xor(v.y, 16, 64);
        v.x += __shfl_xor(v.x, 32, 64);
        v.y += __shfl_xor(v.y, 32, 64);
        if (lane < 16) *reinterpret_cast<d2*>(pe + (e * XZ_T + t) * 32 + 2 * lane) = v;
      }
  }
}


// H = 1: the sixteen lanes of half blockIdx.y (a launch per half: grid.y).  H = 2: BOTH halves on one read of the copy
// (grid.y = 1; a call of more than sixteen lanes whose second half needs residuals from X -- dense points of the model-Gram
// rounds -- used to read the 4 GB twice).
// E > 0 (H = 1, grid.y = 1): 16 + E lanes -- half 0 on the matrix cores, the first E lanes of half 1 on the vector units beside
// them (rowdot_step): seventeen to twenty lanes at the price of sixteen, as in xtr18 / xtr20_mfma_kernel.
// doubles of LDS the body needs: the wavefronts' loss sums, their partial products of one step, the extra lanes' sums
template <int H, int E = 0>
constexpr int rowdot_lds_doubles() {
  return XZ_WAVES * (E > 0 ? 2 : H) * SPLIT_LANES + H * XZ_WAVES * XZ_T * 2 * 4 * 64 + (E > 0 ? XZ_WAVES * E * XZ_T * 32 : 1);
}
// (lds: rowdot_lds_doubles<H, E>() doubles; bx / nbx: this workgroup's row block and their number; by: the half served when a
//  launch serves one half per grid row.  Round 6 ran this body and resid_mfma_body as the two halves of ONE grid: 6 us gained
//  per headline path, whose launch of this kernel returns at once -- and 0.35 ms lost per pass of config 3's group path, where
//  it works: under the 512-thread bound of the shared kernel its 256 registers per wavefront went to scratch memory.)
template <int H, int E = 0>
__device__ __forceinline__ void rowdot_mfma_body(SplitArgs& a, double* lds, int bx, int nbx, int by) {
  static_assert(SPLIT_LANES == 16 && SPLIT_RSTRIDE == 16, "lane slots are the 16 rows of the MFMA A operand");
  static_assert(E == 0 || (H == 1 && (E == 2 || E == 4)), "extras: the first lanes of half 1 beside the sixteen of half 0");
  constexpr int HZ = E > 0 ? 2 : H;   // halves whose lanes this launch serves (masks, the cold start, the loss sums)
  constexpr int EE = E > 0 ? E : 1;
  if (split_off(a)) return;
  const int half0 = (H == 1 && E == 0) ? by : 0;  // first half served here
  unsigned mask[HZ];
  bool any = false;
#pragma unroll
  for (int h = 0; h < HZ; ++h) {
    mask[h] = split_x_mask(a, half0 + h);
    any = any || mask[h] != 0u;
  }
  if (!any) return;
  const int LS = split_slots(a);
  constexpr int RS = HZ * SPLIT_LANES;  // red[wavefront][RS]
  double* red = lds;
  double* part = red + XZ_WAVES * RS;  // 64 KiB per half: the wavefronts' partial products of one step
  double* partE = part + H * XZ_WAVES * XZ_T * 2 * 4 * 64;  // the extra lanes' sums of a step: [wavefront][e][tile][32 rows]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t b = bx;
  const int64_t r0 = b * a.rows_base + (b < a.rows_rem ? b : a.rows_rem);
  const int64_t nrows = a.rows_base + (b < a.rows_rem ? 1 : 0);

  bool all_zero = a.ctl != nullptr;  // cold start: e = -y without reading X (as rowdot_ring_kernel)
  if (all_zero) {  // (lane l of every wavefront reads control block l: one round trip, not one per lane slot)
#pragma unroll
    for (int h = 0; h < HZ; ++h) {
      const int L0 = SPLIT_LANES * (half0 + h);
      const bool in = lane < SPLIT_LANES && L0 + lane < a.n_lanes;
      const bool moved = in && ((mask[h] >> lane) & 1u) && !a.ctl[in ? L0 + lane : 0].zzero;
      all_zero = all_zero && __ballot(moved) == 0ull;
    }
  }
  if (all_zero) {
    // thread = (row of a group of XZ_WAVES * 4, lane slot): a wavefront stores four whole rows of R, 512 contiguous
    // bytes (one thread per row wrote its sixteen slots one by one, sixty-four lines per store instruction: 18 us
    // for the 12.8 MB of the headline problem's first pass)
    const int l = tid & 15;
    const bool has_rw = a.rw != nullptr;
#pragma unroll
    for (int h = 0; h < HZ; ++h) {
      const int L0 = SPLIT_LANES * (half0 + h);
      double* Rh = a.R + (int64_t)(half0 + h) * a.r_plane;
      const bool on = ((mask[h] >> l) & 1u) != 0u;
      double ls = 0.0;
      const double* rwp = has_rw ? a.rw + (int64_t)(L0 + l) * a.rw_stride : a.y;  // (no row weights: any readable address)
      for (int64_t i0 = tid >> 4; i0 < nrows; i0 += 8 * XZ_WAVES * 4) {  // eight rows per round: their loads go out together
        double yv[8], mv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int64_t i = i0 + u * (XZ_WAVES * 4);
          const int64_t row = r0 + (i < nrows ? i : 0);
          yv[u] = a.y[row];
          mv[u] = rwp[row];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int64_t i = i0 + u * (XZ_WAVES * 4);
          if (on && i < nrows) {
            const double e = -yv[u], m = has_rw ? mv[u] : 1.0;
            Rh[(r0 + i) * SPLIT_RSTRIDE + l] = e * m;
            ls = __builtin_fma(e * m, e, ls);
          }
        }
      }
      ls += __shfl_xor(ls, 16, 64);
      ls += __shfl_xor(ls, 32, 64);
      if (lane < SPLIT_LANES) red[wave * RS + h * SPLIT_LANES + lane] = ls;
    }
    __syncthreads();
    if (tid < HZ * SPLIT_LANES && ((mask[tid / SPLIT_LANES] >> (tid % SPLIT_LANES)) & 1u)) {
      double t = 0.0;
      for (int w2 = 0; w2 < XZ_WAVES; ++w2) t += red[w2 * RS + tid];
      a.loss_partial[b * LS + SPLIT_LANES * half0 + tid] = t;
    }
    return;
  }

  // Rows in tiles of 32 (a d2 load = two adjacent rows, sixteen lanes j = one 256-byte segment of a column of XT), the
  // tiles dealt to the workgroups in contiguous, tile-aligned runs: every segment starts on a 256-byte boundary and no
  // workgroup reads past its last tile.  (Until round 3 the blocks were the row-count-balanced ones of the other
  // residual kernels -- 390 or 391 rows at n = 100k -- rounded OUT to 13 tiles from an arbitrary even row: PMC
  // FETCH_SIZE 4.69 GB per launch for the 4.01 GB of XT, 0.75 ms at the 6.3 TB/s the memory system sustains;
  // profiles/r03a_rowdot_counters.json.)
  const int j = lane & 15, q = lane >> 4;
  const int64_t tiles_all = (a.n + 31) >> 5;
  const int64_t tb = tiles_all / nbx, tr = tiles_all % nbx;
  const int64_t t_lo = b * tb + (b < tr ? b : tr);
  const int T = (int)(tb + (b < tr ? 1 : 0));
  // A step = up to XZ_T adjacent tiles, taken by ALL wavefronts together: wavefront w contracts ITS quarter of the
  // columns for every tile of the step, the four partial products meet in LDS and wavefront t finishes tile t.  (One
  // wavefront per group of tiles, all columns -- the first layout -- left the wavefronts of a 13-tile block with 4, 4,
  // 4 and 1 tiles; here a block's time is proportional to its tiles, and z is read once per step, not once per
  // wavefront.)  Steps of equal size (13 tiles: 4 + 3 + 3 + 3), so that no step runs with a single tile's loads in flight.
  const int nsteps = (T + XZ_T - 1) / XZ_T;
  const int ngroups = (int)(a.ld >> 4);
  const int gb = ngroups / XZ_WAVES, gr = ngroups % XZ_WAVES;
  const int g_lo = wave * gb + (wave < gr ? wave : gr);
  const int g_n = gb + (wave < gr ? 1 : 0);
  const double* zp[H];
#pragma unroll
  for (int h = 0; h < H; ++h) {
    const int L = SPLIT_LANES * (half0 + h) + j;
    zp[h] = a.z + (int64_t)(L < a.n_lanes ? L : a.n_lanes - 1) * a.ld + 4 * q + 16 * (int64_t)g_lo;
  }
  const double* zpe[EE];  // (E > 0) extra lane e = lane 16 + e of the call
#pragma unroll
  for (int e = 0; e < EE; ++e) {
    const int L = SPLIT_LANES + e;
    zpe[e] = a.z + (int64_t)(L < a.n_lanes ? L : a.n_lanes - 1) * a.ld + 4 * q + 16 * (int64_t)g_lo;
  }
  double loss[H][4];  // of lane slots q, q + 4, q + 8, q + 12 over this lane's rows
#pragma unroll
  for (int h = 0; h < H; ++h)
#pragma unroll
    for (int r = 0; r < 4; ++r) loss[h][r] = 0.0;
  double lossE = 0.0;  // (E > 0, q < E) of extra lane q over this lane's rows
  int t_at = 0;
  for (int st = 0; st < nsteps; ++st) {
    const int nt = __builtin_amdgcn_readfirstlane(T / nsteps + (st < T % nsteps ? 1 : 0));
    const int64_t row_s = 32 * (t_lo + t_at);  // first row of the step
    // (one instantiation per number of tiles: with `if (t < nt)` around the loads and products of a tile the compiler
    //  could not count the loads in flight and put s_waitcnt vmcnt(0) before every pair of MFMAs -- the batch just
    //  asked for included: no overlap of loads and products at all)
    const double* xt0 = a.XT + (((t_lo + t_at) * a.ld + 4 * q + 16 * (int64_t)g_lo) << 5) + 2 * j;
    double* mine = part + (size_t)wave * (H * XZ_T * 2 * 4 * 64) + lane;
    double* pe = partE + (E > 0 ? (size_t)wave * (E * XZ_T * 32) : 0);
    switch (nt) {
      case 1: rowdot_step<1, H, E>(xt0, a.ld, zp, zpe, g_n, mine, pe); break;
      case 2: rowdot_step<2, H, E>(xt0, a.ld, zp, zpe, g_n, mine, pe); break;
      case 3: rowdot_step<3, H, E>(xt0, a.ld, zp, zpe, g_n, mine, pe); break;
      default: rowdot_step<4, H, E>(xt0, a.ld, zp, zpe, g_n, mine, pe); break;
    }
    __syncthreads();
    if (wave < nt) {  // wavefront t finishes tile t: result register r of lane l is lane slot (l >> 4) + 4 r, row 2 (l & 15) + e
      const int t = wave;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int64_t row = row_s + 32 * (int64_t)t + 2 * j + e;
        const bool in = row < a.n;
        const double yi = a.y[in ? row : 0];
#pragma unroll
        for (int h = 0; h < H; ++h) {
          const int L0 = SPLIT_LANES * (half0 + h);
          double* Rh = a.R + (int64_t)(half0 + h) * a.r_plane;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int l = q + 4 * r;
            double v = 0.0;
#pragma unroll
            for (int w2 = 0; w2 < XZ_WAVES; ++w2) v += part[(size_t)w2 * (H * XZ_T * 2 * 4 * 64) + (((h * XZ_T + t) * 2 + e) * 4 + r) * 64 + lane];
            if (in && ((mask[h] >> l) & 1u)) {
              const double m = a.rw ? a.rw[(int64_t)(L0 + l) * a.rw_stride + row] : 1.0;
              const double err = v - yi;
              const double res = err * m;
              Rh[row * SPLIT_RSTRIDE + l] = res;
              loss[h][r] = __builtin_fma(res, err, loss[h][r]);
            }
          }
        }
        if constexpr (E > 0) {  // lane group q < E finishes extra lane q of the tile: plane 1 of R, slot q
          if (q < E && in && ((mask[1] >> q) & 1u)) {
            double v = 0.0;
#pragma unroll
            for (int w2 = 0; w2 < XZ_WAVES; ++w2) v += partE[(size_t)w2 * (E * XZ_T * 32) + (q * XZ_T + t) * 32 + 2 * j + e];
            const double m = a.rw ? a.rw[(int64_t)(SPLIT_LANES + q) * a.rw_stride + row] : 1.0;
            const double err = v - yi;
            const double res = err * m;
            a.R[a.r_plane + row * SPLIT_RSTRIDE + q] = res;
            lossE = __builtin_fma(res, err, lossE);
          }
        }
      }
    }
    __syncthreads();  // (the next step overwrites the partial products)
    t_at += nt;
  }
#pragma unroll
  for (int h = 0; h < H; ++h)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double t = loss[h][r];
#pragma unroll
      for (int off = 8; off >= 1; off >>= 1) t += __shfl_xor(t, off, 64);
      if (j == 0) red[wave * RS + h * SPLIT_LANES + q + 4 * r] = t;
    }
  if constexpr (E > 0) {
    double t = lossE;
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) t += __shfl_xor(t, off, 64);
    if (j == 0) red[wave * RS + SPLIT_LANES + q] = t;  // (q >= E: zero; slots beyond 16 + 3 are never read)
  }
  __syncthreads();
  if (tid < HZ * SPLIT_LANES && ((mask[tid / SPLIT_LANES] >> (tid % SPLIT_LANES)) & 1u)) {
    double t = 0.0;
    for (int w2 = 0; w2 < XZ_WAVES; ++w2) t += red[w2 * RS + tid];
    a.loss_partial[b * LS + SPLIT_LANES * half0 + tid] = t;
  }
}

#define SLM_ROWDOT_KERNEL(name, H, E)                                                                  \
  static __global__ __launch_bounds__(XZ_WAVES * 64) void name(SplitArgs a) {                          \
    __shared__ double lds[rowdot_lds_doubles<H, E>()];                                                 \
    rowdot_mfma_body<H, E>(a, lds, (int)blockIdx.x, (int)gridDim.x, (int)blockIdx.y);                  \
  }
SLM_ROWDOT_KERNEL(rowdot_mfma_kernel, 1, 0)
SLM_ROWDOT_KERNEL(rowdot32_mfma_kernel, 2, 0)
SLM_ROWDOT_KERNEL(rowdot18_mfma_kernel, 1, 2)
SLM_ROWDOT_KERNEL(rowdot20_mfma_kernel, 1, 4)
#undef SLM_ROWDOT_KERNEL

// ---------------------------------------------------------------------------------------------
// The residuals from the gathered columns on the matrix cores: R[row][l] = w_l,row (XW_row . zW_l - y_row)
// is the (n x K)(K x 16) product XW zW.  D[i][j] += sum_k A[i][k] B[k][j] with i = row, j = lane slot,
// k = position in W: lane l = (i = l & 15, q = l >> 4) loads XW[row_i][16 g + 4 q .. 4 q + 3] (32 bytes: a
// load instruction is 16 rows x one full 128-byte line) and its four doubles feed four MFMAs, step m
// contracting positions 16 g + 4 q + m against zW[16 g + 4 q + m][j] from the LDS image.  Result register
// r of lane l is row (l >> 4) + 4 r, lane slot l & 15: a row of R is stored as one 128-byte segment.
// (resid_ws_kernel walks a row per thread: every lane of a load is its own cache line, 83 us at K = 268.)
// ---------------------------------------------------------------------------------------------
constexpr int RM_WAVES = 8;   // (4: 30.6 us per pass on the headline path, 8 or 16: 26.6 us -- six 16-row tiles per wavefront were a chain of six)
constexpr int RM_U = 4;  // 16-position groups per batch (two batches in flight)

// H = 1: the sixteen lanes of half blockIdx.y.  H = 2: BOTH halves of a call of more than sixteen lanes on one read of the
// gathered columns (a launch per half read them twice: 66 us against 40 on the headline path's eighteen lanes).
template <int H>
constexpr int resid_lds_doubles() { return H * WS_KCAP * SPLIT_LANES + RM_WAVES * H * SPLIT_LANES; }
template <int H>
__device__ __forceinline__ void resid_mfma_body(SplitArgs& a, double* lds, int bx, int by) {
  static_assert(SPLIT_LANES == 16 && SPLIT_RSTRIDE == 16, "lane slots are the 16 columns of the MFMA B operand");
  if (split_off(a)) return;
  const int half0 = H == 1 ? by : 0;  // (grid.y: the halves of the call, one launch each; H = 2: both here)
  unsigned mask[H];
  bool any = false;
#pragma unroll
  for (int h = 0; h < H; ++h) {
    mask[h] = split_ws_mask(a, half0 + h);
    any = any || mask[h] != 0u;
  }
  if (!any) return;
  double* zw = lds;                                 // [H][WS_KCAP][SPLIT_LANES]: 64 KiB per half
  double* lsum = lds + H * WS_KCAP * SPLIT_LANES;   // [RM_WAVES][H * SPLIT_LANES]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int K = a.ws->K;  // multiple of 16; positions >= Kreal hold zero columns
  const int64_t b = bx;
  const int64_t r0 = b * a.rows_base + (b < a.rows_rem ? b : a.rows_rem);
  const int64_t nrows = a.rows_base + (b < a.rows_rem ? 1 : 0);
  const int64_t rend = r0 + nrows;
#pragma unroll
  for (int h = 0; h < H; ++h) {
    const int L0 = SPLIT_LANES * (half0 + h);
    for (int e = tid; e < K * SPLIT_LANES; e += RM_WAVES * 64) {
      const int k = e >> 4, l = e & 15;
      const int j = a.idx[k];
      zw[(h * WS_KCAP + k) * SPLIT_LANES + l] = (j >= 0 && L0 + l < a.n_lanes) ? a.z[(int64_t)(L0 + l) * a.ld + j] : 0.0;
    }
  }
  __syncthreads();
  const int i16 = lane & 15, q = lane >> 4;
  const bool has_rw = a.rw != nullptr;
  const double* rwp = has_rw ? a.rw : a.y;  // (no row weights: any readable address)
  const int ngroups = K >> 4;
  const int ntiles = (int)((nrows + 15) >> 4);
  double loss[H];  // of lane slot i16 (of each half) over this lane's rows
#pragma unroll
  for (int h = 0; h < H; ++h) loss[h] = 0.0;
  for (int t = wave; t < ntiles; t += RM_WAVES) {
    const int64_t row0 = r0 + 16 * (int64_t)t;
    const int64_t rl = row0 + i16 < rend ? row0 + i16 : rend - 1;  // (rows past the block are computed and dropped)
    const double* xp = a.XW + rl * WS_KCAP + 4 * q;
    slm_d4 acc[H];
#pragma unroll
    for (int h = 0; h < H; ++h) acc[h] = slm_d4{0.0, 0.0, 0.0, 0.0};
    slm_d4 xa[RM_U], xb[RM_U];
    // targets and row weights of the four rows this lane finishes: asked for now, with the first columns (at the end
    // of the tile each pair was a round trip of its own before the row could be stored)
    double yv[4], mv[H][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t row = row0 + q + 4 * r;
      const int64_t rc = row < rend ? row : rend - 1;
      yv[r] = a.y[rc];
#pragma unroll
      for (int h = 0; h < H; ++h) {
        const int L = SPLIT_LANES * (half0 + h) + i16;
        mv[h][r] = rwp[has_rw ? (int64_t)(L < a.n_lanes ? L : 0) * a.rw_stride + rc : 0];
      }
    }
    // (loads without conditions -- a group beyond K is read from the last one and not used: with the loads under
    //  `if (g0 + u < ngroups)` the compiler could not count them and drained the queue before every batch of MFMAs)
    auto load = [&](slm_d4(&xv)[RM_U], int g0) {
#pragma unroll
      for (int u = 0; u < RM_U; ++u) xv[u] = *reinterpret_cast<const slm_d4*>(xp + 16 * min(g0 + u, ngroups - 1));
    };
    auto compute = [&](const slm_d4(&xv)[RM_U], int g0) {
#pragma unroll
      for (int u = 0; u < RM_U; ++u)
        if (g0 + u < ngroups) {
#pragma unroll
          for (int h = 0; h < H; ++h) {
            const double* zr = zw + (h * WS_KCAP + 16 * (g0 + u) + 4 * q) * SPLIT_LANES + i16;
#pragma unroll
            for (int m = 0; m < 4; ++m)
              acc[h] = __builtin_amdgcn_mfma_f64_16x16x4f64(xv[u][m], zr[m * SPLIT_LANES], acc[h], 0, 0, 0);
          }
        }
    };
    load(xa, 0);
    for (int g0 = 0; g0 < ngroups; g0 += 2 * RM_U) {
      load(xb, g0 + RM_U);
      compute(xa, g0);
      load(xa, g0 + 2 * RM_U);
      compute(xb, g0 + RM_U);
    }
#pragma unroll
    for (int h = 0; h < H; ++h)
      if ((mask[h] >> i16) & 1u) {
        double* Rh = a.R + (int64_t)(half0 + h) * a.r_plane;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t row = row0 + q + 4 * r;
          if (row < rend) {
            const double m = has_rw ? mv[h][r] : 1.0;
            const double err = acc[h][r] - yv[r];
            const double res = err * m;
            Rh[row * SPLIT_RSTRIDE + i16] = res;
            loss[h] = __builtin_fma(res, err, loss[h]);
          }
        }
      }
  }
#pragma unroll
  for (int h = 0; h < H; ++h) {
    loss[h] += __shfl_xor(loss[h], 16, 64);
    loss[h] += __shfl_xor(loss[h], 32, 64);
    if (lane < SPLIT_LANES) lsum[wave * (H * SPLIT_LANES) + h * SPLIT_LANES + lane] = loss[h];
  }
  __syncthreads();
  if (tid < H * SPLIT_LANES && ((mask[tid / SPLIT_LANES] >> (tid % SPLIT_LANES)) & 1u)) {
    double t = 0.0;
    for (int w2 = 0; w2 < RM_WAVES; ++w2) t += lsum[w2 * (H * SPLIT_LANES) + tid];
    a.loss_partial[b * split_slots(a) + SPLIT_LANES * half0 + tid] = t;
  }
}

static __global__ __launch_bounds__(RM_WAVES * 64) void resid_mfma_kernel(SplitArgs a) {
  __shared__ double lds[resid_lds_doubles<1>()];
  resid_mfma_body<1>(a, lds, (int)blockIdx.x, (int)blockIdx.y);
}
static __global__ __launch_bounds__(RM_WAVES * 64) void resid32_mfma_kernel(SplitArgs a) {
  __shared__ double lds[resid_lds_doubles<2>()];
  resid_mfma_body<2>(a, lds, (int)blockIdx.x, 0);
}

}  // namespace slm

// Model Gram: what the lanes that outgrow the working set iterate on between two passes over X.
//
// The working set (ws_kernels.hpp) holds 512 columns: on them the Gram is exact in fp64 and a path point costs one pass.
// A lane whose solution has more non-zeros than that used to finish with plain proximal-gradient steps, every step two
// reads of X (rowdot_mfma + xtr_mfma): 22-56 passes for the dense end of a 50-alpha path.  The quadratic model such a lane
// needs is the WHOLE Gram G = X^T W X / n -- 36 ms of fp64 matrix-core time at n = 100k, p = 5k, as much as the passes it
// would save.  But the model only has to PROPOSE the next point: every proposal is verified by a pass over X in fp64 under
// the unchanged acceptance test and stopping rule of fista_tail_kernel, exactly like a point of the working set's model
// solver.  So the model Gram is built where the chip is twenty times faster: X is scaled column by column, rounded to
// fp16 and multiplied on v_mfma_f32_32x32x16_f16 (fp32 accumulation over chunks of rows, the chunks summed in fp64) --
// G~ = G + E with ||E|| ~ 1e-4 ||G|| on a standardised design.  An outer round is then
//     pass over X:  g0 = grad f(z0) exactly                      (the gradient every decision is taken on)
//     inner:        x ~ argmin g0.(x - z0) + 1/2 (x - z0)^T G~ (x - z0) + penalty(x)   by proximal-gradient steps, each
//                   ONE read of G~ for all sixteen lanes (cov_gz_mfma_kernel's loop: 200 MB, not 2 x 4 GB) + mg_step_kernel
//     next pass:    verifies x (the tail kernel's test on the true objective) and is the expansion point of the next round
// and the error contracts by ||G^-1 E|| ~ 1e-4 per round: two rounds per path point from a start sixteen points up the
// path, whatever the number of non-zeros.  A proposal the true objective rejects costs the lane nothing but the round
// (the tail kernel falls back on its own step); two rejections and the lane finishes that path point with plain steps.
//
// Reference counterpart: none -- cvxpy hands the whole problem to an interior-point solver
// (src/sparselm/model/_base.py:512-519); the construction is an inexact proximal Newton method with a low-precision
// Hessian and exact gradients (iterative refinement), restated for the engine's path state machine.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cov_kernels.hpp"
#include "host_logic.hpp"
#include "split_kernels.hpp"
#include "tail_kernels.hpp"
#include "ws_kernels.hpp"

namespace slm {

typedef _Float16 mg_h8 __attribute__((ext_vector_type(8)));
typedef float mg_f16v __attribute__((ext_vector_type(16)));
typedef unsigned int mg_u4 __attribute__((ext_vector_type(4)));

constexpr int MG_TILE = 128;               // result tile of a workgroup (four wavefronts, 64 x 64 each)
constexpr int MG_BK = 64;                  // rows of X per step of the product
constexpr int MG_LDS_STRIDE = MG_BK + 8;   // halves per LDS row: 144 bytes -- sixteen rows fall on sixteen different 16-byte slots
constexpr int MG_MAX_LD = 16384;           // the Gram is 4 ld^2 bytes (fp32): 1 GB at most; beyond, slm_dataset_model_gram refuses
                                           // (SLM_ERR_UNSUPPORTED) and solves go on with plain steps: tests/test_width_limits_gpu.py
constexpr int MG_CHUNK_ROWS = 6400;        // rows per fp32 accumulation (chunks are summed in fp64): 100 steps of MG_BK
constexpr int MG_ROUNDS_PER_POINT = 12;    // rounds a lane may spend on one path point before it is left to plain steps
constexpr int MG_REJECT_LIMIT = 2;         // proposals for one path point the true objective may reject before the point is left to plain steps
constexpr double MG_ETA = 5e-5;            // inner stop relative to the move ||x - z0||: below the model's own error (~1.5e-4 of the move)

struct MgLane {
  int32_t active;     // the inner iteration of this round runs for the lane
  int32_t settled;    // ... and has met its tolerance
  int32_t iters;      // inner iterations of this round
  int32_t spectral;   // the iteration is still in its opening spectral steps
  int32_t have_prev;
  int32_t proposed;   // a proposal of this lane awaits the verdict of the next pass
  int32_t rej_saved;  // PathCtl::rejects when it was made: the verdict on a proposal is read from the counter and does not
                      // count towards the spectral scheme's own fallback (BB_REJECT_LIMIT), mg_finish_kernel / mg_begin_kernel
  int32_t rejected;   // proposals for the lane's current path point that the true objective rejected
  int32_t bad;        // the model went astray (a non-finite iterate): the lane is left to plain steps for the rest of the solve
  int32_t point;      // path point of the lane's last round ...
  int32_t rounds_pt;  // ... and rounds spent on it
  int32_t rq_n;
  double L, Ls, t, rq_min;
};

struct MgCtl {
  int32_t rounds;       // (lane, round) pairs in which the inner iteration ran
  int32_t inner_iters;  // their iterations
  int32_t settled;      // ... of which met the inner tolerance
  int32_t rejected;     // proposals the true objective rejected
  int32_t most_iters;   // most inner iterations any lane needed in the last round (the host sizes the next round's queue)
  int32_t pad_[3];
  MgLane lane[SLM_MAX_LANES];
};

// ---------------------------------------------------------------------------------------------
// building the model Gram
// ---------------------------------------------------------------------------------------------
// (1) column scales: cmax[j] = max_i |X[i][j]| as the bits of a non-negative double -- integer maxima commute, so the
// result does not depend on the order of the atomics.  grid: row blocks; thread t owns the column pairs 2 t + 512 c.
static __global__ __launch_bounds__(256) void mg_colmax_kernel(const double* __restrict__ X, int64_t n, int64_t ld, int64_t rows_per_block,
                                                               unsigned long long* __restrict__ cmax) {
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < n ? r0 + rows_per_block : n;
  for (int64_t c0 = 2 * (int64_t)threadIdx.x; c0 < ld; c0 += 512) {
    double m0 = 0.0, m1 = 0.0;
    const double* xp = X + r0 * ld + c0;
    int64_t i = r0;
    for (; i + 4 <= r1; i += 4) {
      d2 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const d2*>(xp + (int64_t)u * ld);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        m0 = fmax(m0, fabs(v[u].x));
        m1 = fmax(m1, fabs(v[u].y));
      }
      xp += 4 * ld;
    }
    for (; i < r1; ++i) {
      const d2 v = *reinterpret_cast<const d2*>(xp);
      m0 = fmax(m0, fabs(v.x));
      m1 = fmax(m1, fabs(v.y));
      xp += ld;
    }
    if (m0 > 0.0) atomicMax(&cmax[c0], (unsigned long long)__double_as_longlong(m0));
    if (m1 > 0.0) atomicMax(&cmax[c0 + 1], (unsigned long long)__double_as_longlong(m1));
  }
}

// the power of two a column is divided by before rounding: |x| / 2^e < 1
__device__ __forceinline__ int mg_scale_exp(unsigned long long bits) {
  const double m = __longlong_as_double((long long)bits);
  if (!(m > 0.0) || !isfinite(m)) return 0;
  int e;
  (void)frexp(m, &e);  // m = f 2^e, 1/2 <= f < 1
  return e;
}

// (2) the fp16 operand: XTh[j][i] = fp16(sqrt(w_i) X[i][j] / 2^e_j), column-major ([p_pad][n_pad]: the rows of X are the
// contraction index of the product and have to be contiguous for both of its operands), from the column-major tiled copy
// XT (XT[((i >> 5) ld + j) 32 + (i & 31)]: a thread reads 8 consecutive rows of one column, 64 bytes, and writes their 16).
// grid (n_pad / 64, p_pad / 256); thread (column jl = tid >> 3 of a group of 32, eighth q = tid & 7 of a pair of row tiles).
struct MgConvArgs {
  const double* XT;
  const unsigned long long* cmax;
  const double* rw;  // row weights of the dataset (nullptr: ones)
  int64_t n, ld, row_tiles;
  _Float16* XTh;
  int64_t n_pad, p_pad;
};
static __global__ __launch_bounds__(256) void mg_convert_kernel(MgConvArgs a) {
  const int jl = threadIdx.x >> 3, q = threadIdx.x & 7;
  const int64_t rt = 2 * (int64_t)blockIdx.x + (q >> 2);
  const int r8 = 8 * (q & 3);
  const int64_t i0 = rt * 32 + r8;
  double sw[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) sw[u] = (a.rw != nullptr && i0 + u < a.n) ? sqrt(a.rw[i0 + u]) : 1.0;
#pragma unroll 2
  for (int g = 0; g < 8; ++g) {
    const int64_t j = (int64_t)blockIdx.y * 256 + 32 * g + jl;
    if (j >= a.p_pad) break;
    mg_h8 h;
#pragma unroll
    for (int u = 0; u < 8; ++u) h[u] = (_Float16)0.0f;
    if (j < a.ld && rt < a.row_tiles) {
      const double* src = a.XT + ((rt * a.ld + j) << 5) + r8;
      d2 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const d2*>(src + 2 * u);
      const double inv = ldexp(1.0, -mg_scale_exp(a.cmax[j]));
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const double x0 = fmin(fmax(v[u].x * inv * sw[2 * u], -65504.0), 65504.0);
        const double x1 = fmin(fmax(v[u].y * inv * sw[2 * u + 1], -65504.0), 65504.0);
        h[2 * u] = (_Float16)(float)x0;
        h[2 * u + 1] = (_Float16)(float)x1;
      }
    }
    if (i0 < a.n_pad) *reinterpret_cast<mg_h8*>(a.XTh + j * a.n_pad + i0) = h;
  }
}

// (3) the product: P[chunk][tile] = M_I M_J^T over the rows of one chunk, M = XTh, for the 128 x 128 tiles (I, J <= I) of the
// lower triangle.  Four wavefronts, each a 64 x 64 corner as 2 x 2 tiles of v_mfma_f32_32x32x16_f16; both operands come
// through LDS in steps of MG_BK rows (128 bytes of every one of the tile's 128 columns: full lines), the loads of the next
// step are in flight while the current one is multiplied.  The fragment a lane needs -- 8 consecutive rows of one column --
// is 16 bytes of an LDS row; rows are 144 bytes apart, so the sixteen lanes of a ds_read_b128 group hit sixteen slots.
// Work items are numbered chunk-major and dealt to the XCDs in contiguous ranges: the workgroups of an XCD work on one
// chunk of rows at a time, whose 128-byte pieces of all columns (640 KB at p = 5k) stay in its L2.
struct MgSyrkArgs {
  const _Float16* M;  // [p_pad][n_pad]
  int64_t n_pad;
  int64_t k_chunk;    // rows per chunk (multiple of MG_BK)
  int n_tiles;        // tiles of the lower triangle
  int n_chunks;
  float* P;           // [n_chunks][n_tiles][128 x 128]
};
static __global__ __launch_bounds__(256) void mg_syrk_f16_kernel(MgSyrkArgs a) {
  __shared__ __attribute__((aligned(16))) _Float16 As[MG_TILE * MG_LDS_STRIDE];
  __shared__ __attribute__((aligned(16))) _Float16 Bs[MG_TILE * MG_LDS_STRIDE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int64_t m;
  {
    const int64_t total = gridDim.x, lin = blockIdx.x;
    const int64_t xcd = lin & 7, slot = lin >> 3, base = total >> 3, rem = total & 7;
    m = xcd * base + (xcd < rem ? xcd : rem) + slot;
  }
  const int chunk = (int)(m / a.n_tiles), tile = (int)(m - (int64_t)chunk * a.n_tiles);
  int I, J;
  slm_host::triangle_tile_fast(tile, &I, &J);  // (host_logic.hpp: tile t = (I, J <= I) of the lower triangle)
  const int64_t k0 = (int64_t)chunk * a.k_chunk;
  const int64_t k1 = k0 + a.k_chunk < a.n_pad ? k0 + a.k_chunk : a.n_pad;
  // staging: thread -> (row (tid >> 3) + 32 u, 16-byte piece tid & 7)
  const int srow = tid >> 3, spc = tid & 7;
  const _Float16* Ag = a.M + ((int64_t)I * MG_TILE + srow) * a.n_pad + 8 * spc;
  const _Float16* Bg = a.M + ((int64_t)J * MG_TILE + srow) * a.n_pad + 8 * spc;
  mg_u4 ra[4], rb[4];
  auto gload = [&](int64_t k) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      ra[u] = *reinterpret_cast<const mg_u4*>(Ag + (int64_t)(32 * u) * a.n_pad + k);
      rb[u] = *reinterpret_cast<const mg_u4*>(Bg + (int64_t)(32 * u) * a.n_pad + k);
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      *reinterpret_cast<mg_u4*>(As + (srow + 32 * u) * MG_LDS_STRIDE + 8 * spc) = ra[u];
      *reinterpret_cast<mg_u4*>(Bs + (srow + 32 * u) * MG_LDS_STRIDE + 8 * spc) = rb[u];
    }
  };
  const int wr = wave >> 1, wc = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  mg_f16v acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.0f;
  if (k0 < k1) {
    gload(k0);
    lstore();
    __syncthreads();
    for (int64_t k = k0; k < k1; k += MG_BK) {
      const bool more = k + MG_BK < k1;
      if (more) gload(k + MG_BK);
#pragma unroll
      for (int kk = 0; kk < MG_BK / 16; ++kk) {
        mg_h8 af[2], bf[2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
          af[mi] = *reinterpret_cast<const mg_h8*>(As + (64 * wr + 32 * mi + r) * MG_LDS_STRIDE + 16 * kk + 8 * h);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          bf[ni] = *reinterpret_cast<const mg_h8*>(Bs + (64 * wc + 32 * ni + r) * MG_LDS_STRIDE + 16 * kk + 8 * h);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[mi], bf[ni], acc[mi][ni], 0, 0, 0);
      }
      __syncthreads();
      if (more) {
        lstore();
        __syncthreads();
      }
    }
  }
  float* out = a.P + ((int64_t)chunk * a.n_tiles + tile) * (MG_TILE * MG_TILE);
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = 64 * wr + 32 * mi + (e & 3) + 8 * (e >> 2) + 4 * h;
        const int col = 64 * wc + 32 * ni + r;
        out[row * MG_TILE + col] = acc[mi][ni][e];
      }
}

// The same product with the operands brought into LDS by the DMA path (global_load_lds_dwordx4: no staging registers, no
// ds_write -- the 16-byte LDS stores of the kernel above cost more LDS cycles than its fragment reads), two buffers, the
// next step's tiles in flight while the current one is multiplied, one barrier per step.  A DMA instruction fills 1 KiB of
// LDS in lane order, so the tiles are stored unpadded ([128 rows][64 halves]) and the 16-byte pieces of a row are swizzled
// instead -- piece c of row R sits in slot c ^ ((R >> 1) & 7), applied to the SOURCE address of the load and to the
// fragment read -- which puts the sixteen rows of a ds_read_b128 group on sixteen different slots again.
static __global__ __launch_bounds__(256) void mg_syrk_f16_dma_kernel(MgSyrkArgs a) {
  __shared__ __attribute__((aligned(1024))) _Float16 lds[2][2][MG_TILE * MG_BK];  // [buffer][operand]: 64 KiB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int64_t m;
  {
    const int64_t total = gridDim.x, lin = blockIdx.x;
    const int64_t xcd = lin & 7, slot = lin >> 3, base = total >> 3, rem = total & 7;
    m = xcd * base + (xcd < rem ? xcd : rem) + slot;
  }
  const int chunk = (int)(m / a.n_tiles), tile = (int)(m - (int64_t)chunk * a.n_tiles);
  int I, J;
  slm_host::triangle_tile_fast(tile, &I, &J);  // (host_logic.hpp: tile t = (I, J <= I) of the lower triangle)
  const int64_t k0 = (int64_t)chunk * a.k_chunk;
  const int64_t k1 = k0 + a.k_chunk < a.n_pad ? k0 + a.k_chunk : a.n_pad;
  // DMA: wave w, instruction q fills piece P = 4 w + q (rows 8 P ... 8 P + 7); lane -> (row 8 P + (lane >> 3), slot lane & 7)
  const _Float16* srcA[4];
  const _Float16* srcB[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = 8 * (4 * wave + q) + (lane >> 3);
    const int piece = (lane & 7) ^ ((row >> 1) & 7);
    srcA[q] = a.M + ((int64_t)I * MG_TILE + row) * a.n_pad + 8 * piece;
    srcB[q] = a.M + ((int64_t)J * MG_TILE + row) * a.n_pad + 8 * piece;
  }
  auto dma = [&](int buf, int64_t k) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcA[q] + k),
                                       (__attribute__((address_space(3))) void*)(&lds[buf][0][(4 * wave + q) * 512]), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcB[q] + k),
                                       (__attribute__((address_space(3))) void*)(&lds[buf][1][(4 * wave + q) * 512]), 16, 0, 0);
    }
  };
  const int wr = wave >> 1, wc = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  mg_f16v acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.0f;
  // fragment addresses (halves) inside an operand's tile, per 16-row step kk: row R, piece 2 kk + h in slot (2 kk + h) ^ ((R >> 1) & 7)
  int ra[2], rb[2], sa[2], sb[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int RA = 64 * wr + 32 * t + r, RB = 64 * wc + 32 * t + r;
    ra[t] = RA * MG_BK; sa[t] = (RA >> 1) & 7;
    rb[t] = RB * MG_BK; sb[t] = (RB >> 1) & 7;
  }
  if (k0 < k1) {
    dma(0, k0);
    __syncthreads();
    int cur = 0;
    for (int64_t k = k0; k < k1; k += MG_BK) {
      if (k + MG_BK < k1) dma(cur ^ 1, k + MG_BK);
      const _Float16* As = lds[cur][0];
      const _Float16* Bs = lds[cur][1];
#pragma unroll
      for (int kk = 0; kk < MG_BK / 16; ++kk) {
        mg_h8 af[2], bf[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          af[t] = *reinterpret_cast<const mg_h8*>(As + ra[t] + 8 * ((2 * kk + h) ^ sa[t]));
          bf[t] = *reinterpret_cast<const mg_h8*>(Bs + rb[t] + 8 * ((2 * kk + h) ^ sb[t]));
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[mi], bf[ni], acc[mi][ni], 0, 0, 0);
      }
      __syncthreads();  // (drains the DMA of the next step, and every wavefront is done with this step's buffer)
      cur ^= 1;
    }
  }
  float* out = a.P + ((int64_t)chunk * a.n_tiles + tile) * (MG_TILE * MG_TILE);
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = 64 * wr + 32 * mi + (e & 3) + 8 * (e >> 2) + 4 * h;
        const int col = 64 * wc + 32 * ni + r;
        out[row * MG_TILE + col] = acc[mi][ni][e];
      }
}

// (4) G~[i][j] = 2^(e_i + e_j) / n_eff * sum over the chunks (in order, in fp64) of P[chunk][tile][i][j], and its mirror.
// grid (n_tiles, 16): a 32 x 32 corner of the tile per workgroup; of a diagonal tile only the lower half is taken (and
// mirrored), so every entry of G~ is written once and the matrix is symmetric to the bit.
struct MgReduceArgs {
  const float* P;
  const unsigned long long* cmax;
  int n_tiles, n_chunks;
  int64_t ld;
  double inv_n;
  float* G;  // [ld][ld]  (fp32: the entries are good to 1e-4 of the columns' scales; sums in fp64 up to the store)
};
static __global__ __launch_bounds__(256) void mg_reduce_kernel(MgReduceArgs a) {
  __shared__ double tile[32][33];
  const int t = blockIdx.x;
  int I, J;
  slm_host::triangle_tile_fast(t, &I, &J);
  const int sy = blockIdx.y >> 2, sx = blockIdx.y & 3;
  if (I == J && sx > sy) return;  // (above the diagonal)
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int64_t jg = (int64_t)J * MG_TILE + 32 * sx + tx;
  const int ej = jg < a.ld ? mg_scale_exp(a.cmax[jg]) : 0;
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int row = 32 * sy + ty + 8 * rr, col = 32 * sx + tx;
    const int64_t ig = (int64_t)I * MG_TILE + row;
    const float* src = a.P + (int64_t)t * (MG_TILE * MG_TILE) + row * MG_TILE + col;
    double s = 0.0;
    for (int c = 0; c < a.n_chunks; ++c) s += (double)src[(int64_t)c * a.n_tiles * (MG_TILE * MG_TILE)];
    const int ei = ig < a.ld ? mg_scale_exp(a.cmax[ig]) : 0;
    const double v = s * ldexp(a.inv_n, ei + ej);
    tile[ty + 8 * rr][tx] = (double)(float)v;  // (the mirrored entry is the stored one: symmetric to the bit)
    const bool lower = !(I == J && col > row);
    if (lower && ig < a.ld && jg < a.ld) a.G[ig * a.ld + jg] = (float)v;
  }
  __syncthreads();
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    // mirror: G[j][i] for the corner's entries (i, j), written along i
    const int col = 32 * sx + ty + 8 * rr, row = 32 * sy + tx;
    const int64_t ig = (int64_t)I * MG_TILE + row, jg2 = (int64_t)J * MG_TILE + col;
    const bool strictly_lower = !(I == J && col >= row);
    if (strictly_lower && ig < a.ld && jg2 < a.ld) a.G[jg2 * a.ld + ig] = (float)tile[tx][ty + 8 * rr];
  }
}

// ---------------------------------------------------------------------------------------------
// the inner iteration
// ---------------------------------------------------------------------------------------------
struct MgArgs {
  MgCtl* mg;
  WsCtl* ws;              // nullable: WsCtl::served tells which lanes the working set's model solver moved this round
  const double* partial;  // [halves][row sets][nblk][16][ld] of the products G~_s D (a half: sixteen lanes, the width of the product)
  int64_t part_stride;    // doubles between the partial sums of two row sets
  int32_t set_of[SLM_MAX_LANES];  // row set of lane l
  int n_sets;
  int64_t z_plane;        // doubles between the two halves' planes of Z (ld x 16)
  int nblk;
  double* gd;             // [16][ld] the product, summed over its row blocks (mg_gsum_kernel)
  double* Z;              // [halves][ld][16] D = v - z0, lane-minor: the B operand of the product, a plane per half of the lanes
  double* x;              // [16][ld] iterate
  double* v;              // [16][ld] point the model gradient is evaluated at
  double* vprev;          // [16][ld]
  double* gvprev;         // [16][ld]
};

// which lanes still iterate (one round of loads, a ballot)
// does a lane of this half still iterate -- on the row set of this grid slice (blockIdx.z)?  (the products of the other
// row sets are theirs: of a grid's five folds and thirty-two lanes most (set, half) pairs have nobody)
__device__ __forceinline__ bool mg_any_active(const MgCtl* mg, int half, const CovBatch& cb) {
  const int l = threadIdx.x & 63;
  int on = 0;
  if (l < SPLIT_LANES)
    on = (mg->lane[SPLIT_LANES * half + l].active != 0) & (mg->lane[SPLIT_LANES * half + l].settled == 0) &
         (cb.set_of[SPLIT_LANES * half + l] == (int)blockIdx.z);
  return __ballot(on != 0) != 0ull;
}

// the product of an inner iteration for one half of the lanes: cov_gz_mfma_kernel's loop on (G~, that half's plane of Z),
// skipped when no lane of the half iterates any more.  (G~ is stored in fp32 -- it is good to 1e-4 -- and widened on load:
// half the footprint, sixteen row sets within 3 GB at p = 5 000.  The product itself stays in fp64: an fp32 form on
// v_mfma_f32_16x16x4_f32 was built and measured -- the soak law's seven dense-ended paths 93.8 ms against 91.5, config 4's
// dense grid 0.317 s against 0.322: the launch is bound by neither the matrix cores nor the Gram's bytes but by the 250
// short workgroups of a 35 us kernel -- and removed.)
static __global__ __launch_bounds__(XTR_WAVES * 64, 2) void mg_gz_kernel(SplitArgs a, CovBatch cb, const MgCtl* mg, int half) {
  if (a.done != nullptr && *a.done != 0) return;
  if (!mg_any_active(mg, half, cb)) return;
  cov_gz_body<1, float>(a, cb);
}
// both halves of a call of more than sixteen lanes on ONE read of the Gram (a.R: plane 0 of Z, a.r_plane on: plane 1)
static __global__ __launch_bounds__(XTR_WAVES * 64, 1) void mg_gz32_kernel(SplitArgs a, CovBatch cb, const MgCtl* mg) {
  if (a.done != nullptr && *a.done != 0) return;
  const bool on0 = mg_any_active(mg, 0, cb), on1 = mg_any_active(mg, 1, cb);
  if (!on0 && !on1) return;
  cov_gz_body<2, float>(a, cb, (on0 ? 1u : 0u) | (on1 ? 2u : 0u));  // (and of those only the halves with lanes of this row set)
}

// The product's partial sums folded over the row blocks, in block order: gd[l][j] = sum_b partial[b][l][j].  A kernel of
// its own, grid (ld / 256, lanes): the 16 MB of partial sums are then read by the whole chip -- inside mg_step_kernel
// every lane's workgroup pulled its megabyte through ONE compute unit, 20 us of a 33 us call.
static __global__ __launch_bounds__(256) void mg_gsum_kernel(MgArgs m, int64_t ld, const int* done) {
  if (done != nullptr && *done != 0) return;
  const int lane_id = blockIdx.y;
  const MgLane* ml = &m.mg->lane[lane_id];
  if (ml->active == 0 || ml->settled != 0) return;
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= ld) return;
  const int half = lane_id / SPLIT_LANES, l16 = lane_id % SPLIT_LANES;
  const double* part = m.partial + ((int64_t)half * m.n_sets + m.set_of[lane_id]) * m.part_stride + (int64_t)l16 * ld + j;
  const int64_t bstride = (int64_t)SPLIT_LANES * ld;
  double acc = 0.0;
  int b = 0;
  for (; b + 8 <= m.nblk; b += 8) {
    double q[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) q[u] = part[(int64_t)(b + u) * bstride];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += q[u];
  }
  for (; b < m.nblk; ++b) acc += part[(int64_t)b * bstride];
  m.gd[(int64_t)lane_id * ld + j] = acc;
}

// Start of a round, one workgroup per lane, after the pass's tail kernel and the working set's model solver: does the lane
// take part, and if so x = v = z (the tail kernel's own next point), D = z - z0 with (z0, g0) = (zprev, gprev), the last
// point whose true gradient the lane has seen.
static __global__ __launch_bounds__(TAIL_THREADS) void mg_begin_kernel(TailArgs a, MgArgs m) {
  const int lane_id = blockIdx.x;
  PathCtl* ctl = a.ctl + lane_id;
  MgLane* ml = &m.mg->lane[lane_id];
  const int tid = threadIdx.x;
  const bool live = !(ctl->done != 0 || ctl->idle != 0 || a.gdone[0] != 0);
  const int served = m.ws != nullptr ? m.ws->served[lane_id] : 0;
  const int proposed = ml->proposed, rej_saved = ml->rej_saved, rejects = ctl->rejects;
  const int bad = ml->bad;
  const int point_now = ctl->point + ctl->pt_off;
  const bool same_point = ml->point == point_now;
  int rounds_pt = same_point ? ml->rounds_pt : 0;
  int rejected = same_point ? ml->rejected : 0;  // (counted per path point: the next point is another problem)
  __syncthreads();  // (everything above is read before anything below is written)
  // (the counter was set to zero with the proposal: non-zero now = the tail kernel's test on the true objective said no)
  const bool was_rejected = proposed && rejects > 0;
  if (was_rejected) rejected += 1;
  // spectral lanes only: their acceptance test on the true objective is what makes a proposal safe to make (an accelerated
  // lane -- the fallback of the tail kernel -- takes every point it is given)
  const bool active = live && !served && !bad && rejected < MG_REJECT_LIMIT && !a.provisional && ctl->total_iter >= 1 &&
                      rounds_pt < MG_ROUNDS_PER_POINT && ctl->mode == 1;
  if (tid == 0) {
    if (m.ws != nullptr) m.ws->served[lane_id] = 0;
    if (was_rejected) atomicAdd(&m.mg->rejected, 1);
    if (proposed) ctl->rejects = rej_saved;  // (the lane's own count, as it stood)
    ml->proposed = 0;
    ml->rejected = rejected;
    if (!same_point) {  // (also for a lane that sits this round out: the counts are those of its current point)
      ml->point = point_now;
      ml->rounds_pt = 0;
    }
    ml->active = active ? 1 : 0;
    ml->settled = 0;
    ml->iters = 0;
    if (active) {
      const double L = fmax(ctl->L, ctl->Lhat);
      ml->L = L;
      ml->Ls = fmin(L, fmax(ctl->ak, L / WS_BB_MAX_STEP));
      ml->t = 1.0;
      ml->spectral = 1;
      ml->have_prev = 0;
      ml->rq_n = 0;
      ml->rq_min = 0.0;
      ml->point = point_now;
      ml->rounds_pt = rounds_pt + 1;
      atomicAdd(&m.mg->rounds, 1);
    }
  }
  if (lane_id == 0 && tid == 0) m.mg->most_iters = 0;
  if (!active) return;
  const int64_t off = (int64_t)lane_id * a.ld;
  const double* z = a.z + off;
  const double* z0 = a.zprev + off;
  double* x = m.x + off;
  double* v = m.v + off;
  for (int j0 = tid; j0 < (int)a.ld; j0 += 4 * TAIL_THREADS) {
    double zj[4], zo[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = j0 + u * TAIL_THREADS;
      const int jj = j < a.p ? j : 0;
      zj[u] = z[jj];
      zo[u] = z0[jj];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = j0 + u * TAIL_THREADS;
      if (j < a.p) {
        x[j] = zj[u];
        v[j] = zj[u];
        m.Z[(int64_t)(lane_id / SPLIT_LANES) * m.z_plane + (int64_t)j * SPLIT_RSTRIDE + lane_id % SPLIT_LANES] = zj[u] - zo[u];
      } else if (j < (int)a.ld) {
        m.Z[(int64_t)(lane_id / SPLIT_LANES) * m.z_plane + (int64_t)j * SPLIT_RSTRIDE + lane_id % SPLIT_LANES] = 0.0;
      }
    }
  }
}

// One inner iteration of every active lane, a workgroup per lane: the model gradient at v from the product's partial
// sums, the proximal step, and the bookkeeping of ws_refine_lane's loop (ws_kernels.hpp) -- spectral steps first, then
// accelerated steps with restart, the curvature guard on L -- on all p coordinates.  Streaming like
// fista_tail_stream_kernel: nothing per feature lives in registers across a workgroup sum; the candidate sits in an LDS image.
template <int E>
__global__ __launch_bounds__(TAIL_THREADS) void mg_step_kernel(TailArgs a, MgArgs m) {
  __shared__ double red[9][TAIL_WAVES];
  constexpr int US_LDS = 16 * TAIL_THREADS;
  __shared__ double us_lds[US_LDS];
  const int lane_id = blockIdx.x;
  PathCtl* ctl = a.ctl + lane_id;
  MgLane* ml = &m.mg->lane[lane_id];
  if (a.gdone[0] != 0 || ml->active == 0 || ml->settled != 0) return;
  const int tid = threadIdx.x;
  const int p = a.p, G = a.G;
  const int64_t off = (int64_t)lane_id * a.ld;
  const double* z0 = a.zprev + off;
  const double* g0 = a.gprev + off;
  a.a0 += off; a.b0 += off; a.d0 += off;
  a.gscale += (int64_t)lane_id * G;
  double* x = m.x + off;
  double* v = m.v + off;
  double* vp = m.vprev + off;
  double* gvp = m.gvprev + off;
  double* us = p <= US_LDS ? us_lds : a.uscratch + off;
  const slm_path_point pt = a.pts[ctl->pt_off + ctl->point];
  const bool group_pen = (pt.sb != 0.0) || (pt.sd != 0.0);
  const double tol = ctl->tol;
  double L = ml->L, Ls = ml->Ls, t = ml->t, rq_min = ml->rq_min;
  int rq_n = ml->rq_n;
  bool spectral = ml->spectral != 0;
  const bool have_prev = ml->have_prev != 0;
  const int it = ml->iters;

  // ---- model gradient at v, the step's argument into the image, curvature along the last move of v --------------
  //  s[4] = ||v - v_prev||^2   s[5] = ||gv - gv_prev||^2   s[6] = <v - v_prev, gv - gv_prev>
  double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const double inv_ls = 1.0 / Ls;
  // the model gradient: g0 + the product (mg_gsum_kernel has folded its partial sums)
  const double* gdv = m.gd + off;
  tail_for<E>(tid, p, [&](int j, bool ok) {
    const double gv = g0[j] + gdv[j];
    const double vj = v[j];
    if (ok) {
      if (have_prev) {
        const double dv = vj - vp[j], dg = gv - gvp[j];
        s[4] = __builtin_fma(dv, dv, s[4]);
        s[5] = __builtin_fma(dg, dg, s[5]);
        s[6] = __builtin_fma(dv, dg, s[6]);
      }
      gvp[j] = gv;
      vp[j] = vj;
      us[j] = vj - gv * inv_ls;
    }
  });
  // ---- prox of the lane's penalty at its current path point, step 1 / Ls, in the image --------------------------
  tail_for<E>(tid, p, [&](int j, bool ok) {
    double uu = soft(us[j], inv_ls * pt.sa * a.a0[j]);
    if (group_pen && a.singleton) {
      const double nrm = fabs(uu);
      const double sc = nrm > 0.0 ? fmax(0.0, 1.0 - inv_ls * pt.sb * a.b0[j] / nrm) : 0.0;
      uu *= sc / (1.0 + inv_ls * pt.sd * a.d0[j]);
    }
    if (ok) us[j] = uu;
  });
  if (group_pen && !a.singleton) {
    __syncthreads();
    for_each_group_sumsq(us, a.order, a.gstart, G, a.team, [&](int g, double ss) {
      const double nrm = sqrt(ss);
      a.gscale[g] = (nrm > 0.0 ? fmax(0.0, 1.0 - inv_ls * pt.sb * a.b0[g] / nrm) : 0.0) / (1.0 + inv_ls * pt.sd * a.d0[g]);
    });
    __syncthreads();
    tail_for<E>(tid, p, [&](int j, bool ok) {
      const double val = us[j] * a.gscale[a.gid[j]];
      if (ok) us[j] = val;
    });
  }
  //  s[0] = ||u - v||^2  s[1] = ||u||^2  s[2] = (v - u).(u - x)  s[3] = #non-finite  s[7] = ||u - z0||^2
  tail_for<E>(tid, p, [&](int j, bool ok) {
    const double u = us[j], vj = v[j], xj = x[j], zo = z0[j];
    if (ok) {
      const double r = u - vj, d = u - zo;
      s[0] = __builtin_fma(r, r, s[0]);
      s[1] = __builtin_fma(u, u, s[1]);
      s[2] = __builtin_fma(-r, u - xj, s[2]);
      if (!isfinite(u)) s[3] += 1.0;
      s[7] = __builtin_fma(d, d, s[7]);
    }
  });
  block_sum<8>(s, red);
  if (s[3] > 0.0 || !isfinite(s[0])) {  // the model went astray: the lane keeps the tail kernel's own point
    if (tid == 0) {
      ml->active = 0;
      ml->bad = 1;
    }
    return;
  }
  if (have_prev && s[4] > 1e-20 * s[1] && s[4] > 0.0) {  // (a move at the rounding level measures nothing)
    const double rq = s[6] / s[4];
    if (rq > 0.0 && (rq_n == 0 || rq < rq_min)) rq_min = rq;
    rq_n += 1;
  }
  bool redo = false;
  if (have_prev && s[4] > 1e-20 * s[1] && s[4] > 0.0 && sqrt(s[5] / s[4]) > L) {  // the bound was too low (a move at the rounding level measures nothing)
    L = 1.05 * sqrt(s[5] / s[4]);
    if (!spectral) {  // an accelerated step of 1 / L was too long: again from x
      Ls = L;
      redo = true;
    }
  }
  double mom = 0.0, t_new = 1.0;
  bool inner_conv = false;
  if (!redo) {
    if (spectral) {
      const double rq = (have_prev && s[4] > 1e-20 * s[1]) ? s[6] / s[4] : L;
      Ls = fmin(L, fmax(rq, L / WS_BB_MAX_STEP));
      if (it + 1 >= WS_BB_ITERS) {
        spectral = false;
        Ls = L;
      }
    }
    inner_conv = sqrt(s[0]) <= fmax(WS_INNER_TOL * tol * sqrt(s[1]), MG_ETA * sqrt(s[7]));
    const bool restart = s[2] > 0.0;
    const double t_use = restart ? 1.0 : t;
    t_new = 0.5 * (1.0 + sqrt(1.0 + 4.0 * t_use * t_use));
    mom = spectral ? 0.0 : (t_use - 1.0) / t_new;
  }
  // ---- the next point -----------------------------------------------------------------------------------------
  tail_for<E>(tid, p, [&](int j, bool ok) {
    const double u = us[j], xj = x[j], zo = z0[j];
    const double xn = redo ? xj : u;
    const double vn = redo ? xj : u + mom * (u - xj);
    if (ok) {
      x[j] = xn;
      v[j] = vn;
      m.Z[(int64_t)(lane_id / SPLIT_LANES) * m.z_plane + (int64_t)j * SPLIT_RSTRIDE + lane_id % SPLIT_LANES] = vn - zo;
    }
  });
  if (tid == 0) {
    ml->L = L;
    ml->Ls = Ls;
    ml->t = redo ? 1.0 : t_new;
    ml->spectral = spectral ? 1 : 0;
    ml->have_prev = 1;
    ml->rq_n = rq_n;
    ml->rq_min = rq_min;
    ml->iters = it + 1;
    if (inner_conv) ml->settled = 1;
    atomicMax(&m.mg->most_iters, it + 1);
  }
}

// End of a round: the iterate becomes the lane's next evaluation point -- a CANDIDATE of the spectral scheme when the lane
// has a base (the tail kernel's acceptance test on the true objective decides, fista_tail_kernel), the start of the path
// point when it has none, the restart point of an accelerated lane.
static __global__ __launch_bounds__(TAIL_THREADS) void mg_finish_kernel(TailArgs a, MgArgs m) {
  __shared__ double red[2][TAIL_WAVES];
  __shared__ double us_lds[16 * TAIL_THREADS];
  const int lane_id = blockIdx.x;
  PathCtl* ctl = a.ctl + lane_id;
  MgLane* ml = &m.mg->lane[lane_id];
  if (a.gdone[0] != 0 || ml->active == 0 || ml->iters == 0) return;
  const int tid = threadIdx.x;
  const int p = a.p, G = a.G;
  const int64_t off = (int64_t)lane_id * a.ld;
  a.a0 += off; a.b0 += off; a.d0 += off;
  const double* x = m.x + off;
  double* z = a.z + off;
  double* beta = a.beta + off;
  double* us = p <= 16 * TAIL_THREADS ? us_lds : a.uscratch + off;
  const slm_path_point pt = a.pts[ctl->pt_off + ctl->point];
  const bool group_pen = (pt.sb != 0.0) || (pt.sd != 0.0);
  const int mode = ctl->mode;
  double pen[1] = {0.0};
  for (int j0 = tid; j0 < p; j0 += 4 * TAIL_THREADS) {
    double xj[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = j0 + u * TAIL_THREADS;
      xj[u] = x[j < p ? j : 0];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = j0 + u * TAIL_THREADS;
      if (j < p) {
        z[j] = xj[u];
        if (mode == 0) beta[j] = xj[u];
        us[j] = xj[u];
        const double ax = fabs(xj[u]);
        pen[0] += pt.sa * a.a0[j] * ax;
        if (group_pen && a.singleton) pen[0] += pt.sb * a.b0[j] * ax + 0.5 * pt.sd * a.d0[j] * ax * ax;
      }
    }
  }
  if (group_pen && !a.singleton) {
    __syncthreads();
    for_each_group_sumsq(us, a.order, a.gstart, G, a.team, [&](int g, double ss) {
      pen[0] += pt.sb * a.b0[g] * sqrt(ss) + 0.5 * pt.sd * a.d0[g] * ss;
    });
  }
  block_sum<1>(pen, red);
  if (tid == 0) {
    ctl->pen_z = pen[0];
    ctl->zzero = 0;
    ctl->zsup = 0;  // (the point is not supported on the working set: its residual comes from X)
    if (mode == 0) ctl->t = 1.0;
    // strong convexity on the face, from the iteration's own moves on the model (as ws_refine_lane reports it)
    ctl->mu = ml->rq_n >= 3 ? 0.5 * ml->rq_min : 0.0;
    ml->proposed = 1;
    ml->rej_saved = ctl->rejects;
    ctl->rejects = 0;
    atomicAdd(&m.mg->inner_iters, ml->iters);
    if (ml->settled) atomicAdd(&m.mg->settled, 1);
  }
}

}  // namespace slm

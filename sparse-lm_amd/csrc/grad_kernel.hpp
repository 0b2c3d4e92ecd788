// Fused single-pass gradient of the least-squares loss for gfx950 (MI355X):
//
//     partial[b][:] = sum_{rows i owned by workgroup b}  w_i (x_i . z - y_i) x_i
//
// i.e. the X^T (X z - y) that every FISTA iteration needs (the arithmetic the reference delegates
// to cvxpy's solver at src/sparselm/model/_base.py:516-518 for the objective built at
// src/sparselm/model/_lasso.py:109-121), with X read from HBM exactly ONCE per evaluation.
//
// Design (HBM-bound, 0.5 flop/byte):
//   * X is row-major with a leading dimension padded to 16 doubles; a workgroup of W wavefronts
//     (64 lanes each) owns a contiguous range of rows and ALL columns.  Lane t of the workgroup
//     owns the 16-byte column chunks {c*T + t}, so every wave-instruction is one fully coalesced
//     1 KiB `global_load_dwordx4`.
//   * R rows at a time are held in VGPRs between the two uses (dot with z, then rank-R update of
//     the per-lane gradient accumulators); the next R rows are already in flight in a second
//     register set, so the workgroup's single barrier per step never drains the memory pipe.
//   * The R dot products are reduced with a 64-wide butterfly per wavefront and a W-entry LDS
//     exchange (one barrier; the exchange buffer is double-buffered so no second barrier is
//     needed).  Every lane then holds bit-identical residuals.
//   * Each workgroup writes one partial gradient row; a deterministic second kernel sums them
//     (no fp64 atomics => bit-reproducible results).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace slm {

typedef double d2 __attribute__((ext_vector_type(2)));

struct GradArgs {
  const double* X;       // [n][ld], pad columns are zero
  const double* y;       // [n]
  const double* rw;      // [n] row weights or nullptr
  const double* z;       // [ld], pad entries are zero
  double* partial;       // [gridDim.x][ld]
  double* loss_partial;  // [gridDim.x]   sum_i w_i (x_i.z - y_i)^2 over the block's rows
  const int* done;       // early-exit flag of the path state machine (nullable)
  int64_t n;
  int64_t ld;            // doubles, multiple of 16
  int64_t rows_base;     // n / gridDim.x   (host-computed: no 64-bit division on the device)
  int64_t rows_rem;      // n % gridDim.x
  int p2;                // ld / 2: number of 16-byte chunks per row
};

#ifndef SLM_NT_LOADS
#define SLM_NT_LOADS 1
#endif

__device__ __forceinline__ d2 load_x(const d2* p) {
#if SLM_NT_LOADS
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}

__device__ __forceinline__ double wave_sum_all(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

template <int C, int R>
struct RowSet {
  d2 x[R][C];
  double y[R];
  double m[R];  // row multiplier: row weight, or 0 for rows past the end of the block's range
};

template <int W, int C, int R>
__global__ __launch_bounds__(W * 64) void grad_fused_kernel(GradArgs a) {
  constexpr int T = W * 64;
  if (a.done != nullptr && *a.done != 0) return;

  __shared__ double red[2][R][W];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // contiguous, balanced row ranges
  const int64_t b = blockIdx.x;
  const int64_t base = a.rows_base, rem = a.rows_rem;
  const int64_t r0 = b * base + (b < rem ? b : rem);
  const int64_t nrows = base + (b < rem ? 1 : 0);

  int cidx[C];
  uint32_t coff[C];  // byte offset of the lane's chunk inside a row (rows are < 4 GiB)
  bool valid[C];
  d2 zr[C], acc[C];
  const d2* zv = reinterpret_cast<const d2*>(a.z);
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int ci = c * T + tid;
    valid[c] = ci < a.p2;
    cidx[c] = valid[c] ? ci : a.p2 - 1;
    coff[c] = (uint32_t)cidx[c] * 16u;
    d2 zz = zv[cidx[c]];
    zr[c] = valid[c] ? zz : d2{0.0, 0.0};  // clamped duplicates contribute nothing to the dots
    acc[c] = d2{0.0, 0.0};
  }
  double loss = 0.0;

  auto load_rows = [&](RowSet<C, R>& s, int64_t step) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t i = step * R + r;
      const bool live = i < nrows;
      const int64_t row = r0 + (live ? i : nrows - 1);
      s.y[r] = a.y[row];
      double m = 1.0;
      if (a.rw != nullptr) m = a.rw[row];
      s.m[r] = live ? m : 0.0;
      const char* rp = reinterpret_cast<const char*>(a.X + row * a.ld);  // wave-uniform base
#pragma unroll
      for (int c = 0; c < C; ++c) s.x[r][c] = load_x(reinterpret_cast<const d2*>(rp + coff[c]));
    }
  };

  auto process = [&](const RowSet<C, R>& s, int parity) {
    double dot[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      double t = 0.0;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        t = __builtin_fma(s.x[r][c].x, zr[c].x, t);
        t = __builtin_fma(s.x[r][c].y, zr[c].y, t);
      }
      dot[r] = wave_sum_all(t);
    }
    if constexpr (W > 1) {
      if (lane == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) red[parity][r][wave] = dot[r];
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < R; ++r) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < W; ++w) t += red[parity][r][w];
        dot[r] = t;
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const double e = dot[r] - s.y[r];
      const double res = e * s.m[r];
      loss = __builtin_fma(res, e, loss);
#pragma unroll
      for (int c = 0; c < C; ++c) {
        acc[c].x = __builtin_fma(res, s.x[r][c].x, acc[c].x);
        acc[c].y = __builtin_fma(res, s.x[r][c].y, acc[c].y);
      }
    }
  };

  if (nrows > 0) {
    const int64_t nsteps = (nrows + R - 1) / R;
    RowSet<C, R> sa, sb;
    load_rows(sa, 0);
    for (int64_t s = 0; s < nsteps; s += 2) {
      const bool has1 = s + 1 < nsteps;
      if (has1) load_rows(sb, s + 1);
      process(sa, 0);
      if (has1) {
        if (s + 2 < nsteps) load_rows(sa, s + 2);
        process(sb, 1);
      }
    }
  }

  d2* out = reinterpret_cast<d2*>(a.partial + b * a.ld);
#pragma unroll
  for (int c = 0; c < C; ++c)
    if (valid[c]) out[cidx[c]] = acc[c];
  if (tid == 0) a.loss_partial[b] = loss;
}

// ---------------------------------------------------------------------------------------------
// Deterministic cross-workgroup reduction:  g[j] = scale * sum_b partial[b][j],  loss likewise.
// 256 threads = 16 column lanes x 16 row slices; one workgroup per 16 columns (128-byte segments).
// The loss sum lands in g[ld] so that a single all-reduce covers gradient and loss in the
// row-sharded mode.
// ---------------------------------------------------------------------------------------------
struct ReduceArgs {
  const double* partial;
  const double* loss_partial;
  double* g;  // [ld + 16]
  const int* done;
  int nblk;
  int64_t ld;
  double scale;       // 1/n
  double loss_scale;  // 1/(2n)
};

__global__ __launch_bounds__(256) void reduce_partials_kernel(ReduceArgs a) {
  if (a.done != nullptr && *a.done != 0) return;
  __shared__ double lds[16][17];
  const int tid = threadIdx.x;
  const int cl = tid & 15, slice = tid >> 4;
  const int64_t col = (int64_t)blockIdx.x * 16 + cl;
  const bool loss_block = (int64_t)blockIdx.x * 16 >= a.ld;  // the extra trailing block
  double s = 0.0;
  if (!loss_block) {
    for (int b = slice; b < a.nblk; b += 16) s += a.partial[(int64_t)b * a.ld + col];
  } else {
    for (int b = tid; b < a.nblk; b += 256) s += a.loss_partial[b];
  }
  lds[slice][cl] = s;
  __syncthreads();
  if (!loss_block) {
    if (slice == 0) {
      double t = 0.0;
#pragma unroll
      for (int k = 0; k < 16; ++k) t += lds[k][cl];
      a.g[col] = t * a.scale;
    }
  } else if (tid == 0) {
    double t = 0.0;
    for (int k = 0; k < 16; ++k)
      for (int c = 0; c < 16; ++c) t += lds[k][c];
    a.g[a.ld] = t * a.loss_scale;
  }
}

}  // namespace slm

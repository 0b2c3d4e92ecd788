// Data-movement and synthetic-data kernels (one-off per dataset; not on the per-iteration path).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "grad_kernel.hpp"   // d2
#include "tail_kernels.hpp"  // mix64, block_sum

namespace slm {

// F-order (column-major, leading dimension = n) -> padded row-major.  32x32 LDS tile.
static __global__ __launch_bounds__(256) void transpose_f2c_kernel(const double* __restrict__ src, int64_t n,
                                                            int64_t p, double* __restrict__ dst,
                                                            int64_t ld) {
  __shared__ double tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int64_t i0 = (int64_t)blockIdx.x * 32, j0 = (int64_t)blockIdx.y * 32;
  for (int k = ty; k < 32; k += 8) {  // read: consecutive lanes walk rows i (contiguous in F-order)
    const int64_t i = i0 + tx, j = j0 + k;
    tile[k][tx] = (i < n && j < p) ? src[j * n + i] : 0.0;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {  // write: consecutive lanes walk columns j (contiguous in C-order)
    const int64_t i = i0 + k, j = j0 + tx;
    if (i < n && j < p) dst[i * ld + j] = tile[tx][k];
  }
}

// The column-major copy of X the working set gathers from and rowdot_mfma_kernel streams, in tiles of 32 rows:
//   XT[((i >> 5) * ld + j) * 32 + (i & 31)] = X[i][j]
// -- inside a tile of rows a column is 256 contiguous bytes (what a gather moves per request, what sixteen lanes of an
// MFMA operand load cover), the columns of a tile follow each other (four columns = one KiB, a batch of sixteen = 4 KiB
// per request group), and the tiles of a block of rows follow each other: every reader walks its part linearly.  As a
// plain [ld][n] matrix (until round 3) the 256-byte pieces of consecutive columns lay 8 n bytes apart -- a new DRAM row
// and, every third column, a new 2 MB page per piece.  Rows beyond n inside the last tile are written as zeros.
// grid ((n + 31) / 32, (ld + 31) / 32).
static __global__ __launch_bounds__(256) void tile_columns_kernel(const double* __restrict__ X, int64_t n, int64_t ld,
                                                           double* __restrict__ XT) {
  __shared__ double tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int64_t i0 = (int64_t)blockIdx.x * 32, j0 = (int64_t)blockIdx.y * 32;
  for (int k = ty; k < 32; k += 8) {  // read: consecutive lanes walk columns j (contiguous in X)
    const int64_t i = i0 + k, j = j0 + tx;
    tile[k][tx] = (i < n && j < ld) ? X[i * ld + j] : 0.0;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {  // write: consecutive lanes walk the 32 rows of the tile (contiguous in XT)
    const int64_t j = j0 + k;
    if (j < ld) XT[((int64_t)blockIdx.x * ld + j) * 32 + tx] = tile[tx][k];
  }
}

// Read-only stream over a buffer (slm_dataset_read_ceiling): the rate the memory system delivers to plain 16-byte loads
// with nothing to do but add them up -- the ceiling a pass over X is to be read against on THIS device (SURVEY Appendix D;
// the microarchitecture guide's 6.29 TB/s is a float4 COPY, half of whose traffic is stores).  Every workgroup walks a
// contiguous part of the buffer, a wavefront 1 KiB per load instruction, U instructions in flight per thread; one sum per
// workgroup so that nothing is optimised away.  The host tries several (workgroups per CU, U, cached / non-temporal) and
// reports the best: a ceiling is the best stream found, not one guess at it.
// RR: the workgroups take the chunks of U x 4 KiB in turn (at any moment the chip reads one window of the buffer) instead
// of a contiguous part each
template <int U, bool NT, bool RR = false>
__global__ __launch_bounds__(256) void read_stream_kernel(const double* __restrict__ src, int64_t count2 /* 16-byte pieces */,
                                                          double* __restrict__ sink) {
  const d2* p = reinterpret_cast<const d2*>(src);
  const int64_t per = (count2 + gridDim.x - 1) / gridDim.x;
  const int64_t lo = RR ? (int64_t)blockIdx.x * (U * 256) : (int64_t)blockIdx.x * per;
  const int64_t hi = RR ? count2 : (lo + per < count2 ? lo + per : count2);
  const int64_t step = RR ? (int64_t)gridDim.x * (U * 256) : (int64_t)U * 256;
  double a0 = 0.0, a1 = 0.0;
  int64_t i = lo + threadIdx.x;
  for (; i + (int64_t)(U - 1) * 256 < hi; i += step) {
    d2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * 256) : p[i + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      a0 += v[u].x;
      a1 += v[u].y;
    }
  }
  if (!RR || blockIdx.x == 0) {  // (the tail: a few KiB at most)
    if (RR) i = (count2 / (U * 256)) * (U * 256) + threadIdx.x;
    for (; i < hi; i += 256) {
      const d2 v = p[i];
      a0 += v.x;
      a1 += v.y;
    }
  }
  const double t = wave_sum_all(a0 + a1);
  __shared__ double w[4];
  if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = t;
  __syncthreads();
  if (threadIdx.x == 0) sink[blockIdx.x] = (w[0] + w[1]) + (w[2] + w[3]);
}

// Row-major copy with different leading dimensions (device -> device), pad columns left untouched.
static __global__ __launch_bounds__(256) void copy_rows_kernel(const double* __restrict__ src, int64_t n,
                                                        int64_t p, int64_t lds_, double* __restrict__ dst,
                                                        int64_t ld) {
  const int64_t total = n * p;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e / p, j = e - i * p;
    dst[i * ld + j] = src[i * lds_ + j];
  }
}

// out[l] = sum over the row blocks of loss_partial[blk][l] (the split pass's residual kernels leave sum_i w_i e_i^2 of every
// block there), in block order: the weighted SSE of sixteen coefficient vectors after ONE read of X (slm_eval_sse)
static __global__ void sse_from_blocks_kernel(const double* __restrict__ loss_partial, int nblk, int lanes_stride, double* __restrict__ out) {
  const int l = threadIdx.x;
  if (l >= lanes_stride) return;
  double s = 0.0;
  for (int b = 0; b < nblk; ++b) s += loss_partial[(int64_t)b * lanes_stride + l];
  out[l] = s;
}

// flags |= 1 where v holds a NaN, |= 2 where it holds an infinity (slm_dataset_nonfinite)
static __global__ __launch_bounds__(256) void nonfinite_kernel(const double* __restrict__ v, int64_t count, int* __restrict__ flags) {
  int f = 0;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += (int64_t)gridDim.x * blockDim.x) {
    const double x = v[e];
    if (x != x) f |= 1;
    else if (x - x != 0.0) f |= 2;
  }
  if (f) atomicOr(flags, f);
}

// lane 0's copy of a per-lane vector -> lanes 1 .. n_lanes-1 (stride ld)
static __global__ __launch_bounds__(256) void broadcast_lanes_kernel(double* v, int64_t count, int64_t ld, int n_lanes) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < count;
       e += (int64_t)gridDim.x * blockDim.x) {
    const double x = v[e];
    for (int l = 1; l < n_lanes; ++l) v[(int64_t)l * ld + e] = x;
  }
}

// Everything a solve has to reset before its first pass, in ONE launch (it used to be some fifteen small
// fills, copies and broadcasts, each a dependent 4-8 us step on the stream: 0.15 ms per solve).  Per lane l and
// vector (a, b, d): mode 0 = the host uploaded it (leave it), 1 = fill with 1.0, 2 = copy lane 0's.
// beta_mode: 0 = zero, 1 = the host uploaded a warm start.  z <- beta, zprev = gprev = 0.
// carry != 0 (a "carried start", solve_core): every lane starts where the dataset's last solve left it -- the point
// zprev, whose gradient gprev and loss the tail kernels kept -- and g receives that gradient: the first step of the
// solve needs no pass over the data.
struct SetupArgs {
  double *beta, *z, *zprev, *gprev, *a0, *b0, *d0;
  double* g;              // [n_lanes][ld + 16], written when carry != 0
  int carry;
  double carry_loss[SLM_MAX_LANES];
  unsigned char* infos;   // zeroed: infos_bytes bytes (multiple of 8)
  int64_t infos_bytes;
  int64_t ld, p, G;
  int n_lanes, max_lanes;
  unsigned char a_mode[SLM_MAX_CELLS], b_mode[SLM_MAX_CELLS], d_mode[SLM_MAX_CELLS], beta_mode[SLM_MAX_CELLS];
};

__device__ __forceinline__ void solve_setup_body(const SetupArgs& s) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int64_t e = t0; e < (int64_t)s.max_lanes * s.ld; e += stride) {
    const int l = (int)(e / s.ld);
    const int64_t j = e - (int64_t)l * s.ld;
    double bv = 0.0;
    if (l < s.n_lanes && s.beta_mode[l] == 1 && j < s.p) bv = s.beta[e];
    if (s.carry && l < s.n_lanes) {
      bv = j < s.p ? s.zprev[e] : 0.0;
      s.g[(int64_t)l * (s.ld + 16) + j] = j < s.p ? s.gprev[e] : 0.0;
      if (j == 0) s.g[(int64_t)l * (s.ld + 16) + s.ld] = s.carry_loss[l];
    }
    s.beta[e] = bv;
    s.z[e] = bv;
    s.zprev[e] = 0.0;
    s.gprev[e] = 0.0;
    if (l < s.n_lanes) {
      if (j < s.p) {
        if (s.a_mode[l] == 1) s.a0[e] = 1.0;
        else if (s.a_mode[l] == 2) s.a0[e] = s.a0[j];
      }
      if (j < s.G) {
        if (s.b_mode[l] == 1) s.b0[e] = 1.0;
        else if (s.b_mode[l] == 2) s.b0[e] = s.b0[j];
        if (s.d_mode[l] == 1) s.d0[e] = 1.0;
        else if (s.d_mode[l] == 2) s.d0[e] = s.d0[j];
      }
    }
  }
  unsigned long long* q = reinterpret_cast<unsigned long long*>(s.infos);
  for (int64_t e = t0; e < s.infos_bytes / 8; e += stride) q[e] = 0ull;
}
static __global__ __launch_bounds__(256) void solve_setup_kernel(SetupArgs s) { solve_setup_body(s); }

static __global__ __launch_bounds__(256) void fill_kernel(double* dst, int64_t count, double value) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < count;
       e += (int64_t)gridDim.x * blockDim.x)
    dst[e] = value;
}

// sums[0] = sum_i w_i, sums[1] = sum_i w_i y_i   (one workgroup; w == nullptr => ones)
static __global__ __launch_bounds__(1024) void weighted_sums_kernel(const double* __restrict__ y,
                                                             const double* __restrict__ w, int64_t n,
                                                             double* __restrict__ sums) {
  __shared__ double red[2][TAIL_WAVES];
  double s[2] = {0.0, 0.0};
  for (int64_t i = threadIdx.x; i < n; i += 1024) {
    const double wi = w ? w[i] : 1.0;
    s[0] += wi;
    s[1] = __builtin_fma(wi, y[i], s[1]);
  }
  block_sum<2>(s, red);
  if (threadIdx.x == 0) {
    sums[0] = s[0];
    sums[1] = s[1];
  }
}

// X[i][j] -= xmean[j] (j < p only: pad columns stay zero), y[i] -= ymean
static __global__ __launch_bounds__(256) void center_kernel(double* __restrict__ X, double* __restrict__ y, int64_t n,
                                                     int64_t p, int64_t ld, const double* __restrict__ xmean,
                                                     double ymean) {
  const int64_t chunks = ld / 2;
  const int64_t total = n * chunks;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e / chunks, c = e - i * chunks;
    d2* px = reinterpret_cast<d2*>(X + i * ld) + c;
    d2 v = *px;
    const int64_t j = 2 * c;
    if (j < p) v.x -= xmean[j];
    if (j + 1 < p) v.y -= xmean[j + 1];
    *px = v;
    if (c == 0) y[i] -= ymean;
  }
}

// Two independent N(0,1) draws from a 64-bit counter (Box-Muller on two 53-bit uniforms).
__device__ __forceinline__ void normal_pair(uint64_t key, uint64_t ctr, double& n0, double& n1) {
  const uint64_t h1 = mix64(key ^ mix64(ctr + 0x632be59bd9b4e019ull));
  const uint64_t h2 = mix64(h1 + 0x9e3779b97f4a7c15ull);
  const double u1 = ((double)(h1 >> 11) + 0.5) * (1.0 / 9007199254740992.0);
  const double u2 = ((double)(h2 >> 11) + 0.5) * (1.0 / 9007199254740992.0);
  const double r = sqrt(-2.0 * log(u1));
  double s, c;
  sincospi(2.0 * u2, &s, &c);
  n0 = r * c;
  n1 = r * s;
}

// X_ij ~ N(0,1) keyed by (seed, global row, column pair): independent of the launch geometry and of
// how rows are sharded over ranks.
static __global__ __launch_bounds__(256) void synth_x_kernel(double* __restrict__ X, int64_t n, int64_t p,
                                                      int64_t ld, uint64_t seed, int64_t row_offset) {
  const int64_t pairs = (p + 1) / 2;
  const int64_t total = n * pairs;
  const uint64_t key = mix64(seed ^ 0x5851f42d4c957f2dull);
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e / pairs, jp = e - i * pairs;
    double a, b;
    normal_pair(key, (uint64_t)(row_offset + i) * (uint64_t)pairs + (uint64_t)jp, a, b);
    double* row = X + i * ld;
    row[2 * jp] = a;
    if (2 * jp + 1 < p) row[2 * jp + 1] = b;
  }
}

// y_i = x_i . coef + noise_sd * N(0,1): one wavefront per row.
static __global__ __launch_bounds__(256) void synth_y_kernel(const double* __restrict__ X, int64_t n, int64_t p,
                                                      int64_t ld, const double* __restrict__ coef,
                                                      double noise_sd, uint64_t seed,
                                                      int64_t row_offset, double* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const uint64_t key = mix64(seed ^ 0x2545f4914f6cdd1dull);
  for (int64_t i = wave; i < n; i += nwaves) {
    const double* row = X + i * ld;
    double s = 0.0;
    for (int64_t j = lane; j < p; j += 64) s = __builtin_fma(row[j], coef[j], s);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) {
      double e0, e1;
      normal_pair(key, (uint64_t)(row_offset + i), e0, e1);
      y[i] = s + noise_sd * e0;
    }
  }
}

}  // namespace slm

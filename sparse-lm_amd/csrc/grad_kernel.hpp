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
  const double* rw;      // row weights: nullptr, [n] (rw_stride == 0) or [B][rw_stride] per lane
  const double* z;       // [B][ld], pad entries are zero
  double* partial;       // [gridDim.x][B][ld]
  double* loss_partial;  // [gridDim.x][B]   sum_i w_i (x_i.z - y_i)^2 over the block's rows
  const int* done;       // early-exit flag of the path state machine (nullable)
  int64_t n;
  int64_t ld;            // doubles, multiple of 16
  int64_t rows_base;     // n / gridDim.x   (host-computed: no 64-bit division on the device)
  int64_t rows_rem;      // n % gridDim.x
  int64_t rw_stride;     // 0: all lanes share rw[]; otherwise lane b reads rw[b*rw_stride + row]
  int p2;                // ld / 2: number of 16-byte chunks per row
  const int* skip = nullptr;  // non-null and *skip != 0: this pass is not needed (light_kernels.hpp): return at once
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

// v + (v moved by a DPP pattern); lanes the pattern does not feed (or rows masked off) add 0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, true);
  const int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, true);
  return v + __hiloint2double(hi2, lo2);
}

// Sum over the 64 lanes with DPP moves only (no LDS round trips, short dependent chain):
// inclusive scan inside each row of 16 lanes (row_shr 1,2,4,8), then row_bcast15 / row_bcast31
// carry the row totals forward.  The total ends up in LANE 63 (other lanes hold prefixes).
__device__ __forceinline__ double wave_sum_lane63(double v) {
  v = dpp_add<0x111, 0xf>(v);  // row_shr:1
  v = dpp_add<0x112, 0xf>(v);  // row_shr:2
  v = dpp_add<0x114, 0xf>(v);  // row_shr:4
  v = dpp_add<0x118, 0xf>(v);  // row_shr:8
  v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
  v = dpp_add<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
  return v;
}

// Sum over aligned groups of W lanes (W = 2, 4, 8), every lane of the group getting the same bits
// (the pairings are commutative): row_half_mirror, then quad permutes [3,2,1,0] and [1,0,3,2].
template <int W>
__device__ __forceinline__ double group_sum_all(double t) {
  static_assert(W == 1 || W == 2 || W == 4 || W == 8, "W must be 1, 2, 4 or 8");
  if constexpr (W == 8) {
    t = dpp_add<0x141, 0xf>(t);  // row_half_mirror: i <-> 7 - i
    t = dpp_add<0x1B, 0xf>(t);   // quad_perm [3,2,1,0]
    t = dpp_add<0xB1, 0xf>(t);   // quad_perm [1,0,3,2]
  } else if constexpr (W == 4) {
    t = dpp_add<0xB1, 0xf>(t);   // quad_perm [1,0,3,2]
    t = dpp_add<0x4E, 0xf>(t);   // quad_perm [2,3,0,1]
  } else if constexpr (W == 2) {
    t = dpp_add<0xB1, 0xf>(t);
  }
  return t;
}

__device__ __forceinline__ double read_lane63(double v) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}

template <int C, int R, int B>
struct RowSet {
  d2 x[R][C];
  double y[R];
  double m[R][B];  // row multiplier per lane: row weight, or 0 for rows past the block's range
};

// B = number of independent problems ("lanes": alpha sub-paths, CV folds) that share ONE pass over
// X.  Lane b has its own point z_b, row weights and partial gradient; the X registers are shared, so
// HBM traffic per launch is that of a single gradient while B gradients come out.
template <int W, int C, int R, int B>
__global__ __launch_bounds__(W * 64) void grad_fused_kernel(GradArgs a) {
  constexpr int T = W * 64;
  if (a.done != nullptr && *a.done != 0) return;
  if (a.skip != nullptr && *a.skip != 0) return;

  __shared__ double red[2][R][B][W];

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
  d2 zr[B][C], acc[B][C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int ci = c * T + tid;
    valid[c] = ci < a.p2;
    cidx[c] = valid[c] ? ci : a.p2 - 1;
    coff[c] = (uint32_t)cidx[c] * 16u;
#pragma unroll
    for (int l = 0; l < B; ++l) {
      const d2 zz = reinterpret_cast<const d2*>(a.z + l * a.ld)[cidx[c]];
      zr[l][c] = valid[c] ? zz : d2{0.0, 0.0};  // clamped duplicates contribute nothing to the dots
      acc[l][c] = d2{0.0, 0.0};
    }
  }
  double loss[B];
#pragma unroll
  for (int l = 0; l < B; ++l) loss[l] = 0.0;

  auto load_rows = [&](RowSet<C, R, B>& s, int64_t step) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t i = step * R + r;
      const bool live = i < nrows;
      const int64_t row = r0 + (live ? i : nrows - 1);
      s.y[r] = a.y[row];
#pragma unroll
      for (int l = 0; l < B; ++l) {
        double m = 1.0;
        if (a.rw != nullptr) m = a.rw[l * a.rw_stride + row];
        s.m[r][l] = live ? m : 0.0;
      }
      const char* rp = reinterpret_cast<const char*>(a.X + row * a.ld);  // wave-uniform base
#pragma unroll
      for (int c = 0; c < C; ++c) s.x[r][c] = load_x(reinterpret_cast<const d2*>(rp + coff[c]));
    }
  };

  auto process = [&](const RowSet<C, R, B>& s, int parity) {
    double dot[R][B];
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
      for (int l = 0; l < B; ++l) {
        double t = 0.0;
#pragma unroll
        for (int c = 0; c < C; ++c) {
          t = __builtin_fma(s.x[r][c].x, zr[l][c].x, t);
          t = __builtin_fma(s.x[r][c].y, zr[l][c].y, t);
        }
        dot[r][l] = wave_sum_lane63(t);
        if constexpr (W == 1) dot[r][l] = read_lane63(dot[r][l]);
      }
    }
    if constexpr (W > 1) {
      if (lane == 63) {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
          for (int l = 0; l < B; ++l) red[parity][r][l][wave] = dot[r][l];
      }
      __syncthreads();
      // lane i picks up wavefront (i mod W)'s partial (ONE LDS read per value) and every aligned
      // group of W lanes folds them with DPP moves: no LDS round trips after the barrier and the
      // same bits in every lane
#pragma unroll
      for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int l = 0; l < B; ++l) dot[r][l] = group_sum_all<W>(red[parity][r][l][lane & (W - 1)]);
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
      for (int l = 0; l < B; ++l) {
        const double e = dot[r][l] - s.y[r];
        const double res = e * s.m[r][l];
        loss[l] = __builtin_fma(res, e, loss[l]);
#pragma unroll
        for (int c = 0; c < C; ++c) {
          acc[l][c].x = __builtin_fma(res, s.x[r][c].x, acc[l][c].x);
          acc[l][c].y = __builtin_fma(res, s.x[r][c].y, acc[l][c].y);
        }
      }
    }
  };

  if (nrows > 0) {
    const int64_t nsteps = (nrows + R - 1) / R;
    RowSet<C, R, B> sa, sb;
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

#pragma unroll
  for (int l = 0; l < B; ++l) {
    d2* out = reinterpret_cast<d2*>(a.partial + (b * B + l) * a.ld);
#pragma unroll
    for (int c = 0; c < C; ++c)
      if (valid[c]) out[cidx[c]] = acc[l][c];
    if (tid == 0) a.loss_partial[b * B + l] = loss[l];
  }
}

// ---------------------------------------------------------------------------------------------
// LDS-ring variant: the same single pass, but the rows in flight wait in LDS instead of VGPRs.
//
// Each wavefront streams ITS OWN 16-byte chunks of the next D rows straight into LDS with
// `global_load_lds_dwordx4` (no VGPR destination) and later reads exactly those chunks back, so the
// ring needs no cross-wave synchronisation: the only ordering is the wave's own counted
// `s_waitcnt vmcnt(D*C)`.  With the prefetch depth decoupled from the register budget, the register
// file holds only one row (4C VGPRs) next to the per-lane z and accumulators (8C VGPRs per lane),
// which is what lets four and more lanes keep two to three rows (80-120 KB per CU) in flight.
// Rules followed (cdna_hip_programming.md, "Pipelining across barriers"): all LDS in ONE array, raw
// `s_barrier` + `lgkmcnt(0)` for the dot exchange (a `__syncthreads()` would drain the DMA queue),
// counted `vmcnt`, no VGPR-destination global loads inside the loop (y / row weights are scalar loads).
// ---------------------------------------------------------------------------------------------
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// Scalar (SMEM) load of one double through a wave-uniform pointer.  Written in asm because, next to
// LDS-DMA traffic, hipcc turns `y[row]` into a VGPR load and then waits `vmcnt(0)` on it, draining
// the ring every row.  The caller must `s_waitcnt lgkmcnt(0)` before using the value.
__device__ __forceinline__ uint64_t smem_load_u64(const double* p) {
  uint64_t v;
  asm volatile("s_load_dwordx2 %0, %1, 0x0" : "=s"(v) : "s"(p) : "memory");
  return v;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int W, int C, int B, int D>
__global__ __launch_bounds__(W * 64) void grad_ring_kernel(GradArgs a) {
  constexpr int T = W * 64;
  constexpr int SLOT = T * C * 16;             // bytes of one ring slot (one padded row)
  constexpr int RING = (D + 1) * SLOT;
  constexpr int RED = 2 * B * W * 8;
  static_assert(RING + RED <= 160 * 1024, "ring does not fit the 160 KiB LDS");
  static_assert(D * C < 64, "too many DMA loads in flight for vmcnt");
  if (a.done != nullptr && *a.done != 0) return;
  if (a.skip != nullptr && *a.skip != 0) return;

  __shared__ __attribute__((aligned(16))) char smem[RING + RED];
  double* red = reinterpret_cast<double*>(smem + RING);  // [2][B][W]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  const int64_t b = blockIdx.x;
  const int64_t base = a.rows_base, rem = a.rows_rem;
  const int64_t r0 = b * base + (b < rem ? b : rem);
  const int64_t nrows = base + (b < rem ? 1 : 0);

  uint32_t coff[C];  // byte offset of the lane's chunk inside a global row (clamped)
  bool valid[C];
  d2 zr[B][C], acc[B][C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int ci = c * T + tid;
    valid[c] = ci < a.p2;
    const int cc = valid[c] ? ci : a.p2 - 1;
    coff[c] = (uint32_t)cc * 16u;
#pragma unroll
    for (int l = 0; l < B; ++l) {
      const d2 zz = reinterpret_cast<const d2*>(a.z + l * a.ld)[cc];
      zr[l][c] = valid[c] ? zz : d2{0.0, 0.0};
      acc[l][c] = d2{0.0, 0.0};
    }
  }
  double loss[B];
#pragma unroll
  for (int l = 0; l < B; ++l) loss[l] = 0.0;

  // DMA one row into ring slot `slot`: C wave-instructions of 1 KiB each
  auto issue_row = [&](int64_t i, int slot) {
    const char* rp = reinterpret_cast<const char*>(a.X + (r0 + i) * a.ld);
#pragma unroll
    for (int c = 0; c < C; ++c) {
      char* dst = smem + slot * SLOT + (c * T + wave * 64) * 16;  // wave-uniform; hardware adds lane*16
      __builtin_amdgcn_global_load_lds((gptr_t)(rp + coff[c]), (lptr_t)dst, 16, 0, SLM_NT_LOADS ? 2 : 0);
    }
  };

  if (nrows > 0) {
    // prologue: D rows in flight
#pragma unroll
    for (int k = 0; k < D; ++k)
      if (k < nrows) issue_row(k, k);
    int slot = 0;        // slot of row i
    int slot_in = D;     // slot row i + D goes into
    for (int64_t i = 0; i < nrows; ++i) {
      const int64_t left = nrows - 1 - i;  // rows after i
      if (left >= D) {
        issue_row(i + D, slot_in);
        wait_vmcnt<D * C>();
      } else {
        // tail: fewer than D rows remain in flight behind row i
        if (D >= 3 && left == 2) wait_vmcnt<(D >= 3 ? 2 : 0) * C>();
        else if (D >= 2 && left == 1) wait_vmcnt<(D >= 2 ? 1 : 0) * C>();
        else wait_vmcnt<0>();
      }
      // scalar operands of this row (SMEM loads; waited for together with the LDS reads below)
      const int64_t row = r0 + i;
      uint64_t yi_bits = smem_load_u64(a.y + row);
      uint64_t m_bits[B];
      const bool has_rw = a.rw != nullptr;
      if (has_rw) {
#pragma unroll
        for (int l = 0; l < B; ++l) m_bits[l] = smem_load_u64(a.rw + l * a.rw_stride + row);
      }

      d2 x[C];
#pragma unroll
      for (int c = 0; c < C; ++c)
        x[c] = *reinterpret_cast<const d2*>(smem + slot * SLOT + (c * T + tid) * 16);
      // (the "+s" operands make the loaded SGPRs outputs of the wait: the compiler, which believes the
      // load asm delivered them at once, can neither read nor recycle them before this point)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(yi_bits) : : "memory");
      if (has_rw) {
#pragma unroll
        for (int l = 0; l < B; ++l) asm volatile("" : "+s"(m_bits[l]));
      }
      const double yi = __longlong_as_double((long long)yi_bits);
      double m[B];
#pragma unroll
      for (int l = 0; l < B; ++l) m[l] = has_rw ? __longlong_as_double((long long)m_bits[l]) : 1.0;

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
#pragma unroll
      for (int l = 0; l < B; ++l) {
        const double e = dot[l] - yi;
        const double res = e * m[l];
        loss[l] = __builtin_fma(res, e, loss[l]);
#pragma unroll
        for (int c = 0; c < C; ++c) {
          acc[l][c].x = __builtin_fma(res, x[c].x, acc[l][c].x);
          acc[l][c].y = __builtin_fma(res, x[c].y, acc[l][c].y);
        }
      }
      slot = (slot == D) ? 0 : slot + 1;
      slot_in = (slot_in == D) ? 0 : slot_in + 1;
    }
  }

#pragma unroll
  for (int l = 0; l < B; ++l) {
    d2* out = reinterpret_cast<d2*>(a.partial + (b * B + l) * a.ld);
#pragma unroll
    for (int c = 0; c < C; ++c)
      if (valid[c]) out[c * T + tid] = acc[l][c];
    if (tid == 0) a.loss_partial[b * B + l] = loss[l];
  }
}

// ---------------------------------------------------------------------------------------------
// Two-pass fallback for rows too long for the fused kernels (p > 10 240): X is read twice, so it
// tops out near half the fused kernels' rate, but it has no limit on p.  One lane only.
//   pass A  rowdot_kernel:  r_i = w_i (x_i . z - y_i)   (one wavefront per row, z re-read from L2)
//   pass B  xtr_kernel<C>:  partial[blk][tile] = sum_{i in blk} r_i x_i[tile]
// Both use the same contiguous row blocks, so the reduce kernel sees the layout of the fused path.
// ---------------------------------------------------------------------------------------------
struct TwoPassArgs {
  const double* X;
  const double* y;
  const double* rw;  // nullptr or [n]
  const double* z;   // [ld]
  double* r;         // [n] weighted residuals
  double* partial;   // [gridDim.x][ld]
  double* loss_partial;
  const int* done;
  int64_t n, ld, rows_base, rows_rem;
  int p2;
  const int* skip = nullptr;
};

static __global__ __launch_bounds__(256) void rowdot_kernel(TwoPassArgs a) {
  if (a.done != nullptr && *a.done != 0) return;
  if (a.skip != nullptr && *a.skip != 0) return;
  __shared__ double lsum[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t b = blockIdx.x;
  const int64_t r0 = b * a.rows_base + (b < a.rows_rem ? b : a.rows_rem);
  const int64_t nrows = a.rows_base + (b < a.rows_rem ? 1 : 0);
  const d2* zv = reinterpret_cast<const d2*>(a.z);
  double loss = 0.0;
  for (int64_t i = wave; i < nrows; i += 4) {
    const int64_t row = r0 + i;
    const d2* xv = reinterpret_cast<const d2*>(a.X + row * a.ld);
    double t0 = 0.0, t1 = 0.0;
    int ci = lane;
    for (; ci + 64 < a.p2; ci += 128) {  // two independent chains
      const d2 xa = load_x(xv + ci), xb = load_x(xv + ci + 64);
      const d2 za = zv[ci], zb = zv[ci + 64];
      t0 = __builtin_fma(xa.x, za.x, t0);
      t0 = __builtin_fma(xa.y, za.y, t0);
      t1 = __builtin_fma(xb.x, zb.x, t1);
      t1 = __builtin_fma(xb.y, zb.y, t1);
    }
    if (ci < a.p2) {
      const d2 xa = load_x(xv + ci), za = zv[ci];
      t0 = __builtin_fma(xa.x, za.x, t0);
      t0 = __builtin_fma(xa.y, za.y, t0);
    }
    const double dot = wave_sum_lane63(t0 + t1);
    if (lane == 63) {
      const double e = dot - a.y[row];
      const double res = e * (a.rw != nullptr ? a.rw[row] : 1.0);
      a.r[row] = res;
      loss = __builtin_fma(res, e, loss);
    }
  }
  if (lane == 63) lsum[wave] = loss;
  __syncthreads();
  if (threadIdx.x == 0) a.loss_partial[b] = (lsum[0] + lsum[1]) + (lsum[2] + lsum[3]);
}

// grid = (row blocks, column tiles of 512*C chunks)
template <int C>
__global__ __launch_bounds__(512) void xtr_kernel(TwoPassArgs a) {
  if (a.done != nullptr && *a.done != 0) return;
  if (a.skip != nullptr && *a.skip != 0) return;
  constexpr int T = 512;
  const int tid = threadIdx.x;
  const int64_t b = blockIdx.x;
  const int64_t r0 = b * a.rows_base + (b < a.rows_rem ? b : a.rows_rem);
  const int64_t nrows = a.rows_base + (b < a.rows_rem ? 1 : 0);
  const int tile0 = blockIdx.y * (T * C);
  int cidx[C];
  bool valid[C];
  d2 acc[C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int ci = tile0 + c * T + tid;
    valid[c] = ci < a.p2;
    cidx[c] = valid[c] ? ci : a.p2 - 1;
    acc[c] = d2{0.0, 0.0};
  }
  int64_t i = 0;
  for (; i + 1 < nrows; i += 2) {  // two rows in flight
    const double ra = a.r[r0 + i], rb = a.r[r0 + i + 1];
    const d2* xa = reinterpret_cast<const d2*>(a.X + (r0 + i) * a.ld);
    const d2* xb = reinterpret_cast<const d2*>(a.X + (r0 + i + 1) * a.ld);
    d2 va[C], vb[C];
#pragma unroll
    for (int c = 0; c < C; ++c) va[c] = load_x(xa + cidx[c]);
#pragma unroll
    for (int c = 0; c < C; ++c) vb[c] = load_x(xb + cidx[c]);
#pragma unroll
    for (int c = 0; c < C; ++c) {
      acc[c].x = __builtin_fma(ra, va[c].x, acc[c].x);
      acc[c].y = __builtin_fma(ra, va[c].y, acc[c].y);
      acc[c].x = __builtin_fma(rb, vb[c].x, acc[c].x);
      acc[c].y = __builtin_fma(rb, vb[c].y, acc[c].y);
    }
  }
  if (i < nrows) {
    const double ra = a.r[r0 + i];
    const d2* xa = reinterpret_cast<const d2*>(a.X + (r0 + i) * a.ld);
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const d2 v = load_x(xa + cidx[c]);
      acc[c].x = __builtin_fma(ra, v.x, acc[c].x);
      acc[c].y = __builtin_fma(ra, v.y, acc[c].y);
    }
  }
  d2* out = reinterpret_cast<d2*>(a.partial + b * a.ld);
#pragma unroll
  for (int c = 0; c < C; ++c)
    if (valid[c]) out[cidx[c]] = acc[c];
}

// ---------------------------------------------------------------------------------------------
// Deterministic cross-workgroup reduction per lane:  g_l[j] = scale_l * sum_b partial[b][l][j], loss
// likewise.  256 threads = 16 column lanes x 16 row slices; one workgroup per 16 columns (128-byte
// segments).  The loss sum lands in g_l[ld] so that a single all-reduce covers gradient and loss in
// the row-sharded mode.
// ---------------------------------------------------------------------------------------------
struct ReduceArgs {
  const double* partial;       // [nblk][B][ld]
  const double* loss_partial;  // [nblk][B]
  double* g;                   // [B][ld + 16]
  const int* done;
  int nblk;
  int nblk_loss;  // row blocks of loss_partial (the split pass's residual kernels keep their own blocks)
  int n_lanes;
  int64_t ld;
  double scale[32];       // 1/n_eff per lane (SLM_MAX_LANES)
  double loss_scale[32];  // 1/(2 n_eff) per lane
  const int* skip = nullptr;  // non-null and *skip != 0: the gradients are there already (light_kernels.hpp): return at once
};

// grid = (ld/16 + 1, n_lanes)
static __global__ __launch_bounds__(256) void reduce_partials_kernel(ReduceArgs a) {
  if (a.done != nullptr && *a.done != 0) return;
  if (a.skip != nullptr && *a.skip != 0) return;
  __shared__ double lds[16][17];
  const int tid = threadIdx.x;
  const int cl = tid & 15, slice = tid >> 4;
  const int lane = blockIdx.y, B = a.n_lanes;
  const int64_t col = (int64_t)blockIdx.x * 16 + cl;
  const bool loss_block = (int64_t)blockIdx.x * 16 >= a.ld;  // the extra trailing block
  double s = 0.0;
  if (!loss_block) {
    for (int b = slice; b < a.nblk; b += 16) s += a.partial[((int64_t)b * B + lane) * a.ld + col];
  } else {
    for (int b = tid; b < a.nblk_loss; b += 256) s += a.loss_partial[(int64_t)b * B + lane];
  }
  lds[slice][cl] = s;
  __syncthreads();
  double* g = a.g + (int64_t)lane * (a.ld + 16);
  if (!loss_block) {
    if (slice == 0) {
      double t = 0.0;
#pragma unroll
      for (int k = 0; k < 16; ++k) t += lds[k][cl];
      g[col] = t * a.scale[lane];
    }
  } else if (tid == 0) {
    double t = 0.0;
    for (int k = 0; k < 16; ++k)
      for (int c = 0; c < 16; ++c) t += lds[k][c];
    g[a.ld] = t * a.loss_scale[lane];
  }
}

}  // namespace slm

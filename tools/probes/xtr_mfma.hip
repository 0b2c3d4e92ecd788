// Can X^T R for 16 / 32 / 48 residual columns ("lanes") stay on the HBM roofline when the products go
// through v_mfma_f64_16x16x4_f64 instead of vector FMAs?  (standalone probe; nothing in the library uses it)
//
//   G[col][lane] = sum_row X[row][col] * R[row][lane]
//
// One MFMA contracts 4 rows: A[i][k] = X[row_k][col_i] (16 columns), B[k][j] = R[row_k][lane_j] (16 lanes).
// Lane l of the wavefront holds A[i = l & 15][k = l >> 4], so one 16-byte load per lane brings rows
// row0 .. row0+3, 32 consecutive columns each (256 contiguous bytes per row); the two doubles of a lane feed
// two MFMAs (even / odd columns).  R is stored [n][16 * LT] so the B operand is one 8-byte load per lane.
//
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/xtr_mfma tools/probes/xtr_mfma.hip
// run:   tools/probes/xtr_mfma [n] [p2]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

#define CHECK(x)                                                                      \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      std::exit(1);                                                                   \
    }                                                                                 \
  } while (0)

#ifndef PROBE_WAVES
#define PROBE_WAVES 4
#endif
constexpr int WAVES = PROBE_WAVES;       // wavefronts per workgroup
constexpr int CW = 128;        // columns per wavefront (8 output tiles of 16)
constexpr int CB = WAVES * CW; // columns per workgroup

// LT: lane tiles (16 lanes each); U: 4-row steps per batch (two batches in flight)
template <int LT, int U>
__global__ __launch_bounds__(WAVES * 64, WAVES <= 4 ? 2 : 1) void xtr_mfma_kernel(const double* __restrict__ X, int64_t ld,
                                                                  const double* __restrict__ R, int64_t n,
                                                                  int rows_per_blk, double* __restrict__ part,
                                                                  int p2) {
  constexpr int RS = 16 * LT;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int col0 = ((int)blockIdx.x * WAVES + wave) * CW;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_blk;
  const int kq = lane >> 4, i16 = lane & 15;
  const int nb = rows_per_blk / (4 * U);  // host guarantees divisibility and r0 + rows_per_blk <= n
  d4 acc[8][LT];
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int t = 0; t < LT; ++t) acc[c][t] = d4{0.0, 0.0, 0.0, 0.0};
  const double* xp = X + (r0 + kq) * ld + col0 + 2 * i16;
  const double* rp = R + (r0 + kq) * RS + i16;
  d2 xa[U][4], xb[U][4];
  double ra[U][LT], rb[U][LT];
  auto load = [&](d2(&xv)[U][4], double(&rv)[U][LT], const double* xq, const double* rq) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int c = 0; c < 4; ++c) xv[u][c] = __builtin_nontemporal_load(reinterpret_cast<const d2*>(xq + (int64_t)u * 4 * ld + 32 * c));
#pragma unroll
      for (int t = 0; t < LT; ++t) rv[u][t] = rq[u * 4 * RS + 16 * t];
    }
  };
  auto compute = [&](d2(&xv)[U][4], double(&rv)[U][LT]) {
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int t = 0; t < LT; ++t) {
          acc[2 * c][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(xv[u][c].x, rv[u][t], acc[2 * c][t], 0, 0, 0);
          acc[2 * c + 1][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(xv[u][c].y, rv[u][t], acc[2 * c + 1][t], 0, 0, 0);
        }
  };
  load(xa, ra, xp, rp);
  for (int b = 0; b < nb; b += 2) {
    if (b + 1 < nb) load(xb, rb, xp + (int64_t)(b + 1) * 4 * U * ld, rp + (int64_t)(b + 1) * 4 * U * RS);
    compute(xa, ra);
    if (b + 2 < nb) load(xa, ra, xp + (int64_t)(b + 2) * 4 * U * ld, rp + (int64_t)(b + 2) * 4 * U * RS);
    if (b + 1 < nb) compute(xb, rb);
  }
  // result register r of lane l is D[i = (l >> 4) + 4 r][j = l & 15]; tile 2c+e holds columns col0 + 32 c + 2 i + e
  double* out = part + ((int64_t)blockIdx.y * p2 + col0) * RS;
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int t = 0; t < LT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int col = 32 * c + 2 * (kq + 4 * r) + e;
          out[(int64_t)col * RS + 16 * t + i16] = acc[2 * c + e][t][r];
        }
}


// ---------------------------------------------------------------------------------------------------------
// Variant: rows DMA'd into an LDS ring (global_load_lds_dwordx4: one instruction = 1 KiB contiguous of ONE row,
// no VGPR destination), MFMA operands read back from LDS in the A / B layouts.  Asks whether the 4 rows x 256 B
// request shape of the register variant above costs HBM efficiency against 1 KiB per request.  Each wavefront
// streams ITS OWN 128-column strip (1 KiB per row) and its own copy of the four residual rows of a step (512 B),
// so the ring needs no cross-wave sync, only counted vmcnt.  Q steps (4 rows each) in flight, ring of Q + 1 steps.
// ---------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int Q, int WGS_PER_CU>
__global__ __launch_bounds__(WAVES * 64, WGS_PER_CU) void xtr_lds_kernel(const double* __restrict__ X, int64_t ld,
                                                                         const double* __restrict__ R, int64_t n,
                                                                         int rows_per_blk, double* __restrict__ part,
                                                                         int p2) {
  constexpr int RS = 16;
  constexpr int SLOTS = Q + 1;                 // steps the ring holds
  constexpr int XSTEP = 4 * 1024;              // bytes of X per step and wave (4 rows x 128 columns)
  constexpr int RSTEP = 4 * RS * 8;            // bytes of R per step (4 rows x 16 lanes)
  constexpr int WAVE_BYTES = SLOTS * (XSTEP + RSTEP);
  __shared__ __attribute__((aligned(16))) char smem[WAVES * WAVE_BYTES];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int col0 = ((int)blockIdx.x * WAVES + wave) * CW;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_blk;
  const int kq = lane >> 4, i16 = lane & 15;
  const int nsteps = rows_per_blk / 4;
  char* xring = smem + wave * WAVE_BYTES;
  char* rring = xring + SLOTS * XSTEP;
  d4 acc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) acc[c] = d4{0.0, 0.0, 0.0, 0.0};
  const char* xg = reinterpret_cast<const char*>(X + r0 * ld + col0) + lane * 16;       // this lane's 16 bytes of a row
  const char* rg = reinterpret_cast<const char*>(R + r0 * RS) + (lane & 31) * 16;       // 32 lanes cover 4 rows of R
  auto issue = [&](int step, int slot) {  // 5 VMEM operations: R (half a wavefront), then the four rows
    if (lane < 32)
      __builtin_amdgcn_global_load_lds((gptr_t)(rg + (int64_t)step * RSTEP), (lptr_t)(rring + slot * RSTEP), 16, 0, 0);
#pragma unroll
    for (int k = 0; k < 4; ++k)
      __builtin_amdgcn_global_load_lds((gptr_t)(xg + ((int64_t)step * 4 + k) * ld * 8), (lptr_t)(xring + slot * XSTEP + k * 1024), 16, 0, 2);
  };
  auto consume = [&](int slot) {
    const double rv = *reinterpret_cast<const double*>(rring + slot * RSTEP + (kq * RS + i16) * 8);
    d2 xv[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) xv[c] = *reinterpret_cast<const d2*>(xring + slot * XSTEP + kq * 1024 + (32 * c + 2 * i16) * 8);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      acc[2 * c] = __builtin_amdgcn_mfma_f64_16x16x4f64(xv[c].x, rv, acc[2 * c], 0, 0, 0);
      acc[2 * c + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(xv[c].y, rv, acc[2 * c + 1], 0, 0, 0);
    }
  };
#pragma unroll
  for (int q = 0; q < Q; ++q)
    if (q < nsteps) issue(q, q);
  int slot = 0, slot_in = Q;
  for (int step = 0; step < nsteps; ++step) {
    const int left = nsteps - 1 - step;  // steps after this one
    if (left >= Q) {
      issue(step + Q, slot_in);
      wait_vmcnt<5 * Q>();
    } else if (Q >= 4 && left == 3) wait_vmcnt<(Q >= 4 ? 15 : 0)>();
    else if (Q >= 3 && left == 2) wait_vmcnt<(Q >= 3 ? 10 : 0)>();
    else if (Q >= 2 && left == 1) wait_vmcnt<(Q >= 2 ? 5 : 0)>();
    else wait_vmcnt<0>();
    consume(slot);
    // (the LDS reads of this step must have been issued before a later DMA may overwrite the slot: program order;
    //  the slot written next is the one consumed a whole ring turn ago)
    slot = (slot == Q) ? 0 : slot + 1;
    slot_in = (slot_in == Q) ? 0 : slot_in + 1;
  }
  double* out = part + ((int64_t)blockIdx.y * p2 + col0) * RS;
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int col = 32 * c + 2 * (kq + 4 * r) + e;
        out[(int64_t)col * RS + i16] = acc[2 * c + e][r];
      }
}

__global__ void reduce_kernel(const double* part, int nblk, int64_t count, double* G) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  double s = 0.0;
  for (int b = 0; b < nblk; ++b) s += part[(int64_t)b * count + i];
  G[i] = s;
}

__global__ void fill_kernel(double* v, int64_t count, uint64_t seed) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  uint64_t z = (uint64_t)i * 0x9E3779B97F4A7C15ull + seed;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  v[i] = (double)(int64_t)(z >> 11) * (1.0 / 9007199254740992.0) - 0.5;
}

// VAR 0: register double buffering (U steps per batch); VAR 1: LDS ring with Q = U steps in flight, 2 workgroups
// per CU; VAR 2: the same with one workgroup per CU
template <int LT, int U, int VAR = 0>
static void run(const double* X, int64_t ld, int64_t n, int p2, int nblk, bool verify) {
  constexpr int RS = 16 * LT;
  const int rows_per_blk = (int)(n / nblk);
  if ((int64_t)rows_per_blk * nblk != n || rows_per_blk % (VAR == 0 ? 4 * U : 4) != 0 || p2 % CB != 0 || ld < p2) {
    std::printf("LT=%d U=%d: shape not covered (n=%lld nblk=%d p2=%d)\n", LT, U, (long long)n, nblk, p2);
    return;
  }
  double *R, *part, *G;
  CHECK(hipMalloc(&R, n * RS * sizeof(double)));
  CHECK(hipMalloc(&part, (int64_t)nblk * p2 * RS * sizeof(double)));
  CHECK(hipMalloc(&G, (int64_t)p2 * RS * sizeof(double)));
  fill_kernel<<<(unsigned)((n * RS + 255) / 256), 256>>>(R, n * RS, 777);
  const dim3 grid(p2 / CB, nblk), block(WAVES * 64);
  const int64_t count = (int64_t)p2 * RS;
  auto kernel = [&]() {
    if constexpr (VAR == 0) xtr_mfma_kernel<LT, U><<<grid, block>>>(X, ld, R, n, rows_per_blk, part, p2);
    else if constexpr (VAR == 1) xtr_lds_kernel<U, 2><<<grid, block>>>(X, ld, R, n, rows_per_blk, part, p2);
    else xtr_lds_kernel<U, 1><<<grid, block>>>(X, ld, R, n, rows_per_blk, part, p2);
  };
  auto launch = [&]() {
    kernel();
    reduce_kernel<<<(unsigned)((count + 255) / 256), 256>>>(part, nblk, count, G);
  };
  launch();
  CHECK(hipDeviceSynchronize());
  if (verify) {
    std::vector<double> hX((size_t)n * ld), hR((size_t)n * RS), hG((size_t)count);
    CHECK(hipMemcpy(hX.data(), X, hX.size() * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(hR.data(), R, hR.size() * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(hG.data(), G, hG.size() * 8, hipMemcpyDeviceToHost));
    double worst = 0.0;
    for (int col = 0; col < p2; col += 7)
      for (int l = 0; l < RS; ++l) {
        double s = 0.0;
        for (int64_t i = 0; i < n; ++i) s += hX[(size_t)i * ld + col] * hR[(size_t)i * RS + l];
        worst = std::fmax(worst, std::fabs(s - hG[(size_t)col * RS + l]));
      }
    std::printf("VAR=%d LT=%d U=%d verify: max abs err %.3e (n=%lld)\n", VAR, LT, U, worst, (long long)n);
  } else {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int reps = 20;
    float ms_k = 0.f, ms_all = 0.f;
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) kernel();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms_k, e0, e1));
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms_all, e0, e1));
    const double bytes = 8.0 * ((double)n * p2 + (double)n * RS + (double)p2 * RS);
    std::printf("VAR=%d LT=%d (%2d lanes) U=%d nblk=%3d: kernel %.3f ms = %.0f GB/s (X only: %.0f GB/s), with reduce %.3f ms; %.1f TFLOP/s\n",
                VAR, LT, RS, U, nblk, ms_k / reps, bytes / (ms_k / reps) * 1e-6, 8.0 * n * p2 / (ms_k / reps) * 1e-6,
                ms_all / reps, 2.0 * n * p2 * RS / (ms_k / reps) * 1e-9);
  }
  CHECK(hipFree(R));
  CHECK(hipFree(part));
  CHECK(hipFree(G));
}

int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? std::atoll(argv[1]) : 100000;
  const int p2 = argc > 2 ? std::atoi(argv[2]) : 5120;
  const int64_t ld = argc > 4 ? std::atoll(argv[4]) : p2;  // leading dimension (>= p2): 40 960-byte rows alias in HBM
  {  // small verification problem first
    const int64_t nv = 1600;
    double* Xv;
    CHECK(hipMalloc(&Xv, nv * ld * sizeof(double)));
    fill_kernel<<<(unsigned)((nv * ld + 255) / 256), 256>>>(Xv, nv * ld, 1);
    run<1, 2>(Xv, ld, nv, p2, 4, true);
    run<2, 2>(Xv, ld, nv, p2, 4, true);
    run<3, 1>(Xv, ld, nv, p2, 2, true);
    run<2, 1>(Xv, ld, nv, p2, 2, true);
#if PROBE_WAVES == 4
    run<1, 3, 1>(Xv, ld, nv, p2, 4, true);
    run<1, 2, 1>(Xv, ld, nv, p2, 2, true);
    run<1, 4, 2>(Xv, ld, nv, p2, 4, true);
#endif
    CHECK(hipFree(Xv));
  }
  double* X;
  CHECK(hipMalloc(&X, n * ld * sizeof(double)));
  fill_kernel<<<(unsigned)((n * ld + 255) / 256), 256>>>(X, n * ld, 1);
  CHECK(hipDeviceSynchronize());
  if (argc > 3) {  // A/B of the request shapes: the production mapping against the LDS ring, same rows per block
    for (int rep = 0; rep < 2; ++rep)
      for (int nblk : {48, 24, 32, 16}) {
        run<1, 2, 0>(X, ld, n, p2, nblk, false);
#if PROBE_WAVES == 4
        run<1, 3, 1>(X, ld, n, p2, nblk, false);
        run<1, 2, 1>(X, ld, n, p2, nblk, false);
        run<1, 4, 2>(X, ld, n, p2, nblk, false);
#endif
      }
    CHECK(hipFree(X));
    return 0;
  }
  for (int nblk : {50, 25, 100}) {
    run<1, 1>(X, ld, n, p2, nblk, false);
    run<1, 2>(X, ld, n, p2, nblk, false);
    run<2, 2>(X, ld, n, p2, nblk, false);
    run<3, 1>(X, ld, n, p2, nblk, false);
    run<2, 1>(X, ld, n, p2, nblk, false);
    run<1, 4>(X, ld, n, p2, nblk, false);
  }
  CHECK(hipFree(X));
  return 0;
}

// How does the accumulate-only pass scale with the number of lane slots?  (standalone probe of the
// vector-FMA kernel the library used before xtr_mfma_kernel: rows DMA'd through an LDS ring, residuals
// by scalar loads)
#include "../../sparse-lm_amd/csrc/split_kernels.hpp"
#include <cstdio>
#include <cstring>
#include <vector>
using namespace slm;

// 64-byte / 16-byte scalar loads: eight / two of the residuals of one row.  The caller waits
// lgkmcnt(0) before use.  EVERY element of the result must be used: the compiler treats the asm output
// as available at once and recycles registers of the tuple it considers dead while the load is still
// in flight (a 64-byte load for lanes 8-9 had its upper SGPRs reused for the LDS ring address: the
// DMA then went to whatever the load wrote there).
typedef uint32_t slm_u32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t slm_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ slm_u32x16 smem_load_64B(const double* p) {
  slm_u32x16 v;
  asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(v) : "s"(p) : "memory");
  return v;
}
__device__ __forceinline__ slm_u32x4 smem_load_16B(const double* p) {
  slm_u32x4 v;
  asm volatile("s_load_dwordx4 %0, %1, 0x0" : "=s"(v) : "s"(p) : "memory");
  return v;
}



// same loop as xtr_ring_kernel, any B <= 10 (the residual row is always loaded in full)
template <int W, int C, int B, int D, int AUX = 2>
__global__ __launch_bounds__(W * 64) void xtr_probe_kernel(SplitArgs a) {
  constexpr int T = W * 64;
  constexpr int SLOT = T * C * 16;
  constexpr int RING = (D + 1) * SLOT;
  __shared__ __attribute__((aligned(16))) char smem[RING];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t b = blockIdx.x;
  const int64_t r0 = b * a.rows_base + (b < a.rows_rem ? b : a.rows_rem);
  const int64_t nrows = a.rows_base + (b < a.rows_rem ? 1 : 0);
  uint32_t coff[C];
  d2 acc[B][C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int ci = c * T + tid;
    coff[c] = (uint32_t)(ci < a.p2 ? ci : a.p2 - 1) * 16u;
#pragma unroll
    for (int l = 0; l < B; ++l) acc[l][c] = d2{0.0, 0.0};
  }
  auto issue_row = [&](int64_t i, int slot) {
    const char* rp = reinterpret_cast<const char*>(a.X + (r0 + i) * a.ld);
#pragma unroll
    for (int c = 0; c < C; ++c) {
      char* dst = smem + slot * SLOT + (c * T + wave * 64) * 16;
      __builtin_amdgcn_global_load_lds((gptr_t)(rp + coff[c]), (lptr_t)dst, 16, 0, AUX);
    }
  };
  for (int k = 0; k < D; ++k) if (k < nrows) issue_row(k, k);
  int slot = 0, slot_in = D;
  for (int64_t i = 0; i < nrows; ++i) {
    slm_u32x16 ra = smem_load_64B(a.R + (r0 + i) * SPLIT_RSTRIDE);
    slm_u32x4 rb = smem_load_16B(a.R + (r0 + i) * SPLIT_RSTRIDE + 8);
    const int64_t left = nrows - 1 - i;
    if (left >= D) { issue_row(i + D, slot_in); wait_vmcnt<D * C>(); } else wait_vmcnt<0>();
    d2 x[C];
#pragma unroll
    for (int c = 0; c < C; ++c) x[c] = *reinterpret_cast<const d2*>(smem + slot * SLOT + (c * T + tid) * 16);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(ra), "+s"(rb) : : "memory");
#pragma unroll
    for (int l = 0; l < B; ++l) {
      const double res = l < 8 ? __hiloint2double((int)ra[2 * (l & 7) + 1], (int)ra[2 * (l & 7)])
                               : __hiloint2double((int)rb[2 * (l & 7) + 1], (int)rb[2 * (l & 7)]);
#pragma unroll
      for (int c = 0; c < C; ++c) {
        acc[l][c].x = __builtin_fma(res, x[c].x, acc[l][c].x);
        acc[l][c].y = __builtin_fma(res, x[c].y, acc[l][c].y);
      }
    }
    slot = (slot == D) ? 0 : slot + 1;
    slot_in = (slot_in == D) ? 0 : slot_in + 1;
  }
#pragma unroll
  for (int l = 0; l < B; ++l) {
    d2* out = reinterpret_cast<d2*>(a.partial + (b * B + l) * a.ld);
#pragma unroll
    for (int c = 0; c < C; ++c)
      if (c * T + tid < a.p2) out[c * T + tid] = acc[l][c];
  }
}

// variant: wait for and consume one 16-byte chunk column at a time instead of the whole row
template <int C, int K> struct ChunkWait {
  template <int D> static __device__ __forceinline__ void go() { wait_vmcnt<D * C + (C - 1 - K)>(); }
};
template <int W, int C, int B, int D>
__global__ __launch_bounds__(W * 64) void xtr_chunk_kernel(SplitArgs a) {
  constexpr int T = W * 64;
  constexpr int SLOT = T * C * 16;
  constexpr int RING = (D + 1) * SLOT;
  __shared__ __attribute__((aligned(16))) char smem[RING];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t b = blockIdx.x;
  const int64_t r0 = b * a.rows_base + (b < a.rows_rem ? b : a.rows_rem);
  const int64_t nrows = a.rows_base + (b < a.rows_rem ? 1 : 0);
  uint32_t coff[C];
  d2 acc[B][C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const int ci = c * T + tid;
    coff[c] = (uint32_t)(ci < a.p2 ? ci : a.p2 - 1) * 16u;
#pragma unroll
    for (int l = 0; l < B; ++l) acc[l][c] = d2{0.0, 0.0};
  }
  auto issue_row = [&](int64_t i, int slot) {
    const char* rp = reinterpret_cast<const char*>(a.X + (r0 + i) * a.ld);
#pragma unroll
    for (int c = 0; c < C; ++c) {
      char* dst = smem + slot * SLOT + (c * T + wave * 64) * 16;
      __builtin_amdgcn_global_load_lds((gptr_t)(rp + coff[c]), (lptr_t)dst, 16, 0, 2);
    }
  };
  for (int k = 0; k < D; ++k) if (k < nrows) issue_row(k, k);
  int slot = 0, slot_in = D;
  for (int64_t i = 0; i < nrows; ++i) {
    slm_u32x16 ra = smem_load_64B(a.R + (r0 + i) * SPLIT_RSTRIDE);
    slm_u32x4 rb = smem_load_16B(a.R + (r0 + i) * SPLIT_RSTRIDE + 8);
    const int64_t left = nrows - 1 - i;
    const bool steady = left >= D;
    if (steady) issue_row(i + D, slot_in);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(ra), "+s"(rb) : : "memory");
    double res[B];
#pragma unroll
    for (int l = 0; l < B; ++l)
      res[l] = l < 8 ? __hiloint2double((int)ra[2 * (l & 7) + 1], (int)ra[2 * (l & 7)])
                     : __hiloint2double((int)rb[2 * (l & 7) + 1], (int)rb[2 * (l & 7)]);
#pragma unroll
    for (int c = 0; c < C; ++c) {
      if (steady) {
        if (c == 0) wait_vmcnt<D * C + (C - 1)>();
        if (c == 1) wait_vmcnt<D * C + (C - 2 > 0 ? C - 2 : 0)>();
        if (c == 2) wait_vmcnt<D * C + (C - 3 > 0 ? C - 3 : 0)>();
        if (c == 3) wait_vmcnt<D * C + (C - 4 > 0 ? C - 4 : 0)>();
        if (c >= 4) wait_vmcnt<D * C>();
      } else {
        wait_vmcnt<0>();
      }
      const d2 x = *reinterpret_cast<const d2*>(smem + slot * SLOT + (c * T + tid) * 16);
#pragma unroll
      for (int l = 0; l < B; ++l) {
        acc[l][c].x = __builtin_fma(res[l], x.x, acc[l][c].x);
        acc[l][c].y = __builtin_fma(res[l], x.y, acc[l][c].y);
      }
    }
    slot = (slot == D) ? 0 : slot + 1;
    slot_in = (slot_in == D) ? 0 : slot_in + 1;
  }
#pragma unroll
  for (int l = 0; l < B; ++l) {
    d2* out = reinterpret_cast<d2*>(a.partial + (b * B + l) * a.ld);
#pragma unroll
    for (int c = 0; c < C; ++c)
      if (c * T + tid < a.p2) out[c * T + tid] = acc[l][c];
  }
}

static void runchunk(const SplitArgs& a, int nblk) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((xtr_chunk_kernel<8, 5, 10, 2>), dim3(nblk), dim3(512), 0, 0, a);
  (void)hipEventRecord(e0, 0);
  for (int r = 0; r < 30; ++r) hipLaunchKernelGGL((xtr_chunk_kernel<8, 5, 10, 2>), dim3(nblk), dim3(512), 0, 0, a);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("chunk-wise B=10 D=2: %.4f ms  %.0f GB/s\n", ms / 30, 8.0 * a.n * (a.ld) / (ms / 30) / 1e6);
}

template <int W, int C, int B, int D>
static void runw(const SplitArgs& a, int nblk, const char* tag) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((xtr_probe_kernel<W, C, B, D>), dim3(nblk), dim3(W * 64), 0, 0, a);
  (void)hipEventRecord(e0, 0);
  for (int r = 0; r < 30; ++r) hipLaunchKernelGGL((xtr_probe_kernel<W, C, B, D>), dim3(nblk), dim3(W * 64), 0, 0, a);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%s W=%d C=%d B=%d D=%d: %.4f ms  %.0f GB/s\n", tag, W, C, B, D, ms / 30, 8.0 * a.n * (a.ld) / (ms / 30) / 1e6);
}

template <int AUX>
static void runaux(const SplitArgs& a, int nblk) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((xtr_probe_kernel<8, 5, 10, 2, AUX>), dim3(nblk), dim3(512), 0, 0, a);
  (void)hipEventRecord(e0, 0);
  for (int r = 0; r < 30; ++r) hipLaunchKernelGGL((xtr_probe_kernel<8, 5, 10, 2, AUX>), dim3(nblk), dim3(512), 0, 0, a);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("aux=%d B=10 D=2: %.4f ms  %.0f GB/s\n", AUX, ms / 30, 8.0 * a.n * (a.ld) / (ms / 30) / 1e6);
}

template <int B, int D>
static void run(const SplitArgs& a, int nblk, const char* tag) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((xtr_probe_kernel<8, 5, B, D>), dim3(nblk), dim3(512), 0, 0, a);
  (void)hipEventRecord(e0, 0);
  for (int r = 0; r < 30; ++r) hipLaunchKernelGGL((xtr_probe_kernel<8, 5, B, D>), dim3(nblk), dim3(512), 0, 0, a);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%s B=%d D=%d: %.4f ms  %.0f GB/s\n", tag, B, D, ms / 30, 8.0 * a.n * (a.ld) / (ms / 30) / 1e6);
}

int main() {
  const int64_t n = 100000, ld = 5008;
  double *X, *R, *P;
  (void)hipMalloc(&X, n * ld * 8); (void)hipMalloc(&R, n * 16 * 8); (void)hipMalloc(&P, (size_t)512 * 10 * ld * 8);  // largest grid x most lanes
  (void)hipMemset(X, 0, n * ld * 8); (void)hipMemset(R, 0, n * 16 * 8);
  SplitArgs a; memset(&a, 0, sizeof(a));
  a.X = X; a.R = R; a.partial = P; a.n = n; a.ld = ld; a.p2 = (int)(ld / 2);
  for (int nblk : {256}) {
    a.rows_base = n / nblk; a.rows_rem = n % nblk;
    printf("nblk=%d\n", nblk);
    runaux<0>(a, nblk); runaux<1>(a, nblk); runaux<2>(a, nblk); runaux<3>(a, nblk);
    runchunk(a, nblk); runaux<2>(a, nblk); runchunk(a, nblk);
  }
  return 0;
}

// Reading K of every LD doubles of a row-major n x LD matrix (the gathered columns XW, LD = 512): what does the row stride cost?
// A wavefront per row, lane per column chunk -- ws_xty_partial_kernel's pattern -- for K in {96, 176, 256} at LD = 512, 256, K.
// build: hipcc --offload-arch=gfx950 -O3 xw_stride.hip -o xw_stride
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(512) void read_rows(const double* X, long n, int ld, int K, double* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long rows = (n + gridDim.x - 1) / gridDim.x;
  const long i0 = blockIdx.x * rows, i1 = i0 + rows < n ? i0 + rows : n;
  double acc[4] = {0, 0, 0, 0};
#pragma unroll 4
  for (long i = i0 + wave; i < i1; i += 8)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int k = lane + 64 * c;
      if (64 * c < K) acc[c] += k < K ? X[i * ld + k] : 0.0;
    }
  double s = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  if (s == 12345.678) out[0] = s;
}
int main() {
  const long n = 100000;
  double* X; double* out;
  hipMalloc(&X, sizeof(double) * n * 512);
  hipMalloc(&out, 64);
  hipMemset(X, 0, sizeof(double) * n * 512);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int K : {96, 176, 256}) {
    for (int ld : {512, 256, K}) {
      if (ld < K) continue;
      for (int grid : {512, 1024}) {
        for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(read_rows, dim3(grid), dim3(512), 0, 0, X, n, ld, K, out);
        hipEventRecord(a);
        for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(read_rows, dim3(grid), dim3(512), 0, 0, X, n, ld, K, out);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("K %3d ld %3d grid %4d: %7.1f us per read, %6.2f TB/s of useful bytes\n", K, ld, grid, 1e3 * ms / 20, n * K * 8.0 / (ms / 20 * 1e-3) / 1e12);
      }
    }
  }
  return 0;
}

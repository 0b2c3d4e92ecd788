// ws_xty_partial_kernel's loop in isolation (K = 96, 256 of 512 columns, n = 100 000) against variants: what keeps it at 2.5 TB/s?
// build: hipcc --offload-arch=gfx950 -O3 xty_probe.hip -o xty_probe
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int KCAP = 512;
template <int VARIANT>
__global__ __launch_bounds__(512) void xty(const double* XW, const double* y, long n, const int* Kp, double* part) {
  __shared__ double red[8][KCAP];
  __shared__ double red_yy[8];
  const int K = *Kp;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long rows = (n + gridDim.x - 1) / gridDim.x;
  const long i0 = (long)blockIdx.x * rows, i1 = i0 + rows < n ? i0 + rows : n;
  double acc[KCAP / 64];
#pragma unroll
  for (int c = 0; c < KCAP / 64; ++c) acc[c] = 0.0;
  double yy = 0.0;
  if (VARIANT == 0) {  // as in the engine
#pragma unroll 4
    for (long i = i0 + wave; i < i1; i += 8) {
      const double yi = y[i];
      yy = __builtin_fma(yi, yi, yy);
#pragma unroll
      for (int c = 0; c < KCAP / 64; ++c) {
        const int k = lane + 64 * c;
        if (64 * c < K) acc[c] = __builtin_fma(k < K ? XW[i * KCAP + k] : 0.0, yi, acc[c]);
      }
    }
  } else {  // the number of chunks decided ONCE, outside the row loop: loads of four rows issued together
    const int nc = (K + 63) >> 6;
    auto body = [&](auto NC) {
      constexpr int C = decltype(NC)::value;
      for (long i = i0 + wave; i < i1; i += 32) {
        double xv[4][C], yv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const long ii = i + 8 * r < i1 ? i + 8 * r : i;
          yv[r] = i + 8 * r < i1 ? y[ii] : 0.0;
#pragma unroll
          for (int c = 0; c < C; ++c) {
            const int k = lane + 64 * c;
            xv[r][c] = XW[ii * KCAP + (k < K ? k : 0)];
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          yy = __builtin_fma(yv[r], yv[r], yy);
#pragma unroll
          for (int c = 0; c < C; ++c) acc[c] = __builtin_fma(lane + 64 * c < K ? xv[r][c] : 0.0, yv[r], acc[c]);
        }
      }
    };
    switch (nc) {
      case 1: body(std::integral_constant<int, 1>{}); break;
      case 2: body(std::integral_constant<int, 2>{}); break;
      case 3: body(std::integral_constant<int, 3>{}); break;
      case 4: body(std::integral_constant<int, 4>{}); break;
      case 5: body(std::integral_constant<int, 5>{}); break;
      case 6: body(std::integral_constant<int, 6>{}); break;
      case 7: body(std::integral_constant<int, 7>{}); break;
      default: body(std::integral_constant<int, 8>{}); break;
    }
  }
#pragma unroll
  for (int c = 0; c < KCAP / 64; ++c) red[wave][lane + 64 * c] = acc[c];
  if (lane == 0) red_yy[wave] = yy;
  __syncthreads();
  double* out = part + (long)blockIdx.x * (KCAP + 1);
  const int k = threadIdx.x;
  if (k < K) {
    double sum = 0.0;
#pragma unroll
    for (int w = 0; w < 8; ++w) sum += red[w][k];
    out[k] = sum;
  }
  if (k == KCAP - 1) {
    double sum = 0.0;
    for (int w = 0; w < 8; ++w) sum += red_yy[w];
    out[KCAP] = sum;
  }
}
template <int V>
void run(const double* X, const double* y, long n, int* Kd, double* part, int K, int grid) {
  hipMemcpy(Kd, &K, 4, hipMemcpyHostToDevice);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(xty<V>, dim3(grid), dim3(512), 0, 0, X, y, n, Kd, part);
  hipEventRecord(a);
  for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(xty<V>, dim3(grid), dim3(512), 0, 0, X, y, n, Kd, part);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("variant %d K %3d grid %4d: %7.1f us, %5.2f TB/s\n", V, K, grid, 1e3 * ms / 20, n * K * 8.0 / (ms / 20 * 1e-3) / 1e12);
}
int main() {
  const long n = 100000;
  double *X, *y, *part; int* Kd;
  hipMalloc(&X, sizeof(double) * n * KCAP); hipMalloc(&y, sizeof(double) * n); hipMalloc(&part, sizeof(double) * 2048 * (KCAP + 1)); hipMalloc(&Kd, 4);
  hipMemset(X, 0, sizeof(double) * n * KCAP); hipMemset(y, 0, sizeof(double) * n);
  for (int K : {96, 176, 256, 384})
    for (int grid : {512, 1024}) {
      run<0>(X, y, n, Kd, part, K, grid);
      run<1>(X, y, n, Kd, part, K, grid);
    }
  return 0;
}

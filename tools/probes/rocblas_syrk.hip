// What the BLAS library of the box charges for a full Gram at the headline shape, first call included:
// C = A A^T for the row-major n x ld matrix X read as the column-major ld x n matrix A (rocblas_dsyrk / dgemm).
// hipcc --offload-arch=gfx950 -O2 tools/probes/rocblas_syrk.hip -lrocblas -o /tmp/rocblas_syrk
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void fill(double* x, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    x[i] = (double)((i * 2654435761u) & 0xffff) / 65536.0 - 0.5;
}
int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 100000, p = argc > 2 ? atoi(argv[2]) : 5000, ld = (p + 15) / 16 * 16;
  double *X, *G;
  hipMalloc(&X, sizeof(double) * (size_t)n * ld);
  hipMalloc(&G, sizeof(double) * (size_t)ld * ld);
  fill<<<1024, 256>>>(X, (size_t)n * ld);
  hipDeviceSynchronize();
  double t0 = now();
  rocblas_handle h;
  rocblas_create_handle(&h);
  printf("create_handle %.1f ms\n", 1e3 * (now() - t0));
  const double one = 1.0, zero = 0.0;
  for (int rep = 0; rep < 4; ++rep) {
    t0 = now();
    rocblas_status st = rocblas_dsyrk(h, rocblas_fill_lower, rocblas_operation_none, ld, n, &one, X, ld, &zero, G, ld);
    hipDeviceSynchronize();
    printf("dsyrk   rep %d: %.1f ms (status %d)\n", rep, 1e3 * (now() - t0), (int)st);
  }
  for (int rep = 0; rep < 3; ++rep) {
    t0 = now();
    rocblas_status st = rocblas_dgemm(h, rocblas_operation_none, rocblas_operation_transpose, ld, ld, n, &one, X, ld, X, ld, &zero, G, ld);
    hipDeviceSynchronize();
    printf("dgemm   rep %d: %.1f ms (status %d)\n", rep, 1e3 * (now() - t0), (int)st);
  }
  const int nb = n / 5;
  for (int rep = 0; rep < 3; ++rep) {
    t0 = now();
    rocblas_dgemm(h, rocblas_operation_none, rocblas_operation_transpose, ld, ld, nb, &one, X, ld, X, ld, &zero, G, ld);
    hipDeviceSynchronize();
    printf("dgemm on a fifth of the rows rep %d: %.1f ms\n", rep, 1e3 * (now() - t0));
  }
  rocblas_destroy_handle(h);
  return 0;
}

// Periodic host-side stall probe: batches of tiny launches + a stream sync; prints outlier batches.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
__global__ void tiny(int* p) { if (p && threadIdx.x == 12345) *p = 1; }
int main(int argc, char** argv) {
  const int batch = argc > 1 ? atoi(argv[1]) : 128, reps = argc > 2 ? atoi(argv[2]) : 400;
  const int nonblocking = argc > 3 ? atoi(argv[3]) : 1;
  hipStream_t s;
  hipStreamCreateWithFlags(&s, nonblocking ? hipStreamNonBlocking : hipStreamDefault);
  int* d; hipMalloc(&d, 4);
  long launches = 0; double worst = 0; int n_out = 0;
  for (int r = 0; r < reps; ++r) {
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < batch; ++i) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, d);
    hipStreamSynchronize(s);
    double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    launches += batch;
    if (ms > 10.0) { printf("batch %d (launch %ld): %.2f ms\n", r, launches, ms); ++n_out; }
    if (ms > worst) worst = ms;
  }
  printf("batch=%d reps=%d outliers=%d worst=%.2f ms\n", batch, reps, n_out, worst);
  return 0;
}

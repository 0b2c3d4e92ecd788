// What a kernel of the chain between two passes costs the stream beyond its own code: 16 workgroups of 1024
// threads that do (next to) nothing, with and without per-lane scratch and a large static LDS block, launched
// back to back (stream time per launch from events), and the same after a kernel that sweeps 1 GB of HBM.
//   hipcc --offload-arch=gfx950 -O3 -o dispatch_cost dispatch_cost.hip && ./dispatch_cost
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));             \
      return 1;                                                                \
    }                                                                          \
  } while (0)

template <int SCRATCH_DOUBLES, int LDS_DOUBLES, bool TOUCH = true>
__global__ __launch_bounds__(1024) void probe_kernel(double* out, const int* sel, unsigned long long* ticks) {
  const unsigned long long t0 = wall_clock64();
  __shared__ double lds[LDS_DOUBLES > 0 ? LDS_DOUBLES : 1];
  double acc = 0.0;
  if (LDS_DOUBLES > 0) {
    lds[threadIdx.x % LDS_DOUBLES] = threadIdx.x;
    __syncthreads();
    acc += lds[(threadIdx.x * 7) % LDS_DOUBLES];
  }
  if (SCRATCH_DOUBLES > 0 && (TOUCH || sel[3] == 12345)) {  // (!TOUCH: allocated, never written)
    double priv[SCRATCH_DOUBLES > 0 ? SCRATCH_DOUBLES : 1];
#pragma unroll 1
    for (int i = 0; i < SCRATCH_DOUBLES; ++i) priv[i] = i * 0.5 + threadIdx.x;
    acc += priv[sel[threadIdx.x & 3] % SCRATCH_DOUBLES];  // (dynamic index: the array stays in scratch)
  }
  if (acc == -1.0) out[threadIdx.x] = acc;
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] += wall_clock64() - t0;
}

__global__ __launch_bounds__(256) void sweep_kernel(const double* x, double* out, size_t n) {
  double acc = 0.0;
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += x[i];
  if (acc == -1.0) out[0] = acc;
}

template <typename F>
static int timed(const char* what, hipStream_t s, F launch, int reps, unsigned long long* d_ticks, F* between = nullptr) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 5; ++i) launch();
  CK(hipStreamSynchronize(s));
  CK(hipMemsetAsync(d_ticks, 0, 8, s));
  CK(hipEventRecord(e0, s));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1, s));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long tk = 0;
  CK(hipMemcpy(&tk, d_ticks, 8, hipMemcpyDeviceToHost));
  std::printf("%-58s %7.2f us of stream per launch, %6.2f us inside the kernel\n", what, 1e3 * ms / reps, tk * 1e-2 / reps);
  return 0;
}

int main() {
  hipStream_t s;
  CK(hipStreamCreate(&s));
  double* out;
  int* sel;
  unsigned long long* ticks;
  double* big;
  const size_t nbig = 128ull << 20;  // 1 GB
  CK(hipMalloc(&out, 8 * 1024));
  CK(hipMalloc(&sel, 16));
  CK(hipMalloc(&ticks, 8));
  CK(hipMalloc(&big, nbig * 8));
  CK(hipMemset(big, 0, nbig * 8));
  CK(hipMemset(sel, 0, 16));
  const dim3 g(16), b(1024);
  auto k00 = [&] { hipLaunchKernelGGL((probe_kernel<0, 0>), g, b, 0, s, out, sel, ticks); };
  auto k10 = [&] { hipLaunchKernelGGL((probe_kernel<62, 0>), g, b, 0, s, out, sel, ticks); };
  auto k01 = [&] { hipLaunchKernelGGL((probe_kernel<0, 16384>), g, b, 0, s, out, sel, ticks); };
  auto k11 = [&] { hipLaunchKernelGGL((probe_kernel<62, 16384>), g, b, 0, s, out, sel, ticks); };
  auto k20 = [&] { hipLaunchKernelGGL((probe_kernel<400, 0>), g, b, 0, s, out, sel, ticks); };
  auto k10n = [&] { hipLaunchKernelGGL((probe_kernel<62, 0, false>), g, b, 0, s, out, sel, ticks); };
  auto k20n = [&] { hipLaunchKernelGGL((probe_kernel<400, 0, false>), g, b, 0, s, out, sel, ticks); };
  if (timed("496 bytes of scratch per lane, never touched", s, k10n, 200, ticks)) return 1;
  if (timed("3 200 bytes of scratch per lane, never touched", s, k20n, 200, ticks)) return 1;
  for (int rep = 0; rep < 2; ++rep) {
    if (timed("nothing", s, k00, 200, ticks)) return 1;
    if (timed("496 bytes of scratch per lane", s, k10, 200, ticks)) return 1;
    if (timed("128 KB of LDS", s, k01, 200, ticks)) return 1;
    if (timed("496 bytes of scratch per lane + 128 KB of LDS", s, k11, 200, ticks)) return 1;
    if (timed("3 200 bytes of scratch per lane", s, k20, 200, ticks)) return 1;
  }
  // alternating with a kernel that needs no scratch (does the queue's scratch state toggle?) and with a sweep of HBM
  auto alt0 = [&] { k00(); k11(); };
  if (timed("nothing, then scratch + LDS (per pair)", s, alt0, 100, ticks)) return 1;
  auto sw = [&] { hipLaunchKernelGGL(sweep_kernel, dim3(2048), dim3(256), 0, s, big, out, nbig); };
  auto sw_only = [&] { sw(); };
  auto sw00 = [&] { sw(); k00(); };
  auto sw11 = [&] { sw(); k11(); };
  if (timed("1 GB sweep", s, sw_only, 20, ticks)) return 1;
  if (timed("1 GB sweep, then nothing (per pair)", s, sw00, 20, ticks)) return 1;
  if (timed("1 GB sweep, then scratch + LDS (per pair)", s, sw11, 20, ticks)) return 1;
  return 0;
}

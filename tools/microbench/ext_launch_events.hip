// What do the events of hipExtLaunchKernelGGL cost a stream of back-to-back kernels?  200 launches of a ~10 us kernel per
// trial: plain launches; every launch with a stop event; with a start and a stop event; with a start event only; plain launches
// with a hipEventRecord behind each.  Printed: us per launch (one HIP-event pair around the 200).
// Build: hipcc -O3 --offload-arch=gfx950 tools/microbench/ext_launch_events.hip -o tools/microbench/ext_launch_events
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdio>
#include <vector>

__global__ void k_work(float* p, int n, int iters) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = p[i];
  for (int k = 0; k < iters; ++k) v = __builtin_fmaf(v, 1.0000001f, 1e-9f);
  p[i] = v;
}

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e__ = (x);                                                      \
    if (e__ != hipSuccess) {                                                   \
      printf("%s: %s\n", #x, hipGetErrorString(e__));                          \
      return 1;                                                                \
    }                                                                          \
  } while (0)

int main() {
  const int n = 131072, iters = 2400, L = 200;
  float* a;
  CK(hipMalloc(&a, n * 4));
  CK(hipMemset(a, 0, n * 4));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  std::vector<hipEvent_t> ev(2 * L);
  for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableSystemFence));
  hipEvent_t t0, t1;
  CK(hipEventCreate(&t0));
  CK(hipEventCreate(&t1));
  const dim3 g(n / 256), b(256);
  const char* names[] = {"plain", "stop event", "start + stop", "start event", "plain + record"};
  for (int mode = 0; mode < 5; ++mode) {
    float best = 1e9f, span = 0.f;
    for (int rep = 0; rep < 6; ++rep) {
      CK(hipStreamSynchronize(s));
      CK(hipEventRecord(t0, s));
      for (int i = 0; i < L; ++i) {
        if (mode == 0 || mode == 4)
          hipLaunchKernelGGL(k_work, g, b, 0, s, a, n, iters);
        else
          hipExtLaunchKernelGGL(k_work, g, b, 0, s, (mode == 2 || mode == 3) ? ev[2 * i] : nullptr, (mode == 1 || mode == 2) ? ev[2 * i + 1] : nullptr, 0,
                                a, n, iters);
        if (mode == 4) CK(hipEventRecord(ev[2 * i + 1], s));
      }
      CK(hipEventRecord(t1, s));
      CK(hipEventSynchronize(t1));
      float ms = 0.f;
      CK(hipEventElapsedTime(&ms, t0, t1));
      best = ms < best ? ms : best;
      if (mode == 1 || mode == 2) CK(hipEventElapsedTime(&span, ev[1], ev[2 * L - 1]));  // first stop -> last stop
    }
    printf("%-16s %7.2f us per launch", names[mode], best * 1e3f / L);
    if (mode == 1 || mode == 2) printf("   (first stop -> last stop: %7.2f us per launch)", span * 1e3f / (L - 1));
    printf("\n");
  }
  // does a stop event know its kernel's duration?  elapsed(start event, stop event) of ONE launch against the kernel's time
  CK(hipStreamSynchronize(s));
  hipExtLaunchKernelGGL(k_work, g, b, 0, s, ev[0], ev[1], 0, a, n, iters);
  CK(hipEventSynchronize(ev[1]));
  float d = 0.f;
  CK(hipEventElapsedTime(&d, ev[0], ev[1]));
  printf("one launch, elapsed(start event, stop event) = %.2f us\n", d * 1e3f);
  return 0;
}

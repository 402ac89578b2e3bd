// What does it cost the MAIN stream to let a SIDE stream start work behind one of its kernels?  (The overlapped return gather:
// snapshot kernel on the main stream -> exchange on the side stream; tools/episode_end_cost.py books 17-20 us per episode for it
// at 131 072 envs where the 512 KB copy itself costs 2 us in line.)  One iteration on the main stream = a long kernel (the
// episode's last rollout launch), a short kernel (the snapshot), the hand-over, another kernel (the reset); the side stream runs a
// short kernel behind the hand-over.  Hand-over variants:
//   none       : nothing handed over (floor)
//   event      : hipEventRecord(main) + hipStreamWaitEvent(side)                      (what comm.hip does)
//   event_nofence : the same with an event created with hipEventDisableSystemFence | hipEventDisableTiming
//   writevalue : hipStreamWriteValue32(main, sig, epoch) + hipStreamWaitValue32(side, sig, epoch, GTE)   (signal memory)
//   kernelflag : the short kernel's LAST block stores epoch to sig itself; the side stream waits with hipStreamWaitValue32
//                -- nothing extra is queued on the main stream
//   stopevent  : the short kernel is launched with hipExtLaunchKernelGGL(..., stopEvent = ev): the event IS the kernel's own
//                completion signal, no barrier packet behind it; the side stream waits for that event
//   stopevent_nofence : the same with an event created with hipEventDisableSystemFence | hipEventDisableTiming
// Printed: us per iteration on the main stream's timeline (events around 200 back-to-back iterations) and the side kernels' count.
// Build: hipcc -O3 --offload-arch=gfx950 tools/microbench/cross_stream.hip -o tools/microbench/cross_stream
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdio>
#include <cstdlib>

__global__ void k_long(float* p, int n, int iters) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = p[i];
  for (int k = 0; k < iters; ++k) v = __builtin_fmaf(v, 1.0000001f, 1e-9f);
  p[i] = v;
}
__global__ void k_copy(float* dst, const float* src, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}
// the copy whose last block announces completion in memory (device-scope release; the waiter is the command processor)
__global__ void k_copy_flag(float* dst, const float* src, int n, unsigned* counter, unsigned* sig, unsigned epoch) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    const unsigned done = atomicAdd(counter, 1u) + 1u;
    if (done == gridDim.x) {
      *counter = 0u;
      __threadfence();
      __hip_atomic_store(sig, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
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
  const int n = 131072, iters_long = 6000;  // ~25 us of dependent FMAs: one multi-step rollout launch
  float *a, *b, *c, *d;
  CK(hipMalloc(&a, n * 4));
  CK(hipMalloc(&b, n * 4));
  CK(hipMalloc(&c, n * 4));
  CK(hipMalloc(&d, n * 4));
  CK(hipMemset(a, 0, n * 4));
  unsigned* counter;
  CK(hipMalloc(&counter, 4));
  CK(hipMemset(counter, 0, 4));
  int can = 0;
  CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
  unsigned* sig = nullptr;
  if (can) {
    if (hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory) != hipSuccess) {
      (void)hipGetLastError();
      can = 0;
    } else {
      *sig = 0;  // signal memory is host-visible
    }
  }
  printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
  hipStream_t mainS, side;
  CK(hipStreamCreateWithFlags(&mainS, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
  hipEvent_t ev, ev_nf, t0, t1;
  CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&ev_nf, hipEventDisableTiming | hipEventDisableSystemFence));
  CK(hipEventCreate(&t0));
  CK(hipEventCreate(&t1));
  const dim3 g(n / 256), blk(256);
  unsigned epoch = 0;
  const char* names[] = {"none", "event", "event_nofence", "writevalue", "kernelflag", "stopevent", "stopevent_nofence"};
  for (int mode = 0; mode < 7; ++mode) {
    if ((mode == 3 || mode == 4) && !can) continue;
    auto iteration = [&]() -> hipError_t {
      hipLaunchKernelGGL(k_long, g, blk, 0, mainS, a, n, iters_long);
      ++epoch;
      if (mode == 4)
        hipLaunchKernelGGL(k_copy_flag, g, blk, 0, mainS, b, a, n, counter, sig, epoch);
      else if (mode >= 5)
        hipExtLaunchKernelGGL(k_copy, g, blk, 0, mainS, nullptr, mode == 5 ? ev : ev_nf, 0, b, (const float*)a, n);
      else
        hipLaunchKernelGGL(k_copy, g, blk, 0, mainS, b, a, n);
      hipError_t e = hipSuccess;
      if (mode == 1) {
        e = hipEventRecord(ev, mainS);
        if (e == hipSuccess) e = hipStreamWaitEvent(side, ev, 0);
      } else if (mode == 2) {
        e = hipEventRecord(ev_nf, mainS);
        if (e == hipSuccess) e = hipStreamWaitEvent(side, ev_nf, 0);
      } else if (mode == 3) {
        e = hipStreamWriteValue32(mainS, sig, epoch, 0);
        if (e == hipSuccess) e = hipStreamWaitValue32(side, sig, epoch, hipStreamWaitValueGte, 0xFFFFFFFFu);
      } else if (mode == 4) {
        e = hipStreamWaitValue32(side, sig, epoch, hipStreamWaitValueGte, 0xFFFFFFFFu);
      } else if (mode >= 5) {
        e = hipStreamWaitEvent(side, mode == 5 ? ev : ev_nf, 0);
      }
      if (e != hipSuccess) return e;
      if (mode != 0) hipLaunchKernelGGL(k_copy, g, blk, 0, side, c, b, n);   // the "exchange"
      hipLaunchKernelGGL(k_copy, g, blk, 0, mainS, d, a, n);                 // the "reset"
      return hipGetLastError();
    };
    for (int w = 0; w < 20; ++w) CK(iteration());
    CK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipEventRecord(t0, mainS));
      for (int it = 0; it < 200; ++it) CK(iteration());
      CK(hipEventRecord(t1, mainS));
      CK(hipEventSynchronize(t1));
      CK(hipDeviceSynchronize());
      float ms = 0.f;
      CK(hipEventElapsedTime(&ms, t0, t1));
      best = ms < best ? ms : best;
    }
    printf("%-14s %8.2f us per iteration on the main stream\n", names[mode], best * 1e3f / 200);
  }
  return 0;
}

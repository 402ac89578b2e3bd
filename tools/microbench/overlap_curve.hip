// How well does the chip overlap per-thread arithmetic with the step kernel's streaming pattern when every wave
// runs [27 row loads] -> [X FMAs] -> [37 row stores] and a launch is only two residency rounds deep (1 M threads)?
// Prints T(X); perfect overlap would give max(T(0), T_alu(X)), none gives their sum.
// Build: hipcc -O3 --offload-arch=gfx950 tools/microbench/overlap_curve.hip -o tools/microbench/overlap_curve
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

constexpr int RD = 27, WR = 37, BLOCK = 256;

template <bool MEM>
__global__ __launch_bounds__(BLOCK) void k(const float* __restrict__ in, float* __restrict__ out, long n, long ld, int x) {
  const long i = (long)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  float v[RD], acc = 0.f;
#pragma unroll
  for (int r = 0; r < RD; ++r) {
    v[r] = MEM ? in[(long)r * ld + i] : (float)(i & 255) * 1e-3f + r;
    acc += v[r];
  }
  float a0 = acc, a1 = acc + 1.f, a2 = acc + 2.f, a3 = acc + 3.f;
  for (int it = 0; it < x; it += 4) {  // x FMAs, four independent chains
    a0 = __builtin_fmaf(a0, 0.999f, 0.001f);
    a1 = __builtin_fmaf(a1, 0.998f, 0.002f);
    a2 = __builtin_fmaf(a2, 0.997f, 0.003f);
    a3 = __builtin_fmaf(a3, 0.996f, 0.004f);
  }
  acc = a0 + a1 + a2 + a3;
  if (MEM) {
#pragma unroll
    for (int r = 0; r < WR; ++r) __builtin_nontemporal_store(acc + v[r % RD], out + (long)r * ld + i);
  } else if (acc == -1.2345f) {
    out[i] = acc;
  }
}

int main(int argc, char** argv) {
  const long n = argc > 1 ? atol(argv[1]) : (1L << 20);
  const long ld = (n + 255) / 256 * 256 + 1024;
  float *in, *out;
  (void)hipMalloc(&in, (size_t)RD * ld * 4);
  (void)hipMalloc(&out, (size_t)WR * ld * 4);
  (void)hipMemset(in, 0, (size_t)RD * ld * 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const dim3 grid((unsigned)((n + BLOCK - 1) / BLOCK));
  for (int x : {0, 256, 512, 768, 1024, 1536, 2048}) {
    float t[2];
    for (int mem = 0; mem < 2; ++mem) {
      auto launch = [&]() {
        if (mem)
          hipLaunchKernelGGL(k<true>, grid, dim3(BLOCK), 0, 0, in, out, n, ld, x);
        else
          hipLaunchKernelGGL(k<false>, grid, dim3(BLOCK), 0, 0, in, out, n, ld, x);
      };
      for (int w = 0; w < 5; ++w) launch();
      (void)hipEventRecord(e0);
      for (int r = 0; r < 100; ++r) launch();
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      t[mem] = ms * 10.f;  // us per launch
    }
    printf("n=%ld X=%4d FMAs/thread: arithmetic only %.1f us, with the stream %.1f us\n", n, x, t[0], t[1]);
  }
  return 0;
}

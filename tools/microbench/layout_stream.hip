// Micro-benchmark behind DESIGN.md's layout discussion: the step kernel's memory pattern (27 rows read, 37 rows
// written with non-temporal stores, 4-byte elements, one thread per env) under two resident layouts:
//   soa   : row r of env i at  r * ld + i                     (what the engine uses)
//   tiled : row r of env i at (i / 256) * R * 256 + r * 256 + (i % 256)   (AoSoA, one 256-env tile contiguous)
// Build: hipcc -O3 --offload-arch=gfx950 tools/microbench/layout_stream.hip -o /tmp/layout_stream
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

constexpr int RD = 27, WR = 37, BLOCK = 256;

// MODE 0: all stores non-temporal.  MODE 1: the engine's mix -- 6 rows (goals, alive, return) rewritten IN PLACE over
// the rows they were read from with default-policy stores, 30 rows non-temporal, one row as bytes (done flags).
template <bool TILED, int MODE = 0>
__global__ __launch_bounds__(BLOCK) void stream(float* __restrict__ in, float* __restrict__ out, long n, long ld) {
  const long i = (long)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  float acc = 0.f, v[RD];
#pragma unroll
  for (int r = 0; r < RD; ++r) {
    const long idx = TILED ? ((long)blockIdx.x * RD + r) * BLOCK + threadIdx.x : (long)r * ld + i;
    v[r] = in[idx];
    acc += v[r];
  }
#pragma unroll
  for (int r = 0; r < WR; ++r) {
    const long idx = TILED ? ((long)blockIdx.x * WR + r) * BLOCK + threadIdx.x : (long)r * ld + i;
    if (MODE == 1 && r < 6)
      in[TILED ? ((long)blockIdx.x * RD + r) * BLOCK + threadIdx.x : (long)r * ld + i] = acc + v[r];
    else if (MODE == 1 && r == 6)
      __builtin_nontemporal_store((unsigned char)(acc > 1.0f), reinterpret_cast<unsigned char*>(out) + idx);
    else
      __builtin_nontemporal_store(acc + v[r % RD], out + idx);
  }
}

int main(int argc, char** argv) {
  const long n = argc > 1 ? atol(argv[1]) : (1L << 20);
  const long ld = (n + 255) / 256 * 256 + 1024;
  float *in, *out;
  hipMalloc(&in, (size_t)RD * ld * 4);
  hipMalloc(&out, (size_t)WR * ld * 4);
  hipMemset(in, 0, (size_t)RD * ld * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const dim3 grid((unsigned)((n + BLOCK - 1) / BLOCK));
  const int reps = n > (1L << 24) ? 20 : 200;
  for (int tiled = 0; tiled < 3; ++tiled) {
    auto launch = [&]() {
      if (tiled == 1)
        hipLaunchKernelGGL((stream<true, 0>), grid, dim3(BLOCK), 0, 0, in, out, n, ld);
      else if (tiled == 2)
        hipLaunchKernelGGL((stream<false, 1>), grid, dim3(BLOCK), 0, 0, in, out, n, ld);
      else
        hipLaunchKernelGGL((stream<false, 0>), grid, dim3(BLOCK), 0, 0, in, out, n, ld);
    };
    for (int w = 0; w < 5; ++w) launch();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps, bytes = (double)(RD + WR) * 4 * n;
    printf("n=%ld %s: %.1f us per pass, %.0f GB/s\n", n, tiled == 1 ? "tiled          " : tiled == 2 ? "soa, engine mix" : "soa, all nt    ", us, bytes / us / 1e3);
  }
  return 0;
}

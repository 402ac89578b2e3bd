// Does the access WIDTH limit the step kernel's streaming pattern in the pure-HBM regime?  The engine moves one dword
// per lane per row (27 rows read, 37 written, one thread per env).  This micro-benchmark streams the same rows and
// bytes three ways:
//   dword : one thread per env, 4 B per lane per row                       (the engine's pattern)
//   wide  : one thread per FOUR consecutive envs, 16 B per lane per row    (global_load/store_dwordx4)
//   lds   : one thread per env, but the block moves each row's 1 KiB segment with dwordx4 (64 lanes x 16 B) and
//           transposes through LDS (what a wide-access step kernel would have to do to keep one env per thread)
// Build: hipcc -O3 --offload-arch=gfx950 tools/microbench/wide_stream.hip -o tools/microbench/wide_stream
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

constexpr int RD = 27, WR = 37, BLOCK = 256;

__global__ __launch_bounds__(BLOCK) void k_dword(const float* __restrict__ in, float* __restrict__ out, long n, long ld) {
  const long i = (long)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  float acc = 0.f, v[RD];
#pragma unroll
  for (int r = 0; r < RD; ++r) {
    v[r] = in[(long)r * ld + i];
    acc += v[r];
  }
#pragma unroll
  for (int r = 0; r < WR; ++r) __builtin_nontemporal_store(acc + v[r % RD], out + (long)r * ld + i);
}

__global__ __launch_bounds__(BLOCK) void k_wide(const float4* __restrict__ in, float4* __restrict__ out, long n4, long ld4) {
  const long i = (long)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n4) return;
  float acc = 0.f;
  float4 v[RD];
#pragma unroll
  for (int r = 0; r < RD; ++r) {
    v[r] = in[(long)r * ld4 + i];
    acc += v[r].x + v[r].y + v[r].z + v[r].w;
  }
#pragma unroll
  for (int r = 0; r < WR; ++r) {
    float4 o = v[r % RD];
    o.x += acc;
    o.y += acc;
    o.z += acc;
    o.w += acc;
    __builtin_nontemporal_store(o.x, &out[(long)r * ld4 + i].x);
    __builtin_nontemporal_store(o.y, &out[(long)r * ld4 + i].y);
    __builtin_nontemporal_store(o.z, &out[(long)r * ld4 + i].z);
    __builtin_nontemporal_store(o.w, &out[(long)r * ld4 + i].w);
  }
}

// rows through LDS: wave w of the block moves rows w, w+4, ... as 64 lanes x 16 B
__global__ __launch_bounds__(BLOCK) void k_lds(const float* __restrict__ in, float* __restrict__ out, long n, long ld) {
  __shared__ float4 tile[RD][BLOCK / 4];
  const long base = (long)blockIdx.x * BLOCK;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r = wave; r < RD; r += 4) tile[r][lane] = *reinterpret_cast<const float4*>(in + (long)r * ld + base + 4 * lane);
  __syncthreads();
  float acc = 0.f, v[RD];
  const float* t = reinterpret_cast<const float*>(tile);
#pragma unroll
  for (int r = 0; r < RD; ++r) {
    v[r] = t[r * BLOCK + threadIdx.x];
    acc += v[r];
  }
  __syncthreads();
  float* tw = reinterpret_cast<float*>(tile);
  // outputs in two batches of <= RD rows through the same tile
  for (int b = 0; b < WR; b += RD) {
    const int rows = (WR - b < RD) ? WR - b : RD;
#pragma unroll
    for (int r = 0; r < RD; ++r)
      if (r < rows) tw[r * BLOCK + threadIdx.x] = acc + v[(b + r) % RD];
    __syncthreads();
    for (int r = wave; r < rows; r += 4) {
      const float4 o = tile[r][lane];
      float* dst = out + (long)(b + r) * ld + base + 4 * lane;
      __builtin_nontemporal_store(o.x, dst);
      __builtin_nontemporal_store(o.y, dst + 1);
      __builtin_nontemporal_store(o.z, dst + 2);
      __builtin_nontemporal_store(o.w, dst + 3);
    }
    __syncthreads();
  }
}

int main(int argc, char** argv) {
  long sizes[] = {1L << 20, 1L << 22, 1L << 24};
  for (long n : sizes) {
    const long ld = n + 1024;
    float *in, *out;
    if (hipMalloc(&in, (size_t)RD * ld * 4) != hipSuccess || hipMalloc(&out, (size_t)WR * ld * 4) != hipSuccess) return 1;
    hipMemset(in, 0, (size_t)RD * ld * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int reps = n > (1L << 23) ? 20 : 100;
    for (int mode = 0; mode < 3; ++mode) {
      auto launch = [&]() {
        if (mode == 0)
          hipLaunchKernelGGL(k_dword, dim3((unsigned)(n / BLOCK)), dim3(BLOCK), 0, 0, in, out, n, ld);
        else if (mode == 1)
          hipLaunchKernelGGL(k_wide, dim3((unsigned)(n / 4 / BLOCK)), dim3(BLOCK), 0, 0, (const float4*)in, (float4*)out, n / 4, ld / 4);
        else
          hipLaunchKernelGGL(k_lds, dim3((unsigned)(n / BLOCK)), dim3(BLOCK), 0, 0, in, out, n, ld);
      };
      for (int w = 0; w < 5; ++w) launch();
      hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int r = 0; r < reps; ++r) launch();
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double us = ms * 1e3 / reps, bytes = (double)(RD + WR) * 4 * n;
      printf("n=%ld %-6s %8.1f us per pass  %6.0f GB/s\n", n, mode == 0 ? "dword" : mode == 1 ? "wide" : "lds", us, bytes / us / 1e3);
    }
    hipFree(in);
    hipFree(out);
  }
  return 0;
}

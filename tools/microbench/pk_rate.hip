// Is packed f32 (v_pk_fma_f32 / v_pk_mul_f32: two f32 per lane per instruction) worth using in the sub-step
// recurrence?  The recurrence is a rotation per joint: (s, c) <- (s cd + c sd, c cd - s sd), i.e. 2 mul + 2 fma,
// and the kernel walks a forward and a backward chain that are independent: a natural pair.
//   scalar : CH independent rotation chains per lane, v_mul_f32 / v_fma_f32
//   packed : the same chains two by two in float2 registers, v_pk_mul_f32 / v_pk_fma_f32
// Every wave runs ITERS rotations per chain; the grid fills every SIMD with W waves.  Prints ns per rotation-pair
// per wave and the ratio.  Same arithmetic, same results (checked).
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize tools/microbench/pk_rate.hip -o tools/microbench/pk_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int CH = 8;  // chains per lane (scalar) = 4 packed pairs

__global__ __launch_bounds__(256) void k_scalar(const float* __restrict__ in, float* __restrict__ out, int iters) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  float s[CH], c[CH], sd[CH], cd[CH];
#pragma unroll
  for (int j = 0; j < CH; ++j) {
    s[j] = in[(4 * j + 0) * 256 + threadIdx.x];
    c[j] = in[(4 * j + 1) * 256 + threadIdx.x];
    sd[j] = in[(4 * j + 2) * 256 + threadIdx.x];
    cd[j] = in[(4 * j + 3) * 256 + threadIdx.x];
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const float c2 = __builtin_fmaf(c[j], cd[j], -(s[j] * sd[j]));
      s[j] = __builtin_fmaf(s[j], cd[j], c[j] * sd[j]);
      c[j] = c2;
    }
  }
  float acc = 0.f;
#pragma unroll
  for (int j = 0; j < CH; ++j) acc += s[j] + c[j];
  out[i] = acc;
}

__global__ __launch_bounds__(256) void k_packed(const float* __restrict__ in, float* __restrict__ out, int iters) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  f2 s[CH / 2], c[CH / 2], sd[CH / 2], cd[CH / 2];
#pragma unroll
  for (int j = 0; j < CH / 2; ++j) {
    s[j] = f2{in[(8 * j + 0) * 256 + threadIdx.x], in[(8 * j + 4) * 256 + threadIdx.x]};
    c[j] = f2{in[(8 * j + 1) * 256 + threadIdx.x], in[(8 * j + 5) * 256 + threadIdx.x]};
    sd[j] = f2{in[(8 * j + 2) * 256 + threadIdx.x], in[(8 * j + 6) * 256 + threadIdx.x]};
    cd[j] = f2{in[(8 * j + 3) * 256 + threadIdx.x], in[(8 * j + 7) * 256 + threadIdx.x]};
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < CH / 2; ++j) {
      const f2 c2 = __builtin_elementwise_fma(c[j], cd[j], -(s[j] * sd[j]));
      s[j] = __builtin_elementwise_fma(s[j], cd[j], c[j] * sd[j]);
      c[j] = c2;
    }
  }
  float acc = 0.f;
#pragma unroll
  for (int j = 0; j < CH / 2; ++j) {
    acc += s[j].x + c[j].x;
    acc += s[j].y + c[j].y;
  }
  out[i] = acc;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 4096;
  const int waves_per_simd = argc > 2 ? atoi(argv[2]) : 4;
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int blocks = p.multiProcessorCount * waves_per_simd;  // 4 waves per block = one per SIMD
  std::vector<float> h(32 * 256);
  for (size_t k = 0; k < h.size(); ++k) {
    const int row = (int)(k / 256) % 4;
    const float a = 0.001f * (float)(k % 977);
    h[k] = row == 0 ? sinf(a) : row == 1 ? cosf(a) : row == 2 ? sinf(0.01f + a * 1e-3f) : cosf(0.01f + a * 1e-3f);
  }
  float *in, *o1, *o2;
  hipMalloc(&in, h.size() * 4);
  hipMalloc(&o1, (size_t)blocks * 256 * 4);
  hipMalloc(&o2, (size_t)blocks * 256 * 4);
  hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms[2] = {0, 0};
  for (int rep = 0; rep < 3; ++rep) {
    for (int v = 0; v < 2; ++v) {
      hipEventRecord(e0);
      if (v == 0)
        hipLaunchKernelGGL(k_scalar, dim3(blocks), dim3(256), 0, 0, in, o1, iters);
      else
        hipLaunchKernelGGL(k_packed, dim3(blocks), dim3(256), 0, 0, in, o2, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms[v], e0, e1);
    }
  }
  std::vector<float> a((size_t)blocks * 256), b((size_t)blocks * 256);
  hipMemcpy(a.data(), o1, a.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(b.data(), o2, b.size() * 4, hipMemcpyDeviceToHost);
  size_t diff = 0;
  for (size_t k = 0; k < a.size(); ++k) diff += a[k] != b[k];
  // per SIMD: waves_per_simd waves x iters x CH rotations x 4 VALU ops (scalar) or 2 packed pairs x 4 ops
  const double rot = (double)waves_per_simd * iters * CH;
  printf("{\"iters\": %d, \"waves_per_simd\": %d, \"scalar_ms\": %.4f, \"packed_ms\": %.4f, \"ratio\": %.3f, "
         "\"scalar_ns_per_rotation_per_simd\": %.3f, \"packed_ns_per_rotation_per_simd\": %.3f, \"mismatching_outputs\": %zu}\n",
         iters, waves_per_simd, ms[0], ms[1], ms[0] / ms[1], ms[0] * 1e6 / rot, ms[1] * 1e6 / rot, diff);
  return diff != 0;
}

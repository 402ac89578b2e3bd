// Does recasting the batched 4x4 DH chain as a dense contraction on the matrix cores pay?  (BASELINE.json north_star: "MFMA
// used only if recasting the batched 4x4 chain as a dense contraction actually pays per rocprof")
//
// What the reference does per sub-step pose (manytor.py:50-52, `m = m.dot(h)`, 75 x per step through :188): the product
// H0 H1 H2 H3 of four 4x4 homogeneous transforms; the step consumes the z of the translation after three and after four
// joints (manytor.py:191).  Two kernels evaluate exactly that for the same poses of the same envs:
//
//   closed : the shipped form -- mt::chain_z<Ref4Table> (manytor_amd/csrc/kernels.h): with the compile-time DH table every
//            product with a 0 / +-1 entry folds away, ~20 VALU instructions per pose, one env per lane
//   mfma   : the three chain products as dense 4x4 x 4x4 contractions on v_mfma_f32_4x4x1_16b_f32 (16 independent 4x4
//            blocks per instruction = 16 envs per instruction, K = 1: four instructions per product), f32 in / f32
//            accumulate -- the only MFMA precision that can hold the 1e-4 position tolerance.  Formulated transposed,
//            M^T <- H^T M^T, so that the accumulator layout of one product (VGPR k, lane 4 b + j = M^T[k][j] of block b) IS
//            the B operand of the next one and nothing has to be shuffled between products; the A operand (column i of H
//            in lane 4 b + i) is built from the env's sin / cos with per-lane constants (2 FMAs + 2 constants per product),
//            and the env's sin / cos reach the four lanes of its block through ds_bpermute.  64 envs per wave = 4 groups of 16.
//
// Both kernels advance the poses with the same angle-addition recurrence as the step kernel (rotate_pose), P poses per
// launch, and keep the running min of the two z values.  Every SIMD of the chip is filled (W waves each).  Printed: ns per
// pose per wave (64 envs) as seen by a SIMD, env-poses per second of the chip, instruction mix, and -- on a P = 1 launch
// with no rotation -- the max error of both forms against the fp64 product of the same inputs.
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -I include tools/microbench/mfma_chain.hip -o tools/microbench/mfma_chain
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../manytor_amd/csrc/kernels.h"

using namespace mt;
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int D = 4;

// per-lane constants of the A operand of joint n: A_0 = p c + q s, A_1 = p s - q c, A_2 = r2, A_3 = r3 with i = lane % 4
struct LaneConst {
  float p, q, r2, r3;
};
__device__ __forceinline__ LaneConst lane_const(int n, int i) {
  const float a = Ref4Table::a(n), d = Ref4Table::d(n), sa = Ref4Table::sa(n), ca = Ref4Table::ca(n);
  LaneConst k;
  k.p = i == 0 ? 1.f : (i == 3 ? a : 0.f);
  k.q = i == 1 ? -ca : (i == 2 ? sa : 0.f);
  k.r2 = i == 1 ? sa : (i == 2 ? ca : (i == 3 ? d : 0.f));
  k.r3 = i == 3 ? 1.f : 0.f;
  return k;
}

// ---- closed form: the shipped per-pose arithmetic ------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_closed(const float* __restrict__ in, float* __restrict__ out, int poses, int rotate) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int n = gridDim.x * 256;
  float s[D], c[D], sd[D], cd[D];
#pragma unroll
  for (int j = 0; j < D; ++j) {
    s[j] = in[(4 * j + 0) * n + i];
    c[j] = in[(4 * j + 1) * n + i];
    sd[j] = in[(4 * j + 2) * n + i];
    cd[j] = in[(4 * j + 3) * n + i];
  }
  const Ref4Table t{};
  float zo, ze, zmin = 1e30f;
  for (int p = 0; p < poses; ++p) {
    if (rotate) rotate_pose<D, D, +1>(s, c, sd, cd);
    chain_z<Ref4Table>(s, c, t, zo, ze);
    zmin = fminf(zmin, fminf(zo, ze));
  }
  out[i] = zmin;
  out[n + i] = zo;
  out[2 * n + i] = ze;
}

// ---- the same chain as three dense 4x4 products per pose on the matrix cores ----------------------------------------
__global__ __launch_bounds__(256) void k_mfma(const float* __restrict__ in, float* __restrict__ out, int poses, int rotate) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int n = gridDim.x * 256;
  const int lane = threadIdx.x & 63, li = lane & 3, lb = lane >> 2;
  float s[D], c[D], sd[D], cd[D];
#pragma unroll
  for (int j = 0; j < D; ++j) {
    s[j] = in[(4 * j + 0) * n + i];
    c[j] = in[(4 * j + 1) * n + i];
    sd[j] = in[(4 * j + 2) * n + i];
    cd[j] = in[(4 * j + 3) * n + i];
  }
  LaneConst k1 = lane_const(1, li), k2 = lane_const(2, li), k3 = lane_const(3, li);
  // M1^T = H0^T as the first B operand: VGPR k, lane (b, j) = H0[j][k]
  const float e0 = li == 0, e1 = li == 1, e2 = li == 2, e3 = li == 3;
  const float a0 = Ref4Table::a(0), d0 = Ref4Table::d(0), sa0 = Ref4Table::sa(0), ca0 = Ref4Table::ca(0);
  float zmin[4] = {1e30f, 1e30f, 1e30f, 1e30f}, zo_last[4], ze_last[4];
  for (int p = 0; p < poses; ++p) {
    if (rotate) rotate_pose<D, D, +1>(s, c, sd, cd);
#pragma unroll
    for (int g = 0; g < 4; ++g) {                       // 16 envs per MFMA: envs 16 g .. 16 g + 15 of the wave
      const int src = 16 * g + lb;                      // the env of this lane's block
      const float c0 = __shfl(c[0], src), s0 = __shfl(s[0], src);
      v4f m;
      m.x = e0 * c0 + e1 * s0;
      m.y = ca0 * (e1 * c0 - e0 * s0) + sa0 * e2;
      m.z = sa0 * (e0 * s0 - e1 * c0) + ca0 * e2;
      m.w = a0 * (e0 * c0 + e1 * s0) + d0 * e2 + e3;
      float zo = 0.f;
#pragma unroll
      for (int j = 1; j < D; ++j) {
        const LaneConst k = j == 1 ? k1 : (j == 2 ? k2 : k3);
        const float cj = __shfl(c[j], src), sj = __shfl(s[j], src);
        const float A0 = __builtin_fmaf(k.p, cj, k.q * sj), A1 = __builtin_fmaf(k.p, sj, -(k.q * cj));
        v4f acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(A0, m.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(A1, m.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(k.r2, m.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(k.r3, m.w, acc, 0, 0, 0);
        m = acc;
        if (j == 2) zo = m.w;                           // M3[2][3] = M3^T[3][2]: VGPR 3 of lane (b, 2)
      }
      zo_last[g] = zo;
      ze_last[g] = m.w;
      zmin[g] = fminf(zmin[g], fminf(zo, m.w));
    }
  }
  if (li == 2) {                                        // lane (b, 2) of group g holds env 16 g + b of the wave
    const int base = i - lane;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int e = base + 16 * g + lb;
      out[e] = zmin[g];
      out[n + e] = zo_last[g];
      out[2 * n + e] = ze_last[g];
    }
  }
}

static void host_chain(const float* s, const float* c, double& zo, double& ze) {
  const double A[4] = {0, 0, 0, 27.0}, Dd[4] = {4.3, 0, 24.3, 0}, SA[4] = {-1, 1, -1, 1}, CA[4] = {0, 0, 0, 0};
  double M[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
  for (int j = 0; j < 4; ++j) {
    const double ct = c[j], st = s[j];
    const double H[4][4] = {{ct, -st * CA[j], st * SA[j], A[j] * ct}, {st, ct * CA[j], -ct * SA[j], A[j] * st}, {0, SA[j], CA[j], Dd[j]}, {0, 0, 0, 1}};
    double R[4][4];
    for (int r = 0; r < 4; ++r)
      for (int q = 0; q < 4; ++q) {
        double v = 0;
        for (int k = 0; k < 4; ++k) v += M[r][k] * H[k][q];
        R[r][q] = v;
      }
    for (int r = 0; r < 4; ++r)
      for (int q = 0; q < 4; ++q) M[r][q] = R[r][q];
    if (j == 2) zo = M[2][3];
  }
  ze = M[2][3];
}

int main(int argc, char** argv) {
  const int waves_per_simd = argc > 1 ? atoi(argv[1]) : 8;
  const int poses = argc > 2 ? atoi(argv[2]) : 2400;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 1;
  const int cus = prop.multiProcessorCount;
  const int blocks = cus * waves_per_simd;  // 4 waves per block = one per SIMD
  const int n = blocks * 256;
  std::vector<float> h((size_t)16 * n);
  srand(7);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < 4; ++j) {
      const double ang = (rand() % 360 - 180) * M_PI / 180.0, dl = ((rand() % 3000) / 100.0 - 15.0) / 24.0 * M_PI / 180.0;
      h[(size_t)(4 * j + 0) * n + i] = (float)sin(ang);
      h[(size_t)(4 * j + 1) * n + i] = (float)cos(ang);
      h[(size_t)(4 * j + 2) * n + i] = (float)sin(dl);
      h[(size_t)(4 * j + 3) * n + i] = (float)cos(dl);
    }
  float *din, *dout;
  if (hipMalloc(&din, h.size() * 4) != hipSuccess || hipMalloc(&dout, (size_t)3 * n * 4) != hipSuccess) return 1;
  hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  std::vector<float> o((size_t)3 * n);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  double ns[2] = {0, 0}, err[2] = {0, 0};
  for (int mode = 0; mode < 2; ++mode) {
    auto launch = [&](int p, int rot) {
      if (mode == 0)
        hipLaunchKernelGGL(k_closed, dim3(blocks), dim3(256), 0, 0, din, dout, p, rot);
      else
        hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), 0, 0, din, dout, p, rot);
    };
    // accuracy: one pose, no rotation, against the fp64 product of the same (sin, cos) inputs
    launch(1, 0);
    hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) {
      float s[4], c[4];
      for (int j = 0; j < 4; ++j) {
        s[j] = h[(size_t)(4 * j + 0) * n + i];
        c[j] = h[(size_t)(4 * j + 1) * n + i];
      }
      double zo, ze;
      host_chain(s, c, zo, ze);
      err[mode] = fmax(err[mode], fmax(fabs(o[(size_t)n + i] - zo), fabs(o[(size_t)2 * n + i] - ze)));
    }
    for (int w = 0; w < 3; ++w) launch(poses, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int reps = 5;
    for (int r = 0; r < reps; ++r) launch(poses, 1);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ns[mode] = ms * 1e6 / reps / poses / waves_per_simd;  // SIMD time per wave-pose (64 envs)
    printf("%-6s  %d CUs x 4 SIMDs x %d waves, %d poses: %8.3f ms per launch, %7.2f ns per pose per wave (64 envs), %.3e env-poses/s, "
           "max |z - fp64| %.2e\n",
           mode == 0 ? "closed" : "mfma", cus, waves_per_simd, poses, ms / reps, ns[mode], (double)n * poses / (ms * 1e-3 / reps), err[mode]);
  }
  printf("ratio mfma / closed: %.2f x the time per pose\n", ns[1] / ns[0]);
  return (err[0] < 1e-4 && err[1] < 1e-4) ? 0 : 2;
}

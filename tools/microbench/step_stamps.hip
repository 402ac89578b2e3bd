// Where does one step launch spend its time at small batch sizes?  A DIAGNOSTIC build of the library's own step
// kernel (kernels.h compiled with -DMT_STAMPS: s_memtime stamps at the phase boundaries, one lane per wave) run on
// synthetic state, stand-alone (no engine, no ABI).  Reports, per batch size, the median over waves of every phase in
// shader cycles and the span first-wave-in .. last-wave-out.  Stamps fence the schedule, so read SHARES, not totals.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -DMT_STAMPS -I include -fno-signed-zeros -ffinite-math-only \
//        -fno-slp-vectorize -ffp-contract=off tools/microbench/step_stamps.hip -o /tmp/step_stamps
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../manytor_amd/csrc/kernels.h"

using namespace mt;

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e = (x);                                                            \
    if (e != hipSuccess) {                                                         \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                       \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

static DhConst ref_table() {
  DhConst t{};
  const float a[4] = {0, 0, 0, 27.0f}, d[4] = {4.3f, 0, 24.3f, 0}, sa[4] = {-1, 1, -1, 1}, off[4] = {0, 0, 0, -90};
  for (int j = 0; j < MT_MAX_DOF; ++j) t.ca[j] = 1.f;
  for (int j = 0; j < 4; ++j) {
    t.a[j] = a[j];
    t.d[j] = d[j];
    t.sa[j] = sa[j];
    t.ca[j] = 0.f;
    t.off_deg[j] = off[j];
  }
  return t;
}

int main(int argc, char** argv) {
  const int K = 7, D = 4;
  std::vector<long> sizes = {16384, 65536, 131072, 262144, 1048576};
  if (argc > 1) {
    sizes.clear();
    for (int i = 1; i < argc; ++i) sizes.push_back(atol(argv[i]));
  }
  for (long n : sizes) {
    const long ld = (n + 255) / 256 * 256 + (n > 16384 ? 1024 : 0);
    const long waves = (n + 63) / 64;
    size_t rows = 2 * D + 6 * K + 3 + 6 + 2;  // generous
    char* arena;
    CK(hipMalloc(&arena, rows * ld * 4 + 4096));
    CK(hipMemset(arena, 0, rows * ld * 4 + 4096));
    unsigned long long* stamps;
    CK(hipMalloc(&stamps, waves * 8 * 8));
    StepArgs a{};
    float* f = (float*)arena;
    a.actions = f; f += D * ld;
    a.goals = f; f += D * ld;
    a.points = f; f += 3 * K * ld;
    a.obs = f; f += 3 * K * ld;
    a.alive = (uint32_t*)f; f += ld;
    a.total_reward = f; f += ld;
    a.reward = (int32_t*)f; f += ld;
    a.done = (uint8_t*)f; f += ld;
    a.done_bits = (unsigned long long*)f; f += ld;
    a.ee = f; f += 3 * ld;
    a.episodes = (uint32_t*)f; f += ld;
    a.last_return = f; f += ld;
    a.ring = nullptr;
    a.bad_actions = (uint32_t*)f;
    a.ring_slots = 0;
    a.n = n; a.ld = ld; a.env_base = 0; a.K = K; a.S = 25; a.tol = 8.0f; a.inv_sm1 = 1.0f / 24.0f; a.flags = 0;
    a.seed_lo = 0x5EED; a.seed_hi = 0; a.major = 0;
    a.dh = ref_table();
    a.stamps = stamps;
    const dim3 grid((unsigned)((n + kBlock - 1) / kBlock)), block(kBlock);
    hipLaunchKernelGGL((reset_kernel<4, true, false>), grid, block, 0, 0, a, 51.3f);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int w = 0; w < 60; ++w) {
      a.major = w;
      hipLaunchKernelGGL((step_kernel<Ref4Table, true, 0, false>), grid, block, 0, 0, a);
    }
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int w = 0; w < 40; ++w) {
      a.major = 60 + w;
      hipLaunchKernelGGL((step_kernel<Ref4Table, true, 0, false>), grid, block, 0, 0, a);
    }
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(waves * 8);
    CK(hipMemcpy(h.data(), stamps, waves * 8 * 8, hipMemcpyDeviceToHost));
    const char* names[5] = {"load pose + draw action", "kinematics (25 poses)", "7 targets (load, observe, store)",
                            "state stores issued", "stores acknowledged"};
    std::vector<double> ph[5];
    unsigned long long t_min = ~0ull, t_max = 0;
    std::vector<double> life;
    for (long w = 0; w < waves; ++w) {
      const unsigned long long* s = &h[w * 8];
      for (int p = 0; p < 5; ++p) ph[p].push_back((double)(s[p + 1] - s[p]));
      life.push_back((double)(s[5] - s[0]));
      t_min = std::min(t_min, s[0]);
      t_max = std::max(t_max, s[5]);
    }
    auto med = [](std::vector<double>& v) {
      std::sort(v.begin(), v.end());
      return v[v.size() / 2];
    };
    printf("n = %ld  (%ld waves, %.2f waves per SIMD)   %.2f us per launch (HIP events, stamped build)\n", n, waves,
           waves / 1024.0, ms * 1e3 / 40);
    double tot = med(life);
    for (int p = 0; p < 5; ++p) printf("   %-36s %8.0f cycles  %5.1f %%\n", names[p], med(ph[p]), 100.0 * med(ph[p]) / tot);
    printf("   %-36s %8.0f cycles\n", "wave lifetime (median)", tot);
    printf("   %-36s %8.0f cycles  (first wave in .. last wave out, last launch)\n", "kernel span", (double)(t_max - t_min));
    CK(hipFree(arena));
    CK(hipFree(stamps));
  }
  return 0;
}

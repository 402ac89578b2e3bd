// How much arithmetic can hide under the step's memory operations?  The streaming yardstick of the library (stream_probe_kernel:
// the rows of one mt_step_random read / rewritten / nt-written per env, one env per lane) with ALU vector instructions of
// independent fused multiply-adds between its loads and its stores -- the simplest possible schedule: every load first, then
// the arithmetic, then every store.  If this kernel with ~1 000 VALU instructions (what step_kernel<Ref4Table> executes per
// wave) runs close to max(memory-only, arithmetic-only), the step kernel's distance from its own memory-only time is a matter
// of its schedule; if it runs near the step kernel's time, the step kernel already is where this chip puts such a mix.
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize tools/microbench/probe_alu.hip -o tools/microbench/probe_alu
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

constexpr int D = 4, K = 7, RD = D + 3 * K + 2, WP = D + 2, WN = 3 * K + 4, BLOCK = 256;

template <int ALU>
__global__ __launch_bounds__(BLOCK) void k_probe(float* state, float* out, unsigned char* bytes, long n, long ld) {
  const long i = (long)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  float v[RD], acc = 0.f;
#pragma unroll
  for (int r = 0; r < RD; ++r) {
    v[r] = state[(long)r * ld + i];
    acc += v[r];
  }
  // ALU instructions in 8 independent chains (enough ILP for 4-cycle issue at >= 2 waves per SIMD)
  float c0 = v[0], c1 = v[1], c2 = v[2], c3 = v[3], c4 = v[4], c5 = v[5], c6 = v[6], c7 = v[7];
  const float m = 1.0000001f, a = 1e-9f;
  for (int it = 0; it < ALU / 8; ++it) {
    c0 = __builtin_fmaf(c0, m, a);
    c1 = __builtin_fmaf(c1, m, a);
    c2 = __builtin_fmaf(c2, m, a);
    c3 = __builtin_fmaf(c3, m, a);
    c4 = __builtin_fmaf(c4, m, a);
    c5 = __builtin_fmaf(c5, m, a);
    c6 = __builtin_fmaf(c6, m, a);
    c7 = __builtin_fmaf(c7, m, a);
  }
  acc = (acc + c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7) * 1e-30f;
#pragma unroll
  for (int r = 0; r < WN; ++r) __builtin_nontemporal_store(v[r % RD] + acc, out + (long)r * ld + i);
#pragma unroll
  for (int r = 0; r < WP; ++r) {
    const int row = r < D ? r : RD - 2 + (r - D);
    state[(long)row * ld + i] = v[row] + acc;
  }
  __builtin_nontemporal_store((unsigned char)(acc > 1.f ? 1 : 0), bytes + i);
}

// The same work with TWO envs per thread, software-pipelined: the loads of env B are in flight under the arithmetic of env A,
// the stores of env A drain under the arithmetic of env B (each env gets its own ALU instructions).
template <int ALU>
__global__ __launch_bounds__(BLOCK) void k_probe2(float* state, float* out, unsigned char* bytes, long n, long ld) {
  const long half = n / 2;
  const long i = (long)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= half) return;
  const long j = i + half;
  float va[RD], vb[RD], acca = 0.f, accb = 0.f;
#pragma unroll
  for (int r = 0; r < RD; ++r) va[r] = state[(long)r * ld + i];
#pragma unroll
  for (int r = 0; r < RD; ++r) vb[r] = state[(long)r * ld + j];
#pragma unroll
  for (int r = 0; r < RD; ++r) acca += va[r];
  const float m = 1.0000001f, a = 1e-9f;
  {
    float c0 = va[0], c1 = va[1], c2 = va[2], c3 = va[3], c4 = va[4], c5 = va[5], c6 = va[6], c7 = va[7];
    for (int it = 0; it < ALU / 8; ++it) {
      c0 = __builtin_fmaf(c0, m, a);
      c1 = __builtin_fmaf(c1, m, a);
      c2 = __builtin_fmaf(c2, m, a);
      c3 = __builtin_fmaf(c3, m, a);
      c4 = __builtin_fmaf(c4, m, a);
      c5 = __builtin_fmaf(c5, m, a);
      c6 = __builtin_fmaf(c6, m, a);
      c7 = __builtin_fmaf(c7, m, a);
    }
    acca = (acca + c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7) * 1e-30f;
  }
#pragma unroll
  for (int r = 0; r < WN; ++r) __builtin_nontemporal_store(va[r % RD] + acca, out + (long)r * ld + i);
#pragma unroll
  for (int r = 0; r < WP; ++r) {
    const int row = r < D ? r : RD - 2 + (r - D);
    state[(long)row * ld + i] = va[row] + acca;
  }
  __builtin_nontemporal_store((unsigned char)(acca > 1.f ? 1 : 0), bytes + i);
#pragma unroll
  for (int r = 0; r < RD; ++r) accb += vb[r];
  {
    float c0 = vb[0], c1 = vb[1], c2 = vb[2], c3 = vb[3], c4 = vb[4], c5 = vb[5], c6 = vb[6], c7 = vb[7];
    for (int it = 0; it < ALU / 8; ++it) {
      c0 = __builtin_fmaf(c0, m, a);
      c1 = __builtin_fmaf(c1, m, a);
      c2 = __builtin_fmaf(c2, m, a);
      c3 = __builtin_fmaf(c3, m, a);
      c4 = __builtin_fmaf(c4, m, a);
      c5 = __builtin_fmaf(c5, m, a);
      c6 = __builtin_fmaf(c6, m, a);
      c7 = __builtin_fmaf(c7, m, a);
    }
    accb = (accb + c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7) * 1e-30f;
  }
#pragma unroll
  for (int r = 0; r < WN; ++r) __builtin_nontemporal_store(vb[r % RD] + accb, out + (long)r * ld + j);
#pragma unroll
  for (int r = 0; r < WP; ++r) {
    const int row = r < D ? r : RD - 2 + (r - D);
    state[(long)row * ld + j] = vb[row] + accb;
  }
  __builtin_nontemporal_store((unsigned char)(accb > 1.f ? 1 : 0), bytes + j);
}

template <int ALU>
static float run2(float* state, float* out, unsigned char* bytes, long n, long ld, hipStream_t st, int reps) {
  auto pass = [&]() { hipLaunchKernelGGL((k_probe2<ALU>), dim3((unsigned)(n / 2 / BLOCK)), dim3(BLOCK), 0, st, state, out, bytes, n, ld); };
  for (int w = 0; w < 5; ++w) pass();
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, st);
  for (int r = 0; r < reps; ++r) pass();
  (void)hipEventRecord(e1, st);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / reps;
}

template <int ALU>
static float run(float* state, float* out, unsigned char* bytes, long n, long ld, int chains, hipStream_t* st, int reps) {
  const long per = n / chains;
  auto pass = [&]() {
    for (int c = 0; c < chains; ++c)
      hipLaunchKernelGGL((k_probe<ALU>), dim3((unsigned)(per / BLOCK)), dim3(BLOCK), 0, st[c], state + c * per, out + c * per,
                         bytes + c * per, per, ld);
  };
  for (int w = 0; w < 5; ++w) pass();
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, st[0]);
  for (int r = 0; r < reps; ++r) pass();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e1, st[0]);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / reps;
}

int main(int argc, char** argv) {
  const long n = argc > 1 ? atol(argv[1]) : 1048576;
  const long ld = n + 256;
  float *state, *out;
  unsigned char* bytes;
  if (hipMalloc(&state, (size_t)RD * ld * 4) != hipSuccess || hipMalloc(&out, (size_t)WN * ld * 4) != hipSuccess ||
      hipMalloc(&bytes, (size_t)ld) != hipSuccess)
    return 1;
  (void)hipMemset(state, 0, (size_t)RD * ld * 4);
  hipStream_t st[2];
  (void)hipStreamCreateWithFlags(&st[0], hipStreamNonBlocking);
  (void)hipStreamCreateWithFlags(&st[1], hipStreamNonBlocking);
  const double gb = (double)(8 * D + 24 * K + 33) * n / 1e9;
  printf("%ld envs, %.1f MB per pass; us per pass (GB/s) for ALU = fused multiply-adds per thread between the loads and the stores\n", n, gb * 1e3);
  for (int chains = 1; chains <= 2; ++chains) {
    const float t0 = run<0>(state, out, bytes, n, ld, chains, st, 200), t256 = run<256>(state, out, bytes, n, ld, chains, st, 200),
                t512 = run<512>(state, out, bytes, n, ld, chains, st, 200), t768 = run<768>(state, out, bytes, n, ld, chains, st, 200),
                t1024 = run<1024>(state, out, bytes, n, ld, chains, st, 200), t1536 = run<1536>(state, out, bytes, n, ld, chains, st, 200),
                t2048 = run<2048>(state, out, bytes, n, ld, chains, st, 200);
    printf("%d launch(es) per pass: ALU 0: %6.2f (%5.0f)  256: %6.2f  512: %6.2f  768: %6.2f  1024: %6.2f  1536: %6.2f  2048: %6.2f\n", chains, t0,
           gb / (t0 * 1e-6), t256, t512, t768, t1024, t1536, t2048);
  }
  {
    const float t0 = run2<0>(state, out, bytes, n, ld, st[0], 200), t256 = run2<256>(state, out, bytes, n, ld, st[0], 200),
                t512 = run2<512>(state, out, bytes, n, ld, st[0], 200), t768 = run2<768>(state, out, bytes, n, ld, st[0], 200),
                t1024 = run2<1024>(state, out, bytes, n, ld, st[0], 200), t1536 = run2<1536>(state, out, bytes, n, ld, st[0], 200),
                t2048 = run2<2048>(state, out, bytes, n, ld, st[0], 200);
    printf("two envs per thread, pipelined (ALU per env): ALU 0: %6.2f (%5.0f)  256: %6.2f  512: %6.2f  768: %6.2f  1024: %6.2f  1536: %6.2f  2048: %6.2f\n",
           t0, gb / (t0 * 1e-6), t256, t512, t768, t1024, t1536, t2048);
  }
  return 0;
}

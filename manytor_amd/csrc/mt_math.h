// Device math for the ManyTor step kernels (gfx950).  fp32 throughout.
//
// Trigonometry is done on DEGREES because that is the unit of the state
// (manytor.py:39 converts per call): the quadrant reduction x - 90*rint(x/90)
// is exact in fp32, so integer-degree poses (every action the reference's
// action_sample produces, manytor.py:216) carry no argument-rounding error at
// all.  Coefficients come from tools/gen_poly.py.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace mt {

// Every kernel is compiled with -ffp-contract=off and spells its fused multiply-adds out, so that a formula
// rounds the same way wherever it is inlined: step_kernel (both action sources), rollout_kernel, observe_kernel
// ... produce bit-identical numbers for the same inputs by construction, not by luck of instruction selection.
// pmul / pfma additionally drop products with an operand that is a compile-time 0 or +-1 (the entries of a
// static DH table and of the identity the chain starts from).
__device__ __forceinline__ float pmul(float a, float b) {
  if (__builtin_constant_p(a)) {
    if (a == 0.f) return 0.f;
    if (a == 1.f) return b;
    if (a == -1.f) return -b;
  }
  if (__builtin_constant_p(b)) {
    if (b == 0.f) return 0.f;
    if (b == 1.f) return a;
    if (b == -1.f) return -a;
  }
  return a * b;
}
// a * b + c
__device__ __forceinline__ float pfma(float a, float b, float c) {
  if (__builtin_constant_p(a)) {
    if (a == 0.f) return c;
    if (a == 1.f) return b + c;
    if (a == -1.f) return c - b;
  }
  if (__builtin_constant_p(b)) {
    if (b == 0.f) return c;
    if (b == 1.f) return a + c;
    if (b == -1.f) return c - a;
  }
  if (__builtin_constant_p(c) && c == 0.f) return a * b;
  return __builtin_fmaf(a, b, c);
}

// a * b + K for a compile-time K, as ONE instruction wherever it is inlined (v_fmaak_f32: the addend is a literal of the
// instruction).  Left to the compiler, a Horner chain whose coefficients are used twice in a block -- the two atan of an
// observation, the sin and cos polynomials of two poses -- keeps the literals only in the first chain: the second gets a
// v_mov_b32 of the coefficient plus a v_fmac_f32, two issue slots per term (49 extra instructions per wave-step in the
// reference arm's step kernel).  Same operation (a fused multiply-add, round once), hence the same bits as __builtin_fmaf.
template <uint32_t KBITS>
__device__ __forceinline__ float fma_lit(float a, float b) {
  if (__builtin_constant_p(a) && __builtin_constant_p(b)) return __builtin_fmaf(a, b, __builtin_bit_cast(float, KBITS));
  float r;
  asm("v_fmaak_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "n"(KBITS));
  return r;
}
#define MT_FMA_LIT(a, b, K) ::mt::fma_lit<__builtin_bit_cast(uint32_t, (float)(K))>((a), (b))

// sin and cos of x degrees, |x| <= 720 or so.  Max abs error ~8e-8.
__device__ __forceinline__ void sincos_deg(float x, float& s, float& c) {
  const float q = __builtin_rintf(x * (1.0f / 90.0f));
  const float f = __builtin_fmaf(q, -90.0f, x);  // exact: |f| <= 45
  const float z = f * f;
  float ps = -9.621979952e-17f;
  ps = MT_FMA_LIT(ps, z, 1.349391605e-11f);
  ps = MT_FMA_LIT(ps, z, -8.860952789e-07f);
  ps = MT_FMA_LIT(ps, z, 1.745329238e-02f);
  ps *= f;
  float pc = 2.099184062e-19f;
  pc = MT_FMA_LIT(pc, z, -3.925190319e-14f);
  pc = MT_FMA_LIT(pc, z, 3.866319265e-09f);
  pc = MT_FMA_LIT(pc, z, -1.523087121e-04f);
  pc = __builtin_fmaf(pc, z, 1.0f);
  const int qi = (int)q;
  // quadrant rotation: (s,c) -> q&1 ? (c,-s) : (s,c); q&2 ? negate both
  const bool odd = qi & 1;
  float ss = odd ? pc : ps;
  float cc = odd ? -ps : pc;
  const unsigned flip = (unsigned)(qi & 2) << 30;  // sign bit
  s = __uint_as_float(__float_as_uint(ss) ^ flip);
  c = __uint_as_float(__float_as_uint(cc) ^ flip);
}

// Hardware transcendental path: v_sin_f32 / v_cos_f32 take REVOLUTIONS.
// Quarter-rate, ~1e-6 absolute; only used where a 1e-3 guard band applies
// (the ground flag of intermediate sub-steps) and only under MT_FLAG_HW_TRIG.
__device__ __forceinline__ void sincos_deg_hw(float x, float& s, float& c) {
  const float rev = x * (1.0f / 360.0f);
  s = __builtin_amdgcn_sinf(rev);
  c = __builtin_amdgcn_cosf(rev);
}

// atan2(y, x) in DEGREES for y >= 0, x >= 0 (first quadrant only: the
// reference takes |differences| first, manytor.py:18).  atan2(0,0) = 0 like
// math.atan2.  Max abs error ~1e-5 degrees.
__device__ __forceinline__ float atan2_deg_q1(float y, float x) {
  const float mx = fmaxf(x, y);
  const float mn = fminf(x, y);
  // mx == 0 implies mn == 0, and 0 * rcp(1e-30) = 0: atan2(0, 0) = 0 without a compare/select
  const float q = mn * __builtin_amdgcn_rcpf(fmaxf(mx, 1e-30f));
  const float w = q * q;
  // atan(q)/q on [0,1], already scaled to degrees (tools/gen_poly.py ATAN x 180/pi)
  float p = -4.668773152e-03f * 57.29577951308232f;
  p = MT_FMA_LIT(p, w, 2.416618913e-02f * 57.29577951308232f);
  p = MT_FMA_LIT(p, w, -5.936710164e-02f * 57.29577951308232f);
  p = MT_FMA_LIT(p, w, 9.906096756e-02f * 57.29577951308232f);
  p = MT_FMA_LIT(p, w, -1.401658505e-01f * 57.29577951308232f);
  p = MT_FMA_LIT(p, w, 1.996923536e-01f * 57.29577951308232f);
  p = MT_FMA_LIT(p, w, -3.333196044e-01f * 57.29577951308232f);
  p = MT_FMA_LIT(p, w, 9.999998808e-01f * 57.29577951308232f);
  const float a = p * q;  // degrees, in [0, 45]
  return (y > x) ? 90.0f - a : a;
}

// sin and cos of a SMALL angle in degrees (|x| <= 45): no quadrant reduction.  Used for the per-sub-step
// increment delta = (action - goal) / (S - 1), |delta| <= 360 / 24 = 15 degrees for S = 25; the caller falls
// back to sincos_deg when S is small enough for |delta| to exceed 45.
__device__ __forceinline__ void sincos_deg_small(float f, float& s, float& c) {
  const float z = f * f;
  float ps = -9.621979952e-17f;
  ps = MT_FMA_LIT(ps, z, 1.349391605e-11f);
  ps = MT_FMA_LIT(ps, z, -8.860952789e-07f);
  ps = MT_FMA_LIT(ps, z, 1.745329238e-02f);
  s = ps * f;
  float pc = 2.099184062e-19f;
  pc = MT_FMA_LIT(pc, z, -3.925190319e-14f);
  pc = MT_FMA_LIT(pc, z, 3.866319265e-09f);
  pc = MT_FMA_LIT(pc, z, -1.523087121e-04f);
  c = __builtin_fmaf(pc, z, 1.0f);
}

}  // namespace mt

// HIP kernels of the ManyTor step engine (gfx950, wave64).
//
// One thread = one environment.  State is struct-of-arrays with row stride
// `ld`, so every load/store below is a fully coalesced row access
// (lane l of a wave touches element 64*w + l of a row).
//
// Reference semantics restated here (file:line = /root/reference/manytor.py):
//   dh()               :25-32   one DH row = Rz(theta) Tz(d) Tx(a) Rx(alpha)
//   fk()               :35-53   product of the first `mode` rows, degrees in
//   get_observations() :141-153 from joints_coordinates[-2] (the elbow)
//   is_done()          :155-173 from joints_coordinates[-1] (the end effector)
//   action()           :175-213 25 interpolated sub-steps, ground flag, reward
//   step()             :255-260
//   reset()            :219-253
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mt_math.h"
#include "philox.h"
#include "step_args.h"

namespace mt {

// Write-only outputs of a step (observations, reward, done, end effector) are never re-read by the next
// step.  MT_NT_STORES marks them non-temporal so they do not displace the re-read state (goals, targets,
// alive mask, return) from L2 / Infinity Cache.
#ifndef MT_NT_STORES
#define MT_NT_STORES 1
#endif
// Row access: every SoA row is addressed as (wave-uniform row base) + (32-bit per-lane BYTE offset), which maps
// onto the `global_load/store v, v_off, s[base:base+1]` form -- the row base stays in SGPRs and no 64-bit
// per-lane address arithmetic is needed.  Byte offsets fit 32 bits because mt_create caps n_envs below 2^30.
// The row base MUST be the same in every lane.  It is passed through readfirstlane (free on a value that already
// sits in SGPRs): without that the optimiser folds `base + row * ld` and the lane offset into one per-lane 64-bit
// address per row and keeps all of them in VGPRs for the whole kernel (14 registers for 7 joint rows plus 64-bit
// VALU adds; measured: step_kernel<Dh7Table> 86 -> VGPRs, see profiles/r02_kernel_resources.txt).
template <typename T>
using global_ptr = __attribute__((address_space(1))) T*;  // a pointer known to be global memory: global_*, not flat_*

__device__ __forceinline__ global_ptr<char> uniform_row(const void* row) {
  const uint64_t v = reinterpret_cast<uint64_t>(row);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return (global_ptr<char>)(((uint64_t)hi << 32) | lo);
}
// The lane offset has to reach instruction selection as a 32-BIT value in the basic block of the access: selection works
// per block, and an offset that was zero-extended once at the top and carried into other blocks as a 64-bit register can
// no longer be proven to fit the `saddr + zext(voffset)` form -- every access then costs a v_lshl_add_u64 and a 64-bit
// VGPR pair (96 of them in the prefetch kernel of the reference arm, 8 % of its instructions: round 3).  The empty asm
// "redefines" the offset in place before every access -- the SAME register, no instruction -- so the extension is
// re-materialised, and folded into the access, where it is used.  Hence the offset travels by reference: a kernel keeps
// ONE variable per element size (o4 = 4 i for the 32-bit rows, o1 = i for the byte rows) and every accessor renews it.
//   LaneOffset<true>  : renewed at every access (1 106 -> 1 013 VALU instructions in that kernel; 3-5 % less time where
//                       the step is latency- or issue-bound: <= 262 144 envs, the 7-joint arm, the fused episodes)
//   LaneOffset<false> : left to the optimiser (the 64-bit adds stay).  Where the reference arm's step is HBM-bound
//                       (>= ~400 k envs per launch) THIS form is the faster one by 0.7-2 % -- same memory operations in
//                       the same order, so the cause is the waves' timing against the memory system, not the code
//                       (profiles/r03_variants.md section 8) -- and step_kernel's FLAT instantiation keeps it.
#ifndef MT_PLAIN_LANE_OFFSET
#define MT_PLAIN_LANE_OFFSET 0  // 1: no renewal anywhere (A/B builds: tools/ab_lane_offset.sh)
#endif
template <bool RENEW>
struct LaneOffset {
  uint32_t v;
};
template <bool RENEW>
__device__ __forceinline__ uint32_t lane_offset(LaneOffset<RENEW>& o) {
#if !MT_PLAIN_LANE_OFFSET
  if (RENEW) asm volatile("" : "+v"(o.v));
#endif
  return o.v;
}
// MT_BUFFER_ROWS: the same access as a BUFFER instruction -- `buffer_load/store v, v_off, s[rsrc:rsrc+3], 0 offen`: the row
// base sits in a 128-bit resource descriptor in SGPRs (built by the scalar unit: base, no stride, no range check, the raw
// 32-bit data format of gfx94x / gfx950), the lane's byte offset is the VGPR operand by construction.  Neither the 64-bit
// per-lane adds of the plain form nor the renewal asm of the saddr form (a scheduling barrier at every access) are needed.
#ifndef MT_BUFFER_ROWS
#define MT_BUFFER_ROWS 0
#endif
#if MT_BUFFER_ROWS
constexpr int kRsrcWord3 = 0x00020000;  // raw buffer, DATA_FORMAT = 32 (gfx90a / gfx94x / gfx950)
constexpr int kAuxNt = 2;               // cache policy: non-temporal (the `nt` bit of gfx94x)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_rsrc(const void* row) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)uniform_row(row), 0, -1, kRsrcWord3);  // num_records = 2^32 - 1: no clamp
}
template <typename T, bool R>
__device__ __forceinline__ T ldr(const T* row, LaneOffset<R>& o) {
  static_assert(sizeof(T) == 4 || sizeof(T) == 1, "rows hold 32-bit words or bytes");
  if constexpr (sizeof(T) == 4)
    return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b32(row_rsrc(row), (int)o.v, 0, 0));
  else
    return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b8(row_rsrc(row), (int)o.v, 0, 0));
}
template <typename T, bool R>
__device__ __forceinline__ void str(T* row, LaneOffset<R>& o, T v) {
  static_assert(sizeof(T) == 4 || sizeof(T) == 1, "rows hold 32-bit words or bytes");
  if constexpr (sizeof(T) == 4)
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), row_rsrc(row), (int)o.v, 0, 0);
  else
    __builtin_amdgcn_raw_buffer_store_b8(__builtin_bit_cast(uint8_t, v), row_rsrc(row), (int)o.v, 0, 0);
}
template <typename T, bool R>
__device__ __forceinline__ void str_stream(T* row, LaneOffset<R>& o, T v) {
  static_assert(sizeof(T) == 4 || sizeof(T) == 1, "rows hold 32-bit words or bytes");
  constexpr int aux = MT_NT_STORES ? kAuxNt : 0;
  if constexpr (sizeof(T) == 4)
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), row_rsrc(row), (int)o.v, 0, aux);
  else
    __builtin_amdgcn_raw_buffer_store_b8(__builtin_bit_cast(uint8_t, v), row_rsrc(row), (int)o.v, 0, aux);
}
#else
template <typename T, bool R>
__device__ __forceinline__ T ldr(const T* row, LaneOffset<R>& o) {
  return *(global_ptr<const T>)(uniform_row(row) + lane_offset(o));
}
template <typename T, bool R>
__device__ __forceinline__ void str(T* row, LaneOffset<R>& o, T v) {
  *(global_ptr<T>)(uniform_row(row) + lane_offset(o)) = v;
}
template <typename T, bool R>
__device__ __forceinline__ void str_stream(T* row, LaneOffset<R>& o, T v) {
  global_ptr<T> p = (global_ptr<T>)(uniform_row(row) + lane_offset(o));
#if MT_NT_STORES
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}
#endif

// ---------------------------------------------------------------------------
// DH table views.  The chain code below is written once against this
// interface; `RtTable` reads the per-launch constants (SGPRs or LDS), the
// static tables are compile-time constants, so after unrolling every product
// with a 0 / +-1 entry disappears and the chain collapses to the arm's closed
// form (for the reference arm: z_elbow = d0 + d2 c1, z_ee = z_elbow +
// a3 (c1 s3' ... )).  The host picks a static table only
// when the configured table matches it exactly (engine.hip: match_static).
// ---------------------------------------------------------------------------
// Which rows of joints_coordinates the observation (manytor.py:143: [2]) and the pickup (:162: [3]) are measured from,
// and whose z the ground test looks at (:191: both).  The reference arm fixes them to the last two rows, and so do the
// tables below (kFrames == false: D - 2 and D - 1 at compile time).  RtTableF takes them from the launch constants
// (mt_config.obs_frame / ee_frame); row 0 is the all-zero origin row (manytor.py:189).
template <int D_>
struct RtTable {
  static constexpr int D = D_;
  static constexpr bool kFrames = false;
  const DhConst& c;
  __device__ __forceinline__ float a(int j) const { return c.a[j]; }
  __device__ __forceinline__ float d(int j) const { return c.d[j]; }
  __device__ __forceinline__ float sa(int j) const { return c.sa[j]; }
  __device__ __forceinline__ float ca(int j) const { return c.ca[j]; }
  __device__ __forceinline__ float off(int j) const { return c.off_deg[j]; }
};

template <int D_>
struct RtTableF {
  static constexpr int D = D_;
  static constexpr bool kFrames = true;
  const DhConst& c;
  __device__ __forceinline__ float a(int j) const { return c.a[j]; }
  __device__ __forceinline__ float d(int j) const { return c.d[j]; }
  __device__ __forceinline__ float sa(int j) const { return c.sa[j]; }
  __device__ __forceinline__ float ca(int j) const { return c.ca[j]; }
  __device__ __forceinline__ float off(int j) const { return c.off_deg[j]; }
  __device__ __forceinline__ int fo() const { return c.fo; }
  __device__ __forceinline__ int fe() const { return c.fe; }
};

#define MT_SEL8(j, v0, v1, v2, v3, v4, v5, v6, v7) \
  ((j) == 0 ? (v0) : (j) == 1 ? (v1) : (j) == 2 ? (v2) : (j) == 3 ? (v3) : (j) == 4 ? (v4) : (j) == 5 ? (v5) : (j) == 6 ? (v6) : (v7))

// The reference arm, manytor.py:42-48: rows (a, alpha, d, theta offset) =
// (0,-pi/2,4.3,0) (0,pi/2,0,0) (0,-pi/2,24.3,0) (27,pi/2,0,-pi/2).
struct Ref4Table {
  static constexpr int D = 4;
  static constexpr bool kFrames = false;
  __host__ __device__ static constexpr float a(int j) { return MT_SEL8(j, 0.f, 0.f, 0.f, 27.0f, 0.f, 0.f, 0.f, 0.f); }
  __host__ __device__ static constexpr float d(int j) { return MT_SEL8(j, 4.3f, 0.f, 24.3f, 0.f, 0.f, 0.f, 0.f, 0.f); }
  __host__ __device__ static constexpr float sa(int j) { return MT_SEL8(j, -1.f, 1.f, -1.f, 1.f, 0.f, 0.f, 0.f, 0.f); }
  __host__ __device__ static constexpr float ca(int j) { return MT_SEL8(j, 0.f, 0.f, 0.f, 0.f, 1.f, 1.f, 1.f, 1.f); }
  __host__ __device__ static constexpr float off(int j) { return MT_SEL8(j, 0.f, 0.f, 0.f, -90.f, 0.f, 0.f, 0.f, 0.f); }
};

// The 7-joint table of BASELINE.json configs[4] (manytor_amd/engine.py DH7_TABLE, fixture F7).
struct Dh7Table {
  static constexpr int D = 7;
  static constexpr bool kFrames = false;
  __host__ __device__ static constexpr float a(int j) { return MT_SEL8(j, 0.f, 0.f, 4.5f, -4.5f, 0.f, 8.8f, 0.f, 0.f); }
  __host__ __device__ static constexpr float d(int j) { return MT_SEL8(j, 34.0f, 0.f, 40.0f, 0.f, 40.0f, 0.f, 12.6f, 0.f); }
  __host__ __device__ static constexpr float sa(int j) { return MT_SEL8(j, -1.f, 1.f, 1.f, -1.f, -1.f, 1.f, 0.f, 0.f); }
  __host__ __device__ static constexpr float ca(int j) { return MT_SEL8(j, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 1.f, 1.f); }
  __host__ __device__ static constexpr float off(int j) { return MT_SEL8(j, 0.f, 0.f, 0.f, 0.f, 0.f, -90.f, 0.f, 0.f); }
};

// ---------------------------------------------------------------------------
// DH chain.  R = [X Y Z] (columns), p = origin.  One joint:
//   X' = X c + Y s ;  T = Y c - X s ;  Y' = T ca + Z sa ;  Z' = Z ca - T sa
//   p' = p + a X' + d Z
// which is R' = R * M(theta, alpha), p' = p + R * (a c, a s, d) for the matrix
// M of manytor.py:28-31.
// ---------------------------------------------------------------------------
template <class Tbl>
__device__ __forceinline__ void chain_all(const float (&s)[Tbl::D], const float (&c)[Tbl::D], const Tbl& t,
                                          float (&p)[Tbl::D][3]) {
  constexpr int D = Tbl::D;
  float X[3] = {1.f, 0.f, 0.f}, Y[3] = {0.f, 1.f, 0.f}, Z[3] = {0.f, 0.f, 1.f};
  float o[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < D; ++j) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const float nx = pfma(X[q], c[j], pmul(Y[q], s[j]));
      const float tt = pfma(Y[q], c[j], -pmul(X[q], s[j]));
      o[q] = pfma(t.a(j), nx, pfma(t.d(j), Z[q], o[q]));
      X[q] = nx;
      Y[q] = pfma(tt, t.ca(j), pmul(Z[q], t.sa(j)));
      Z[q] = pfma(Z[q], t.ca(j), -pmul(tt, t.sa(j)));
      p[j][q] = o[q];
    }
  }
}

// z components only, of the frames after D-1 and after D joints: what the
// ground test of manytor.py:191 needs.  Joint 0's angle drops out (the z row of
// the identity is (0,0,1)), which the compiler sees after unrolling.
template <class Tbl>
__device__ __forceinline__ void chain_z(const float (&s)[Tbl::D], const float (&c)[Tbl::D], const Tbl& t, float& z_obs,
                                        float& z_ee) {
  constexpr int D = Tbl::D;
  float x = 0.f, y = 0.f, z = 1.f, o = 0.f;
  z_obs = 0.f;
  if constexpr (Tbl::kFrames) z_ee = 0.f;
#pragma unroll
  for (int j = 0; j < D; ++j) {
    const float nx = pfma(x, c[j], pmul(y, s[j]));
    const float tt = pfma(y, c[j], -pmul(x, s[j]));
    o = pfma(t.a(j), nx, pfma(t.d(j), z, o));
    x = nx;
    y = pfma(tt, t.ca(j), pmul(z, t.sa(j)));
    z = pfma(z, t.ca(j), -pmul(tt, t.sa(j)));
    if constexpr (Tbl::kFrames) {  // row 0 of joints_coordinates is the origin: z = 0 whatever joint 0 does
      if (j >= 1 && j == t.fo()) z_obs = o;
      if (j >= 1 && j == t.fe()) z_ee = o;
    } else {
      if (j == D - 2 && D > 2) z_obs = o;
    }
  }
  if constexpr (!Tbl::kFrames) z_ee = o;
}

// The observation frame and the pickup frame out of the chain's frame origins.
template <class Tbl>
__device__ __forceinline__ void pick_frames(const Tbl& t, const float (&p)[Tbl::D][3], float (&el)[3], float (&e)[3]) {
  constexpr int D = Tbl::D;
  if constexpr (Tbl::kFrames) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      el[q] = 0.f;
      e[q] = 0.f;
    }
#pragma unroll
    for (int j = 1; j < D; ++j)
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        el[q] = (j == t.fo()) ? p[j][q] : el[q];
        e[q] = (j == t.fe()) ? p[j][q] : e[q];
      }
  } else {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      el[q] = (D > 2) ? p[D - 2][q] : 0.f;  // joints_coordinates[-2]; row 0 is zeros (manytor.py:189)
      e[q] = p[D - 1][q];
    }
  }
}

// One target: observation triple (manytor.py:150-152, :17-22) and pickup test
// (manytor.py:160-168).
__device__ __forceinline__ void observe_target(const float (&el)[3], float x, float y, float z, float& dist, float& r,
                                               float& th) {
  const float m0 = fabsf(el[0] - x), m1 = fabsf(el[1] - y), m2 = fabsf(el[2] - z);
  const float h2 = __builtin_fmaf(m0, m0, m1 * m1);
  const float h = __builtin_amdgcn_sqrtf(h2);            // v_sqrt_f32, 1 ulp: no refinement sequence
  dist = __builtin_amdgcn_sqrtf(__builtin_fmaf(m2, m2, h2));
  r = atan2_deg_q1(m0, m1);
  th = atan2_deg_q1(h, m2);
}

__device__ __forceinline__ bool within_box(const float (&e)[3], float x, float y, float z, float tol) {
  return (fabsf(e[0] - x) <= tol) && (fabsf(e[1] - y) <= tol) && (fabsf(e[2] - z) <= tol);
}

// One target inside step(): obs2 triple (taken before the pickup, manytor.py:204), pickup test
// (manytor.py:206), and the zeroing of a target that died earlier (manytor.py:148).
template <bool ABLATE_OBS, bool R>
__device__ __forceinline__ void step_target(const StepArgs& a, int64_t ld, LaneOffset<R>& o4, int k, uint32_t am, uint32_t& nam,
                                            const float (&el)[3], const float (&e)[3], float x, float y, float z) {
  const bool al = (am >> k) & 1u;
  float dist = 0.f, r = 0.f, th = 0.f;
  if (al) {
    if (ABLATE_OBS) {
      dist = x + el[0];
      r = y + el[1];
      th = z + el[2];
    } else {
      observe_target(el, x, y, z, dist, r, th);
    }
    if (within_box(e, x, y, z, a.tol)) nam &= ~(1u << k);
  } else if ((x != 0.f) | (y != 0.f) | (z != 0.f)) {
    float* row = a.points + (int64_t)(3 * k) * ld;
    str(row, o4, 0.f);
    str(row + ld, o4, 0.f);
    str(row + 2 * ld, o4, 0.f);
  }
  float* orow = a.obs + (int64_t)(3 * k) * ld;
  str_stream(orow, o4, dist);
  str_stream(orow + ld, o4, r);
  str_stream(orow + 2 * ld, o4, th);
}

// ---------------------------------------------------------------------------
// step: Environment.step() for all envs (manytor.py:255-260 + :175-213).
//   Tbl    : RtTable<D> (any table, constants per launch) or a static table
//   SAMPLE : draw the action in-kernel (== sample_actions_kernel then step)
//   TRIG   : how sin/cos of the S-2 interior sub-step poses are obtained
//            0 angle-addition recurrence: the poses are g + k*delta, so
//              (c,s)_{k+1} = (c,s)_k rotated by delta -- 4 FMA-class ops per joint
//              instead of a sincos.  Run forward from the previous pose (k = 0,
//              exact) and backward from the action (k = S-1, exact) to the middle,
//              which halves the drift (<= 12 rotations, < 1e-4 in z, inside the
//              1e-3 guard band of the ground flag) and gives two independent
//              dependency chains per iteration.
//            1 polynomial sincos at every sub-step (reference-shaped, slowest)
//            2 hardware v_sin_f32/v_cos_f32 at every interior sub-step
//            3, 4 DIAGNOSTIC ablations for profiling (MT_FLAG_ABLATE_*): outputs wrong
//   LDS    : stage the runtime DH constants in LDS instead of SGPRs (measured
//            variant; BASELINE.json's north_star asks for the comparison)
// The first and last pose are always evaluated with the polynomial sincos.
// ---------------------------------------------------------------------------
template <class Tbl>
struct TableMaker {
  static __device__ __forceinline__ Tbl make(const DhConst&) { return Tbl{}; }
};
template <int D>
struct TableMaker<RtTable<D>> {
  static __device__ __forceinline__ RtTable<D> make(const DhConst& c) { return RtTable<D>{c}; }
};
template <int D>
struct TableMaker<RtTableF<D>> {
  static __device__ __forceinline__ RtTableF<D> make(const DhConst& c) { return RtTableF<D>{c}; }
};

// The step index that keys this launch's action draw (see StepArgs::major_base; the address is wave-uniform: s_load).
__device__ __forceinline__ uint32_t step_index(const StepArgs& a) {
  return a.major_base ? a.major + *a.major_base : a.major;
}

// Environment.action_sample for one env (manytor.py:215-217): D integer degrees from one Philox block.
template <int D>
__device__ __forceinline__ void draw_action(uint64_t seed, uint64_t env_id, uint32_t step_idx, float (&act)[D]) {
  static_assert(D <= 8, "one Philox block yields at most 8 joint angles");
  const u32x4 w = stream_block(seed, env_id, kTagAction, step_idx, 0u);
  const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
  for (int j = 0; j < D; ++j) act[j] = (j < 4) ? action_from_word(ws[j]) : action_from_word_digit1(ws[j - 4]);
}

// The kinematic part of Environment.action (manytor.py:178-192) for one env: S poses on the straight line in
// joint space from `g` (previous pose) to `act`.  Returns the elbow and end-effector positions at the final pose
// and the minimum z of those two frames over all S poses (ground flag <=> zmin < 0).  Shared by step_kernel and
// rollout_kernel so both evaluate exactly the same arithmetic.
// Joints whose angle can influence the z of the last two frames (what chain_z computes).  Joint 0 never does (handled
// by the callers: index 0 is skipped everywhere); the LAST joint does only through a(D-1) * X'.z, so for a static
// table whose last row has a = 0 (the 7-joint table: a pure d offset along the previous z) its sine / cosine are never
// needed at the interior sub-steps.  The compiler does not discover this by itself (the rotation is a loop-carried
// cycle): 6 of 57 instructions of the 7-joint sub-step loop.
template <class Tbl>
struct ZJoints {
  static constexpr int value = (Tbl::a(Tbl::D - 1) == 0.f) ? Tbl::D - 1 : Tbl::D;
};
template <int D>
struct ZJoints<RtTable<D>> {
  static constexpr int value = D;
};
template <int D>
struct ZJoints<RtTableF<D>> {
  static constexpr int value = D;
};

// rotate (s, c) of joints 1..JN-1 by +delta (SIGN = +1) or -delta (SIGN = -1): the angle-addition step of the recurrence
template <int D, int JN, int SIGN>
__device__ __forceinline__ void rotate_pose(float (&s)[D], float (&c)[D], const float (&sd)[D], const float (&cd)[D]) {
#pragma unroll
  for (int j = 1; j < JN; ++j) {
    const float c2 = __builtin_fmaf(c[j], cd[j], SIGN > 0 ? -(s[j] * sd[j]) : s[j] * sd[j]);
    s[j] = __builtin_fmaf(s[j], cd[j], SIGN > 0 ? c[j] * sd[j] : -(c[j] * sd[j]));
    c[j] = c2;
  }
}

// sin / cos of the per-sub-step increment.  It is small (|delta| <= 15 degrees for sampled actions at S = 25): when
// every lane of the wave has |delta| <= 45 the quadrant reduction is skipped.  Both paths give identical bits for
// such angles (sincos_deg reduces with q = 0), so a lane's result does not depend on its wave-mates.
template <int D, int JN>
__device__ __forceinline__ void sincos_increment(const float (&st)[D], float (&sd)[D], float (&cd)[D]) {
#pragma unroll
  for (int j = 0; j < D; ++j) {
    sd[j] = 0.f;
    cd[j] = 1.f;
  }
  bool small = true;
#pragma unroll
  for (int j = 1; j < JN; ++j) small &= fabsf(st[j]) <= 45.0f;
  if (__all(small)) {
#pragma unroll
    for (int j = 1; j < JN; ++j) sincos_deg_small(st[j], sd[j], cd[j]);
  } else {
#pragma unroll
    for (int j = 1; j < JN; ++j) sincos_deg(st[j], sd[j], cd[j]);
  }
}

// Sines / cosines of a SAMPLED action (rollout kernels).  draw_action yields integer degrees in [-180, 180), and the
// joint offsets of the compile-time tables are whole degrees in [-90, 0], so action + offset is one of the 450
// integers -270 .. 179: the block fills an LDS table with sincos_deg of exactly those floats once per launch and every
// step looks its D pairs up instead of evaluating D range reductions and polynomials (~30 instructions each, 8-11 %
// of a rollout step).  Same function on the same float => the same bits as computing it in place.
constexpr int kTrigBias = 270;
constexpr int kTrigEntries = 450;
struct SinCos {
  float s, c;
};
template <class Tbl>
struct ActionTrigTable {
  static constexpr bool value = false;
};
template <>
struct ActionTrigTable<Ref4Table> {
  static constexpr bool value = true;
};
template <>
struct ActionTrigTable<Dh7Table> {
  static constexpr bool value = true;
};
template <class Tbl>
constexpr bool whole_degree_offsets() {
  for (int j = 0; j < Tbl::D; ++j) {
    const float o = Tbl::off(j);
    if (o != (float)(int)o || o < -90.f || o > 0.f) return false;
  }
  return true;
}
// The table itself: sincos_deg of exactly the floats -270 .. 179, computed ONCE per handle by this kernel (mt_create) into
// the arena.  Every block that wants it copies the 3 600 bytes into LDS (one 16-byte load per thread from L2, instead
// of two range reductions + polynomials per thread and launch): the load is issued first, the write and the barrier
// come after whatever the kernel can do in between (its pose loads, the Philox block).
constexpr int kTrigVec4 = kTrigEntries * 2 / 4;  // the table as float4s
static_assert(kTrigEntries * 2 % 4 == 0 && kTrigVec4 <= kBlock, "one float4 per thread stages the table");
__global__ __launch_bounds__(kBlock) void fill_trig_table_kernel(float* table) {
  for (int idx = threadIdx.x; idx < kTrigEntries; idx += kBlock) {
    float sv, cv;
    sincos_deg((float)(idx - kTrigBias), sv, cv);
    table[2 * idx] = sv;
    table[2 * idx + 1] = cv;
  }
}
__device__ __forceinline__ float4 trig_table_load(const float* table) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (threadIdx.x < kTrigVec4) v = reinterpret_cast<const float4*>(table)[threadIdx.x];
  return v;
}
// every wave of the block executes this exactly once (the barrier counts arrivals, wherever the s_barrier sits)
// Values that must be COMPLETE, in the instruction stream too, before the next barrier / wait: the compiler is free to sink
// pure arithmetic below a __syncthreads(), and it does so with the whole Philox block of the step kernels -- which then runs
// AFTER the wait for the table load, i.e. after (nearly) every load of the kernel had come back (vmcnt counts in issue
// order, and the conditional prefetch loads make the count conservative), instead of under their latency.
template <int D>
__device__ __forceinline__ void complete_before_here(float (&v)[D]) {
#pragma unroll
  for (int j = 0; j < D; ++j) asm volatile("" : "+v"(v[j]));
}
__device__ __forceinline__ void trig_table_commit(SinCos* lds, const float4& v) {
  if (threadIdx.x < kTrigVec4) reinterpret_cast<float4*>(lds)[threadIdx.x] = v;
  __syncthreads();
}
template <class Tbl, bool TABLE>
__device__ __forceinline__ void action_sincos(const Tbl& t, const float (&act)[Tbl::D], float (&sv)[Tbl::D],
                                              float (&cv)[Tbl::D], const SinCos* trig) {
#pragma unroll
  for (int j = 0; j < Tbl::D; ++j) {
    if constexpr (TABLE) {
      static_assert(whole_degree_offsets<Tbl>(), "the table covers whole-degree offsets in [-90, 0] only");
      const SinCos v = trig[(int)act[j] + ((int)Tbl::off(j) + kTrigBias)];
      sv[j] = v.s;
      cv[j] = v.c;
    } else {
      sincos_deg(act[j] + t.off(j), sv[j], cv[j]);
    }
  }
}

// The kinematic part of Environment.action (manytor.py:178-192) for one env: S poses on the straight line in
// joint space from `g` (previous pose) to `act`.  Returns the elbow and end-effector positions at the final pose
// and the minimum z of those two frames over all S poses (ground flag <=> zmin < 0).  Shared by step_kernel and
// rollout_kernel so both evaluate exactly the same arithmetic.
//
// Two schedules of the recurrence (TRIG == 0), same poses, same arithmetic per pose, hence the same bits (min is
// order-free):
//   interleaved (D <= 5): forward and backward half advance in the same loop iteration -- two independent
//     dependency chains, and 56 VGPRs for the reference arm;
//   sequential (D >= 6): forward half first, then the action's sincos, the full chain and the backward half.  The two
//     halves never hold their (s, c) state at the same time: 4 (D - 1) fewer live registers, which is what takes the
//     7-joint kernel from 97 VGPRs (4 waves/SIMD) to <= 64 (8 waves/SIMD).
//
// PoseCache (rollout_kernel only): inside a rollout the pose a step starts from is the action of the step before, whose
// sines / cosines and frame heights were computed then.  Handing them over saves the k = 0 pose's sincos and z chain
// (JN - 1 sincos + one chain_z per step, 6 % of the instructions for the reference arm).  Same inputs to the same
// functions, so the bits do not change; when any lane of the wave has no valid cache (first step of a launch, right
// after an in-kernel re-arm) the whole wave simply recomputes.
template <int D>
struct PoseCache {
  float s[D], c[D];  // joints 1 .. JN-1 of the pose the next step starts from
  float zmin;        // min z of its last two frames
};

// sines / cosines of joints 1 .. JN-1 at the pose a step starts from: out of the whole-degree table when the host
// knows every angle of the batch to be a whole degree in [-180, 180) (kFlagWholeGoals, wave-uniform), else computed.
// The table holds sincos_deg of the very float g + off, so both ways give the same bits.
template <class Tbl, bool TABLE>
__device__ __forceinline__ void prev_pose_sincos(const Tbl& t, const float (&g)[Tbl::D], float (&sF)[Tbl::D],
                                                 float (&cF)[Tbl::D], const SinCos* trig, bool whole) {
  constexpr int JN = ZJoints<Tbl>::value;
  if constexpr (TABLE) {
    if (whole) {
#pragma unroll
      for (int j = 1; j < JN; ++j) {
        const SinCos v = trig[(int)g[j] + ((int)Tbl::off(j) + kTrigBias)];
        sF[j] = v.s;
        cF[j] = v.c;
      }
      return;
    }
  }
#pragma unroll
  for (int j = 1; j < JN; ++j) sincos_deg(g[j] + t.off(j), sF[j], cF[j]);
}

template <class Tbl, int TRIG, bool CACHED = false, bool TABLE = false>
__device__ __forceinline__ float route_kinematics(const Tbl& t, int S, float inv_sm1, const float (&g)[Tbl::D],
                                                  const float (&act)[Tbl::D], float (&el)[3], float (&e)[3],
                                                  PoseCache<Tbl::D>* cache = nullptr, bool cache_valid = false,
                                                  const SinCos* trig = nullptr, bool prev_whole = false) {
  constexpr int D = Tbl::D;
  constexpr int JN = ZJoints<Tbl>::value;
  constexpr bool kSequential = (TRIG == 0 || TRIG == 5) && D >= 6;
  const bool use_cache = CACHED && __all(cache_valid);
  // route[k] = goals + k * (action - goals) / (S-1), route[S-1] = action (np.linspace, manytor.py:182)
  float st[D];
#pragma unroll
  for (int j = 0; j < D; ++j) st[j] = (act[j] - g[j]) * inv_sm1;

  float zmin, zo, ze;
  if (kSequential) {
    const int nf = (S - 1) / 2;   // forward poses k = 1..nf
    const int nb = S - 2 - nf;    // backward poses k = S-2..nf+1   (nb = nf or nf-1)
    float sd[D], cd[D];
    sincos_increment<D, JN>(st, sd, cd);
    {  // k = 0 (the previous pose, manytor.py:182-192 evaluates it again) and the forward half
      float sF[D], cF[D];
#pragma unroll
      for (int j = 0; j < D; ++j) {
        sF[j] = 0.f;
        cF[j] = 1.f;
      }
      if (use_cache) {
#pragma unroll
        for (int j = 1; j < JN; ++j) {
          sF[j] = cache->s[j];
          cF[j] = cache->c[j];
        }
        zmin = cache->zmin;
      } else {
        prev_pose_sincos<Tbl, TABLE>(t, g, sF, cF, trig, prev_whole);
        chain_z<Tbl>(sF, cF, t, zo, ze);
        zmin = fminf(zo, ze);
      }
#pragma unroll 2
      for (int it = 1; it <= nf; ++it) {
        rotate_pose<D, JN, +1>(sF, cF, sd, cd);
        chain_z<Tbl>(sF, cF, t, zo, ze);
        zmin = fminf(zmin, fminf(zo, ze));
      }
    }
    // k = S-1: the action itself, full chain (positions are consumed by the caller), then the backward half
    float sB[D], cB[D], p[D][3];
    action_sincos<Tbl, TABLE>(t, act, sB, cB, trig);
    chain_all<Tbl>(sB, cB, t, p);
    pick_frames<Tbl>(t, p, el, e);
    zmin = fminf(zmin, fminf(el[2], e[2]));
    if (CACHED) {
#pragma unroll
      for (int j = 1; j < JN; ++j) {
        cache->s[j] = sB[j];
        cache->c[j] = cB[j];
      }
      cache->zmin = fminf(el[2], e[2]);
    }
    sB[0] = 0.f;
    cB[0] = 1.f;
#pragma unroll 2
    for (int it = 1; it <= nb; ++it) {
      rotate_pose<D, JN, -1>(sB, cB, sd, cd);
      chain_z<Tbl>(sB, cB, t, zo, ze);
      zmin = fminf(zmin, fminf(zo, ze));
    }
    return zmin;
  }

  // k = S-1: the action itself, full chain (positions are consumed by the caller)
  float sA[D], cA[D], p[D][3];
  action_sincos<Tbl, TABLE>(t, act, sA, cA, trig);
  chain_all<Tbl>(sA, cA, t, p);
  pick_frames<Tbl>(t, p, el, e);
  zmin = fminf(el[2], e[2]);

  // k = 0: the previous pose (manytor.py:182-192 evaluates it again: a pose left below ground costs -1 twice)
  float sF[D], cF[D];
#pragma unroll
  for (int j = 0; j < D; ++j) {
    sF[j] = 0.f;
    cF[j] = 1.f;
  }
  if (use_cache) {
#pragma unroll
    for (int j = 1; j < JN; ++j) {
      sF[j] = cache->s[j];
      cF[j] = cache->c[j];
    }
    zmin = fminf(zmin, cache->zmin);
  } else {
    prev_pose_sincos<Tbl, TABLE>(t, g, sF, cF, trig, prev_whole);
    chain_z<Tbl>(sF, cF, t, zo, ze);
    zmin = fminf(zmin, fminf(zo, ze));
  }
  if (CACHED) {
#pragma unroll
    for (int j = 1; j < JN; ++j) {
      cache->s[j] = sA[j];
      cache->c[j] = cA[j];
    }
    cache->zmin = fminf(el[2], e[2]);
  }

  if (TRIG == 3 || TRIG == 4) {
    // diagnostic builds: no interior sub-steps
  } else if (TRIG == 0 || TRIG == 5) {
    float sd[D], cd[D];
    sincos_increment<D, JN>(st, sd, cd);
    float sB[D], cB[D];
#pragma unroll
    for (int j = 0; j < D; ++j) {
      sB[j] = sA[j];
      cB[j] = cA[j];
    }
    sB[0] = 0.f;
    cB[0] = 1.f;
    const int nf = (S - 1) / 2;   // forward poses k = 1..nf
    const int nb = S - 2 - nf;    // backward poses k = S-2..nf+1   (nb = nf or nf-1)
    // both halves advance together for nb iterations (no condition inside the loop: the two chains' registers stay where
    // they are), then the forward half's extra pose when the number of interior poses is odd
#pragma unroll 2
    for (int it = 1; it <= nb; ++it) {
      rotate_pose<D, JN, +1>(sF, cF, sd, cd);
      chain_z<Tbl>(sF, cF, t, zo, ze);
      zmin = fminf(zmin, fminf(zo, ze));
      rotate_pose<D, JN, -1>(sB, cB, sd, cd);
      chain_z<Tbl>(sB, cB, t, zo, ze);
      zmin = fminf(zmin, fminf(zo, ze));
    }
    if (nf > nb) {
      rotate_pose<D, JN, +1>(sF, cF, sd, cd);
      chain_z<Tbl>(sF, cF, t, zo, ze);
      zmin = fminf(zmin, fminf(zo, ze));
    }
  } else {
    float gq[D];
#pragma unroll
    for (int j = 0; j < D; ++j) gq[j] = g[j] + t.off(j);
    for (int k = 1; k < S - 1; ++k) {
      const float fk = (float)k;
#pragma unroll
      for (int j = 1; j < JN; ++j) {
        const float pose = __builtin_fmaf(fk, st[j], gq[j]);
        if (TRIG == 2)
          sincos_deg_hw(pose, sF[j], cF[j]);
        else
          sincos_deg(pose, sF[j], cF[j]);
      }
      chain_z<Tbl>(sF, cF, t, zo, ze);
      zmin = fminf(zmin, fminf(zo, ze));
    }
  }
  return zmin;
}

// An env finished its episode `episode` with return `ret` and is being re-armed: keep the return in the ring
// (slot = number of episodes it finished before, modulo the ring size) and in the one-slot last_return row.
__device__ __forceinline__ void record_finished(const StepArgs& a, uint32_t i, uint32_t episode, float ret) {
  LaneOffset<true> o4{i * 4u};
  str(a.last_return, o4, ret);
  if (a.ring_slots)  // the slot differs from lane to lane: plain per-lane addressing, not a uniform row
    a.ring[(int64_t)((episode - a.episode0) % a.ring_slots) * a.ld + i] = ret;
}

// In-kernel phase stamps (cdna_hip_programming.md section 7): only in the diagnostic build of
// tools/microbench/step_stamps.hip (-DMT_STAMPS); in the library the macro is empty and no stamp executes.
#ifdef MT_STAMPS
#define MT_STAMP(a, i, slot)                                                                 \
  do {                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    unsigned long long t__;                                                                  \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");              \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    if ((a).stamps && (threadIdx.x & 63) == 0) (a).stamps[(size_t)((i) >> 6) * 8 + (slot)] = t__; \
  } while (0)
#else
#define MT_STAMP(a, i, slot) \
  do {                       \
  } while (0)
#endif

// Staged actions come from outside (a policy, a host array): anything that is not a finite angle of at most
// 2^15 degrees in magnitude -- NaN, +-inf, garbage -- would poison `goals` for good.  Tested on the bit pattern,
// so the check survives -ffinite-math-only.
__device__ __forceinline__ bool unusable_angle(float v) { return (__float_as_uint(v) & 0x7FFFFFFFu) > 0x47000000u; }

// Asking for 8 waves per SIMD (= 64 VGPRs) only where the default recurrence kernel is within reach of it: the
// 7-joint table lands on 68 without the request and fits 64 with it, no scratch (profiles/r02_kernel_resources.txt).
template <class Tbl, int TRIG>
constexpr int step_min_waves() { return (TRIG == 0 && Tbl::D >= 6 && Tbl::D <= 7) ? 8 : 1; }

//   PF     : number of targets whose coordinates are requested at the very top, ahead of the kinematics (0 or
//            kPrefetch).  In-kernel stamps (tools/microbench/step_stamps.hip) show that with few waves per SIMD the
//            target loop is a chain of exposed load latencies -- 55 % of a wave's lifetime at 65 536 arms -- which the
//            ~3700 cycles of kinematics hide completely if the loads are already in flight.  Costs 3 * PF registers,
//            so the host uses it where occupancy is not the limit (small batches); same arithmetic, same bits.
//            (Staging the same rows through LDS-DMA -- global_load_lds_dword into a [3K][256] tile, no VGPRs -- was
//            measured and is slower at every batch size: profiles/r02_variants.md section 5.)
constexpr int kPrefetch = 8;

//   TT     : sampled actions are whole degrees, and so is the pose they leave behind: with a compile-time table (whole-
//            degree joint offsets) the sines / cosines of BOTH end poses of the route come out of the 450-entry
//            whole-degree table (fill_trig_table_kernel), staged into LDS per block, instead of 2 (JN - 1) + 1 range
//            reductions and polynomials per env: 1 277 -> 1 153 instructions per wave for the reference arm, 1.5-6 % time
//            (most where an env is spread over lanes and the end poses are replicated; profiles/r03_variants.md section 3).
//            Whole batch or a 256-aligned range of it only (threads past the end run up to the barrier: their loads stay
//            inside the rows, which are ld >= round_up(n, 256) long); same bits as the computed values.
//   FLAT   : the rows are addressed through LaneOffset<false> (see there): the HBM-bound launches of the prefetch kernel.
__device__ __forceinline__ void draw_targets_wave(uint64_t seed, uint64_t env0, uint32_t episode, bool need, int K, float radius,
                                                  float* col0, uint8_t* slots);

//   FRESH  : the launch BEGINS with the full random reset mt_reset_random deferred to it (the chained launch-per-step form of
//            mt_rollout, engine.hip): instead of loading pose / alive mask / return / targets the kernel keeps the finished
//            return (MT_F_LAST_RETURN), starts from the zero pose with every target alive, stores the episode index, draws
//            the targets with reset_kernel's wave-cooperative draw into LDS columns and writes them out -- reset_kernel's
//            state, bit for bit, without its launch, without its stores of what this step overwrites anyway (pose, alive
//            mask, return, end effector, reward, done: 41 B per env) and without this step's loads of what it would have
//            written (108 B per env).  Dynamic LDS: [3K][kBlock] floats.
template <class Tbl, bool SAMPLE, int TRIG, bool LDS, int PF = 0, bool TT = false, bool FLAT = false, bool FRESH = false>
__global__ __launch_bounds__(kBlock, (PF ? 1 : step_min_waves<Tbl, TRIG>())) void step_kernel(const StepArgs a) {
  constexpr int D = Tbl::D;
  static_assert(!TT || (SAMPLE && TRIG == 0 && !LDS && ActionTrigTable<Tbl>::value), "the table serves sampled actions of a static table");
  static_assert(!FRESH || TT, "the reset prologue exists for the sampled-action kernels of the static tables");
  extern __shared__ float fresh_stage[];                          // FRESH: the drawn targets, [3K][kBlock]
  __shared__ uint8_t fresh_slots[FRESH ? kBlock / 64 : 1][64];    // FRESH: draw_targets_wave's scratch
  __shared__ DhConst sh;
  __shared__ __attribute__((aligned(16))) SinCos trig_lds[TT ? kTrigEntries : 1];
  float4 trig_v;
  if (TT) trig_v = trig_table_load(a.trig_table);
  if (LDS) {
    const float* src = reinterpret_cast<const float*>(&a.dh);
    float* dst = reinterpret_cast<float*>(&sh);
    if (threadIdx.x < sizeof(DhConst) / sizeof(float)) dst[threadIdx.x] = src[threadIdx.x];
    __syncthreads();
  }
  const Tbl t = TableMaker<Tbl>::make(LDS ? sh : a.dh);

  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;  // 32-bit lane offset: rows are addressed as uniform base + i
  LaneOffset<!FLAT> o4{i * 4u}, o1{i};  // byte offsets of this lane in the 32-bit rows / the byte rows
  if (!TT && i >= a.n) return;  // (TT, and with it FRESH: every thread stays up to the barrier / takes part in the draw)
  const int64_t ld = a.ld;
  // Read ahead of every row access: behind one (its lane offset renewal is an opaque asm to the memory analysis) the
  // word is no longer provably unclobbered, and the s_load turns into a vector load with an s_waitcnt vmcnt(0) behind
  // it -- which also waits for every prefetched row (measured: +2 % per step at 1 M envs under graph replay).
  const uint32_t step_no = SAMPLE ? step_index(a) : 0u;

  // Long arms (D >= 6) are register-bound: their kernel keeps nothing alive across the sub-step loops that it can
  // fetch or store on the other side of them (alive mask and return loaded after, new goals stored before).
  constexpr bool kLean = D >= 6;
  MT_STAMP(a, i, 0);
  float g[D], act[D];
#pragma unroll
  for (int j = 0; j < D; ++j) g[j] = FRESH ? 0.f : ldr(a.goals + j * ld, o4);
  float tx[PF ? PF : 1][3];
  if constexpr (FRESH) {
    const bool mine = i < a.n;
    const uint32_t lane = threadIdx.x & 63u;
    if (mine) {
      str(a.last_return, o4, ldr(a.total_reward, o4));  // what reset_kernel<.., RANDOM, false> keeps of the episode that ends here
      str(a.episodes, o4, a.reset_episode);
    }
    draw_targets_wave(((uint64_t)a.reset_seed_hi << 32) | a.reset_seed_lo, (uint64_t)(a.env_base + (i - lane)), a.reset_episode, mine,
                      a.K, a.radius, fresh_stage + (threadIdx.x - lane), fresh_slots[threadIdx.x >> 6]);
    if (mine) {
      const float* col = fresh_stage + threadIdx.x;
#pragma unroll
      for (int k = 0; k < (PF ? PF : 0); ++k)
        if (k < a.K) {
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            tx[k][q] = col[(3 * k + q) * kBlock];
            str(a.points + (int64_t)(3 * k + q) * ld, o4, tx[k][q]);
          }
        }
      for (int r = 3 * PF; r < 3 * a.K; ++r) str(a.points + (int64_t)r * ld, o4, col[r * kBlock]);
    }
  } else if (PF) {
#pragma unroll
    for (int k = 0; k < PF; ++k)
      if (k < a.K) {
        // (Loads under a condition make every later s_waitcnt vmcnt(N) conservative -- the compiler can only count the loads
        // certainly issued behind the one it waits for -- so the kinematics starts once nearly all prefetched rows are back.
        // Unconditional loads, the last row again for the slots past K, give exact counts and were measured: +2 % at
        // 98 304-131 072 envs, nothing elsewhere: profiles/r03_ab_unconditional_prefetch.txt.)
        const float* row = a.points + (int64_t)(3 * k) * ld;
#pragma unroll
        for (int q = 0; q < 3; ++q) tx[k][q] = ldr(row + q * ld, o4);
      }
  }
  // The alive mask and the return are requested now, ahead of the arithmetic that does not need them.
  // (Requesting the target rows here as well was measured: no gain, -2 waves/SIMD -- profiles/r01_variants.md.)
  uint32_t am = 0;
  float total_in = 0.f;
  if (FRESH) {
    am = (a.K >= 32) ? 0xFFFFFFFFu : ((1u << a.K) - 1u);
  } else if (!kLean || PF) {
    am = ldr(a.alive, o4);
    total_in = ldr(a.total_reward, o4);
  }
  if (SAMPLE) {
    // not stored separately: the action taken becomes `goals` below (manytor.py:184), 4D bytes of traffic saved
    draw_action<D>(((uint64_t)a.seed_hi << 32) | a.seed_lo, (uint64_t)(a.env_base + i), step_no, act);
  } else {
    bool bad = false;
#pragma unroll
    for (int j = 0; j < D; ++j) {
      act[j] = ldr(a.actions + j * ld, o4);
      bad |= unusable_angle(act[j]);
    }
    if (bad) {  // the env holds its pose this step; the call is counted (mt_bad_action_count)
#pragma unroll
      for (int j = 0; j < D; ++j) act[j] = g[j];
      atomicAdd(a.bad_actions, 1u);
    }
  }
  MT_STAMP(a, i, 1);  // action known (Philox done / staged action loaded)
  if (TT) {
    // (complete_before_here(act) -- the Philox block ahead of this wait -- was measured here: nothing at >= 262 144 envs,
    // +1 % at 131 072; the lane-split kernels gain 1-2 % and have it: profiles/r03_ab_philox_before_wait.txt)
    trig_table_commit(trig_lds, trig_v);  // the table load has had the pose loads to arrive
    if (i >= a.n) return;
  }
  if (kLean) {  // goals = action (manytor.py:184): the old pose is in registers already
#pragma unroll
    for (int j = 0; j < D; ++j) str(a.goals + j * ld, o4, act[j]);
  }

  float el[3], e[3];
  const float zmin = route_kinematics<Tbl, TRIG, false, TT>(t, a.S, a.inv_sm1, g, act, el, e, nullptr, false, trig_lds,
                                                            (a.flags & kFlagWholeGoals) != 0);
  const bool ground = zmin < 0.f;  // manytor.py:191
  if (a.zmin) str_stream(a.zmin, o4, zmin);  // MT_FLAG_DEBUG_ZMIN: wave-uniform branch on an SGPR pointer, NULL by default
  if (kLean && !PF && !FRESH) {
    am = ldr(a.alive, o4);
    total_in = ldr(a.total_reward, o4);
  }
  MT_STAMP(a, i, 2);  // kinematics done (this waits for the pose loads)

  // obs2 (before pickup, manytor.py:204) and pickup (manytor.py:206) per target
  uint32_t nam = am;
  if (PF) {
#pragma unroll
    for (int k = 0; k < PF; ++k)
      if (k < a.K) step_target<false>(a, ld, o4, k, am, nam, el, e, tx[k][0], tx[k][1], tx[k][2]);
  }
  for (int k = PF; k < a.K; ++k) {
    if (TRIG == 5) {  // DIAGNOSTIC: arithmetic only, no HBM traffic for targets / observations
      float dist, r, th;
      const float x = (float)(i & 63) + (float)k, y = 3.0f + (float)k, z = 5.0f + g[0];
      observe_target(el, x, y, z, dist, r, th);
      if (within_box(e, x, y, z, a.tol)) nam &= ~(1u << k);
      if (dist + r + th == -12345.0f) (a.obs + (int64_t)(3 * k) * ld)[i] = dist;
      continue;
    }
    const float* row = a.points + (int64_t)(3 * k) * ld;
    {
      const float x = ldr(row, o4), y = ldr(row + ld, o4), z = ldr(row + 2 * ld, o4);
      step_target<TRIG == 4>(a, ld, o4, k, am, nam, el, e, x, y, z);
    }
  }

  MT_STAMP(a, i, 3);  // target loop done
  const int32_t rew = ground ? -1 : ((nam != am) ? 1 : 0);  // manytor.py:205-212
  bool done = (nam == 0u);                                    // manytor.py:170-171
  if (a.flags & MT_FLAG_TERMINATE_ON_GROUND) done |= ground;

  if (!kLean) {
#pragma unroll
    for (int j = 0; j < D; ++j) str(a.goals + j * ld, o4, act[j]);
  }
#pragma unroll
  for (int q = 0; q < 3; ++q) str_stream(a.ee + q * ld, o4, e[q]);
  str(a.alive, o4, nam);
  str_stream(a.reward, o4, rew);
  str(a.total_reward, o4, total_in + (float)rew);  // manytor.py:258
  if (a.snap) str(a.snap, o4, total_in + (float)rew);  // (wave-uniform branch on an SGPR pointer, like a.zmin)
  str_stream(a.done, o1, (uint8_t)(done ? 1 : 0));
  const unsigned long long bits = __ballot(done);
  if ((threadIdx.x & 63) == 0) a.done_bits[i >> 6] = bits;
  MT_STAMP(a, i, 4);  // all stores issued
#ifdef MT_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  MT_STAMP(a, i, 5);  // all stores acknowledged
#endif
}

// route_kinematics for an env that is spread over L lanes (step_split_kernel, rollout_split_kernel): the same S poses,
// each with the arithmetic of route_kinematics, but a sub-lane walks only ONE half of the recurrence -- the forward half
// from the previous pose (backward == false) or the backward half from the action -- in a loop body that is the same
// for both (the start state and the sign of sin(delta) ride in registers), and the running min of z is combined over
// the env's sub-lanes at the end (lane = q * (64 / L) + e, so the partners sit 64 / L, 2 * 64 / L ... lanes apart).
// Every sub-lane computes the endpoint sincos and the full chain at the final pose itself (it needs elbow and end
// effector for its targets).  -(s * -sd) == s * sd exactly and min is order-free => the bits of route_kinematics.
template <class Tbl, int L, bool CACHED, bool TABLE = false>
__device__ __forceinline__ float route_kinematics_split(const Tbl& t, int S, float inv_sm1, const float (&g)[Tbl::D],
                                                        const float (&act)[Tbl::D], bool backward, float (&el)[3],
                                                        float (&e)[3], PoseCache<Tbl::D>* cache = nullptr,
                                                        bool cache_valid = false, const SinCos* trig = nullptr,
                                                        bool prev_whole = false) {
  constexpr int D = Tbl::D;
  constexpr int JN = ZJoints<Tbl>::value;
  constexpr int EPW = 64 / L;
  const bool use_cache = CACHED && __all(cache_valid);
  float st[D];
#pragma unroll
  for (int j = 0; j < D; ++j) st[j] = (act[j] - g[j]) * inv_sm1;
  float sA[D], cA[D], p3[D][3];
  action_sincos<Tbl, TABLE>(t, act, sA, cA, trig);
  chain_all<Tbl>(sA, cA, t, p3);
  pick_frames<Tbl>(t, p3, el, e);
  float sF[D], cF[D];
#pragma unroll
  for (int j = 0; j < D; ++j) {
    sF[j] = 0.f;
    cF[j] = 1.f;
  }
  float zo, ze, z0;
  if (use_cache) {
#pragma unroll
    for (int j = 1; j < JN; ++j) {
      sF[j] = cache->s[j];
      cF[j] = cache->c[j];
    }
    z0 = cache->zmin;
  } else {
    prev_pose_sincos<Tbl, TABLE>(t, g, sF, cF, trig, prev_whole);
    chain_z<Tbl>(sF, cF, t, zo, ze);
    z0 = fminf(zo, ze);
  }
  if (CACHED) {
#pragma unroll
    for (int j = 1; j < JN; ++j) {
      cache->s[j] = sA[j];
      cache->c[j] = cA[j];
    }
    cache->zmin = fminf(el[2], e[2]);
  }
  // start state and first z of this sub-lane's half: k = 0 (previous pose) forward, k = S - 1 (the action) backward
  float zmin = backward ? fminf(el[2], e[2]) : z0;
  float sd[D], cd[D];
  sincos_increment<D, JN>(st, sd, cd);
  float sW[D], cW[D];
  sW[0] = 0.f;
  cW[0] = 1.f;
#pragma unroll
  for (int j = 1; j < D; ++j) {
    sW[j] = backward ? sA[j] : sF[j];
    cW[j] = backward ? cA[j] : cF[j];
    sd[j] = backward ? -sd[j] : sd[j];                         // rotate by -delta: the sign rides in the operand
  }
  const int nf = (S - 1) / 2;                                  // forward poses k = 1..nf
  const int nb = S - 2 - nf;                                   // backward poses k = S-2..nf+1 (nb = nf or nf - 1)
  const int mine = backward ? nb : nf;
#pragma unroll 2
  for (int it = 1; it <= nf; ++it) {
    rotate_pose<D, JN, +1>(sW, cW, sd, cd);
    chain_z<Tbl>(sW, cW, t, zo, ze);
    if (it <= mine) zmin = fminf(zmin, fminf(zo, ze));
  }
#pragma unroll
  for (int sh = EPW; sh < 64; sh <<= 1) zmin = fminf(zmin, __shfl_xor(zmin, sh));
  return zmin;
}

// ---------------------------------------------------------------------------
// step, one env spread over L lanes of a wave (L = 2 or 4) -- for batches so small that the chip is mostly idle and a
// launch's time is one wave's dependent chain plus the kernel boundary (profiles/r02_variants.md section 3).
//   lane layout   : lane = q * (64 / L) + e;  e = env within the wave (64 / L envs per wave), q = its sub-lane
//   kinematics    : sub-lanes with even q walk the forward half of the recurrence (from the previous pose), odd q the
//                   backward half (from the action) -- the SAME loop body, the sign of sin(delta) and the start state
//                   differ per lane, so there is no divergence; every pose keeps the arithmetic of route_kinematics
//                   (-(s * -sd) == s * sd exactly), and min over the halves is order-free => bit-identical results
//   targets       : sub-lane q handles targets q, q + L, q + 2L, ...; the cleared alive bits are AND-ed over the
//                   sub-lanes, the running min of z is min-ed (ds_bpermute via __shfl_xor)
//   replicated    : the Philox draw, the endpoint sincos and the full chain at the final pose (every sub-lane needs the
//                   elbow and the end effector for its targets)
//   stores        : each sub-lane its targets' observations; sub-lane 0 the env's state and step outputs
// Only the default trigonometry (TRIG 0); D, the table kinds and both action sources as in step_kernel.
// ---------------------------------------------------------------------------
template <class Tbl, bool SAMPLE, int L, bool TT = false>
__global__ __launch_bounds__(kBlock) void step_split_kernel(const StepArgs a) {
  static_assert(L == 2 || L == 4, "an env is spread over 2 or 4 lanes");
  static_assert(!TT || (SAMPLE && ActionTrigTable<Tbl>::value), "the table serves sampled actions of a static table");
  constexpr int D = Tbl::D;
  constexpr int EPW = 64 / L;                                  // envs per wave
  constexpr int PFS = (kPrefetch + L - 1) / L;                 // targets per sub-lane requested up front
  __shared__ __attribute__((aligned(16))) SinCos trig_lds[TT ? kTrigEntries : 1];   // as in step_kernel<.., TT>
  float4 trig_v;
  if (TT) trig_v = trig_table_load(a.trig_table);
  const Tbl t = TableMaker<Tbl>::make(a.dh);
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  const uint32_t q = lane / EPW;
  const int64_t first = (int64_t)wave * EPW;                   // first env of this wave
  if (first >= a.n) {                                          // whole wave beyond the batch
    if (TT) trig_table_commit(trig_lds, trig_v);               // its arrival at the block's barrier, and its quarter of the table
    return;
  }
  const uint32_t env = (uint32_t)first + lane % EPW;
  const bool live = env < a.n;                                 // tail lanes compute on the last env and store nothing
  const uint32_t i = live ? env : (uint32_t)(a.n - 1);
  const int64_t ld = a.ld;

  float g[D], act[D];
#pragma unroll
  for (int j = 0; j < D; ++j) g[j] = (a.goals + j * ld)[i];
  // this sub-lane's first targets, ahead of the arithmetic (target index p = q + L * m)
  float tx[PFS][3];
#pragma unroll
  for (int m = 0; m < PFS; ++m) {
    const int p = (int)q + L * m;
    if (p < a.K) {
      const float* row = a.points + (int64_t)(3 * p) * ld;
#pragma unroll
      for (int c = 0; c < 3; ++c) tx[m][c] = (row + c * ld)[i];
    }
  }
  const uint32_t am = a.alive[i];
  const float total_in = a.total_reward[i];
  if (SAMPLE) {
    draw_action<D>(((uint64_t)a.seed_hi << 32) | a.seed_lo, (uint64_t)(a.env_base + i), step_index(a), act);
  } else {
    bool bad = false;
#pragma unroll
    for (int j = 0; j < D; ++j) {
      act[j] = (a.actions + j * ld)[i];
      bad |= unusable_angle(act[j]);
    }
    if (bad) {
#pragma unroll
      for (int j = 0; j < D; ++j) act[j] = g[j];
      if (live && q == 0) atomicAdd(a.bad_actions, 1u);
    }
  }

  if (TT) {
    complete_before_here(act);  // Philox under the loads' latency: 32 768 arms 4.76 -> 4.68 us per step, 65 536: 5.84 -> 5.77
    trig_table_commit(trig_lds, trig_v);
  }

  // ---- kinematics: the poses of route_kinematics, one half per sub-lane parity ------------------------------
  float el[3], e[3];
  const float zmin = route_kinematics_split<Tbl, L, false, TT>(t, a.S, a.inv_sm1, g, act, (q & 1u) != 0, el, e, nullptr, false,
                                                                trig_lds, (a.flags & kFlagWholeGoals) != 0);
  const bool ground = zmin < 0.f;  // manytor.py:191
  if (a.zmin && live && q == 0) __builtin_nontemporal_store(zmin, a.zmin + i);  // MT_FLAG_DEBUG_ZMIN

  // ---- this sub-lane's targets ------------------------------------------------------------------------------
  uint32_t nam = am;
  for (int m = 0; (int)q + L * m < a.K; ++m) {
    const int p = (int)q + L * m;
    float* row = a.points + (int64_t)(3 * p) * ld;
    float x, y, z;
    bool have = false;
#pragma unroll
    for (int mm = 0; mm < PFS; ++mm)
      if (m == mm) {
        x = tx[mm][0];
        y = tx[mm][1];
        z = tx[mm][2];
        have = true;
      }
    if (!have) {
      x = row[i];
      y = (row + ld)[i];
      z = (row + 2 * ld)[i];
    }
    const bool al = (am >> p) & 1u;
    float dist = 0.f, r = 0.f, th = 0.f;
    if (al) {
      observe_target(el, x, y, z, dist, r, th);
      if (within_box(e, x, y, z, a.tol)) nam &= ~(1u << p);
    } else if (live && ((x != 0.f) | (y != 0.f) | (z != 0.f))) {  // manytor.py:148
      row[i] = 0.f;
      (row + ld)[i] = 0.f;
      (row + 2 * ld)[i] = 0.f;
    }
    if (live) {
      float* orow = a.obs + (int64_t)(3 * p) * ld;
      __builtin_nontemporal_store(dist, orow + i);
      __builtin_nontemporal_store(r, orow + ld + i);
      __builtin_nontemporal_store(th, orow + 2 * ld + i);
    }
  }
#pragma unroll
  for (int sh = EPW; sh < 64; sh <<= 1) nam &= (uint32_t)__shfl_xor((int)nam, sh);

  const int32_t rew = ground ? -1 : ((nam != am) ? 1 : 0);  // manytor.py:205-212
  bool done = (nam == 0u);                                    // manytor.py:170-171
  if (a.flags & MT_FLAG_TERMINATE_ON_GROUND) done |= ground;
  if (live && q == 0) {
#pragma unroll
    for (int j = 0; j < D; ++j) (a.goals + j * ld)[i] = act[j];
#pragma unroll
    for (int c = 0; c < 3; ++c) __builtin_nontemporal_store(e[c], a.ee + c * ld + i);
    a.alive[i] = nam;
    __builtin_nontemporal_store(rew, a.reward + i);
    a.total_reward[i] = total_in + (float)rew;  // manytor.py:258
    if (a.snap) a.snap[i] = total_in + (float)rew;
    __builtin_nontemporal_store((uint8_t)(done ? 1 : 0), a.done + i);
  }
  // sub-lane 0 occupies lanes 0 .. EPW-1: the low EPW bits of the ballot are this wave's slice of the done_bits word
  const unsigned long long bits = __ballot(done && live && q == 0);
  if (lane == 0) {
    if (L == 2)
      reinterpret_cast<uint32_t*>(a.done_bits)[wave] = (uint32_t)bits;
    else
      reinterpret_cast<uint16_t*>(a.done_bits)[wave] = (uint16_t)bits;
  }
}

// Sub-step trajectory of the whole batch (manytor.py:190 appends joints_coordinates[3] of every
// sub-step to `trajectory`): the end effector at each of the S poses of the route goals -> action, as SoA rows
// [3S][ld] (row 3k + axis).  Launched BEFORE the step kernel of the same call (it needs the previous pose) with the
// same action source, only on handles created with MT_FLAG_TRACE: the timed step kernel is untouched.  Every pose gets
// the full polynomial sincos, like route_trace_kernel.
template <int D, bool SAMPLE>
__global__ __launch_bounds__(kBlock) void trace_kernel(const StepArgs a, float* trace) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  LaneOffset<true> o4{i * 4u};
  if (i >= a.n) return;
  const int64_t ld = a.ld;
  const RtTableF<D> t{a.dh};
  const uint32_t step_no = SAMPLE ? step_index(a) : 0u;  // ahead of the row accesses: see step_kernel
  float g[D], act[D], st[D];
#pragma unroll
  for (int j = 0; j < D; ++j) g[j] = ldr(a.goals + j * ld, o4);
  if (SAMPLE) {
    draw_action<D>(((uint64_t)a.seed_hi << 32) | a.seed_lo, (uint64_t)(a.env_base + i), step_no, act);
  } else {
    bool bad = false;
#pragma unroll
    for (int j = 0; j < D; ++j) {
      act[j] = ldr(a.actions + j * ld, o4);
      bad |= unusable_angle(act[j]);
    }
    if (bad) {
#pragma unroll
      for (int j = 0; j < D; ++j) act[j] = g[j];
    }
  }
#pragma unroll
  for (int j = 0; j < D; ++j) st[j] = (act[j] - g[j]) * a.inv_sm1;
  for (int k = 0; k < a.S; ++k) {
    float s[D], c[D], p[D][3];
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const float pose = (k == a.S - 1) ? act[j] : __builtin_fmaf((float)k, st[j], g[j]);  // np.linspace, manytor.py:182
      sincos_deg(pose + t.off(j), s[j], c[j]);
    }
    chain_all<RtTableF<D>>(s, c, t, p);
    float el[3], e[3];
    pick_frames<RtTableF<D>>(t, p, el, e);
#pragma unroll
    for (int q = 0; q < 3; ++q) str_stream(trace + (int64_t)(3 * k + q) * ld, o4, e[q]);
  }
}

// done_bits word `word` rebuilt from the done bytes: single-env launches (mt_env_step / mt_env_reset) cannot produce
// the 64-env ballot themselves.  One wavefront.
__global__ __launch_bounds__(64) void done_bits_word_kernel(const uint8_t* done, int64_t n, int64_t word,
                                                            unsigned long long* done_bits) {
  const int64_t i = word * 64 + threadIdx.x;
  const unsigned long long bits = __ballot(i < n && done[i] != 0);
  if (threadIdx.x == 0) done_bits[word] = bits;
}

// All done_bits words rebuilt from the done bytes (mt_set(MT_F_DONE): restoring a checkpoint).
__global__ __launch_bounds__(kBlock) void done_bits_rebuild_kernel(const uint8_t* done, int64_t n, unsigned long long* done_bits) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  const unsigned long long bits = __ballot(i < n && done[i] != 0);
  if ((threadIdx.x & 63) == 0 && (int64_t)i < n) done_bits[i >> 6] = bits;
}

// One rejection-sampling candidate of manytor.py:229-239: half HALF of a Philox block (philox.h: two candidates
// per block).  Contraction is off so that every op rounds once, exactly as the numpy restatement
// (oracle/philox_ref.py) does.
template <int HALF>
__device__ __forceinline__ bool target_candidate(const u32x4& w, float radius, float& x, float& y, float& z) {
#pragma clang fp contract(off)
  uint32_t fx, fy, fz;
  candidate_fields<HALF>(w, fx, fy, fz);
  const float r2 = 2.0f * radius;
  const float rr = radius * radius;
  const float tx = r2 * u21(fx);
  const float ty = r2 * u21(fy);
  x = tx - radius;
  y = ty - radius;
  z = radius * u21(fz);
  const float xx = x * x, yy = y * y, zz = z * z;
  const float sxy = xx + yy;
  const float n2 = sxy + zz;
  return n2 <= rr;
}

// Rejection-sample K targets for one env (manytor.py:229-239) and hand each accepted one to `put(k, x, y, z)`.
// Candidates are taken in the order (block 0, half 0), (block 0, half 1), (block 1, half 0), ...
// The draw loop is bounded; the tail branch is unreachable in practice (acceptance pi/6 per candidate).
template <class Put>
__device__ __forceinline__ void draw_targets(uint64_t seed, uint64_t env_id, uint32_t episode, int K, float radius,
                                             Put&& put) {
  int cnt = 0;
  for (uint32_t blk = 0; blk < 2048u && cnt < K; ++blk) {
    const u32x4 w = stream_block(seed, env_id, kTagTarget, episode, blk);
    float x, y, z;
    if (target_candidate<0>(w, radius, x, y, z)) {
      put(cnt, x, y, z);
      ++cnt;
    }
    if (cnt < K && target_candidate<1>(w, radius, x, y, z)) {
      put(cnt, x, y, z);
      ++cnt;
    }
  }
  for (; cnt < K; ++cnt) put(cnt, 0.f, 0.f, 0.5f * radius);
}

// ---------------------------------------------------------------------------
// action_sample for all envs (manytor.py:215-217), device RNG.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void sample_actions_kernel(float* actions, int64_t n, int64_t ld, int D,
                                                                int64_t env_base, uint64_t seed, uint32_t step_idx) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;  // 32-bit lane offset: rows are addressed as uniform base + i
  if (i >= n) return;
  const u32x4 w = stream_block(seed, (uint64_t)(env_base + i), kTagAction, step_idx, 0u);
  const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
  for (int j = 0; j < D; ++j)
    (actions + (int64_t)j * ld)[i] = (j < 4) ? action_from_word(ws[j]) : action_from_word_digit1(ws[j - 4]);
}

// ---------------------------------------------------------------------------
// reset (manytor.py:219-253).  RANDOM: draw the targets here; otherwise the
// caller has already filled `points`.  ONLY_DONE: re-arm finished envs only.
// ---------------------------------------------------------------------------
// The draw of reset_kernel, shared by the 64 envs of a wave.  Rejection sampling makes the lanes of a wave need different
// numbers of Philox blocks (6.7 on average for K = 7, ~13 for the unluckiest of 64), and a wave runs as long as its
// unluckiest lane.  So: while more than half of the lanes still need targets every lane draws for itself, one block per
// trip; once m <= 32 lanes are still needy, the whole wave works for them -- the 64 lanes are cut into m groups of
// h = 64 / m lanes, group g serves the g-th needy lane T and evaluates T's NEXT h blocks at once (lane r of the group:
// block blk_T + r), the accepted candidates are numbered in (block, half) order with two ballots and written straight
// into T's LDS column.  That is exactly the sequential order of draw_targets, so the targets are the same bits; the wave
// finishes in ~9 trips instead of ~13, and a reset_done with few finished envs in one or two.
// `need` = this lane's env wants targets; `col0` = the LDS column of the wave's lane 0 ([3K][kBlock] floats per block).
__device__ __forceinline__ void draw_targets_wave(uint64_t seed, uint64_t env0, uint32_t episode, bool need, int K, float radius,
                                                  float* col0, uint8_t* slots) {
  const uint32_t lane = threadIdx.x & 63u;
  auto put = [&](uint32_t t, int k, float x, float y, float z) {
    float* cell = col0 + t + 3 * k * kBlock;
    cell[0] = x;
    cell[kBlock] = y;
    cell[2 * kBlock] = z;
  };
  int cnt = need ? 0 : K;
  uint32_t blk = 0;
  unsigned long long M = __ballot(cnt < K);
  for (int trip = 0; M != 0ull && trip < 4096; ++trip) {
    const int m = __popcll(M);
    if (m > 32) {  // every needy lane for itself: one block, two candidates (draw_targets' loop body)
      if (cnt < K) {
        const u32x4 w = stream_block(seed, env0 + lane, kTagTarget, episode, blk);
        float x, y, z;
        if (target_candidate<0>(w, radius, x, y, z)) put(lane, cnt++, x, y, z);
        if (cnt < K && target_candidate<1>(w, radius, x, y, z)) put(lane, cnt++, x, y, z);
        ++blk;
      }
    } else {
      const uint32_t h = 64u / (uint32_t)m;                                   // lanes per needy lane: 2 .. 64
      const bool needy = cnt < K;
      const uint32_t rank = (uint32_t)__popcll(M & ((1ull << lane) - 1ull));  // needy lane -> its group
      if (needy) slots[rank] = (uint8_t)lane;
      __builtin_amdgcn_wave_barrier();
      const uint32_t g = lane / h, r = lane % h;
      const bool serving = g < (uint32_t)m;
      const uint32_t T = serving ? slots[g] : 0u;
      __builtin_amdgcn_wave_barrier();                                        // all reads done before the next trip's writes
      const uint32_t blkT = (uint32_t)__shfl((int)blk, (int)T);
      const int cntT = __shfl(cnt, (int)T);
      const uint32_t epT = (uint32_t)__shfl((int)episode, (int)T);
      bool a0 = false, a1 = false;
      float x0 = 0.f, y0 = 0.f, z0 = 0.f, x1 = 0.f, y1 = 0.f, z1 = 0.f;
      if (serving) {
        const u32x4 w = stream_block(seed, env0 + T, kTagTarget, epT, blkT + r);
        a0 = target_candidate<0>(w, radius, x0, y0, z0);
        a1 = target_candidate<1>(w, radius, x1, y1, z1);
      }
      const unsigned long long A0 = __ballot(a0), A1 = __ballot(a1);
      auto group_mask = [&](uint32_t grp) { return (h == 64u ? ~0ull : ((1ull << h) - 1ull)) << (grp * h); };
      if (serving) {  // number my candidates behind those of the lower lanes of my group: (block, half) order
        const unsigned long long lower = group_mask(g) & ((1ull << lane) - 1ull);
        const int k0 = cntT + __popcll(A0 & lower) + __popcll(A1 & lower);
        if (a0 && k0 < K) put(T, k0, x0, y0, z0);
        const int k1 = k0 + (a0 ? 1 : 0);
        if (a1 && k1 < K) put(T, k1, x1, y1, z1);
      }
      if (needy) {  // what my group found for me
        const unsigned long long gm = group_mask(rank);
        cnt += __popcll(A0 & gm) + __popcll(A1 & gm);
        blk += h;
      }
    }
    M = __ballot(cnt < K);
  }
  for (; cnt < K; ++cnt) put(lane, cnt, 0.f, 0.f, 0.5f * radius);  // unreachable in practice, as in draw_targets
  __builtin_amdgcn_wave_barrier();  // a lane's column was written by other lanes of its wave
}

// The draw of an env that is spread over L lanes of a wave (lane = q * (64 / L) + e: reset_split_kernel, rollout_split_kernel):
// sub-lane q evaluates blocks q, q + L, q + 2 L, ...; in every round the accepted candidates are numbered in block order across
// the env's sub-lanes (their counts travel by shuffle), so the K targets handed to `put(k, x, y, z)` -- by whichever sub-lane
// found them -- are exactly the first K accepted candidates of draw_targets' sequential order: the same bits.  Every sub-lane of
// the env must call it (`go` = the env wants targets; equal in all of them, so they leave the loop together).
template <int L, class Put>
__device__ __forceinline__ void draw_targets_split(uint64_t seed, uint64_t env_id, uint32_t episode, bool go, int K, float radius,
                                                   uint32_t q, uint32_t e, Put&& put) {
  constexpr int EPW = 64 / L;
  int cnt = go ? 0 : K;
  for (uint32_t round = 0; round < 2048u / L && cnt < K; ++round) {
    const u32x4 w = stream_block(seed, env_id, kTagTarget, episode, round * L + q);
    float x0, y0, z0, x1, y1, z1;
    const bool a0 = target_candidate<0>(w, radius, x0, y0, z0);
    const bool a1 = target_candidate<1>(w, radius, x1, y1, z1);
    const int mine = (a0 ? 1 : 0) + (a1 ? 1 : 0);
    int before = 0, total = 0;
#pragma unroll
    for (int qq = 0; qq < L; ++qq) {
      const int c = __shfl(mine, qq * EPW + (int)e);
      total += c;
      before += (qq < (int)q) ? c : 0;
    }
    const int k0 = cnt + before;
    if (a0 && k0 < K) put(k0, x0, y0, z0);
    const int k1 = k0 + (a0 ? 1 : 0);
    if (a1 && k1 < K) put(k1, x1, y1, z1);
    cnt += total;
  }
  if (q == 0)
    for (; cnt < K; ++cnt) put(cnt, 0.f, 0.f, 0.5f * radius);  // unreachable in practice, as in draw_targets
}

template <int D, bool RANDOM, bool ONLY_DONE>
__global__ __launch_bounds__(kBlock) void reset_kernel(const StepArgs a, float radius) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  const bool valid = i < a.n;
  if (!RANDOM && !valid) return;  // with RANDOM every lane of a wave takes part in the draw (draw_targets_wave)
  const int64_t ld = a.ld;
  // done byte: 1 = finished, waiting for this call; 2 = finished and already re-armed inside mt_rollout_fused
  const uint8_t dn = !valid ? (uint8_t)0 : (ONLY_DONE ? a.done[i] : (uint8_t)1);
  const bool go = dn == 1;
  if (ONLY_DONE && dn == 2) a.done[i] = 0;
  uint32_t episode = a.major;
  if (go) {
    float s[D], c[D], p[D][3];
#pragma unroll
    for (int j = 0; j < D; ++j) {
      (a.goals + j * ld)[i] = 0.f;
      sincos_deg(a.dh.off_deg[j], s[j], c[j]);
    }
    const RtTableF<D> t{a.dh};
    chain_all<RtTableF<D>>(s, c, t, p);  // joints_coordinates at the zero pose, manytor.py:224-225
    float el0[3], e0[3];
    pick_frames<RtTableF<D>>(t, p, el0, e0);
#pragma unroll
    for (int q = 0; q < 3; ++q) (a.ee + q * ld)[i] = e0[q];
    if (ONLY_DONE)
      record_finished(a, i, a.episodes[i], a.total_reward[i]);
    else
      a.last_return[i] = a.total_reward[i];
    a.total_reward[i] = 0.f;
    a.reward[i] = 0;
    a.done[i] = 0;
    uint32_t all_alive = (a.K >= 32) ? 0xFFFFFFFFu : ((1u << a.K) - 1u);
    if (!RANDOM) {
      // Targets the caller placed (mt_reset; from device memory they cannot be screened on the host): one with a NaN or an
      // infinite coordinate is dropped -- dead from the start, coordinates zeroed, counted with the unusable actions -- instead
      // of reaching arithmetic that is built with -ffinite-math-only.  Tested on the bit pattern.
      for (int k = 0; k < a.K; ++k) {
        float* row = a.points + (int64_t)(3 * k) * ld;
        const uint32_t bx = __float_as_uint(row[i]), by = __float_as_uint((row + ld)[i]), bz = __float_as_uint((row + 2 * ld)[i]);
        if (((bx & 0x7FFFFFFFu) > 0x7F7FFFFFu) | ((by & 0x7FFFFFFFu) > 0x7F7FFFFFu) | ((bz & 0x7FFFFFFFu) > 0x7F7FFFFFu)) {
          row[i] = 0.f;
          (row + ld)[i] = 0.f;
          (row + 2 * ld)[i] = 0.f;
          all_alive &= ~(1u << k);
          atomicAdd(a.bad_actions, 1u);
        }
      }
    }
    a.alive[i] = all_alive;
    // full reset: the caller names the episode; re-arm of a finished env: its own counter advances by one
    if (ONLY_DONE) episode = a.episodes[i] + 1u;
    a.episodes[i] = episode;
  }
  if (RANDOM) {
    // manytor.py:229-239: uniform in the cube, keep z >= 0 and |p| <= radius.  z is drawn from
    // [0, R) directly (same conditional law).  fp32, one rounding per op, mirrored by oracle/philox_ref.py.
    // Envs of one wave accept their k-th target at different candidates, so storing from inside the draw loop
    // would write partial lines of different rows per instruction.  The accepted targets are parked in the env's
    // LDS column ([3K][kBlock] floats, conflict-free) and written out row by row afterwards.
    extern __shared__ float stage[];
    __shared__ uint8_t slots[kBlock / 64][64];
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t seed = ((uint64_t)a.seed_hi << 32) | a.seed_lo;
    draw_targets_wave(seed, (uint64_t)(a.env_base + (i - lane)), episode, go, a.K, radius, stage + (threadIdx.x - lane),
                      slots[threadIdx.x >> 6]);
    if (go) {
      const float* col = stage + threadIdx.x;
      for (int r = 0; r < 3 * a.K; ++r) (a.points + (int64_t)r * ld)[i] = col[r * kBlock];
    }
  }
  if (valid && (threadIdx.x & 63) == 0) a.done_bits[i >> 6] = 0ull;  // every env of the wave has done == 0 now
}

// reset with the draw of an env's targets spread over L lanes of a wave -- for batches so small that the kernel's time
// is one wave per SIMD walking its ~13 Philox blocks one after the other (11-12 us at <= 65 536 arms, of which the
// launch and the stores are about half).  Lane layout as in step_split_kernel (lane = q * (64 / L) + e).  Sub-lane q of an
// env evaluates blocks q, q + L, q + 2 L, ...; in every round the accepted candidates are numbered in block order
// across the env's sub-lanes (their counts travel by shuffle), so the K targets are exactly the first K accepted
// candidates of draw_targets' sequential order: the same bits as reset_kernel.  The targets are parked in the env's
// LDS column ([3K][kBlock / L]) and written out by the sub-lanes in turn; everything else is done by sub-lane 0.
template <int D, bool ONLY_DONE, int L>
__global__ __launch_bounds__(kBlock) void reset_split_kernel(const StepArgs a, float radius) {
  static_assert(L == 2 || L == 4, "an env is spread over 2 or 4 lanes");
  extern __shared__ float stage[];  // [3K][kBlock / L]
  constexpr int EPW = 64 / L;
  constexpr int EPB = kBlock / L;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  const uint32_t q = lane / EPW;
  const uint32_t e = lane % EPW;
  const int64_t first = (int64_t)wave * EPW;
  if (first >= a.n) return;
  const uint32_t env = (uint32_t)first + e;
  const bool live = env < a.n;  // tail lanes shadow the last env (in their own column) and store nothing
  const uint32_t i = live ? env : (uint32_t)(a.n - 1);
  const int64_t ld = a.ld;
  // every sub-lane reads what it needs of the old state first; sub-lane 0 overwrites it after the draw
  const uint8_t dn = ONLY_DONE ? a.done[i] : (uint8_t)1;
  const bool go = dn == 1;
  const uint32_t old_episode = ONLY_DONE ? a.episodes[i] : 0u;
  const float old_total = a.total_reward[i];
  const uint32_t episode = ONLY_DONE ? old_episode + 1u : a.major;
  float* col = stage + (threadIdx.x >> 6) * EPW + e;
  const uint64_t seed = ((uint64_t)a.seed_hi << 32) | a.seed_lo;
  const uint64_t env_id = (uint64_t)(a.env_base + i);
  const int K = a.K;
  auto put = [&](int k, float x, float y, float z) {
    float* cell = col + 3 * k * EPB;
    cell[0] = x;
    cell[EPB] = y;
    cell[2 * EPB] = z;
  };
  draw_targets_split<L>(seed, env_id, episode, go, K, radius, q, e, put);
  __builtin_amdgcn_wave_barrier();  // the columns were written by the env's sub-lanes; a wave only touches its own
  if (live && go)
    for (int r = (int)q; r < 3 * K; r += L) (a.points + (int64_t)r * ld)[i] = col[r * EPB];
  if (live && q == 0) {
    if (ONLY_DONE && dn == 2) a.done[i] = 0;
    if (go) {
      float s[D], c[D], p[D][3];
#pragma unroll
      for (int j = 0; j < D; ++j) {
        (a.goals + j * ld)[i] = 0.f;
        sincos_deg(a.dh.off_deg[j], s[j], c[j]);
      }
      const RtTableF<D> t{a.dh};
      chain_all<RtTableF<D>>(s, c, t, p);
      float el0[3], e0[3];
      pick_frames<RtTableF<D>>(t, p, el0, e0);
#pragma unroll
      for (int qq = 0; qq < 3; ++qq) (a.ee + qq * ld)[i] = e0[qq];
      if (ONLY_DONE)
        record_finished(a, i, old_episode, old_total);
      else
        a.last_return[i] = old_total;
      a.total_reward[i] = 0.f;
      a.reward[i] = 0;
      a.done[i] = 0;
      a.alive[i] = (K >= 32) ? 0xFFFFFFFFu : ((1u << K) - 1u);
      a.episodes[i] = episode;
    }
  }
  if (lane == 0) a.done_bits[first >> 6] = 0ull;  // every env of the word has done == 0 after this launch
}

// ---------------------------------------------------------------------------
// rollout: T consecutive Environment.step()s with in-kernel random actions in ONE launch (the inner loop of
// test_multi.py:19-21), optionally re-arming an env the moment it finishes.
// Joint angles, alive mask and return stay in registers between steps and the env's targets stay in LDS
// ([3K][256] floats per block, one column per thread, conflict-free), so per step only the outputs are
// written: obs 12K + reward 4 + done 1 + ee 12 bytes per env instead of the 233 a single-step launch moves.
// Per step it runs exactly the arithmetic of step_kernel (shared device functions), so
//   rollout(T)  ==  T x step_random          (auto_reset = 0)
//   rollout(T)  ==  T x (step_random; reset_done)   (auto_reset = 1, state fields)
// bit for bit; tests/test_gpu_parity.py checks both.
// ---------------------------------------------------------------------------
struct RolloutArgs {
  int32_t T;
  uint32_t step0;
  uint32_t auto_reset;
  float radius;
  // Episode boundary folded into the launch (mt_rollout's multi-step form, engine.hip):
  //   reset_first : the launch BEGINS with the full random reset of every env that mt_reset_random(reset seed, reset_episode)
  //                 deferred to it -- last_return <- return, zero pose, all targets alive, episode index, targets drawn from
  //                 the (seed, env, episode) stream in draw_targets' sequential order: reset_kernel<.., RANDOM, false>'s
  //                 state, bit for bit -- instead of loading pose / alive mask / return / targets: no reset launch, no
  //                 re-fetch of what it would have written
  //   snap        : the launch ENDS by storing every env's return to this row as well (the snapshot the overlapped return
  //                 gather would otherwise take with a launch of its own behind it), or NULL
  uint32_t reset_first, reset_episode, reset_seed_lo, reset_seed_hi;
  float* snap;
};

//   RPF : prologue form.  0: the state and all targets are loaded (targets straight into LDS) before the first step starts
//         -- a launch's first step then waits for the table load, the barrier and every state load in turn before its
//         Philox block even begins (~6 us before the first step's outputs at 131 072 envs, tools/rollout_k_sweep.py), which
//         nothing amortises when mt_rollout runs only a few steps per launch.  RPF = kPrefetch: everything is REQUESTED at
//         the top (table, pose, alive mask, return, the first RPF targets into registers), the first step's Philox block and
//         kinematics run under those loads, and the targets go to LDS between the first step's kinematics and its target
//         loop (step 0 is peeled).  Same device functions on the same inputs: the same bits.  The host takes RPF for the
//         short launches of mt_rollout (engine.hip), the long fused launches keep RPF = 0 (24 fewer live registers).
template <class Tbl, int RPF = 0>
__global__ __launch_bounds__(kBlock) void rollout_kernel(const StepArgs a, const RolloutArgs r) {
  extern __shared__ __attribute__((aligned(16))) float tile[];  // [3K][kBlock] (+ the action sin / cos table for the compile-time tables)
  constexpr int D = Tbl::D;
  constexpr bool kTable = ActionTrigTable<Tbl>::value;
  const Tbl t = TableMaker<Tbl>::make(a.dh);
  const SinCos* trig = nullptr;
  float4 trig_v;
  if constexpr (kTable) {
    trig = reinterpret_cast<const SinCos*>(tile + 3 * a.K * kBlock);
    trig_v = trig_table_load(a.trig_table);
    if (!RPF) trig_table_commit(reinterpret_cast<SinCos*>(tile + 3 * a.K * kBlock), trig_v);
  }
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  LaneOffset<true> o4{i * 4u}, o1{i};
  // (threads past the end: with a table and RPF they stay until the barrier -- their loads stay inside the rows, which are
  // ld >= round_up(n, 256) long, and the launch covers the batch or a 256-aligned range of it -- and leave right behind it)
  // (... and with a reset folded into the launch every lane of a wave takes part in the wave-cooperative draw below)
  const bool stay = (kTable && RPF) || r.reset_first != 0;
  if (!stay && i >= a.n) return;
  const int64_t ld = a.ld;
  const uint64_t seed = ((uint64_t)a.seed_hi << 32) | a.seed_lo;
  const uint64_t env_id = (uint64_t)(a.env_base + i);
  float* col = tile + threadIdx.x;
  __shared__ uint8_t draw_slots[kBlock / 64][64];  // draw_targets_wave's scratch (used by launches with reset_first only)

  const uint32_t all_alive = (a.K >= 32) ? 0xFFFFFFFFu : ((1u << a.K) - 1u);
  const bool fresh = r.reset_first != 0;  // (wave-uniform: a launch argument)
  float g[D];
#pragma unroll
  for (int j = 0; j < D; ++j) g[j] = fresh ? 0.f : ldr(a.goals + j * ld, o4);
  uint32_t am = fresh ? all_alive : ldr(a.alive, o4);
  float total = ldr(a.total_reward, o4);
  uint32_t episode = fresh ? r.reset_episode : (r.auto_reset ? ldr(a.episodes, o4) : 0u);
  bool ended = fresh, dirty = fresh;  // a fresh episode writes its episode index and its targets back at the end
  float tx[RPF ? RPF : 1][3];
  if (fresh) {
    const bool mine = i < a.n;  // (threads past the end help with the draw and stay for the barrier, nothing else)
    if (mine) str(a.last_return, o4, total);  // what reset_kernel<.., RANDOM, false> keeps of the episode that ends here
    total = 0.f;
    const uint32_t lane = threadIdx.x & 63u;
    draw_targets_wave(((uint64_t)r.reset_seed_hi << 32) | r.reset_seed_lo, (uint64_t)(a.env_base + (i - lane)), episode, mine, a.K, r.radius,
                      tile + (threadIdx.x - lane), draw_slots[threadIdx.x >> 6]);
    if (!(kTable && RPF) && !mine) return;
  } else if (RPF) {
#pragma unroll
    for (int k = 0; k < RPF; ++k)
      if (k < a.K) {
#pragma unroll
        for (int q = 0; q < 3; ++q) tx[k][q] = ldr(a.points + (int64_t)(3 * k + q) * ld, o4);
      }
  } else {
    for (int k = 0; k < 3 * a.K; ++k) col[k * kBlock] = ldr(a.points + (int64_t)k * ld, o4);
  }
  PoseCache<D> pose;        // sines / cosines and frame heights of the pose the next step starts from
  bool pose_valid = false;  // nothing known about the pose loaded from memory

  // one step behind its kinematics: observation / pickup per target, outputs, re-arm (the body of the loop below)
  auto finish_step = [&](const float (&act)[D], const float (&el)[3], const float (&e)[3], float zmin) {
    const bool ground = zmin < 0.f;
    if (a.zmin) str_stream(a.zmin, o4, zmin);  // MT_FLAG_DEBUG_ZMIN (a step output like reward: the last step's stays)

    uint32_t nam = am;
    for (int k = 0; k < a.K; ++k) {
      float* pk = col + 3 * k * kBlock;
      const float x = pk[0], y = pk[kBlock], z = pk[2 * kBlock];
      const bool al = (am >> k) & 1u;
      float dist = 0.f, rr = 0.f, th = 0.f;
      if (al) {
        observe_target(el, x, y, z, dist, rr, th);
        if (within_box(e, x, y, z, a.tol)) nam &= ~(1u << k);
      } else if ((x != 0.f) | (y != 0.f) | (z != 0.f)) {  // manytor.py:148
        pk[0] = 0.f;
        pk[kBlock] = 0.f;
        pk[2 * kBlock] = 0.f;
        dirty = true;
      }
      float* orow = a.obs + (int64_t)(3 * k) * ld;
      str_stream(orow, o4, dist);
      str_stream(orow + ld, o4, rr);
      str_stream(orow + 2 * ld, o4, th);
    }
    const int32_t rew = ground ? -1 : ((nam != am) ? 1 : 0);
    bool done = (nam == 0u);
    if (a.flags & MT_FLAG_TERMINATE_ON_GROUND) done |= ground;
    total += (float)rew;
    am = nam;
#pragma unroll
    for (int j = 0; j < D; ++j) g[j] = act[j];
#pragma unroll
    for (int q = 0; q < 3; ++q) str_stream(a.ee + q * ld, o4, e[q]);
    str_stream(a.reward, o4, rew);
    // 2 = "finished in this step and already re-armed below": mt_reset_done must not re-arm it a second time
    str_stream(a.done, o1, (uint8_t)(done ? (r.auto_reset ? 2 : 1) : 0));
    const unsigned long long bits = __ballot(done);
    if ((threadIdx.x & 63) == 0) a.done_bits[i >> 6] = bits;

    if (done && r.auto_reset) {  // re-arm: what reset_kernel<.., RANDOM, ONLY_DONE> does in a separate launch
      record_finished(a, i, episode, total);
      ended = true;
      total = 0.f;
      am = all_alive;
      episode += 1u;
#pragma unroll
      for (int j = 0; j < D; ++j) g[j] = 0.f;
      pose_valid = false;  // the cache describes the pose the finished episode ended in, not the zero pose
      draw_targets(seed, env_id, episode, a.K, r.radius, [&](int k, float x, float y, float z) {
        float* pk = col + 3 * k * kBlock;
        pk[0] = x;
        pk[kBlock] = y;
        pk[2 * kBlock] = z;
      });
      dirty = true;
    }
  };

  int s = 0;
  if (RPF) {  // step 0, peeled: its Philox block and kinematics run under the loads requested above
    float act[D], el[3], e[3];
    draw_action<D>(seed, env_id, r.step0, act);
    if constexpr (kTable) {
      complete_before_here(act);
      trig_table_commit(reinterpret_cast<SinCos*>(tile + 3 * a.K * kBlock), trig_v);
      if (i >= a.n) return;
    }
    const float zmin = route_kinematics<Tbl, 0, true, kTable>(t, a.S, a.inv_sm1, g, act, el, e, &pose, false, trig,
                                                              (a.flags & kFlagWholeGoals) != 0);
    pose_valid = true;
    if (!fresh) {  // (a fresh episode's targets were drawn straight into LDS)
#pragma unroll
      for (int k = 0; k < RPF; ++k)
        if (k < a.K) {
#pragma unroll
          for (int q = 0; q < 3; ++q) col[(3 * k + q) * kBlock] = tx[k][q];
        }
      for (int k = 3 * RPF; k < 3 * a.K; ++k) col[k * kBlock] = ldr(a.points + (int64_t)k * ld, o4);
    }
    finish_step(act, el, e, zmin);
    s = 1;
  }
  for (; s < r.T; ++s) {
    float act[D], el[3], e[3];
    draw_action<D>(seed, env_id, r.step0 + (uint32_t)s, act);
    const float zmin = route_kinematics<Tbl, 0, true, kTable>(t, a.S, a.inv_sm1, g, act, el, e, &pose, pose_valid, trig,
                                                              (a.flags & kFlagWholeGoals) != 0 || s > 0);
    pose_valid = true;
    finish_step(act, el, e, zmin);
  }

#pragma unroll
  for (int j = 0; j < D; ++j) str(a.goals + j * ld, o4, g[j]);
  str(a.alive, o4, am);
  str(a.total_reward, o4, total);
  if (r.snap) str(r.snap, o4, total);
  if (ended) str(a.episodes, o4, episode);
  if (dirty)
    for (int k = 0; k < 3 * a.K; ++k) str(a.points + (int64_t)k * ld, o4, col[k * kBlock]);
}

// rollout with one env spread over L lanes (L = 2 or 4): rollout_kernel for batches so small that a step's time is one
// wave's dependent chain.  Lane layout, kinematics halves and target partition as in step_split_kernel; joint angles,
// alive mask, return and episode counter are replicated in the env's sub-lanes (they all see the same combined z-min and
// alive mask, so they stay equal); the targets live in LDS, one column per ENV ([3K][kBlock / L]), written by whichever
// sub-lane owns the target.  A wave only ever touches its own columns: wave_barrier() orders the LDS traffic, no
// block barrier.  Bit-identical to rollout_kernel.
template <class Tbl, int L, int RPF = 0>
__global__ __launch_bounds__(kBlock) void rollout_split_kernel(const StepArgs a, const RolloutArgs r) {
  static_assert(L == 2 || L == 4, "an env is spread over 2 or 4 lanes");
  extern __shared__ __attribute__((aligned(16))) float tile[];  // [3K][kBlock / L] (+ the action sin / cos table for the compile-time tables)
  constexpr int D = Tbl::D;
  constexpr int EPW = 64 / L;
  constexpr int EPB = kBlock / L;  // envs per block = columns of the tile
  constexpr int PFS = (RPF + L - 1) / L;  // targets per sub-lane requested up front (RPF form, see rollout_kernel)
  constexpr bool kTable = ActionTrigTable<Tbl>::value;
  const Tbl t = TableMaker<Tbl>::make(a.dh);
  const SinCos* trig = nullptr;
  float4 trig_v;
  if constexpr (kTable) {
    trig = reinterpret_cast<const SinCos*>(tile + 3 * a.K * EPB);
    trig_v = trig_table_load(a.trig_table);
    if (!RPF) trig_table_commit(reinterpret_cast<SinCos*>(tile + 3 * a.K * EPB), trig_v);
  }
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  const uint32_t q = lane / EPW;
  const int64_t first = (int64_t)wave * EPW;
  if (first >= a.n) {  // whole wave beyond the batch
    if constexpr (kTable) {
      if (RPF) trig_table_commit(reinterpret_cast<SinCos*>(tile + 3 * a.K * EPB), trig_v);  // its arrival at the block's barrier
    }
    return;
  }
  const uint32_t env = (uint32_t)first + lane % EPW;
  const bool live = env < a.n;  // tail lanes work on a copy of the last env (in their own column) and store nothing
  const uint32_t i = live ? env : (uint32_t)(a.n - 1);
  const int64_t ld = a.ld;
  const uint64_t seed = ((uint64_t)a.seed_hi << 32) | a.seed_lo;
  const uint64_t env_id = (uint64_t)(a.env_base + i);
  float* col = tile + (threadIdx.x >> 6) * EPW + lane % EPW;
  const bool backward = (q & 1u) != 0;

  const uint32_t all_alive = (a.K >= 32) ? 0xFFFFFFFFu : ((1u << a.K) - 1u);
  const bool fresh = r.reset_first != 0;  // (wave-uniform: a launch argument; see RolloutArgs)
  float g[D];
#pragma unroll
  for (int j = 0; j < D; ++j) g[j] = fresh ? 0.f : (a.goals + j * ld)[i];
  uint32_t am = fresh ? all_alive : a.alive[i];
  float total = a.total_reward[i];
  uint32_t episode = fresh ? r.reset_episode : (r.auto_reset ? a.episodes[i] : 0u);
  bool ended = fresh, dirty = fresh;
  float tx[PFS ? PFS : 1][3];
  if (fresh) {  // the env's sub-lanes share the draw (reset_split_kernel's), whoever finds a target parks it in the env's column
    if (live && q == 0) a.last_return[i] = total;
    total = 0.f;
    draw_targets_split<L>(((uint64_t)r.reset_seed_hi << 32) | r.reset_seed_lo, env_id, episode, true, a.K, r.radius, q, lane % EPW,
                          [&](int k, float x, float y, float z) {
                            float* pk = col + 3 * k * EPB;
                            pk[0] = x;
                            pk[EPB] = y;
                            pk[2 * EPB] = z;
                          });
    __builtin_amdgcn_wave_barrier();  // a target's cells may have been written by another sub-lane of the env (same wave)
  } else if (RPF) {  // this sub-lane's first targets into registers (target index p = q + L * m)
#pragma unroll
    for (int m = 0; m < PFS; ++m) {
      const int p = (int)q + L * m;
      if (p < a.K) {
        const float* row = a.points + (int64_t)(3 * p) * ld;
#pragma unroll
        for (int c = 0; c < 3; ++c) tx[m][c] = (row + c * ld)[i];
      }
    }
  } else {
    for (int p = (int)q; p < a.K; p += L) {  // this sub-lane's targets into the env's column
      const float* row = a.points + (int64_t)(3 * p) * ld;
#pragma unroll
      for (int c = 0; c < 3; ++c) col[(3 * p + c) * EPB] = (row + c * ld)[i];
    }
  }
  PoseCache<D> pose;
  bool pose_valid = false;

  auto finish_step = [&](const float (&act)[D], const float (&el)[3], const float (&e)[3], float zmin) {
    const bool ground = zmin < 0.f;
    if (a.zmin && live && q == 0) __builtin_nontemporal_store(zmin, a.zmin + i);  // MT_FLAG_DEBUG_ZMIN

    uint32_t nam = am;
    for (int p = (int)q; p < a.K; p += L) {
      float* pk = col + 3 * p * EPB;
      const float x = pk[0], y = pk[EPB], z = pk[2 * EPB];
      const bool al = (am >> p) & 1u;
      float dist = 0.f, rr = 0.f, th = 0.f;
      if (al) {
        observe_target(el, x, y, z, dist, rr, th);
        if (within_box(e, x, y, z, a.tol)) nam &= ~(1u << p);
      } else if ((x != 0.f) | (y != 0.f) | (z != 0.f)) {  // manytor.py:148
        pk[0] = 0.f;
        pk[EPB] = 0.f;
        pk[2 * EPB] = 0.f;
        dirty = true;
      }
      if (live) {
        float* orow = a.obs + (int64_t)(3 * p) * ld;
        __builtin_nontemporal_store(dist, orow + i);
        __builtin_nontemporal_store(rr, orow + ld + i);
        __builtin_nontemporal_store(th, orow + 2 * ld + i);
      }
    }
#pragma unroll
    for (int sh = EPW; sh < 64; sh <<= 1) nam &= (uint32_t)__shfl_xor((int)nam, sh);
    const int32_t rew = ground ? -1 : ((nam != am) ? 1 : 0);
    bool done = (nam == 0u);
    if (a.flags & MT_FLAG_TERMINATE_ON_GROUND) done |= ground;
    total += (float)rew;
    am = nam;
#pragma unroll
    for (int j = 0; j < D; ++j) g[j] = act[j];
    if (live && q == 0) {
#pragma unroll
      for (int c = 0; c < 3; ++c) __builtin_nontemporal_store(e[c], a.ee + c * ld + i);
      __builtin_nontemporal_store(rew, a.reward + i);
      __builtin_nontemporal_store((uint8_t)(done ? (r.auto_reset ? 2 : 1) : 0), a.done + i);
    }
    const unsigned long long bits = __ballot(done && live && q == 0);
    if (lane == 0) {
      if (L == 2)
        reinterpret_cast<uint32_t*>(a.done_bits)[wave] = (uint32_t)bits;
      else
        reinterpret_cast<uint16_t*>(a.done_bits)[wave] = (uint16_t)bits;
    }

    if (done && r.auto_reset) {  // re-arm, as in rollout_kernel; every sub-lane draws, each keeps its own targets
      if (live && q == 0) record_finished(a, i, episode, total);
      ended = true;
      total = 0.f;
      am = all_alive;
      episode += 1u;
#pragma unroll
      for (int j = 0; j < D; ++j) g[j] = 0.f;
      pose_valid = false;
      draw_targets(seed, env_id, episode, a.K, r.radius, [&](int k, float x, float y, float z) {
        if ((uint32_t)k % L == q) {
          float* pk = col + 3 * k * EPB;
          pk[0] = x;
          pk[EPB] = y;
          pk[2 * EPB] = z;
        }
      });
      dirty = true;
    }
  };

  int s = 0;
  if (RPF) {  // step 0, peeled: its Philox block and kinematics run under the loads requested above
    float act[D], el[3], e[3];
    draw_action<D>(seed, env_id, r.step0, act);
    if constexpr (kTable) {
      complete_before_here(act);
      trig_table_commit(reinterpret_cast<SinCos*>(tile + 3 * a.K * EPB), trig_v);
    }
    const float zmin = route_kinematics_split<Tbl, L, true, kTable>(t, a.S, a.inv_sm1, g, act, backward, el, e, &pose, false, trig,
                                                                    (a.flags & kFlagWholeGoals) != 0);
    pose_valid = true;
    if (!fresh) {  // (a fresh episode's targets were drawn straight into LDS)
#pragma unroll
      for (int m = 0; m < PFS; ++m) {
        const int p = (int)q + L * m;
        if (p < a.K) {
#pragma unroll
          for (int c = 0; c < 3; ++c) col[(3 * p + c) * EPB] = tx[m][c];
        }
      }
      for (int p = (int)q + L * PFS; p < a.K; p += L) {
        const float* row = a.points + (int64_t)(3 * p) * ld;
#pragma unroll
        for (int c = 0; c < 3; ++c) col[(3 * p + c) * EPB] = (row + c * ld)[i];
      }
    }
    finish_step(act, el, e, zmin);
    s = 1;
  }
  for (; s < r.T; ++s) {
    float act[D], el[3], e[3];
    draw_action<D>(seed, env_id, r.step0 + (uint32_t)s, act);
    const float zmin = route_kinematics_split<Tbl, L, true, kTable>(t, a.S, a.inv_sm1, g, act, backward, el, e, &pose, pose_valid,
                                                                    trig, (a.flags & kFlagWholeGoals) != 0 || s > 0);
    pose_valid = true;
    finish_step(act, el, e, zmin);
  }

  if (live && q == 0) {
#pragma unroll
    for (int j = 0; j < D; ++j) (a.goals + j * ld)[i] = g[j];
    a.alive[i] = am;
    a.total_reward[i] = total;
    if (r.snap) r.snap[i] = total;
    if (ended) a.episodes[i] = episode;
  }
  if (live && dirty)
    for (int p = (int)q; p < a.K; p += L) {
      float* row = a.points + (int64_t)(3 * p) * ld;
#pragma unroll
      for (int c = 0; c < 3; ++c) (row + c * ld)[i] = col[(3 * p + c) * EPB];
    }
}

// ---------------------------------------------------------------------------
// get_observations() at the current pose (manytor.py:141-153).
// ---------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(kBlock) void observe_kernel(const StepArgs a) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;  // 32-bit lane offset: rows are addressed as uniform base + i
  if (i >= a.n) return;
  const int64_t ld = a.ld;
  float s[D], c[D], p[D][3];
#pragma unroll
  for (int j = 0; j < D; ++j) sincos_deg((a.goals + j * ld)[i] + a.dh.off_deg[j], s[j], c[j]);
  const RtTableF<D> t{a.dh};
  chain_all<RtTableF<D>>(s, c, t, p);
  float el[3], e_unused[3];
  pick_frames<RtTableF<D>>(t, p, el, e_unused);
  const uint32_t am = a.alive[i];
  for (int k = 0; k < a.K; ++k) {
    float* row = a.points + (int64_t)(3 * k) * ld;
    const float x = row[i], y = (row + ld)[i], z = (row + 2 * ld)[i];
    float dist = 0.f, r = 0.f, th = 0.f;
    if ((am >> k) & 1u) {
      observe_target(el, x, y, z, dist, r, th);
    } else if ((x != 0.f) | (y != 0.f) | (z != 0.f)) {
      row[i] = 0.f;
      (row + ld)[i] = 0.f;
      (row + 2 * ld)[i] = 0.f;
    }
    float* orow = a.obs + (int64_t)(3 * k) * ld;
    orow[i] = dist;
    (orow + ld)[i] = r;
    (orow + 2 * ld)[i] = th;
  }
}

// ---------------------------------------------------------------------------
// is_done() at the current pose (manytor.py:155-173).
// ---------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(kBlock) void check_done_kernel(const StepArgs a) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;  // 32-bit lane offset: rows are addressed as uniform base + i
  if (i >= a.n) return;
  const int64_t ld = a.ld;
  float s[D], c[D], p[D][3];
#pragma unroll
  for (int j = 0; j < D; ++j) sincos_deg((a.goals + j * ld)[i] + a.dh.off_deg[j], s[j], c[j]);
  const RtTableF<D> t{a.dh};
  chain_all<RtTableF<D>>(s, c, t, p);
  float el_unused[3], e[3];
  pick_frames<RtTableF<D>>(t, p, el_unused, e);
  uint32_t am = a.alive[i];
  for (int k = 0; k < a.K; ++k) {
    if (!((am >> k) & 1u)) continue;
    const float* row = a.points + (int64_t)(3 * k) * ld;
    if (within_box(e, row[i], (row + ld)[i], (row + 2 * ld)[i], a.tol)) am &= ~(1u << k);
  }
  a.alive[i] = am;
  const bool done = (am == 0u);
  a.done[i] = done ? 1 : 0;
  const unsigned long long bits = __ballot(done);
  if ((threadIdx.x & 63) == 0) a.done_bits[i >> 6] = bits;
}

// joints_coordinates for all envs, env-major (N, D, 3): row 0 zeros, row j the
// frame after j+1 joints (manytor.py:188-189).
template <int D>
__global__ __launch_bounds__(kBlock) void joints_kernel(const StepArgs a, float* out) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;  // 32-bit lane offset: rows are addressed as uniform base + i
  if (i >= a.n) return;
  const int64_t ld = a.ld;
  float s[D], c[D], p[D][3];
#pragma unroll
  for (int j = 0; j < D; ++j) sincos_deg((a.goals + j * ld)[i] + a.dh.off_deg[j], s[j], c[j]);
  chain_all<RtTable<D>>(s, c, RtTable<D>{a.dh}, p);
  float* o = out + (int64_t)i * (3 * D);
#pragma unroll
  for (int j = 0; j < D; ++j)
#pragma unroll
    for (int q = 0; q < 3; ++q) o[3 * j + q] = (j == 0) ? 0.f : p[j][q];
}

// ---------------------------------------------------------------------------
// Layout conversion between the resident SoA rows and the reference's
// env-major arrays.  Off the hot path (host getters / setters only).
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void soa_to_env_major(const T* src, int64_t ld, int rows, int64_t n, T* dst) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;  // 32-bit lane offset: rows are addressed as uniform base + i
  if (i >= n) return;
  for (int r = 0; r < rows; ++r) dst[(int64_t)i * rows + r] = (src + (int64_t)r * ld)[i];
}

template <typename S>
__global__ __launch_bounds__(kBlock) void env_major_to_soa(const S* src, int rows, int64_t n, float* dst, int64_t ld) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;  // 32-bit lane offset: rows are addressed as uniform base + i
  if (i >= n) return;
  for (int r = 0; r < rows; ++r) (dst + (int64_t)r * ld)[i] = (float)src[(int64_t)i * rows + r];
}

template <typename S>
__global__ __launch_bounds__(kBlock) void soa_to_soa_f32(const S* src, int rows, int64_t n, int64_t ld, float* dst) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;  // 32-bit lane offset: rows are addressed as uniform base + i
  if (i >= n) return;
  for (int r = 0; r < rows; ++r) (dst + (int64_t)r * ld)[i] = (float)(src + (int64_t)r * ld)[i];
}

__global__ __launch_bounds__(kBlock) void alive_unpack(const uint32_t* mask, int K, int64_t n, uint8_t* dst) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;  // 32-bit lane offset: rows are addressed as uniform base + i
  if (i >= n) return;
  const uint32_t m = mask[i];
  for (int k = 0; k < K; ++k) dst[(int64_t)i * K + k] = (m >> k) & 1u;
}

__global__ __launch_bounds__(kBlock) void alive_pack(const uint8_t* src, int K, int64_t n, uint32_t* mask) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;  // 32-bit lane offset: rows are addressed as uniform base + i
  if (i >= n) return;
  uint32_t m = 0;
  for (int k = 0; k < K; ++k) m |= (src[(int64_t)i * K + k] ? 1u : 0u) << k;
  mask[i] = m;
}

// ---------------------------------------------------------------------------
// Stateless helpers: fk()/dh() (manytor.py:25-53) and r_theta() (:17-22).
// ---------------------------------------------------------------------------
struct FkArgs {
  const float* angles;  // (n, dof)
  float* out;           // (n, 16) row-major 4x4
  int64_t n;
  int dof, mode, radians;
  DhConst dh;
};

__global__ __launch_bounds__(kBlock) void fk_kernel(const FkArgs a) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;  // 32-bit lane offset: rows are addressed as uniform base + i
  if (i >= a.n) return;
  float X[3] = {1.f, 0.f, 0.f}, Y[3] = {0.f, 1.f, 0.f}, Z[3] = {0.f, 0.f, 1.f}, o[3] = {0.f, 0.f, 0.f};
  for (int j = 0; j < a.mode; ++j) {
    float ang = a.angles[(int64_t)i * a.dof + j];
    if (a.radians) ang *= 57.29577951308232f;
    float s, c;
    sincos_deg(ang + a.dh.off_deg[j], s, c);
    for (int q = 0; q < 3; ++q) {
      const float nx = __builtin_fmaf(X[q], c, Y[q] * s);
      const float tt = __builtin_fmaf(Y[q], c, -(X[q] * s));
      o[q] = __builtin_fmaf(a.dh.a[j], nx, __builtin_fmaf(a.dh.d[j], Z[q], o[q]));
      X[q] = nx;
      Y[q] = __builtin_fmaf(tt, a.dh.ca[j], Z[q] * a.dh.sa[j]);
      Z[q] = __builtin_fmaf(Z[q], a.dh.ca[j], -(tt * a.dh.sa[j]));
    }
  }
  float* m = a.out + (int64_t)i * 16;
  for (int q = 0; q < 3; ++q) {
    m[4 * q + 0] = X[q];
    m[4 * q + 1] = Y[q];
    m[4 * q + 2] = Z[q];
    m[4 * q + 3] = o[q];
  }
  m[12] = 0.f;
  m[13] = 0.f;
  m[14] = 0.f;
  m[15] = 1.f;
}

// Sub-step trajectory export: joints_coordinates at every one of the S poses of the
// route prev -> action (manytor.py:182-190), for a handful of envs the host wants to draw or log.  One thread per
// (env, sub-step); every pose gets the full polynomial sincos (this is not the timed path).
struct TraceArgs {
  const float* prev;    // (n, dof) degrees
  const float* action;  // (n, dof)
  float* out;           // (n, S, dof, 3)
  int64_t n;
  int dof, S;
  DhConst dh;
};

__global__ __launch_bounds__(kBlock) void route_trace_kernel(const TraceArgs a) {
  const int64_t gid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (gid >= a.n * a.S) return;
  const int64_t env = gid / a.S;
  const int k = (int)(gid % a.S);
  const float inv = 1.0f / (float)(a.S - 1);
  float X[3] = {1.f, 0.f, 0.f}, Y[3] = {0.f, 1.f, 0.f}, Z[3] = {0.f, 0.f, 1.f}, o[3] = {0.f, 0.f, 0.f};
  float* out = a.out + gid * (int64_t)(3 * a.dof);
  for (int j = 0; j < a.dof; ++j) {
    const float g = a.prev[env * a.dof + j], act = a.action[env * a.dof + j];
    // np.linspace: start + k * step, last element = stop exactly (manytor.py:182)
    const float pose = (k == a.S - 1) ? act : __builtin_fmaf((float)k, (act - g) * inv, g);
    float s, c;
    sincos_deg(pose + a.dh.off_deg[j], s, c);
    for (int q = 0; q < 3; ++q) {
      const float nx = __builtin_fmaf(X[q], c, Y[q] * s);
      const float tt = __builtin_fmaf(Y[q], c, -(X[q] * s));
      o[q] = __builtin_fmaf(a.dh.a[j], nx, __builtin_fmaf(a.dh.d[j], Z[q], o[q]));
      X[q] = nx;
      Y[q] = __builtin_fmaf(tt, a.dh.ca[j], Z[q] * a.dh.sa[j]);
      Z[q] = __builtin_fmaf(Z[q], a.dh.ca[j], -(tt * a.dh.sa[j]));
      out[3 * j + q] = (j == 0) ? 0.f : o[q];   // row 0 is zeros, row j the frame after j+1 joints (manytor.py:189)
    }
  }
}

__global__ __launch_bounds__(kBlock) void r_theta_kernel(const float* v1, const float* v2, int64_t n, float* out) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;  // 32-bit lane offset: rows are addressed as uniform base + i
  if (i >= n) return;
  const int64_t b3 = 3 * (int64_t)i;
  const float d0 = fabsf(v1[b3] - v2[b3]), d1 = fabsf(v1[b3 + 1] - v2[b3 + 1]), d2 = fabsf(v1[b3 + 2] - v2[b3 + 2]);
  const float h = __builtin_amdgcn_sqrtf(__builtin_fmaf(d0, d0, d1 * d1));
  out[2 * (int64_t)i] = atan2_deg_q1(d0, d1);
  out[2 * (int64_t)i + 1] = atan2_deg_q1(h, d2);
}

// ---------------------------------------------------------------------------
// Streaming yardstick (mt_stream_probe): the step kernel's memory operations with none of its arithmetic -- per env the
// D + 3K + 2 state rows are read (goals, targets, alive mask, return), D + 2 of them are written back in place (plain
// stores: goals, alive mask, return), and 3K + 4 output rows (obs, reward, end effector) plus one byte row (done) are
// written with non-temporal stores: exactly the 8 D + 24 K + 33 bytes per env mt_step_random moves, with the same row
// addressing (SGPR row base + 32-bit lane offset) and one env per lane.  What this kernel reaches on a box is what the
// step kernel's access shape can reach there; bench.py quotes the step against it (roofline.achievable_gbs).
// ---------------------------------------------------------------------------
template <int D, int K>
__global__ __launch_bounds__(kBlock) void stream_probe_kernel(float* state, float* out, uint8_t* bytes, int64_t n, int64_t ld) {
  constexpr int RD = D + 3 * K + 2, WP = D + 2, WN = 3 * K + 4;
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  LaneOffset<true> o4{i * 4u}, o1{i};
  if (i >= n) return;
  float v[RD], acc = 0.f;
#pragma unroll
  for (int r = 0; r < RD; ++r) {
    v[r] = ldr(state + (int64_t)r * ld, o4);
    acc += v[r];
  }
  acc *= 1e-30f;  // keeps every load alive without letting the state drift
#pragma unroll
  for (int r = 0; r < WN; ++r) str_stream(out + (int64_t)r * ld, o4, v[r % RD] + acc);
#pragma unroll
  for (int r = 0; r < WP; ++r) {  // the rows the step rewrites in place: goals (rows 0 .. D-1) and the last two
    const int row = r < D ? r : RD - 2 + (r - D);
    str(state + (int64_t)row * ld, o4, v[row] + acc);
  }
  str_stream(bytes, o1, (uint8_t)(acc > 1.f ? 1 : 0));
}

}  // namespace mt

// Counter-based RNG for the device-side reset / action_sample paths.
//
// The reference draws from numpy's global MT19937 in env order
// (manytor.py:216, :231), which ties results to a serial loop.  Here every
// draw is a pure function of (seed, GLOBAL env id, episode/step, draw index),
// so any sharding of the envs over GPUs gives the same per-env stream.
// Philox-4x32-10 as published (Salmon et al., SC'11); the numpy restatement in
// oracle/philox_ref.py is pinned by the Random123 known-answer vectors and the
// kernels are checked bit-for-bit against it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mt {

constexpr uint32_t kTagAction = 1u;
constexpr uint32_t kTagTarget = 2u;

struct u32x4 {
  uint32_t x, y, z, w;
};

__host__ __device__ __forceinline__ uint32_t mulhi32(uint32_t a, uint32_t b) {
  return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32);  // v_mul_hi_u32 on the device
}

__host__ __device__ __forceinline__ u32x4 philox4x32_10(u32x4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c.x;  // one 32x32->64 multiply per product
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c.z;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    c = u32x4{hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0};
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c;
}

// counter = (env_lo, env_hi[23:0] | tag << 24, major, minor)
__host__ __device__ __forceinline__ u32x4 stream_block(uint64_t seed, uint64_t env_id, uint32_t tag, uint32_t major,
                                                       uint32_t minor) {
  u32x4 c{(uint32_t)env_id, ((uint32_t)(env_id >> 32) & 0x00FFFFFFu) | (tag << 24), major, minor};
  return philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
}

// Integer degrees uniform in [-180, 180) (manytor.py:216) from a 32-bit word read as a fixed-point fraction u:
// digit 0 = floor(360 u), and the leftover fraction 360 u - floor(360 u) (the low half of the 64-bit product) is
// again uniform, so digit 1 = floor(360 * fraction) is a second, independent draw from the same word (joint
// probabilities deviate from 1/360^2 by at most 360^2 / 2^32 = 3e-5 relative).  One Philox block therefore
// covers up to 8 joints: joint j < 4 is digit 0 of word j, joint j >= 4 is digit 1 of word j - 4.
__host__ __device__ __forceinline__ float action_from_word(uint32_t w) { return (float)mulhi32(w, 360u) - 180.0f; }
__host__ __device__ __forceinline__ float action_from_word_digit1(uint32_t w) {
  return (float)mulhi32(w * 360u, 360u) - 180.0f;
}

// u in [0,1) from a 21-bit integer.  One Philox block (128 bits) is cut into six 21-bit fields, i.e. the three
// coordinates of TWO rejection-sampling candidates:
//   candidate 0: x, y, z fields = top 21 bits of words 0, 1, 2
//   candidate 1: x, y, z fields = (low 11 bits of word 0, 1, 2) << 10 | bits [31:22], [21:12], [11:2] of word 3
// All fields are disjoint bit ranges of the block, so the six uniforms are independent.
__host__ __device__ __forceinline__ float u21(uint32_t v) { return (float)v * 4.76837158203125e-07f; }  // 2^-21
template <int HALF>
__host__ __device__ __forceinline__ void candidate_fields(const u32x4& w, uint32_t& fx, uint32_t& fy, uint32_t& fz) {
  if (HALF == 0) {
    fx = w.x >> 11;
    fy = w.y >> 11;
    fz = w.z >> 11;
  } else {
    fx = ((w.x & 0x7FFu) << 10) | (w.w >> 22);
    fy = ((w.y & 0x7FFu) << 10) | ((w.w >> 12) & 0x3FFu);
    fz = ((w.z & 0x7FFu) << 10) | ((w.w >> 2) & 0x3FFu);
  }
}

}  // namespace mt

"""numpy restatement of the device RNG streams -- TEST INFRASTRUCTURE ONLY.

The reference has no device RNG (it uses numpy's global MT19937,
manytor.py:216,231); the build adds a counter-based Philox-4x32-10 generator
for benchmark mode so that results do not depend on how envs are sharded over
GPUs.  This file restates (a) Philox-4x32-10 as published by Salmon et al.,
"Parallel random numbers: as easy as 1, 2, 3" (SC'11), pinned by the Random123
known-answer vectors in tests/test_oracle_golden.py, and (b) the two stream
definitions of manytor_amd/csrc/philox.h, so the HIP kernels can be checked
bit-for-bit.

Stream definitions (key = (seed_lo, seed_hi)):
  counter = (env_lo, (env_hi & 0x00FFFFFF) | tag << 24, major, minor)
  actions : tag 1, major = step index, minor = 0; one block serves up to 8 joints:
            joint j < 4 : float(mulhi32(w_j, 360)) - 180                    (digit 0 of word j)
            joint j >= 4: float(mulhi32(lo32(w_{j-4} * 360), 360)) - 180    (digit 1 of word j-4)
            -> integer degrees in [-180, 180)
  targets : tag 2, major = episode index, minor = block index (0,1,2,...); every block yields TWO candidates,
            taken in the order (block 0, half 0), (block 0, half 1), (block 1, half 0), ...: six disjoint
            21-bit fields of the 128-bit block,
              half 0: fx, fy, fz = w0 >> 11, w1 >> 11, w2 >> 11
              half 1: fx = (w0 & 0x7FF) << 10 | w3 >> 22 ; fy = (w1 & 0x7FF) << 10 | (w3 >> 12) & 0x3FF ;
                      fz = (w2 & 0x7FF) << 10 | (w3 >> 2) & 0x3FF
            u21(f) = f * 2^-21
            x = 2R*u21(fx) - R ; y = 2R*u21(fy) - R ; z = R*u21(fz)   (fp32, one rounding per op)
            accept iff (x*x + y*y) + z*z <= R*R                     (fp32, one rounding per op)
            (z is drawn from [0,R) directly: conditioning the reference's
             uniform(-R,R) on z >= 0, manytor.py:232, gives the same law.)
"""
from __future__ import annotations

import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = 0x9E3779B9
W1 = 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)

TAG_ACTION = 1
TAG_TARGET = 2


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox-4x32-10.  All args broadcastable uint32 arrays.
    Returns 4 uint32 arrays."""
    c0, c1, c2, c3 = (np.asarray(v, dtype=np.uint64) & MASK for v in np.broadcast_arrays(c0, c1, c2, c3))
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK
        c0, c1, c2, c3 = hi1 ^ c1 ^ np.uint64(k0), lo1, hi0 ^ c3 ^ np.uint64(k1), lo0
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return tuple(v.astype(np.uint32) for v in (c0, c1, c2, c3))


def _ctr(env_ids, tag):
    env_ids = np.asarray(env_ids, dtype=np.uint64)
    lo = env_ids & MASK
    hi = ((env_ids >> np.uint64(32)) & np.uint64(0x00FFFFFF)) | np.uint64(tag << 24)
    return lo, hi


def sample_actions(seed, env_ids, step_idx, dof):
    """(N, dof) float32 integer-valued degrees in [-180, 180).  One Philox block per env and step: joint j < 4 is
    digit 0 of word j (floor(360 u)), joint j >= 4 is digit 1 of word j - 4 (floor(360 * frac(360 u)))."""
    lo, hi = _ctr(env_ids, TAG_ACTION)
    out = np.empty((lo.shape[0], dof), dtype=np.float32)
    w = philox4x32_10(lo, hi, np.uint64(step_idx & 0xFFFFFFFF), np.uint64(0), seed & 0xFFFFFFFF, seed >> 32)
    for col in range(dof):
        prod = w[col % 4].astype(np.uint64) * np.uint64(360)
        if col >= 4:
            prod = (prod & MASK) * np.uint64(360)
        out[:, col] = (prod >> np.uint64(32)).astype(np.float32) - np.float32(180.0)
    return out


def candidate_fields(w):
    """The six 21-bit fields of one Philox block (4 uint32 arrays) as two (fx, fy, fz) triples."""
    w0, w1, w2, w3 = (np.asarray(v, dtype=np.uint32) for v in w)
    lowmask, ten = np.uint32(0x7FF), np.uint32(0x3FF)
    half0 = (w0 >> np.uint32(11), w1 >> np.uint32(11), w2 >> np.uint32(11))
    half1 = (((w0 & lowmask) << np.uint32(10)) | (w3 >> np.uint32(22)),
             ((w1 & lowmask) << np.uint32(10)) | ((w3 >> np.uint32(12)) & ten),
             ((w2 & lowmask) << np.uint32(10)) | ((w3 >> np.uint32(2)) & ten))
    return half0, half1


def sample_targets(seed, env_ids, episode_idx, obj_number, radius):
    """(N, K, 3) float32 targets by per-env rejection sampling.  `episode_idx`: one index for all envs or one per env
    (envs re-armed on the device run on their own episode counters)."""
    lo, hi = _ctr(env_ids, TAG_TARGET)
    episode_idx = np.asarray(episode_idx, dtype=np.uint64) & MASK
    n = lo.shape[0]
    r = np.float32(radius)
    r2 = np.float32(2.0) * r
    rr = r * r
    out = np.zeros((n, obj_number, 3), dtype=np.float32)
    cnt = np.zeros(n, dtype=np.int64)
    blk = 0
    scale = np.float32(2.0 ** -21)
    while (cnt < obj_number).any():
        w = philox4x32_10(lo, hi, episode_idx, np.uint64(blk), seed & 0xFFFFFFFF, seed >> 32)
        for fields in candidate_fields(w):
            u = [f.astype(np.float32) * scale for f in fields]
            x = r2 * u[0] - r
            y = r2 * u[1] - r
            z = r * u[2]
            n2 = (x * x + y * y) + z * z
            ok = (n2 <= rr) & (cnt < obj_number)
            idx = np.nonzero(ok)[0]
            out[idx, cnt[idx], 0] = x[idx]
            out[idx, cnt[idx], 1] = y[idx]
            out[idx, cnt[idx], 2] = z[idx]
            cnt[idx] += 1
        blk += 1
    return out

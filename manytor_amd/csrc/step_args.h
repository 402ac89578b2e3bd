// Argument blocks of the step kernels: plain structs shared by the kernels (kernels.h) and the host side
// (engine_internal.h).  No device code in here.
#pragma once
#include <stdint.h>

#include "../../include/manytor_hip.h"

namespace mt {

constexpr int kBlock = 256;  // 4 wavefronts

// Internal bit of StepArgs::flags (never accepted from mt_config): every env's joint angles are whole degrees in
// [-180, 180) -- true after a reset and while only in-kernel sampled actions have been applied since (the host tracks
// it) -- so the kernels may take the sines / cosines of the pose a step starts from out of the whole-degree table.
constexpr uint32_t kFlagWholeGoals = 0x80000000u;

// Per-joint DH constants, uniform over the launch.  Passed by value in the
// kernel arguments, so they live in SGPRs (s_load from the kernarg segment);
// the MT_FLAG_DH_IN_LDS variant copies them to LDS first.
struct DhConst {
  float a[MT_MAX_DOF];
  float d[MT_MAX_DOF];
  float sa[MT_MAX_DOF];       // sin(alpha)
  float ca[MT_MAX_DOF];       // cos(alpha)
  float off_deg[MT_MAX_DOF];  // theta offset, degrees
  int32_t fo, fe;             // rows of joints_coordinates used for the observation / for pickup (0 = the origin row)
};

struct StepArgs {
  float* actions;                 // [D][ld]
  float* goals;                   // [D][ld]
  float* points;                  // [3K][ld]
  uint32_t* alive;                // [ld] bit p = target p alive
  float* total_reward;            // [ld]
  float* obs;                     // [3K][ld]
  int32_t* reward;                // [ld]
  uint8_t* done;                  // [ld]
  unsigned long long* done_bits;  // [ld/64]
  float* ee;                      // [3][ld]
  uint32_t* episodes;             // [ld] episode index of each env (keys its target draws)
  float* last_return;             // [ld] return of the episode that ended at the last (auto-)reset
  float* ring;                    // [ring_slots][ld] returns of the episodes an env finished, slot = finished count % ring_slots
  uint32_t* bad_actions;          // [1] number of (env, step) pairs whose staged action was not a usable angle
  const float* trig_table;        // [450][2] (sin, cos) of the whole degrees -270 .. 179, filled once at mt_create (kernels.h: SinCos)
  float* zmin;                    // [ld] MT_FLAG_DEBUG_ZMIN only (else NULL): the z-minimum the ground test of the last step used
  // the first step launch of an episode (step_kernel<.., FRESH>): the deferred full reset it starts with -- the seed and
  // the episode index of the target stream, the radius of the half ball
  uint32_t reset_seed_lo, reset_seed_hi, reset_episode;
  float radius;
  float* snap;                    // [n] NULL, or (the last step launch of an mt_rollout) the overlapped gather's snapshot row: the returns again
  uint32_t ring_slots, episode0;  // episode0 = episode index every env got at the last full reset
  int64_t n, ld, env_base;
  int32_t K, S;
  float tol, inv_sm1;
  uint32_t flags;
  uint32_t seed_lo, seed_hi, major;  // RNG key + step / episode index
  // Launches replayed from a HIP graph (mt_rollout on small batches): `major` is the launch's offset inside the
  // segment and the segment's first step index sits in this device word, which the host sets before every replay.
  const uint32_t* major_base;
  DhConst dh;
#ifdef MT_STAMPS
  unsigned long long* stamps;  // diagnostic build only (tools/microbench/step_stamps.hip): [waves][8] shader-clock stamps
#endif
};

}  // namespace mt

/*
 * manytor_hip.h -- C ABI of libmanytor_hip.so (gfx950 / MI355X).
 *
 * Batched manipulator-environment step engine: N lock-stepped arms per handle,
 * state resident in HBM as struct-of-arrays, one HIP launch per step.
 *
 * The reference (victorkich/ManyTor, manytor.py) has no FFI boundary: its
 * boundary is the Python class surface of `Environment` / `Multienv`.  Each entry
 * point below names the reference interface it replaces (file:line in the
 * reference checkout) -- the Python host code in manytor_amd/ binds them with
 * ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - every function returns an mt_status (0 = ok, < 0 = error); the message of
 *     the last error is available through mt_last_error().
 *   - plain pointers and sizes only; no C++ / torch types.
 *   - a handle owns one device, one HIP stream (replaceable with
 *     mt_set_stream / mt_use_own_stream) and all of its device buffers.  A handle is not
 *     re-entrant; different handles may be used from different threads.
 *   - launches are asynchronous on the handle's stream; mt_sync() or any
 *     host-destination mt_get() synchronises.
 *   - angles are DEGREES (manytor.py:39), lengths are the DH table's units.
 *   - "env-major" = the reference's array shapes with a leading env axis, e.g.
 *     points (N, K, 3), obs (N, 3K).  "SoA" = the resident device layout, one
 *     row of length `ld` (>= N, multiple of 256) per scalar field component,
 *     e.g. points row (3*k + axis).
 */
#ifndef MANYTOR_HIP_H
#define MANYTOR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define MT_API __attribute__((visibility("default")))
#else
#define MT_API
#endif

#define MT_VERSION 400          /* major*10000 + minor*100 + patch */
#define MT_MAX_DOF 8
#define MT_MAX_TARGETS 32
#define MT_MAX_RETURN_RING 64
#define MT_UNIQUE_ID_BYTES 128     /* = NCCL_UNIQUE_ID_BYTES of RCCL */

typedef struct mt_engine* mt_handle;

typedef enum mt_status {
  MT_OK = 0,
  MT_ERR_INVALID_ARG = -1,
  MT_ERR_HIP = -2,              /* a HIP runtime call failed; see mt_last_error */
  MT_ERR_NO_DEVICE = -3,
  MT_ERR_ALLOC = -4,
  MT_ERR_STATE = -5,            /* call order violated (e.g. step before reset) */
  MT_ERR_UNSUPPORTED = -6
} mt_status;

/* Fields of the resident state (mt_get / mt_set / mt_device_ptr). */
typedef enum mt_field {
  MT_F_ACTIONS = 0,       /* f32  env-major (N, D)      SoA rows: D        staged action, degrees            */
  MT_F_GOALS = 1,         /* f32  (N, D)                rows: D            manytor.py:133 `goals`            */
  MT_F_POINTS = 2,        /* f32  (N, K, 3)             rows: 3K           manytor.py:137 `points`           */
  MT_F_ALIVE = 3,         /* u8   (N, K) via mt_get; device: u32 bitmask per env (bit p = target p alive)    */
  MT_F_OBS = 4,           /* f32  (N, 3K)               rows: 3K           obs2 of the last step / observe   */
  MT_F_REWARD = 5,        /* i32  (N,)                                      reward of the last step           */
  MT_F_DONE = 6,          /* u8   (N,)   done of the last step: 0 running, 1 finished (mt_reset_done will re-arm it),  */
                          /*             2 finished and already re-armed inside mt_rollout_fused(auto_reset)          */
  MT_F_DONE_BITS = 7,     /* u64  (ceil(N/64),) one wavefront ballot per 64 envs, bit l = env 64*w + l       */
  MT_F_EE = 8,            /* f32  (N, 3)                rows: 3            end effector = joints_coordinates[-1] */
  MT_F_TOTAL_REWARD = 9,  /* f32  (N,)                                      manytor.py:138 `total_reward`     */
  MT_F_JOINTS = 10,       /* f32  (N, D, 3) computed on demand from goals (manytor.py:188-189); mt_get only  */
  MT_F_EPISODES = 11,     /* u32  (N,)   episode index of each env: set by a reset, +1 per auto re-arm          */
  MT_F_LAST_RETURN = 12,  /* f32  (N,)   total_reward the env had when it was last reset / re-armed             */
  MT_F_RETURN_RING = 13,  /* f32  (N, R) rows: R = mt_config.return_ring.  The return of the c-th episode an env finished */
                          /*             since the last full reset (re-armed by mt_reset_done or in-kernel) is in slot   */
                          /*             c % R; c = MT_F_EPISODES - (episode of the last full reset)                     */
  MT_F_TRACE = 14,        /* f32  (N, S, 3) rows: 3S.  End effector at each of the S sub-step poses of the last step     */
                          /*             (the rows manytor.py:190 appends to `trajectory`); needs MT_FLAG_TRACE          */
  MT_F_ZMIN = 15,         /* f32  (N,)   DEBUG (needs MT_FLAG_DEBUG_ZMIN): signed minimum z of the observation and pickup     */
                          /*             frames over ALL S sub-step poses of the last step, exactly the value the step     */
                          /*             kernel compares with 0 for the ground flag (manytor.py:191-192)                    */
  MT_F_COUNT = 16
} mt_field;

typedef enum mt_dtype { MT_F32 = 0, MT_F64 = 1, MT_I32 = 2, MT_I64 = 3, MT_U8 = 4, MT_U32 = 5, MT_U64 = 6 } mt_dtype;

typedef enum mt_layout { MT_ENV_MAJOR = 0, MT_SOA = 1 } mt_layout;

/* Engine flags (mt_config.flags).  0 = the default kernels, which reproduce the reference's semantics. */
#define MT_FLAG_TERMINATE_ON_GROUND 0x1u /* done |= ground hit (README.md:44 intent; NOT what manytor.py:170 does)  */
/* Measured alternatives of the step kernel (profiles/r01_variants.md); same results within the fp32 tolerance. */
#define MT_FLAG_HW_TRIG 0x2u             /* v_sin/v_cos at every interior sub-step instead of the recurrence       */
#define MT_FLAG_DH_IN_LDS 0x4u           /* DH constants staged in LDS instead of SGPRs (implies NO_SPECIALIZE)    */
#define MT_FLAG_DIRECT_TRIG 0x8u         /* polynomial sincos at every interior sub-step (no recurrence)           */
#define MT_FLAG_NO_SPECIALIZE 0x10u      /* never use a compile-time DH table, even if the table matches one       */
#define MT_FLAG_TRACE 0x20u              /* keep MT_F_TRACE: every step also writes the end effector of each       */
                                         /* sub-step pose (300 B/env-step at S = 25; off by default, manytor.py:190) */
#define MT_FLAG_DEBUG_ZMIN 0x40u         /* keep MT_F_ZMIN: the step / rollout kernels also store the z-minimum    */
                                         /* their ground test used (4 B/env-step; parity tests pin the sub-step    */
                                         /* recurrence of the timed kernels with it; off by default)               */
/* Profiling builds.  OUTPUTS ARE WRONG ON PURPOSE; never set outside bench.py --ablate. */
#define MT_FLAG_ABLATE_LOOP 0x100u       /* skip the interior sub-steps                                            */
#define MT_FLAG_ABLATE_OBS 0x200u        /* with ABLATE_LOOP: also skip the observation arithmetic (memory only);  */
                                         /* alone: keep all arithmetic, drop target loads / observation stores     */

/* Constructor arguments.  Replaces Environment.__init__/Multienv.__init__
 * (manytor.py:130-139, :77-82) plus the literals the reference hard-codes:
 * DH table manytor.py:42-48, substeps :178, pickup tolerance :162, radius :231. */
typedef struct mt_config {
  int32_t struct_size;          /* = sizeof(mt_config), checked                                  */
  int32_t device;               /* HIP device ordinal                                            */
  int64_t n_envs;               /* envs owned by this handle (this rank's shard)                 */
  int64_t env_id_base;          /* global id of local env 0: keys the device RNG so results do   */
                                /* not depend on the shard count                                 */
  int32_t dof;                  /* 2..MT_MAX_DOF joints                                          */
  int32_t n_targets;            /* 1..MT_MAX_TARGETS  (obj_number)                               */
  int32_t substeps;             /* >= 2; reference 25                                            */
  uint32_t flags;               /* MT_FLAG_*                                                     */
  float pickup_tol;             /* reference 8.0                                                 */
  float radius;                 /* target hemisphere radius, reference 51.3                      */
  float dh_table[MT_MAX_DOF * 4]; /* rows (a, alpha_rad, d, theta_offset_rad), manytor.py:42-48 */
  int32_t return_ring;          /* slots of MT_F_RETURN_RING per env, 0..MT_MAX_RETURN_RING (0 = none) */
  int32_t obs_frame;            /* row of joints_coordinates the observation is measured from: -dof..dof-1, Python   */
                                /* indexing; the reference uses [2] = -2, the elbow (manytor.py:143)                 */
  int32_t ee_frame;             /* row the pickup test uses; reference [3] = -1, the end effector (manytor.py:162).  */
                                /* The ground test looks at the z of both rows (manytor.py:191).  Row 0 = the origin */
  int32_t reserved;             /* must be 0                                                     */
} mt_config;

MT_API int mt_version(void);
MT_API const char* mt_status_string(int status);
/* Message of the last failing call on this handle (or, with h == NULL, of the
 * last failing call on the calling thread that had no handle). */
MT_API const char* mt_last_error(mt_handle h);
MT_API int mt_device_count(int* count);

/* Environment()/Multienv() constructors, manytor.py:130-139 / :77-82. */
MT_API int mt_create(mt_handle* out, const mt_config* cfg);
MT_API int mt_destroy(mt_handle h);
/* Which instantiation mt_step / mt_step_random launch for this handle (the schedule is picked by batch size and table
 * at mt_create; every schedule gives the same bits): e.g. "step_kernel<Ref4Table, trig=0, lds=false, pf=8>",
 * "step_split_kernel<RtTable<5>, L=4>".  For benchmark records and profiles; valid until the next call on the handle. */
MT_API const char* mt_step_kernel_name(mt_handle h);
/* The dispatch of this handle as data: a JSON object with the schedule resolved for every entry point ("step",
 * "chains", "rollout", "fused", "reset"), the MT_* environment overrides that were in effect at mt_create ("overrides")
 * and the library's one table of size thresholds ("policy": engine.hip kPolicy).  Tests assert regimes from it instead
 * of parsing kernel names; valid until the next mt_describe_dispatch on the handle. */
MT_API const char* mt_describe_dispatch(mt_handle h);
/* Run on a caller-owned hipStream_t (e.g. torch's current stream), so that the handle's launches are ordered with
 * the caller's own work on that stream.  NULL means what it means to HIP: the legacy default stream (which is what
 * torch's default stream is).  mt_use_own_stream goes back to the handle's private non-blocking stream, which is
 * NOT ordered against any other stream: readers of the device buffers must then mt_sync() first.  Both calls
 * drain the stream the handle was on. */
MT_API int mt_set_stream(mt_handle h, void* hip_stream);
MT_API int mt_use_own_stream(mt_handle h);
MT_API int mt_sync(mt_handle h);

/* Environment.reset(), manytor.py:219-253, for all envs.
 * mt_reset: targets supplied by the caller (parity mode: the host draws them from
 * numpy's global RNG in the reference's order).  `points` is (N, K, 3) f32
 * env-major or (3K, ld) SoA, in host or device memory.
 * mt_reset_random: targets rejection-sampled on the device (Philox-4x32-10 keyed by
 * seed / global env id / episode), same law as manytor.py:229-239. */
/* Host arrays are screened: a NaN or an infinite coordinate is MT_ERR_INVALID_ARG and nothing is written.  Targets handed
 * over in device memory cannot be screened on the host: the reset kernel drops an unusable one (dead from the start,
 * coordinates zeroed) and counts it (mt_bad_action_count). */
MT_API int mt_reset(mt_handle h, const float* points, int layout, int is_device);
/* On a handle that runs on its OWN stream and whose mt_rollout runs several steps per launch (<= 262 144 envs), the launch of
 * mt_reset_random is deferred: the next mt_rollout performs the reset as the prologue of its first launch (the same state,
 * bit for bit, one launch and one round trip of the state less), and every other entry point -- mt_sync included, which is
 * what the own-stream contract asks for before anything else looks at the state -- launches it first; only the region
 * timers (mt_timer_start / mt_timer_stop*) leave it alone, it belongs to what follows them.  A single-step mt_rollout absorbs it
 * as well.  On a caller's stream (mt_set_stream) the reset is launched by the call itself, in stream order.
 * MT_DEFER_RESET=0 disables the deferral. */
MT_API int mt_reset_random(mt_handle h, uint64_t seed, uint32_t episode);
/* Re-arm only the envs whose done byte is 1: their return goes to MT_F_LAST_RETURN and into MT_F_RETURN_RING, their
 * episode index (MT_F_EPISODES) advances by one and keys the new targets (device RNG), pose and return are zeroed.
 * Envs with done == 2 were already re-armed inside mt_rollout_fused: only their flag is cleared.  Not in the
 * reference (the caller resets everything, test_multi.py:34; reset on done is caller-driven, test_single.py:20-21,32);
 * SURVEY.md 8(f) rank 1. */
MT_API int mt_reset_done(mt_handle h, uint64_t seed);
/* `multienv.environment[i].reset()` (manytor.py:82 + :219-253): reset ONE env of the batch.  `points` = K x 3 host
 * floats, or NULL to draw them on the device (Philox keyed by seed / global env id / `episode`). */
MT_API int mt_env_reset(mt_handle h, int64_t env, const float* points, uint64_t seed, uint32_t episode);

/* Stage the action of the next mt_step: (N, D) env-major or (D, ld) SoA, degrees,
 * any of f32/f64/i32/i64 (Environment.action_sample returns np.int64, manytor.py:216). */
MT_API int mt_set_actions(mt_handle h, const void* actions, int dtype, int layout, int is_device);
/* Environment.action_sample(), manytor.py:215-217, for all envs on the device:
 * integer degrees uniform in [-180, 180), written to the action buffer. */
MT_API int mt_sample_actions(mt_handle h, uint64_t seed, uint32_t step_idx);

/* Environment.step() = get_observations() side effect + action() + return
 * accumulation + is_done(), manytor.py:255-260, :175-213, for all envs, one launch. */
/* On a multi-chain handle (163 840 .. 3 M envs, see mt_rollout) that runs on its OWN stream the step is two launches, one
 * per half of the env range on its own stream; mt_set_actions from DEVICE memory and mt_sample_actions stage each half's
 * rows on the same streams, so a policy loop (set actions / step / ...) keeps the halves independent from call to call
 * (any other call and mt_sync fold them back): 42 -> 37.5 us per step at 1 M envs.  On a caller's stream (mt_set_stream)
 * the step stays one launch: forking behind the caller's work and joining back for a single launch costs more than it
 * hides.  Same bits either way. */
MT_API int mt_step(mt_handle h);
/* One host round trip of Multienv.step (manytor.py:115-122): (N, D) host actions in, obs2 (N, 3K) f32, reward (N,)
 * i32 and done (N,) u8 out, through one page-locked staging buffer and ONE stream synchronisation (instead of the
 * four that mt_set_actions + mt_step + 3 x mt_get cost).  For small batches driven from host code. */
MT_API int mt_step_host(mt_handle h, const void* actions, int dtype, float* obs, int32_t* reward, uint8_t* done);
/* `multienv.environment[i].step(action)` (manytor.py:82,118 + :255-260): step ONE env of the batch with a host action of
 * D degrees; the other envs are untouched.  obs (3K f32), reward, done are written to host memory; synchronises. */
MT_API int mt_env_step(mt_handle h, int64_t env, const float* action, float* obs, int32_t* reward, uint8_t* done);
/* Number of (env, step) pairs so far whose staged action was not a finite angle of magnitude <= 32768 degrees (NaN,
 * +-inf from a diverging policy ...).  Such an env holds its pose for that step instead of poisoning its state.
 * Synchronises.  The reference has no such check (numpy would propagate the NaN into goals, manytor.py:184).
 * Also counts the targets mt_reset dropped because a coordinate handed over in device memory was NaN / infinite. */
MT_API int mt_bad_action_count(mt_handle h, uint64_t* count);
/* The same with the action drawn in-kernel (results bit-identical to mt_sample_actions followed by
 * mt_step).  The drawn action is not stored in MT_F_ACTIONS: it is the new MT_F_GOALS (goals = action after a
 * step, manytor.py:184), which saves 4*D bytes of traffic per env. */
MT_API int mt_step_random(mt_handle h, uint64_t seed, uint32_t step_idx);
/* n_steps x mt_step_random with step indices step_idx0, step_idx0+1, ... (the
 * inner loop of test_multi.py:19-21).  The call exposes the state after n_steps steps and the outputs of the LAST one
 * (obs, reward, done, end effector), so on small shards (<= 262 144 envs) it runs FIVE steps per launch through the
 * kernels of mt_rollout_fused -- joint angles, alive mask and return in registers, targets in LDS between them; every
 * step still computes and writes its outputs -- bit-identical to the launch-per-step sequence, without four of five
 * kernel boundaries and state re-fetches (MT_ROLLOUT_K=1 gives one launch per step back; then batches <= 131 072 envs are
 * replayed from a HIP graph that the handle captures once per segment length).
 * On large batches (above 262 144 and up to 3 M envs; from 163 840 with MT_ROLLOUT_K=1) the call runs as TWO independent chains of launches -- the two halves of the env
 * range (256-aligned) on two streams forked from the handle's stream: a step of env i depends only on env i, so the
 * results are bit-identical, and one half's kernel boundary is hidden behind the other half's kernel (-10 % per step at
 * 1 M envs).  On the handle's own stream the chains stay forked when the call returns: the next mt_rollout continues them,
 * mt_reset_random resets each half behind its own last step, mt_gather_returns_begin snapshots each half on its chain;
 * mt_step / mt_sample_actions / mt_set_actions(device) run per half as well (see mt_step); every other call -- mt_sync,
 * getters, setters, mt_reset_done, timers' begin ... -- folds them back into the
 * handle's stream first, so "mt_sync before foreign reads" means what it did.  On a caller's stream (mt_set_stream) the
 * chains are joined before the call returns: work queued on that stream afterwards sees the completed rollout. */
MT_API int mt_rollout(mt_handle h, int n_steps, uint64_t seed, uint32_t step_idx0);
/* The same n_steps steps in ONE launch: joint angles, alive mask and return stay in registers and the targets
 * in LDS between steps, so a step only writes its outputs (obs, reward, done, end effector; MT_F_* hold the last
 * step's).  auto_reset != 0 re-arms an env in the step it finishes, exactly like mt_reset_done after every step:
 * every finished return lands in MT_F_RETURN_RING, and an env that finished in the LAST step keeps done == 2
 * (finished, already re-armed) so that a following mt_reset_done does not reset it twice.
 * State after the call is bit-identical to the launch-per-step sequence. */
MT_API int mt_rollout_fused(mt_handle h, int n_steps, uint64_t seed, uint32_t step_idx0, int auto_reset);

/* Environment.get_observations(), manytor.py:141-153, at the current pose (also
 * zeroes the coordinates of dead targets, :148).  Result in MT_F_OBS. */
MT_API int mt_observe(mt_handle h);
/* Environment.is_done(), manytor.py:155-173, at the current pose. Result in MT_F_DONE. */
MT_API int mt_check_done(mt_handle h);

/* Copy a field out in the reference's env-major shape (see mt_field) to host
 * (is_device = 0, synchronises) or device memory.  dst_bytes must match. */
MT_API int mt_get(mt_handle h, int field, void* dst, int64_t dst_bytes, int is_device);
/* Overwrite GOALS / POINTS / ALIVE(u8 N,K) / TOTAL_REWARD from host env-major
 * arrays (attribute assignment on the reference objects, e.g. manytor.py:243), and -- for restoring a checkpoint --
 * DONE (u8 N; the ballot words are rebuilt), EPISODES (u32 N), LAST_RETURN (f32 N), RETURN_RING (f32 N,R).
 * Floating-point input is screened: NaN / +-inf anywhere, or a joint angle beyond +-32768 degrees, is
 * MT_ERR_INVALID_ARG and nothing is written (the kernels assume finite state). */
MT_API int mt_set(mt_handle h, int field, const void* src, int64_t src_bytes);
/* The episode index every env was given by the last full reset (finished-episode counts and return-ring slots are
 * relative to it): mt_reset / mt_reset_random set it; a checkpoint restore sets it back with this call. */
MT_API int mt_set_episode_base(mt_handle h, uint32_t episode0);
/* Raw resident buffer: pointer to row 0, number of rows, row stride in elements and element dtype.  Writing through the
 * pointer is the caller's business (order it with the handle's stream).  Asking for MT_F_GOALS tells the library that
 * joint angles may change behind its back: from then on it never assumes them to be whole degrees (the table look-up
 * the sampled-action kernels use for the pose a step starts from is replaced by the computed sines / cosines). */
MT_API int mt_device_ptr(mt_handle h, int field, void** ptr, int64_t* rows, int64_t* ld, int* dtype);

/* ---- multi-GPU: the one exchange of the path ---------------------------------------------------
 * Envs share nothing, so stepping needs no collective.  What is exchanged is what test_multi.py:32 prints: the
 * per-env return, all-gathered over the ranks (one process per GPU) with RCCL over xGMI, straight out of the
 * arena on the handle's stream -- no staging copy, no host round trip.  librccl.so is dlopen'ed on first use, so the
 * library loads and runs on a box without RCCL.
 *   rank 0: mt_comm_unique_id(id)  ->  ship the 128 bytes to the other ranks by any means (the Python host uses the
 *   torch.distributed store)  ->  every rank: mt_comm_init(h, id, rank, world)  [collective, ncclCommInitRank]. */
MT_API int mt_comm_unique_id(void* id_out /* MT_UNIQUE_ID_BYTES */);
MT_API int mt_comm_init(mt_handle h, const void* unique_id, int rank, int world_size);
MT_API int mt_comm_destroy(mt_handle h);
/* All-gather row `row` of a single-precision field (MT_F_TOTAL_REWARD, MT_F_LAST_RETURN: row 0; MT_F_RETURN_RING: a
 * slot) of every rank into dst (device memory, global env order, dst_elems = total number of envs over all ranks;
 * shards may differ in size).  Asynchronous on the handle's stream.  Without mt_comm_init (one GPU) it is a
 * device-to-device copy. */
MT_API int mt_gather_returns(mt_handle h, int field, int row, float* dst, int64_t dst_elems);
/* The same exchange taken off the handle's stream: the row is snapshotted on the handle's stream (so a reset queued
 * next may overwrite it), the all-gather runs on a side stream of the handle, and the call returns at once -- steps and
 * resets queued afterwards overlap with the exchange over xGMI.  mt_gather_returns_wait orders the handle's stream
 * behind the last begun gather; with host_wait != 0 it also blocks the calling thread until dst is complete and, if
 * elapsed_ms is not NULL, reports the device time the exchange took (if mt_sync has already waited for it: the time
 * of that last completed exchange).  dst must not be read, and the communicator not destroyed, before that.  mt_sync
 * also waits for a begun gather.
 * Several gathers may be begun without a wait in between (each with a dst of its own): they run one after the other on the
 * side stream.  The snapshot is double-buffered, and where mt_rollout writes it itself (its last launch, small batches)
 * mt_gather_returns_begin lets the calling thread run at most ONE exchange ahead of the device: if the exchange before the one
 * just begun has not finished yet it waits for it (the device still holds a whole episode of queued work then), so that the
 * next episode needs neither a snapshot launch nor a stream wait between its steps and the gather.  MT_GATHER_THROTTLE=0
 * never waits (the snapshot is then a launch of its own whenever the host is further ahead). */
MT_API int mt_gather_returns_begin(mt_handle h, int field, int row, float* dst, int64_t dst_elems);
MT_API int mt_gather_returns_wait(mt_handle h, int host_wait, float* elapsed_ms);
/* The same without the snapshot: the exchange on the side stream reads the arena row itself, so nothing is copied on the
 * handle's stream.  Meant for MT_F_LAST_RETURN right after the reset that ended an episode (the reset stores every env's
 * finished return there): that row is written by resets only, and the library orders every later reset / re-arm of its
 * own behind the exchange.  For any other row the caller must not let it change before mt_gather_returns_wait. */
MT_API int mt_gather_returns_begin_inplace(mt_handle h, int field, int row, float* dst, int64_t dst_elems);
/* Total number of envs over all ranks of the communicator (n_envs without one). */
MT_API int mt_comm_total_envs(mt_handle h, int64_t* total);
/* Summary of a return row over ALL ranks without moving the row (what a learner logs per episode):
 * sum / min / max of row `row` of `field` (as for mt_gather_returns), the number of envs, and how many of them have
 * their done flag set.  Reduced on the device (double accumulation, fixed order: the same bits on every rank and from
 * run to run), five numbers per rank exchanged over the communicator (all-gather), combined on the host.
 * Synchronous on the handle's stream; collective when a communicator is attached. */
typedef struct mt_return_stats {
  double sum, min, max;
  int64_t count; /* envs over all ranks */
  int64_t done;  /* of which MT_F_DONE != 0 */
} mt_return_stats;
MT_API int mt_reduce_returns(mt_handle h, int field, int row, mt_return_stats* out);

/* HIP-event timer on the handle's stream (wall-clock of test_multi.py:16-18, device side).  Start and stop join the chains
 * but do not launch a deferred mt_reset_random: the next mt_rollout absorbs it INSIDE the timed region. */
MT_API int mt_timer_start(mt_handle h);
MT_API int mt_timer_stop(mt_handle h, float* elapsed_ms);
/* The same end mark WITHOUT joining the chains and without a host wait: one end event on every stream of the handle
 * that may still carry work (its stream, the chain streams while forked); mt_timer_read waits for them and returns the
 * time from mt_timer_start to the LAST of them -- and to the end of an exchange begun with mt_gather_returns_begin that
 * was still pending at the mark.  bench.py brackets each timed region with the pair to report the region's device
 * timeline next to its wall clock. */
MT_API int mt_timer_stop_async(mt_handle h);
MT_API int mt_timer_read(mt_handle h, float* elapsed_ms);
/* Lap timer: any number of begin/end event pairs recorded on the stream WITHOUT host synchronisation;
 * mt_timer_laps_total synchronises once, returns the summed device time of all laps and clears them.  Lets a
 * benchmark time only its step launches inside a longer region without stalling the GPU at every lap.  A lap begins behind
 * everything queued before it (a deferred reset is launched first); while the chains are forked on the handle's own stream it
 * has one begin event per chain and runs from the earliest of them to the latest end event -- no join for the stopwatch. */
MT_API int mt_timer_lap_begin(mt_handle h);
MT_API int mt_timer_lap_end(mt_handle h);
MT_API int mt_timer_laps_total(mt_handle h, float* total_ms, int* n_laps);
/* The same, one figure per lap in recording order (ms[0 .. *n_laps)); clears the laps.  With capacity smaller than the
 * number of laps it only reports *n_laps and fails, leaving the laps in place. */
MT_API int mt_timer_lap_times(mt_handle h, float* ms, int capacity, int* n_laps);

/* Stateless kinematics helpers = the module functions of the reference.
 * mt_fk_batch: fk(mode, goals), manytor.py:35-53 -> 4x4 row-major per pose; with
 * dof = mode = 1, angles_in_radians = 1 it is dh(a, alfa, d, theta), manytor.py:25-32.
 * mt_r_theta_batch: r_theta(v1, v2), manytor.py:17-22 -> (r_deg, theta_deg).
 * Host pointers in, host pointers out; run on `device`. */
MT_API int mt_fk_batch(int device, const float* dh_table, int dof, int mode, const float* angles, int angles_in_radians,
                int64_t n, float* out_mat16);
/* joints_coordinates at each of the `substeps` poses of the straight joint-space route prev -> action
 * (manytor.py:182-190; what the reference appends to `trajectory` and streams to its viewer): host (n, dof) in,
 * host (n, substeps, dof, 3) out.  Off the step path; meant for the few envs one draws or logs. */
MT_API int mt_route_trace(int device, const float* dh_table, int dof, int substeps, const float* prev,
                          const float* action, int64_t n, float* out);
MT_API int mt_r_theta_batch(int device, const float* v1, const float* v2, int64_t n, float* out_r_theta);
/* Measurement aid ("report against the measured bandwidth"): a streaming kernel with exactly the memory
 * operations of one mt_step_random -- dof + 3 targets + 2 rows read, dof + 2 rewritten in place, 3 targets + 4 rows and
 * one byte row written non-temporally, one env per lane, the arena's row pitch -- and none of its arithmetic, run `reps`
 * times over n_envs envs.  us_per_pass = average device time of a pass, bytes_per_pass = (8 dof + 24 targets + 33) x
 * n_envs.  Built for (dof, targets) = (4, 7), (7, 7), (4, 10); allocates and frees its own buffers. */
MT_API int mt_stream_probe(int device, int dof, int n_targets, int64_t n_envs, int reps, float* us_per_pass,
                           int64_t* bytes_per_pass);

#ifdef __cplusplus
}
#endif
#endif /* MANYTOR_HIP_H */

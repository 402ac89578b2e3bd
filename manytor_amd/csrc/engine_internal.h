// Host-side state of one handle and the error/device helpers shared by engine.hip and comm.hip.
// Internal: nothing in here is part of the C ABI (include/manytor_hip.h).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <string>
#include <vector>

#include "step_args.h"

struct mt_comm;  // comm.hip: the RCCL communicator of a handle

struct mt_engine {
  static constexpr int kMaxChains = 4;  // mt_rollout's independent chains of launches (see `chains` below)
  mt_config cfg{};
  int64_t n = 0, ld = 0;
  int D = 0, K = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // mt_timer_stop_async / mt_timer_read: one end event per stream of the handle that may carry work (its stream, the
  // chain streams while forked; a pending exchange on the side stream ends at ev_g1)
  hipEvent_t ev1c[kMaxChains] = {nullptr, nullptr, nullptr, nullptr};  // [0] unused (= ev1)
  int timer_ends = 0;              // end events recorded by the last mt_timer_stop_async: ev1 + ev1c[1 .. timer_ends-1]
  bool timer_with_gather = false;  // ... and an exchange was pending then (its end = ev_g1)
  // mt_timer_lap_*: a lap = one begin event per stream that carries work when it begins (the handle's stream; every chain's
  // while the chains are forked on the own stream) + one end event per stream that carried work of the lap
  // (the handle's stream, and the chain streams while mt_rollout's chains are forked); its time = begin -> the latest end
  struct LapRec {
    uint32_t begin, n_begin, end0, n_end;  // indices into lap_events: begin .. begin + n_begin - 1, end0 .. end0 + n_end - 1
  };
  std::vector<hipEvent_t> lap_events;  // event pool, grown on demand
  size_t lap_events_used = 0;
  std::vector<LapRec> lap_recs;
  bool lap_open = false;
  void* arena = nullptr;
  size_t arena_bytes = 0;
  void* staging = nullptr;
  size_t staging_bytes = 0;
  void* pinned = nullptr;  // page-locked host buffer of mt_step_host
  size_t pinned_bytes = 0;
  mt::StepArgs args{};
  float* trace = nullptr;  // [3S][ld] end-effector position per sub-step (MT_FLAG_TRACE), else NULL
  unsigned long long* spare_bits = nullptr;  // where a single-env launch parks its wavefront ballot
  bool is_reset = false;
  int trig = 0;          // 0 recurrence, 1 polynomial sincos per sub-step, 2 hardware trig per sub-step
  bool lds_table = false;
  bool custom_frames = false;  // obs_frame / ee_frame are not the reference's last two rows: RtTableF kernels
  int split = 0;          // step_split_kernel<..., L>: one env over L = 2 or 4 lanes (0 = one env per lane)
  int rollout_split = 0;  // the same choice for rollout_split_kernel (it pays up to larger batches: mt_create)
  int blocks_per_cu_override = -1;  // MT_BLOCKS_PER_CU: occupancy cap of the PF + TT step launches (engine.hip: step_blocks_per_cu)
  int64_t flat_from = 393216;  // launches of the prefetch kernel over at least this many envs take its FLAT form (kernels.h,
                               // LaneOffset<false>: the HBM-bound regime); MT_FLAT_FROM overrides
  bool prefetch_forced = false;
  bool prefetch = false;  // step_kernel<..., PF = kPrefetch>: target loads requested ahead of the kinematics
  bool goals_exposed = false;  // mt_device_ptr(MT_F_GOALS) was handed out: joint angles may change without the library knowing
  bool trig_steps = false;  // step kernels with TT: end-pose sines / cosines from the whole-degree table (static tables, sampled actions)
  int static_kind = 0;   // 0 runtime table, 1 Ref4Table, 2 Dh7Table
  mt_comm* comm = nullptr;
  // mt_gather_returns_begin / _wait: the exchange runs on a side stream from a snapshot of the row
  hipStream_t side_stream = nullptr;
  hipEvent_t ev_snap = nullptr, ev_g0 = nullptr, ev_g1 = nullptr;
  // The overlapped gather's snapshot is double-buffered: rows 0 / 1 of `snap` ([2][n]) alternate from exchange to exchange, so
  // that the row the NEXT mt_rollout's last launch writes (snap_next) is not the one the exchange still in flight reads.  A
  // row may be written once the exchange that last read it is KNOWN on the host to have finished (snap_free_known; asked with
  // hipEventQuery on ev_gdone[row], never waited for on the device); mt_gather_returns_begin keeps that true by letting the
  // host run at most one exchange ahead of the device (gather_throttle).
  float* snap = nullptr;
  hipEvent_t ev_gdone[2] = {nullptr, nullptr};
  bool snap_free_known[2] = {true, true};
  int snap_next = 0;
  bool gather_throttle = true;
  float* snap_row(int p) const { return snap + (size_t)p * (size_t)n; }
  void* reduce_scratch = nullptr;  // mt_reduce_returns: block partials + per-rank records, grown on demand
  size_t reduce_scratch_bytes = 0;
  bool reset_split = false;  // mt_reset_random / mt_reset_done of the whole batch: reset_split_kernel
  // mt_rollout on small batches: the segment's launches are captured once into a HIP graph and replayed
  struct RolloutGraph {
    int T;
    int chains;
    mt::StepArgs args;  // what the nodes were captured with (major = 0, episode0 = 0): any change rebuilds the graph
    hipGraphExec_t exec[kMaxChains];  // one graph per chain (a graph with parallel branches replays slower than
                                      // independent graphs on independent streams: profiles/r03_variants.md)
  };
  std::vector<RolloutGraph> graphs;
  std::vector<int> graph_seen;      // segment lengths asked for once: a graph is built at the second request
  uint32_t* graph_step0 = nullptr;  // device word: first step index of the segment being replayed
  int graph_mode = -1;              // -1 by batch size (<= graph_max envs), 0 never, 1 always (MT_GRAPH)
  int64_t graph_max = 131072;
  // mt_rollout (and mt_step / mt_sample_actions / mt_set_actions from device memory) as `chains` independent chains of
  // launches (contiguous env ranges on separate streams): a step of env i depends only on env i's previous step, so while
  // one range's kernel drains and its next one is dispatched the other ranges keep the chip busy
  // (tools/multistream_probe.py, profiles/r03_variants.md).  Forked lazily off the handle's stream by the first per-chain
  // call and -- on the handle's own stream -- left forked across calls (see `forked` below): only the MT_ENTER calls and
  // mt_sync join them; mt_device_ptr does not.
  int chains = 1;
  bool chains_forced = false;   // MT_CHAINS was given (the multi-step mt_rollout then runs per chain as well)
  int chain_split = 0;          // the step-kernel schedule of a chain's launches is picked for the CHAIN's env count
  bool chain_prefetch = false;
  int chain_rollout_split = 0;  // ... and the rollout-kernel schedule of a chain's multi-step launches
  int multi_k = 1;              // mt_rollout: steps per launch on small shards (rollout kernels), 1 = one launch per step
  bool rollout_early = true;    // ... with the rollout kernels' RPF prologue (first step under the state loads; static tables)
  // The episode boundary folded into mt_rollout's multi-step launches (kernels.h RolloutArgs):
  bool defer_reset_chained = false;  // ... and into the first launch of each chain of the chained launch-per-step form
  bool defer_reset = true;      // mt_reset_random on such a handle is DEFERRED into the first launch of the next mt_rollout
  bool reset_pending = false;   // ... one is waiting: every other entry point flushes it first (MT_ENTER, flush_pending_reset)
  uint64_t pend_seed = 0;
  uint32_t pend_episode = 0;
  bool snap_in_rollout = true;  // the last launch of an mt_rollout call also stores the returns to the gather's snapshot row
  bool snap_valid = false;      // ... it did, and nothing has touched the returns since: mt_gather_returns_begin skips its copy
  std::string overrides;        // the MT_* overrides choose_dispatch saw ("NAME=value,...")
  std::string describe;         // mt_describe_dispatch's text
  hipStream_t chain_streams[kMaxChains] = {nullptr, nullptr, nullptr, nullptr};  // [0] unused: chain 0 runs on `stream`
  hipEvent_t ev_fork = nullptr, ev_join[kMaxChains] = {nullptr, nullptr, nullptr, nullptr};
  // Chains stay forked ACROSS calls while the handle runs on its own stream: mt_rollout and mt_reset_random enqueue per
  // chain and return; every other entry point joins first (MT_ENTER), mt_sync included, so "sync before foreign reads"
  // keeps its meaning.  On a caller's stream (mt_set_stream) every call joins before it returns, as stream order demands.
  bool forked = false;
  bool lazy_chains = true;  // MT_LAZY_CHAINS=0: every per-chain call joins before it returns, resets are never per chain
  bool gather_pending = false;
  bool gather_inplace = false;  // the pending exchange reads an arena row directly (mt_gather_returns_begin_inplace): resets wait for it
  float last_gather_ms = 0.f;  // device time of the last exchange that was waited for (mt_gather_returns_wait / mt_sync)
  std::string err;
  std::string kernel_name;  // mt_step_kernel_name
};

namespace mt {

// Event flags.  A default HIP event performs a SYSTEM-scope release when it is recorded (cache write-back so that the host
// may inspect what ran before it): between two launches of a stream that is a bubble of 7-12 us (tools/episode_end_cost.py,
// profiles/r04_variants.md section 4: the overlapped gather's one event record cost the episode end of a 131 072-env
// shard 18 us).  The events this library records only order streams of ONE device against each other, or only time them;
// the launches themselves carry the device-scope fences that make a kernel's writes visible to the next one.  MT_EVENT_FENCE
// = 1 gives the default events back (A/B, tools/episode_end_cost.py).
inline unsigned event_flags(bool timing) {
  static const bool fenced = [] {
    const char* e = std::getenv("MT_EVENT_FENCE");
    return e && *e && std::atoi(e) != 0;
  }();
  return (timing ? 0u : (unsigned)hipEventDisableTiming) | (fenced ? 0u : (unsigned)hipEventDisableSystemFence);
}

// Records `msg` where mt_last_error will find it (on the handle, or per thread when there is none).
int fail(mt_handle h, int code, const std::string& msg);

// Makes `device` current for the duration of a call and restores the caller's device afterwards, so that a handle
// living on another GPU never changes what the calling thread (e.g. torch) considers current.
class DeviceGuard {
 public:
  explicit DeviceGuard(int device) {
    if (hipGetDevice(&prev_) != hipSuccess) prev_ = -1;
    err_ = (prev_ == device) ? hipSuccess : hipSetDevice(device);
    changed_ = (err_ == hipSuccess && prev_ != device);
  }
  ~DeviceGuard() {
    if (changed_ && prev_ >= 0) (void)hipSetDevice(prev_);
  }
  hipError_t error() const { return err_; }

 private:
  int prev_ = -1;
  hipError_t err_ = hipSuccess;
  bool changed_ = false;
};

inline dim3 grid_for(int64_t n) { return dim3((unsigned)((n + kBlock - 1) / kBlock)); }

// engine.hip: the handle's stream waits for everything the chain streams still carry (no-op when not forked)
int join_chains(mt_handle h);
// engine.hip: a full reset that mt_reset_random deferred into the next mt_rollout is launched now (no-op without one)
int flush_pending_reset(mt_handle h);

}  // namespace mt

#define MT_HIP(h, call)                                                                            \
  do {                                                                                             \
    hipError_t e__ = (call);                                                                       \
    if (e__ != hipSuccess)                                                                         \
      return mt::fail(h, MT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e__));          \
  } while (0)

#define MT_REQUIRE(h, cond, msg)                              \
  do {                                                        \
    if (!(cond)) return mt::fail(h, MT_ERR_INVALID_ARG, msg); \
  } while (0)

#define MT_ON_DEVICE(h, device)                                                                          \
  mt::DeviceGuard mt_guard__(device);                                                                    \
  if (mt_guard__.error() != hipSuccess)                                                                  \
  return mt::fail(h, MT_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(mt_guard__.error()))

// Entry of every call that touches the handle's device state as a whole: make the handle's device current and fold the
// chains back into the handle's stream, and launch a reset that was deferred into the next mt_rollout.  (mt_rollout and
// mt_reset_random, which work per chain and know about deferred resets, use MT_ON_DEVICE.)
#define MT_ENTER(h)                              \
  MT_ON_DEVICE(h, (h)->cfg.device);              \
  do {                                           \
    (h)->snap_valid = false;                     \
    if ((h)->forked) {                           \
      int rcj__ = mt::join_chains(h);            \
      if (rcj__ != MT_OK) return rcj__;          \
    }                                            \
    if ((h)->reset_pending) {                    \
      int rcf__ = mt::flush_pending_reset(h);    \
      if (rcf__ != MT_OK) return rcf__;          \
    }                                            \
  } while (0)

// The timers' entry: the join of MT_ENTER without the rest.  A deferred reset is work of whatever FOLLOWS the timer call --
// launching it here would put it in front of the region's start event, outside the timeline it belongs to.
#define MT_ENTER_TIMER(h)                        \
  MT_ON_DEVICE(h, (h)->cfg.device);              \
  do {                                           \
    if ((h)->forked) {                           \
      int rcj__ = mt::join_chains(h);            \
      if (rcj__ != MT_OK) return rcj__;          \
    }                                            \
  } while (0)

// comm.hip
void mt_comm_release(mt_handle h);
void mt_gather_release(mt_handle h);

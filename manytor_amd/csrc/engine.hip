// libmanytor_hip.so -- host side of the C ABI declared in include/manytor_hip.h.
//
// Owns the device arena (SoA rows in HBM), the stream, and the launch logic.
// No arithmetic of the step path happens here: everything the reference
// computes in manytor.py:17-53,141-260 runs in the kernels of kernels.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "engine_internal.h"
#include "kernels.h"

using namespace mt;

namespace {
thread_local std::string g_last_error;
}  // namespace

namespace mt {
int fail(mt_handle h, int code, const std::string& msg) {
  if (h)
    h->err = msg;
  else
    g_last_error = msg;
  return code;
}

int join_chains(mt_handle h) {
  if (!h->forked) return MT_OK;
  h->forked = false;
  for (int c = 1; c < mt_engine::kMaxChains; ++c) {
    if (!h->chain_streams[c] || !h->ev_join[c]) continue;
    MT_HIP(h, hipEventRecord(h->ev_join[c], h->chain_streams[c]));
    MT_HIP(h, hipStreamWaitEvent(h->stream, h->ev_join[c], 0));
  }
  return MT_OK;
}
}  // namespace mt

namespace {

DhConst make_dh(const float* table, int dof) {
  DhConst t{};
  for (int j = 0; j < MT_MAX_DOF; ++j) {
    t.ca[j] = 1.f;
  }
  for (int j = 0; j < dof; ++j) {
    const double a = table[4 * j + 0], alpha = table[4 * j + 1], d = table[4 * j + 2], off = table[4 * j + 3];
    t.a[j] = (float)a;
    t.d[j] = (float)d;
    // snap multiples of pi/2 so that e.g. cos(pi/2) is 0, not 6e-17 (the reference's residue)
    double sa = std::sin(alpha), ca = std::cos(alpha);
    if (std::fabs(sa) < 1e-7) sa = 0.0;
    if (std::fabs(ca) < 1e-7) ca = 0.0;
    t.sa[j] = (float)sa;
    t.ca[j] = (float)ca;
    t.off_deg[j] = (float)(off * 180.0 / M_PI);
  }
  t.fo = dof >= 2 ? dof - 2 : 0;  // the reference's rows (manytor.py:143, :162); mt_create overrides them from mt_config
  t.fe = dof - 1;
  return t;
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

int ensure_staging(mt_handle h, size_t bytes) {
  if (h->staging_bytes >= bytes) return MT_OK;
  if (h->staging) {
    MT_HIP(h, hipStreamSynchronize(h->stream));
    MT_HIP(h, hipFree(h->staging));
    h->staging = nullptr;
    h->staging_bytes = 0;
  }
  if (hipMalloc(&h->staging, bytes) != hipSuccess) return fail(h, MT_ERR_ALLOC, "hipMalloc(staging) failed");
  h->staging_bytes = bytes;
  return MT_OK;
}

// ---- kernel dispatch ---------------------------------------------------------------------------
// static_kind: 0 = runtime table (RtTable<D>), 1 = Ref4Table, 2 = Dh7Table
template <class Tbl>
bool table_matches(const DhConst& t, int dof) {
  if (dof != Tbl::D) return false;
  for (int j = 0; j < dof; ++j)
    if (t.a[j] != Tbl::a(j) || t.d[j] != Tbl::d(j) || t.sa[j] != Tbl::sa(j) || t.ca[j] != Tbl::ca(j) ||
        t.off_deg[j] != Tbl::off(j))
      return false;
  return true;
}

int match_static(const DhConst& t, int dof) {
  if (table_matches<Ref4Table>(t, dof)) return 1;
  if (table_matches<Dh7Table>(t, dof)) return 2;
  return 0;
}

// ---- the dispatch policy: ONE table of every size threshold the host picks a schedule by ------------------------------
// Every schedule of a kernel gives the same bits (tests/test_gpu_parity.py: test_step_kernel_variants_are_bit_identical,
// test_rollout_in_independent_chains_equals_plain_launches, ...), so this table is about time only.  The numbers come from
// the sweeps named beside them (tools/variant_sweep.py, size_sweep.py, chain_sweep.py, fused_split_sweep.py,
// ab_blocks_per_cu.sh; profiles/r02_variants.md, r03_variants.md, r04_variants.md).  choose_dispatch() below is the only
// reader; the MT_* environment variables override single entries for experiments and tests, and mt_describe_dispatch()
// reports the table, the overrides and what was resolved for a handle.
struct BlockCap {
  int64_t from, below;  // envs of the launch
  int blocks;           // resident blocks per CU
};
struct DispatchPolicy {
  // step_split_kernel: one env over 4 / 2 lanes while the batch is so small that a launch is one wave's dependent chain
  //   <= 49 152 arms: 4.5-5.5 us per step against 5.3-5.9 with prefetch (49 152: 5.47 against 5.70 with 2 lanes);
  //   <= 65 536: 5.8 against 6.0 (4 lanes: 6.3)                                  [profiles/r03_variant_sweep_final.json]
  int64_t step_split4_max = 49152, step_split2_max = 65536;
  // prefetch (PF = 8) for arms other than the reference arm: up to here (long arms lose more to the lower occupancy than
  // they gain); the reference arm takes it at every size (7.1 against 7.8 us at 131 072, 19.9 against 21.0 at 524 288)
  int64_t prefetch_other_max = 131072;
  // rollout_split_kernel (fused rollout): 4 lanes per env up to 32 768 arms, 2 up to 131 072 (3.60 against 4.50 us per step
  // at 98 304 arms, 4.37 against 4.50 at 131 072; one env per lane wins from 196 608 on)   [r03_fused_split_sweep.json]
  int64_t fused_split4_max = 32768, fused_split2_max = 131072;
  // reset_split_kernel: the target draw over 4 lanes per env (11.2 -> 6.9 us at 16 384 arms, 12.1 -> 10.0 at 65 536, a tie
  // at 131 072, slower above)
  int64_t reset_split_max = 65536;
  // mt_rollout replays a cached HIP graph of its launches up to here (-3..6 % per step; nothing above, where the kernels
  // are the time)                                                                        [r02_graph_replay.json]
  int64_t graph_max = 131072;
  // mt_rollout / mt_step as two chains of launches on two streams (us per step incl. the per-episode reset, one chain ->
  // two: 163 840 envs 8.45 -> 7.15, 262 144: 10.5 -> 8.7, 524 288: 21.5 -> 19.8, 1 M: 39.9 -> 36.3, 2 M: 74.6 -> 69.0;
  // nothing at 4 M; at 98 304-131 072 two chains win back to back but lose from an idle device)     [r03_variants.md 2]
  int64_t chains_min = 163840, chains_max = 3145728;
  // a chain is a batch of span envs, but two kernels are in flight, so the lane-split schedules stop paying earlier
  // (2 x 65 536 envs: 6.85 us with one env per lane against 7.0-7.35 with two lanes; 2 x 49 152: 6.8 with two against 6.95)
  int64_t chain_split4_max = 16384, chain_split2_max = 49152, chain_prefetch_other_max = 131072;
  // launches of the prefetch kernel over at least this many envs take LaneOffset<false> (kernels.h FLAT: the HBM-bound
  // regime, 0.7-2.4 % faster there)                                                      [r03_ab_flat_from.txt]
  int64_t flat_from = 393216;
  // resident blocks per CU of a CHAIN launch of the sampled-action PF + TT kernel (74-76 VGPRs: six fit), keyed by the
  // envs of the launch with two chains in mind (us per step of the batch, no cap -> cap: 327 680 arms 12.3 -> 11.6 (3),
  // 393 216: 14.4 -> 13.3 (3), 524 288: 19.1 -> 18.0 (4), 786 432: 27.4 -> 26.1 (4), 1 048 576: 35.2 -> 34.5 (5); nothing
  // at >= 1 179 648 and <= 262 144 arms)                                                 [r03_variants.md 12]
  BlockCap block_caps[3] = {{147456, 229376, 3}, {229376, 458752, 4}, {458752, 589824, 5}};
  // mt_rollout(T) exposes the state after T steps and the outputs of the last one, so on small shards -- where a launch
  // per step is a kernel boundary plus a re-fetch of the whole state past L2 for 4-6 us of work -- it runs
  // `multi_step_k` steps per launch through the rollout kernels (state in registers / LDS between them; every step
  // still writes its obs / reward / done / ee), bit-identical by construction             [r04_variants.md 1]
  //   us per step, 50-step segments from an idle device (tools/rollout_k_sweep.py, profiles/r04_rollout_k_sweep*.json):
  //                 one launch per step (default of round 3)      k = 4     k = 5     k = 8     k = 50
  //      65 536 envs   5.71 (graph replay)                         4.18      3.99      3.74      3.04
  //      98 304        6.48 (graph replay)                         4.96      4.70      4.35
  //     131 072        6.47 (graph replay)                         5.79      5.50      5.10      4.42
  //     163 840        7.43 (two chains of step kernels)           6.59      6.26      5.88
  //     196 608        8.32 (two chains)                           6.74      6.35      5.97
  //     262 144        8.09 (two chains)                           8.03      7.64      7.27
  //   (k = 2 loses at 131 072: a rollout-kernel launch costs 4-6 us before its first step's outputs.)  The multi-step
  //   launches run on the handle's stream alone: in two chains they are faster back to back at 131 072-163 840 envs
  //   (4.78 / 5.88 us) but slower from an idle device over a 20-step segment (5.83 / 6.81) and slower at >= 196 608.
  int64_t multi_step_max = 262144;
  int multi_step_k = 5;
  // the chained launch-per-step form of the large batches starts an episode with ONE step of the rollout kernel that carries
  // the deferred reset as its prologue
  bool fold_reset_into_chains = false;
  // the angle-addition recurrence is run (S - 1) / 2 rotations from each end of a route; beyond this many the per-pose
  // polynomial sincos kernels take over (drift: 12 rotations 3.3e-5 / 6.6e-5 in z on the 4- / 7-joint arm, 31 rotations
  // 8.6e-5 / 1.3e-4, outside the 1e-4 position tolerance)                                [tests/test_gpu_zmin.py]
  int max_recurrence_rotations = 12;
  // rows of more than this many envs get a 1 KiB pad (rows an exact power of two apart alias onto the same channels)
  int64_t ld_pad_above = 16384;
  int ld_pad_floats = 256;
};
static const DispatchPolicy kPolicy{};

// Blocks per CU for a chain launch of the sampled-action prefetch kernel of a compile-time table.  The cap is applied at
// launch time the usual way, with dynamic LDS nobody touches: a block then claims 1 / cap of the CU's 160 KB.
// MT_BLOCKS_PER_CU overrides (0 = none; 1 and 2 cannot be realised inside the default 64 KB dynamic-LDS limit and are
// raised to 3).
static int step_blocks_per_cu(mt_handle h, int64_t n_launch) {
  if (h->blocks_per_cu_override >= 0) return h->blocks_per_cu_override;
  if (n_launch >= h->n) return 0;  // chain launches only
  for (const BlockCap& c : kPolicy.block_caps)
    if (n_launch >= c.from && n_launch < c.below) return c.blocks;
  return 0;
}
static size_t lds_pad_for_blocks(int blocks_per_cu, size_t static_lds) {
  if (blocks_per_cu <= 0) return 0;
  const size_t per_block = (size_t)163840 / (size_t)blocks_per_cu - 1024;  // safely inside the bracket of that block count
  return per_block > static_lds ? std::min<size_t>(per_block, 65536) - static_lds : 0;  // <= 64 KB in all: no attribute needed
}

// Envs per chain: equal shares rounded up to whole 256-env blocks; the last chain takes what is left (possibly less).
int64_t chain_span(mt_handle h, int chains) {
  const int64_t per = (h->n + chains - 1) / chains;
  return (per + 255) / 256 * 256;
}

static bool env_int(const char* name, long long* v) {  // set AND non-empty (atoi("") would read as 0)
  const char* e = std::getenv(name);
  if (!e || !*e) return false;
  *v = std::atoll(e);
  return true;
}

// Every schedule choice of a handle, from kPolicy, the handle's size / table / flags (h->cfg, h->args.dh must be set) and
// the MT_* overrides.  Called once by mt_create.  The overrides that were seen are kept in h->overrides for
// mt_describe_dispatch.
static void choose_dispatch(mt_handle h) {
  const mt_config* cfg = &h->cfg;
  const DispatchPolicy& P = kPolicy;
  const int64_t n = cfg->n_envs;
  long long v = 0;
  h->overrides.clear();
  auto seen = [&](const char* name) { h->overrides += (h->overrides.empty() ? "" : ",") + std::string(name) + "=" + std::getenv(name); };

  h->trig = (cfg->flags & MT_FLAG_HW_TRIG) ? 2 : ((cfg->flags & MT_FLAG_DIRECT_TRIG) ? 1 : 0);
  if (cfg->flags & MT_FLAG_ABLATE_LOOP) h->trig = (cfg->flags & MT_FLAG_ABLATE_OBS) ? 4 : 3;
  else if (cfg->flags & MT_FLAG_ABLATE_OBS) h->trig = 5;   /* OBS alone = arithmetic-only build */
  // Routes with more rotations per half than the recurrence tolerates take the per-pose polynomial sincos (the
  // MT_FLAG_DIRECT_TRIG kernels: no drift, about twice the arithmetic).
  const bool long_route = h->trig == 0 && (cfg->substeps - 1) / 2 > P.max_recurrence_rotations;
  if (long_route) h->trig = 1;
  h->lds_table = (cfg->flags & MT_FLAG_DH_IN_LDS) != 0 && !long_route;  // the LDS-table variant exists for the recurrence only
  // other rows than the reference's last two: the runtime-table kernel with runtime frames (RtTableF), one env per lane
  h->custom_frames = !(h->args.dh.fo == h->D - 2 && h->args.dh.fe == h->D - 1);
  h->static_kind = ((cfg->flags & (MT_FLAG_NO_SPECIALIZE | MT_FLAG_DH_IN_LDS)) || h->custom_frames) ? 0 : match_static(h->args.dh, h->D);

  // ---- one launch per step: which step kernel for which batch (all variants give the same bits) ----
  h->split = n <= P.step_split4_max ? 4 : (n <= P.step_split2_max ? 2 : 0);
  h->prefetch = h->static_kind == 1 || n <= P.prefetch_other_max;
  h->rollout_split = n <= P.fused_split4_max ? 4 : (n <= P.fused_split2_max ? 2 : 0);
  bool split_forced = false;
  if (env_int("MT_SPLIT", &v)) {
    h->split = (v == 2 || v == 4) ? (int)v : 0;
    h->rollout_split = h->split;
    split_forced = true;
    seen("MT_SPLIT");
  }
  h->prefetch_forced = false;
  if (env_int("MT_PREFETCH", &v)) {
    h->prefetch = v != 0;
    h->prefetch_forced = true;
    seen("MT_PREFETCH");
  }
  h->reset_split = n <= P.reset_split_max;
  if (env_int("MT_RESET_SPLIT", &v)) {
    h->reset_split = v != 0;
    seen("MT_RESET_SPLIT");
  }
  h->graph_mode = -1;
  h->graph_max = P.graph_max;
  if (env_int("MT_GRAPH", &v)) {
    h->graph_mode = v != 0 ? 1 : 0;
    seen("MT_GRAPH");
  }
  h->flat_from = P.flat_from;
  if (env_int("MT_FLAT_FROM", &v)) {
    h->flat_from = std::max<long long>(0, v);
    seen("MT_FLAT_FROM");
  }
  h->blocks_per_cu_override = -1;
  if (env_int("MT_BLOCKS_PER_CU", &v)) {
    // 1 and 2 blocks per CU would need more than the default 64 KB of dynamic LDS per block: the smallest cap the pad can
    // realise is 3, and the kernel-name string reports what is really applied
    const int c = (int)std::max<long long>(0, std::min<long long>(8, v));
    h->blocks_per_cu_override = (c == 1 || c == 2) ? 3 : c;
    seen("MT_BLOCKS_PER_CU");
  }
  // End-pose sines / cosines from the whole-degree table (step kernels with TT; profiles/r03_variants.md section 3)
  h->trig_steps = h->static_kind != 0;
  if (env_int("MT_TRIG_TABLE", &v)) {
    h->trig_steps = v != 0 && h->static_kind != 0;
    seen("MT_TRIG_TABLE");
  }

  // ---- mt_rollout / mt_step as independent chains of launches on separate streams (engine_internal.h) ----
  h->chains = (n >= P.chains_min && n <= P.chains_max) ? 2 : 1;
  h->chains_forced = false;
  if (env_int("MT_CHAINS", &v)) {
    h->chains = (int)std::max<long long>(1, std::min<long long>(mt_engine::kMaxChains, v));
    h->chains_forced = true;
    seen("MT_CHAINS");
  }
  if (n < 2 * 256) h->chains = 1;
  h->lazy_chains = true;
  if (env_int("MT_LAZY_CHAINS", &v)) {  // 0: join at the end of every call
    h->lazy_chains = v != 0;
    seen("MT_LAZY_CHAINS");
  }

  // ---- mt_rollout: several steps per launch on small shards (the rollout kernels; default trigonometry only) ----
  h->multi_k = (n <= P.multi_step_max) ? P.multi_step_k : 1;
  if (env_int("MT_ROLLOUT_K", &v)) {
    h->multi_k = (int)std::max<long long>(1, std::min<long long>(64, v));
    seen("MT_ROLLOUT_K");
  }
  // ... which absorb the episode boundary: a deferred full reset as the first launch's prologue, the gather's snapshot as
  // the last launch's epilogue (kernels.h RolloutArgs)
  h->defer_reset = true;
  if (env_int("MT_DEFER_RESET", &v)) {
    h->defer_reset = v != 0;
    seen("MT_DEFER_RESET");
  }
  // ... and, for the chained launch-per-step form of the large batches, as a one-step launch of the rollout kernel in front
  // of each chain's step launches (profiles/r04_variants.md section 4)
  h->defer_reset_chained = P.fold_reset_into_chains;
  if (env_int("MT_DEFER_RESET_CHAINS", &v)) {
    h->defer_reset_chained = v != 0;
    seen("MT_DEFER_RESET_CHAINS");
  }
  h->gather_throttle = true;
  if (env_int("MT_GATHER_THROTTLE", &v)) {  // 0: mt_gather_returns_begin never waits on the host (a snapshot launch instead)
    h->gather_throttle = v != 0;
    seen("MT_GATHER_THROTTLE");
  }
  h->snap_in_rollout = true;
  if (env_int("MT_ROLLOUT_SNAP", &v)) {
    h->snap_in_rollout = v != 0;
    seen("MT_ROLLOUT_SNAP");
  }
  // ... whose launches take the prologue that runs the first step under the state loads (kernels.h RPF; static tables)
  h->rollout_early = true;
  if (env_int("MT_ROLLOUT_EARLY", &v)) {
    h->rollout_early = v != 0;
    seen("MT_ROLLOUT_EARLY");
  }

  if (h->custom_frames) {
    h->lds_table = false;
    h->split = 0;
    h->rollout_split = 0;
    h->prefetch = false;
  }
  // A chain of a multi-chain call is a batch of chain_span() envs: its launches use the schedule for THAT size
  h->chain_split = h->split;
  h->chain_prefetch = h->prefetch;
  h->chain_rollout_split = h->rollout_split;
  if (h->chains > 1) {
    const int64_t span = chain_span(h, h->chains);
    h->chain_split = span <= P.chain_split4_max ? 4 : (span <= P.chain_split2_max ? 2 : 0);
    h->chain_prefetch = h->static_kind == 1 || span <= P.chain_prefetch_other_max;
    h->chain_rollout_split = span <= P.fused_split4_max ? 4 : (span <= P.fused_split2_max ? 2 : 0);
    if (split_forced) h->chain_split = h->chain_rollout_split = h->split;
    if (h->prefetch_forced) h->chain_prefetch = h->prefetch;
    if (h->custom_frames) {
      h->chain_split = 0;
      h->chain_rollout_split = 0;
      h->chain_prefetch = false;
    }
  }
}

// whole_rows: the launch covers the batch or a 256-aligned range of it (not the single-env view), so threads past the
// end of the range may read on to the end of their block inside the rows (what the TT kernels do before their barrier)
template <class Tbl, bool LDS_OK>
void launch_step_t(mt_handle h, const StepArgs& args, bool sample, bool whole_rows) {
  const dim3 g = grid_for(args.n), b(kBlock);
#define MT_LAUNCH_STEP(SAMPLE_, TRIG_, LDS_) \
  hipLaunchKernelGGL((step_kernel<Tbl, SAMPLE_, TRIG_, LDS_>), g, b, 0, h->stream, args)
  if (LDS_OK && h->lds_table) {
    if constexpr (LDS_OK) {
      if (sample) MT_LAUNCH_STEP(true, 0, true); else MT_LAUNCH_STEP(false, 0, true);
    }
    return;
  }
  // sampled actions of a compile-time table on the whole batch: both end poses' sines / cosines from the table (TT)
  bool tt = false;
  if constexpr (ActionTrigTable<Tbl>::value) tt = sample && h->trig == 0 && h->trig_steps && whole_rows;
  if (h->trig == 0 && h->split) {  // tiny batches: one env over 2 or 4 lanes (kernels.h, step_split_kernel)
    const int64_t per_block = kBlock / h->split;
    const dim3 gs((unsigned)((args.n + per_block - 1) / per_block));
    if constexpr (ActionTrigTable<Tbl>::value) {
      if (tt) {
        if (h->split == 2)
          hipLaunchKernelGGL((step_split_kernel<Tbl, true, 2, true>), gs, b, 0, h->stream, args);
        else
          hipLaunchKernelGGL((step_split_kernel<Tbl, true, 4, true>), gs, b, 0, h->stream, args);
        return;
      }
    }
    if (h->split == 2) {
      if (sample)
        hipLaunchKernelGGL((step_split_kernel<Tbl, true, 2>), gs, b, 0, h->stream, args);
      else
        hipLaunchKernelGGL((step_split_kernel<Tbl, false, 2>), gs, b, 0, h->stream, args);
    } else {
      if (sample)
        hipLaunchKernelGGL((step_split_kernel<Tbl, true, 4>), gs, b, 0, h->stream, args);
      else
        hipLaunchKernelGGL((step_split_kernel<Tbl, false, 4>), gs, b, 0, h->stream, args);
    }
    return;
  }
  if constexpr (ActionTrigTable<Tbl>::value) {
    if (tt) {
      if (h->prefetch && args.n >= h->flat_from)  // HBM-bound launches: kernels.h, LaneOffset<false>; see step_blocks_per_cu
        hipLaunchKernelGGL((step_kernel<Tbl, true, 0, false, kPrefetch, true, true>), g, b,
                           lds_pad_for_blocks(step_blocks_per_cu(h, args.n), kTrigEntries * sizeof(SinCos)), h->stream, args);
      else if (h->prefetch)
        hipLaunchKernelGGL((step_kernel<Tbl, true, 0, false, kPrefetch, true>), g, b,
                           lds_pad_for_blocks(step_blocks_per_cu(h, args.n), kTrigEntries * sizeof(SinCos)), h->stream, args);
      else
        hipLaunchKernelGGL((step_kernel<Tbl, true, 0, false, 0, true>), g, b, 0, h->stream, args);
      return;
    }
  }
  if (h->trig == 0 && h->prefetch) {  // target loads in flight before the kinematics (kernels.h, PF)
    if (args.n >= h->flat_from) {
      if (sample)
        hipLaunchKernelGGL((step_kernel<Tbl, true, 0, false, kPrefetch, false, true>), g, b, 0, h->stream, args);
      else
        hipLaunchKernelGGL((step_kernel<Tbl, false, 0, false, kPrefetch, false, true>), g, b, 0, h->stream, args);
    } else if (sample) {
      hipLaunchKernelGGL((step_kernel<Tbl, true, 0, false, kPrefetch>), g, b, 0, h->stream, args);
    } else {
      hipLaunchKernelGGL((step_kernel<Tbl, false, 0, false, kPrefetch>), g, b, 0, h->stream, args);
    }
    return;
  }
  switch (h->trig) {
    case 1: if (sample) MT_LAUNCH_STEP(true, 1, false); else MT_LAUNCH_STEP(false, 1, false); break;
    case 2: if (sample) MT_LAUNCH_STEP(true, 2, false); else MT_LAUNCH_STEP(false, 2, false); break;
    case 3: if (sample) MT_LAUNCH_STEP(true, 3, false); else MT_LAUNCH_STEP(false, 3, false); break;
    case 4: if (sample) MT_LAUNCH_STEP(true, 4, false); else MT_LAUNCH_STEP(false, 4, false); break;
    case 5: if (sample) MT_LAUNCH_STEP(true, 5, false); else MT_LAUNCH_STEP(false, 5, false); break;
    default: if (sample) MT_LAUNCH_STEP(true, 0, false); else MT_LAUNCH_STEP(false, 0, false); break;
  }
#undef MT_LAUNCH_STEP
}

// The first step of an episode with the deferred reset as the step kernel's own prologue (kernels.h, FRESH): the sampled-
// action kernels of the static tables, one env per lane.  false = no such kernel for this handle's schedule (the caller
// falls back to one step of the rollout kernel).  `args` carries reset_seed / reset_episode / radius.
template <class Tbl>
bool launch_step_fresh_t(mt_handle h, const StepArgs& args) {
  if constexpr (!ActionTrigTable<Tbl>::value) {
    return false;
  } else {
    if (h->trig != 0 || !h->trig_steps || h->split || h->lds_table || h->trace) return false;
    const size_t stage = (size_t)3 * args.K * kBlock * sizeof(float), fixed = kTrigEntries * sizeof(SinCos) + 512;
    if (stage + fixed > 65536) return false;  // (K > 20: above the default dynamic-LDS limit)
    const dim3 g = grid_for(args.n), b(kBlock);
    const size_t lds = std::max(stage, h->prefetch ? lds_pad_for_blocks(step_blocks_per_cu(h, args.n), fixed) : (size_t)0);
    if (h->prefetch && args.n >= h->flat_from)
      hipLaunchKernelGGL((step_kernel<Tbl, true, 0, false, kPrefetch, true, true, true>), g, b, lds, h->stream, args);
    else if (h->prefetch)
      hipLaunchKernelGGL((step_kernel<Tbl, true, 0, false, kPrefetch, true, false, true>), g, b, lds, h->stream, args);
    else
      hipLaunchKernelGGL((step_kernel<Tbl, true, 0, false, 0, true, false, true>), g, b, lds, h->stream, args);
    return true;
  }
}
bool launch_step_fresh(mt_handle h, const StepArgs& args) {
  if (h->custom_frames) return false;
  if (h->static_kind == 1) return launch_step_fresh_t<Ref4Table>(h, args);
  if (h->static_kind == 2) return launch_step_fresh_t<Dh7Table>(h, args);
  return false;
}

template <int D>
void launch_step_d(mt_handle h, const StepArgs& args, bool sample, bool whole_rows) {
  launch_step_t<RtTable<D>, true>(h, args, sample, whole_rows);
}

template <int D>
void launch_trace_d(mt_handle h, const StepArgs& args, float* trace, bool sample) {
  if (sample)
    hipLaunchKernelGGL((trace_kernel<D, true>), grid_for(args.n), dim3(kBlock), 0, h->stream, args, trace);
  else
    hipLaunchKernelGGL((trace_kernel<D, false>), grid_for(args.n), dim3(kBlock), 0, h->stream, args, trace);
}

#define MT_DISPATCH_D(D_, FN, ...)   \
  switch (D_) {                      \
    case 2: FN<2>(__VA_ARGS__); break; \
    case 3: FN<3>(__VA_ARGS__); break; \
    case 4: FN<4>(__VA_ARGS__); break; \
    case 5: FN<5>(__VA_ARGS__); break; \
    case 6: FN<6>(__VA_ARGS__); break; \
    case 7: FN<7>(__VA_ARGS__); break; \
    case 8: FN<8>(__VA_ARGS__); break; \
    default: break;                  \
  }

template <int D>
void launch_step_frames_d(mt_handle h, const StepArgs& args, bool sample) {
  const dim3 g = grid_for(args.n), b(kBlock);
  if (h->trig == 1) {  // long routes (mt_create: more than 12 rotations per half) or MT_FLAG_DIRECT_TRIG
    if (sample)
      hipLaunchKernelGGL((step_kernel<RtTableF<D>, true, 1, false, 0>), g, b, 0, h->stream, args);
    else
      hipLaunchKernelGGL((step_kernel<RtTableF<D>, false, 1, false, 0>), g, b, 0, h->stream, args);
    return;
  }
  if (sample)
    hipLaunchKernelGGL((step_kernel<RtTableF<D>, true, 0, false, 0>), g, b, 0, h->stream, args);
  else
    hipLaunchKernelGGL((step_kernel<RtTableF<D>, false, 0, false, 0>), g, b, 0, h->stream, args);
}

// One env step of the envs `args` describes (the whole batch, or one env of it: args_for_env); `trace` is the
// matching view of the sub-step trace buffer or NULL.
void launch_step(mt_handle h, const StepArgs& args, float* trace, bool sample, bool whole_rows) {
  if (trace) MT_DISPATCH_D(h->D, launch_trace_d, h, args, trace, sample);  // first: it needs the previous pose
  if (h->custom_frames) {
    MT_DISPATCH_D(h->D, launch_step_frames_d, h, args, sample);
    return;
  }
  if (h->static_kind == 1) return launch_step_t<Ref4Table, false>(h, args, sample, whole_rows);
  if (h->static_kind == 2) return launch_step_t<Dh7Table, false>(h, args, sample, whole_rows);
  MT_DISPATCH_D(h->D, launch_step_d, h, args, sample, whole_rows);
}

void launch_step(mt_handle h, bool sample) { launch_step(h, h->args, h->trace, sample, true); }

// ---- multi-chain rollouts ------------------------------------------------------------------------------------------
StepArgs args_for_range(mt_handle h, const StepArgs& base, int64_t off, int64_t cnt);
int check_launch(mt_handle h, const char* what);
int ensure_chains(mt_handle h, int chains) {
  if (!h->ev_fork) MT_HIP(h, hipEventCreateWithFlags(&h->ev_fork, event_flags(false)));
  for (int c = 1; c < chains; ++c) {
    if (!h->chain_streams[c]) MT_HIP(h, hipStreamCreateWithFlags(&h->chain_streams[c], hipStreamNonBlocking));
    if (!h->ev_join[c]) MT_HIP(h, hipEventCreateWithFlags(&h->ev_join[c], event_flags(false)));
  }
  return MT_OK;
}

// One chain's T steps on the handle's CURRENT stream (h->stream): chain c owns envs [c * span, (c + 1) * span).  The
// step-kernel schedule is the one for the chain's env count (a chain is a batch of `span` envs).  `a0` carries seed /
// major_base; step s is launched with major = major0 + s.  sample = false: staged actions (mt_step), T = 1.
void launch_chain(mt_handle h, const StepArgs& a0, uint32_t major0, int T, int chains, int c, bool sample = true) {
  const int64_t span = chain_span(h, chains), off = (int64_t)c * span;
  if (off >= h->n) return;
  StepArgs as = args_for_range(h, a0, off, std::min(span, h->n - off));
  const int keep_split = h->split;
  const bool keep_pf = h->prefetch;
  h->split = h->chain_split;
  h->prefetch = h->chain_prefetch;
  for (int s = 0; s < T; ++s) {
    as.major = major0 + (uint32_t)s;
    launch_step(h, as, nullptr, sample, true);
  }
  h->split = keep_split;
  h->prefetch = keep_pf;
}

// Chain c's FIRST step of an episode, the deferred reset inside it (launch_step_fresh); false: not served.
bool launch_chain_fresh(mt_handle h, const StepArgs& a0, uint32_t major, int chains, int c) {
  const int64_t span = chain_span(h, chains), off = (int64_t)c * span;
  if (off >= h->n) return true;
  StepArgs as = args_for_range(h, a0, off, std::min(span, h->n - off));
  as.major = major;
  const int keep_split = h->split;
  const bool keep_pf = h->prefetch;
  h->split = h->chain_split;
  h->prefetch = h->chain_prefetch;
  const bool served = launch_step_fresh(h, as);
  h->split = keep_split;
  h->prefetch = keep_pf;
  return served;
}

// Chain c's stream (chain 0 runs on the handle's own stream).
hipStream_t chain_stream(mt_handle h, int c) { return c == 0 ? h->stream : h->chain_streams[c]; }

// Start the chains off the handle's stream: everything queued on it so far is ahead of what the chains will run.
int fork_chains(mt_handle h, int chains) {
  if (h->forked) return MT_OK;
  int rc = ensure_chains(h, chains);
  if (rc) return rc;
  const int64_t span = chain_span(h, chains);
  MT_HIP(h, hipEventRecord(h->ev_fork, h->stream));
  for (int c = 1; c < chains; ++c)
    if ((int64_t)c * span < h->n) MT_HIP(h, hipStreamWaitEvent(h->chain_streams[c], h->ev_fork, 0));
  h->forked = true;
  return MT_OK;
}

// After a per-chain call: on the handle's private stream the chains may stay forked (the next per-chain call continues
// them, any other call joins: MT_ENTER); on a caller's stream the call has to be complete in stream order when it returns.
int settle_chains(mt_handle h) { return (h->lazy_chains && h->stream == h->own_stream) ? MT_OK : join_chains(h); }

// The view of ONE env of the batch: every row base moved `env` elements to the right, n = 1.  The wavefront ballot
// of such a launch goes to a spare word; the real done_bits word is rebuilt afterwards (done_bits_word_kernel).
StepArgs args_for_env(mt_handle h, int64_t env) {
  StepArgs a = h->args;
  a.actions += env;
  a.goals += env;
  a.points += env;
  a.alive += env;
  a.total_reward += env;
  a.obs += env;
  a.reward += env;
  a.done += env;
  a.done_bits = h->spare_bits;
  a.ee += env;
  a.episodes += env;
  a.last_return += env;
  if (a.ring) a.ring += env;
  if (a.zmin) a.zmin += env;
  a.snap = nullptr;
  a.n = 1;
  a.env_base += env;
  return a;
}

// The view of a contiguous RANGE of envs [off, off + cnt) of the batch, off a multiple of 256 (whole blocks, whole ballot
// words, 1 KiB-aligned row segments): what one chain of a multi-chain mt_rollout launches on.
StepArgs args_for_range(mt_handle h, const StepArgs& base, int64_t off, int64_t cnt) {
  StepArgs a = base;
  a.actions += off;
  a.goals += off;
  a.points += off;
  a.alive += off;
  a.total_reward += off;
  a.obs += off;
  a.reward += off;
  a.done += off;
  a.done_bits += off / 64;
  a.ee += off;
  a.episodes += off;
  a.last_return += off;
  if (a.ring) a.ring += off;
  if (a.zmin) a.zmin += off;
  if (a.snap) a.snap += off;
  a.n = cnt;
  a.env_base += off;
  return a;
}

template <int D, bool RANDOM, bool ONLY_DONE>
void launch_reset_k(mt_handle h, const StepArgs& args) {
  const size_t lds = RANDOM ? (size_t)3 * args.K * kBlock * sizeof(float) : 0;  // staging columns of the drawn targets
  if (lds > 65536)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&reset_kernel<D, RANDOM, ONLY_DONE>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((reset_kernel<D, RANDOM, ONLY_DONE>), grid_for(args.n), dim3(kBlock), lds, h->stream, args,
                     h->cfg.radius);
}
template <int D, bool ONLY_DONE>
void launch_reset_split(mt_handle h, const StepArgs& args) {  // whole batch, small: 4 lanes per env for the draw
  constexpr int L = 4;
  const int64_t per_block = kBlock / L;
  const size_t lds = (size_t)3 * args.K * per_block * sizeof(float);
  hipLaunchKernelGGL((reset_split_kernel<D, ONLY_DONE, L>), dim3((unsigned)((args.n + per_block - 1) / per_block)),
                     dim3(kBlock), lds, h->stream, args, h->cfg.radius);
}
template <int D>
void launch_reset_d(mt_handle h, const StepArgs& args, int mode) {  // 0 given points, 1 random, 2 random only-done
  if (mode != 0 && h->reset_split && args.n == h->n)
    return mode == 1 ? launch_reset_split<D, false>(h, args) : launch_reset_split<D, true>(h, args);
  if (mode == 0)
    launch_reset_k<D, false, false>(h, args);
  else if (mode == 1)
    launch_reset_k<D, true, false>(h, args);
  else
    launch_reset_k<D, true, true>(h, args);
}

template <int D>
void launch_observe_d(mt_handle h) {
  hipLaunchKernelGGL((observe_kernel<D>), grid_for(h->n), dim3(kBlock), 0, h->stream, h->args);
}
template <int D>
void launch_check_done_d(mt_handle h) {
  hipLaunchKernelGGL((check_done_kernel<D>), grid_for(h->n), dim3(kBlock), 0, h->stream, h->args);
}
template <int D>
void launch_joints_d(mt_handle h, float* out) {
  hipLaunchKernelGGL((joints_kernel<D>), grid_for(h->n), dim3(kBlock), 0, h->stream, h->args, out);
}

// The rollout kernels on the envs `a` describes (the whole batch or a chain's 256-aligned range), `split` lanes per env.
// RPF (kernels.h): the prologue that requests everything up front and runs the first step under the loads -- for the short
// launches of mt_rollout's multi-step form, compile-time tables only (the long fused launches keep RPF = 0).
template <class Tbl, int L, int RPF>
void launch_rollout_split_t(mt_handle h, const StepArgs& a, const RolloutArgs& r) {
  const int64_t per_block = kBlock / L;
  const size_t lds = (size_t)3 * h->K * per_block * sizeof(float) + (ActionTrigTable<Tbl>::value ? kTrigEntries * sizeof(SinCos) : 0);
  if (lds > 65536)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rollout_split_kernel<Tbl, L, RPF>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((rollout_split_kernel<Tbl, L, RPF>), dim3((unsigned)((a.n + per_block - 1) / per_block)), dim3(kBlock), lds,
                     h->stream, a, r);
}

template <class Tbl, int RPF>
void launch_rollout_t(mt_handle h, const StepArgs& a, int split, const RolloutArgs& r) {
  if (split == 4) return launch_rollout_split_t<Tbl, 4, RPF>(h, a, r);
  if (split == 2) return launch_rollout_split_t<Tbl, 2, RPF>(h, a, r);
  // 21.5 KB at K = 7, 96 KB at K = 32 (of 160 KB), + 3.6 KB for the action sin / cos table of the compile-time tables
  const size_t lds = (size_t)3 * h->K * kBlock * sizeof(float) + (ActionTrigTable<Tbl>::value ? kTrigEntries * sizeof(SinCos) : 0);
  if (lds > 65536)  // above the default dynamic-LDS limit the kernel has to be told
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rollout_kernel<Tbl, RPF>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((rollout_kernel<Tbl, RPF>), grid_for(a.n), dim3(kBlock), lds, h->stream, a, r);
}

void launch_rollout(mt_handle h, const StepArgs& a, int split, const RolloutArgs& r, bool early = false) {
  if (h->static_kind == 1) return early ? launch_rollout_t<Ref4Table, kPrefetch>(h, a, split, r) : launch_rollout_t<Ref4Table, 0>(h, a, split, r);
  if (h->static_kind == 2) return early ? launch_rollout_t<Dh7Table, kPrefetch>(h, a, split, r) : launch_rollout_t<Dh7Table, 0>(h, a, split, r);
  switch (h->D) {
    case 2: launch_rollout_t<RtTable<2>, 0>(h, a, split, r); break;
    case 3: launch_rollout_t<RtTable<3>, 0>(h, a, split, r); break;
    case 4: launch_rollout_t<RtTable<4>, 0>(h, a, split, r); break;
    case 5: launch_rollout_t<RtTable<5>, 0>(h, a, split, r); break;
    case 6: launch_rollout_t<RtTable<6>, 0>(h, a, split, r); break;
    case 7: launch_rollout_t<RtTable<7>, 0>(h, a, split, r); break;
    default: launch_rollout_t<RtTable<8>, 0>(h, a, split, r); break;
  }
}

// The rollout kernels implement the default trigonometry and the reference's frame rows only.
bool fusable(mt_handle h) { return h->trig == 0 && !h->lds_table && !h->trace && !h->custom_frames; }

// Does mt_rollout(n_steps >= 2) on this handle run k steps per launch through the rollout kernels right now?
bool rollout_is_multi_step(mt_handle h) { return h->multi_k > 1 && fusable(h); }
// Does it run as chains of one launch per step whose FIRST launch can be the rollout kernel with the reset as its prologue?
bool chained_rollout_absorbs_reset(mt_handle h) {
  const bool graph = h->graph_mode > 0 || (h->graph_mode < 0 && h->n <= h->graph_max);
  return h->defer_reset_chained && !rollout_is_multi_step(h) && h->chains > 1 && h->lazy_chains && !h->trace && !graph && fusable(h);
}


int check_launch(mt_handle h, const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(h, MT_ERR_HIP, std::string(what) + " launch: " + hipGetErrorString(e));
  return MT_OK;
}

struct FieldInfo {
  void* ptr;
  int rows;      // SoA rows
  int dtype;     // element dtype of the resident buffer
  size_t esize;  // element size
};

int field_info(mt_handle h, int field, FieldInfo* fi) {
  const StepArgs& a = h->args;
  switch (field) {
    case MT_F_ACTIONS: *fi = {a.actions, h->D, MT_F32, 4}; return MT_OK;
    case MT_F_GOALS: *fi = {a.goals, h->D, MT_F32, 4}; return MT_OK;
    case MT_F_POINTS: *fi = {a.points, 3 * h->K, MT_F32, 4}; return MT_OK;
    case MT_F_ALIVE: *fi = {a.alive, 1, MT_U32, 4}; return MT_OK;
    case MT_F_OBS: *fi = {a.obs, 3 * h->K, MT_F32, 4}; return MT_OK;
    case MT_F_REWARD: *fi = {a.reward, 1, MT_I32, 4}; return MT_OK;
    case MT_F_DONE: *fi = {a.done, 1, MT_U8, 1}; return MT_OK;
    case MT_F_DONE_BITS: *fi = {a.done_bits, 1, MT_U64, 8}; return MT_OK;
    case MT_F_EE: *fi = {a.ee, 3, MT_F32, 4}; return MT_OK;
    case MT_F_TOTAL_REWARD: *fi = {a.total_reward, 1, MT_F32, 4}; return MT_OK;
    case MT_F_EPISODES: *fi = {a.episodes, 1, MT_U32, 4}; return MT_OK;
    case MT_F_LAST_RETURN: *fi = {a.last_return, 1, MT_F32, 4}; return MT_OK;
    case MT_F_RETURN_RING:
      if (!a.ring) return fail(h, MT_ERR_STATE, "MT_F_RETURN_RING: the handle was created with return_ring = 0");
      *fi = {a.ring, (int)a.ring_slots, MT_F32, 4};
      return MT_OK;
    case MT_F_TRACE:
      if (!h->trace) return fail(h, MT_ERR_STATE, "MT_F_TRACE: the handle was created without MT_FLAG_TRACE");
      *fi = {h->trace, 3 * h->cfg.substeps, MT_F32, 4};
      return MT_OK;
    case MT_F_ZMIN:
      if (!a.zmin) return fail(h, MT_ERR_STATE, "MT_F_ZMIN: the handle was created without MT_FLAG_DEBUG_ZMIN");
      *fi = {a.zmin, 1, MT_F32, 4};
      return MT_OK;
    default: return fail(h, MT_ERR_INVALID_ARG, "unknown or non-resident field");
  }
}

}  // namespace

// =================================================================================================
// Host arrays handed to mt_set / mt_reset are screened on the bit pattern before they reach the arena (the kernels are
// built with -ffinite-math-only): `limit` = largest accepted magnitude as a float bit pattern.
static int64_t first_unusable(const float* v, int64_t count, uint32_t limit) {
  for (int64_t i = 0; i < count; ++i) {
    uint32_t bits;
    std::memcpy(&bits, v + i, 4);
    if ((bits & 0x7FFFFFFFu) > limit) return i;
  }
  return -1;
}
// Resets write MT_F_LAST_RETURN: an exchange that is still reading that row in place has to finish first.
static int order_behind_inplace_gather(mt_handle h, hipStream_t stream) {
  if (h->gather_pending && h->gather_inplace) MT_HIP(h, hipStreamWaitEvent(stream, h->ev_g1, 0));
  return MT_OK;
}

// Chains a whole-batch call may run as, right now: the handle's chain count unless the sub-step trace is on or the caller
// is capturing the handle's stream into a graph of their own (they get the plain single-stream sequence).
// per_step_call: mt_step / mt_sample_actions / mt_set_actions -- ONE launch per range and call.  On a caller's stream every
// call has to fork behind the caller's work and join back before it returns, two cross-stream dependencies for one launch:
// measured with a torch policy on the same stream (examples/policy_loop.py, 1 M arms) 144.4 us per loop iteration in two
// chains against 127.3 in one (profiles/r04_policy_loop_chains_on_torch_stream.txt), so there the step stays one launch.
// On the handle's own stream the chains stay forked from call to call and the step is 42 -> 37.5 us at 1 M arms.
static int usable_chains(mt_handle h, bool per_step_call = false) {
  if (h->chains <= 1 || h->trace) return 1;
  if (per_step_call && !(h->lazy_chains && h->stream == h->own_stream)) return 1;
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(h->stream, &st) != hipSuccess || st != hipStreamCaptureStatusNone) {
    (void)hipGetLastError();
    return 1;
  }
  return h->chains;
}

// fn(c, off, cnt) once per chain with h->stream switched to the chain's stream; forks first, settles afterwards.
template <class F>
static int per_chain(mt_handle h, int chains, const char* what, F&& fn) {
  int rc = fork_chains(h, chains);
  if (rc) return rc;
  const int64_t span = chain_span(h, chains);
  hipStream_t root = h->stream;
  for (int c = 0; c < chains; ++c) {
    const int64_t off = (int64_t)c * span;
    if (off >= h->n) continue;
    h->stream = c == 0 ? root : h->chain_streams[c];
    fn(c, off, std::min(span, h->n - off));
  }
  h->stream = root;
  rc = check_launch(h, what);
  if (rc) return rc;
  return settle_chains(h);
}

constexpr uint32_t kMaxAngleBits = 0x47000000u;   // 32768.0f: the bound of unusable_angle (kernels.h)
constexpr uint32_t kMaxFiniteBits = 0x7F7FFFFFu;  // FLT_MAX

extern "C" {

int mt_version(void) { return MT_VERSION; }

const char* mt_status_string(int status) {
  switch (status) {
    case MT_OK: return "ok";
    case MT_ERR_INVALID_ARG: return "invalid argument";
    case MT_ERR_HIP: return "HIP runtime error";
    case MT_ERR_NO_DEVICE: return "no usable HIP device";
    case MT_ERR_ALLOC: return "device allocation failed";
    case MT_ERR_STATE: return "invalid call order";
    case MT_ERR_UNSUPPORTED: return "unsupported";
    default: return "unknown status";
  }
}

const char* mt_last_error(mt_handle h) { return h ? h->err.c_str() : g_last_error.c_str(); }

int mt_device_count(int* count) {
  MT_REQUIRE(nullptr, count != nullptr, "count is NULL");
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) {
    *count = 0;
    (void)hipGetLastError();
    return fail(nullptr, MT_ERR_NO_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
  }
  *count = c;
  return MT_OK;
}

int mt_create(mt_handle* out, const mt_config* cfg) {
  MT_REQUIRE(nullptr, out != nullptr && cfg != nullptr, "out/cfg is NULL");
  *out = nullptr;
  MT_REQUIRE(nullptr, cfg->struct_size == (int32_t)sizeof(mt_config), "mt_config.struct_size mismatch");
  MT_REQUIRE(nullptr, cfg->n_envs >= 1, "n_envs must be >= 1");
  MT_REQUIRE(nullptr, cfg->n_envs <= ((int64_t)1 << 30) - 256, "n_envs must be < 2^30 per handle (32-bit row offsets)");
  MT_REQUIRE(nullptr, cfg->dof >= 2 && cfg->dof <= MT_MAX_DOF, "dof must be in 2..8");
  MT_REQUIRE(nullptr, cfg->n_targets >= 1 && cfg->n_targets <= MT_MAX_TARGETS, "n_targets must be in 1..32");
  MT_REQUIRE(nullptr, cfg->substeps >= 2, "substeps must be >= 2");
  MT_REQUIRE(nullptr, cfg->pickup_tol >= 0.f && cfg->radius > 0.f, "pickup_tol/radius out of range");
  // the kernels are built with -ffinite-math-only: every float that reaches them is screened on its bit pattern at the door
  MT_REQUIRE(nullptr, first_unusable(&cfg->pickup_tol, 1, kMaxFiniteBits) < 0 && first_unusable(&cfg->radius, 1, kMaxFiniteBits) < 0,
             "pickup_tol / radius must be finite");
  MT_REQUIRE(nullptr, first_unusable(cfg->dh_table, (int64_t)cfg->dof * 4, kMaxFiniteBits) < 0, "dh_table holds a NaN or an infinity");
  MT_REQUIRE(nullptr, cfg->env_id_base >= 0, "env_id_base must be >= 0");
  MT_REQUIRE(nullptr, cfg->return_ring >= 0 && cfg->return_ring <= MT_MAX_RETURN_RING, "return_ring must be in 0..64");
  MT_REQUIRE(nullptr, cfg->reserved == 0, "mt_config.reserved must be 0");
  MT_REQUIRE(nullptr, cfg->obs_frame >= -cfg->dof && cfg->obs_frame < cfg->dof, "obs_frame must be in -dof..dof-1");
  MT_REQUIRE(nullptr, cfg->ee_frame >= -cfg->dof && cfg->ee_frame < cfg->dof, "ee_frame must be in -dof..dof-1");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    (void)hipGetLastError();
    return fail(nullptr, MT_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU path)");
  }
  MT_REQUIRE(nullptr, cfg->device >= 0 && cfg->device < ndev, "device ordinal out of range");

  mt_engine* h = new (std::nothrow) mt_engine();
  if (!h) return fail(nullptr, MT_ERR_ALLOC, "out of host memory");
  h->cfg = *cfg;
  h->n = cfg->n_envs;
  h->ld = (int64_t)align_up((size_t)cfg->n_envs, 256);
  // Rows that are an exact power of two apart alias onto the same HBM channels, so every row gets a pad.  1 KiB
  // (256 floats): round 1 used 4 KiB, which is fine at 1 M arms but pathological at 262 144 (row stride 1 MiB + 4 KiB:
  // 14.7 us per step against 11.7 with 1 KiB; tools/pad_sweep.py, profiles/r02_variants.md section 6); 1 KiB is within
  // 1 % of the best pad at every size tried.  MT_LD_PAD (floats) overrides it for experiments.
  int64_t pad = (cfg->n_envs > kPolicy.ld_pad_above) ? kPolicy.ld_pad_floats : 0;
  if (const char* env = std::getenv("MT_LD_PAD")) pad = (int64_t)align_up((size_t)std::atoll(env), 64);
  h->ld += pad;
  h->D = cfg->dof;
  h->K = cfg->n_targets;

  auto bail = [&](int code, const std::string& msg) {
    g_last_error = msg;
    if (h->arena) (void)hipFree(h->arena);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
    return code;
  };
#define MT_HIP_C(call)                                                                       \
  do {                                                                                       \
    hipError_t e__ = (call);                                                                 \
    if (e__ != hipSuccess) return bail(MT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e__)); \
  } while (0)

  DeviceGuard guard(cfg->device);
  if (guard.error() != hipSuccess) return bail(MT_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(guard.error()));
  MT_HIP_C(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
  h->stream = h->own_stream;
  MT_HIP_C(hipEventCreateWithFlags(&h->ev0, event_flags(true)));
  MT_HIP_C(hipEventCreateWithFlags(&h->ev1, event_flags(true)));

  // arena layout: every row block starts on a 1 KiB boundary
  const size_t ld = (size_t)h->ld, D = (size_t)h->D, K = (size_t)h->K;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = off;
    off = align_up(off + bytes, 1024);
    return o;
  };
  const size_t o_act = take(D * ld * 4), o_goal = take(D * ld * 4), o_pts = take(3 * K * ld * 4),
               o_obs = take(3 * K * ld * 4), o_alive = take(ld * 4), o_tot = take(ld * 4), o_rew = take(ld * 4),
               o_done = take(ld), o_bits = take(ld / 64 * 8), o_ee = take(3 * ld * 4), o_epi = take(ld * 4),
               o_last = take(ld * 4), o_ring = take((size_t)cfg->return_ring * ld * 4), o_misc = take(256),
               o_trace = take((cfg->flags & MT_FLAG_TRACE) ? (size_t)cfg->substeps * 3 * ld * 4 : 0),
               o_zmin = take((cfg->flags & MT_FLAG_DEBUG_ZMIN) ? ld * 4 : 0), o_trig = take(kTrigEntries * sizeof(SinCos));
  h->arena_bytes = off;
  // (One experiment with hipExtMallocWithFlags(hipDeviceMallocContiguous) instead -- a physically contiguous arena -- cost 5 %
  // at 1 M arms and was dropped the same day: profiles/r04_variants.md section 3.)
  if (hipMalloc(&h->arena, h->arena_bytes) != hipSuccess) {
    (void)hipGetLastError();
    return bail(MT_ERR_ALLOC, "hipMalloc of the state arena failed (" + std::to_string(h->arena_bytes) + " bytes)");
  }
  MT_HIP_C(hipMemsetAsync(h->arena, 0, h->arena_bytes, h->stream));
  char* base = (char*)h->arena;
  StepArgs& a = h->args;
  a.actions = (float*)(base + o_act);
  a.goals = (float*)(base + o_goal);
  a.points = (float*)(base + o_pts);
  a.obs = (float*)(base + o_obs);
  a.alive = (uint32_t*)(base + o_alive);
  a.total_reward = (float*)(base + o_tot);
  a.reward = (int32_t*)(base + o_rew);
  a.done = (uint8_t*)(base + o_done);
  a.done_bits = (unsigned long long*)(base + o_bits);
  a.ee = (float*)(base + o_ee);
  a.episodes = (uint32_t*)(base + o_epi);
  a.last_return = (float*)(base + o_last);
  a.ring = cfg->return_ring ? (float*)(base + o_ring) : nullptr;
  a.ring_slots = (uint32_t)cfg->return_ring;
  a.episode0 = 0;
  a.bad_actions = (uint32_t*)(base + o_misc);                 // word 0 of the misc block
  h->spare_bits = (unsigned long long*)(base + o_misc + 64);  // ballot sink of single-env launches
  h->trace = (cfg->flags & MT_FLAG_TRACE) ? (float*)(base + o_trace) : nullptr;
  a.zmin = (cfg->flags & MT_FLAG_DEBUG_ZMIN) ? (float*)(base + o_zmin) : nullptr;
  a.snap = nullptr;  // set on copies of the arguments only: the last step launch of an mt_rollout (chained form)
  a.trig_table = (const float*)(base + o_trig);
  hipLaunchKernelGGL(fill_trig_table_kernel, dim3(1), dim3(kBlock), 0, h->stream, (float*)(base + o_trig));
  MT_HIP_C(hipGetLastError());
  a.n = h->n;
  a.ld = h->ld;
  a.env_base = cfg->env_id_base;
  a.K = h->K;
  a.S = cfg->substeps;
  a.tol = cfg->pickup_tol;
  a.inv_sm1 = 1.0f / (float)(cfg->substeps - 1);
  a.flags = cfg->flags & ~kFlagWholeGoals;  // the internal bit is the host's to set (resets / sampled steps), never the caller's
  a.dh = make_dh(cfg->dh_table, h->D);
  a.dh.fo = (cfg->obs_frame + h->D) % h->D;
  a.dh.fe = (cfg->ee_frame + h->D) % h->D;
  // other rows than the reference's last two: the runtime-table kernel with runtime frames (RtTableF), one env per lane
  choose_dispatch(h);
#undef MT_HIP_C
  *out = h;
  return MT_OK;
}

int mt_destroy(mt_handle h) {
  if (!h) return MT_OK;
  DeviceGuard guard(h->cfg.device);
  (void)hipStreamSynchronize(h->stream);
  mt_gather_release(h);  // drains the side stream before the communicator goes
  mt_comm_release(h);
  for (auto& g : h->graphs)
    for (hipGraphExec_t x : g.exec)
      if (x) (void)hipGraphExecDestroy(x);
  for (int c = 1; c < mt_engine::kMaxChains; ++c) {
    if (h->chain_streams[c]) {
      (void)hipStreamSynchronize(h->chain_streams[c]);
      (void)hipStreamDestroy(h->chain_streams[c]);
    }
    if (h->ev_join[c]) (void)hipEventDestroy(h->ev_join[c]);
    if (h->ev1c[c]) (void)hipEventDestroy(h->ev1c[c]);
  }
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  if (h->graph_step0) (void)hipFree(h->graph_step0);
  if (h->staging) (void)hipFree(h->staging);
  if (h->pinned) (void)hipHostFree(h->pinned);
  if (h->arena) (void)hipFree(h->arena);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  for (hipEvent_t e : h->lap_events) (void)hipEventDestroy(e);  // (the whole pool)
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
  delete h;
  return MT_OK;
}

const char* mt_step_kernel_name(mt_handle h) {
  if (!h) return "";
  const char* tbl = h->static_kind == 1 ? "Ref4Table" : (h->static_kind == 2 ? "Dh7Table" : nullptr);
  std::string table = tbl ? tbl : (h->custom_frames ? "RtTableF<" : "RtTable<") + std::to_string(h->D) + ">";
  if (h->trig == 0 && !h->lds_table && h->split)
    h->kernel_name = "step_split_kernel<" + table + ", L=" + std::to_string(h->split) + (h->trig_steps ? ", tt=1>" : ">");
  else
    h->kernel_name = "step_kernel<" + table + ", trig=" + std::to_string(h->trig) + ", lds=" + (h->lds_table ? "true" : "false") +
                     ", pf=" + ((h->trig == 0 && !h->lds_table && h->prefetch) ? std::to_string(kPrefetch) : std::string("0")) +
                     ((h->trig == 0 && !h->lds_table && h->trig_steps) ? ", tt=1" : "") +
                     ((h->trig == 0 && !h->lds_table && h->prefetch && h->cfg.n_envs >= h->flat_from) ? ", flat=1>" : ">");
  const bool multi_one_chain = h->multi_k > 1 && fusable(h) && !h->chains_forced;  // mt_rollout does not use the chains then
  if (h->chains > 1 && !multi_one_chain)  // what mt_rollout launches instead: the schedule for a chain's env count, on row views
    h->kernel_name += " [mt_rollout: " + std::to_string(h->chains) + " chains of " + std::to_string(chain_span(h, h->chains)) +
                      " envs, " + (h->trig == 0 && !h->lds_table && h->chain_split ? "L=" + std::to_string(h->chain_split)
                                   : std::string(h->trig == 0 && !h->lds_table && h->chain_prefetch
                                                     ? (chain_span(h, h->chains) >= h->flat_from ? "pf=8, flat=1" : "pf=8") : "pf=0")) +
                      ((h->trig == 0 && !h->lds_table && !h->chain_split && h->chain_prefetch && h->trig_steps &&
                        step_blocks_per_cu(h, chain_span(h, h->chains)) > 0)
                           ? ", " + std::to_string(step_blocks_per_cu(h, chain_span(h, h->chains))) + " blocks/CU" : std::string()) + "]";
  if (h->multi_k > 1 && fusable(h)) {  // what mt_rollout launches on small shards instead of one step kernel per step
    const int sp = (h->chains > 1 && h->chains_forced) ? h->chain_rollout_split : h->rollout_split;
    h->kernel_name += " [mt_rollout: " + std::to_string(h->multi_k) + " steps per launch, " +
                      (sp ? "rollout_split_kernel L=" + std::to_string(sp) : std::string("rollout_kernel")) + "]";
  }
  return h->kernel_name.c_str();
}

const char* mt_describe_dispatch(mt_handle h) {
  if (!h) return "";
  const DispatchPolicy& P = kPolicy;
  auto num = [](long long v) { return std::to_string(v); };
  auto b = [](bool v) { return std::string(v ? "true" : "false"); };
  const int64_t span = chain_span(h, h->chains);
  const char* tbl = h->static_kind == 1 ? "Ref4Table" : (h->static_kind == 2 ? "Dh7Table" : nullptr);
  const std::string table = tbl ? tbl : (h->custom_frames ? "RtTableF<" : "RtTable<") + std::to_string(h->D) + ">";
  const bool rec = h->trig == 0 && !h->lds_table;   // the default recurrence kernels: the only ones with schedules
  const bool multi = h->multi_k > 1 && fusable(h);
  const bool graph = h->graph_mode > 0 || (h->graph_mode < 0 && h->n <= h->graph_max);
  std::string caps = "[";
  for (const BlockCap& c : P.block_caps) caps += std::string(caps.size() > 1 ? "," : "") + "[" + num(c.from) + "," + num(c.below) + "," + num(c.blocks) + "]";
  caps += "]";
  std::string& d = h->describe;
  d = "{\"n_envs\":" + num(h->n) + ",\"ld\":" + num(h->ld) + ",\"dof\":" + num(h->D) + ",\"targets\":" + num(h->K) +
      ",\"table\":\"" + table + "\",\"trig\":" + num(h->trig) + ",\"lds_table\":" + b(h->lds_table) + ",\"custom_frames\":" + b(h->custom_frames) +
      // one launch per step: mt_step / mt_step_random on a one-chain handle
      ",\"step\":{\"lanes_per_env\":" + num(rec && h->split ? h->split : 1) + ",\"prefetch\":" + b(rec && !h->split && h->prefetch) +
      ",\"trig_table\":" + b(rec && h->trig_steps) + ",\"flat\":" + b(rec && !h->split && h->prefetch && h->n >= h->flat_from) + "}" +
      // per-chain calls: mt_rollout, mt_step, mt_sample_actions, mt_set_actions(device), mt_reset_random
      ",\"chains\":{\"count\":" + num(h->chains) + ",\"span\":" + num(h->chains > 1 ? span : h->n) + ",\"lazy\":" + b(h->lazy_chains) +
      ",\"lanes_per_env\":" + num(rec && h->chain_split ? h->chain_split : 1) + ",\"prefetch\":" + b(rec && !h->chain_split && h->chain_prefetch) +
      ",\"flat\":" + b(rec && !h->chain_split && h->chain_prefetch && span >= h->flat_from) +
      ",\"blocks_per_cu\":" + num(h->chains > 1 && rec && !h->chain_split && h->chain_prefetch && h->trig_steps ? step_blocks_per_cu(h, span) : 0) + "}" +
      ",\"rollout\":{\"form\":\"" + (multi ? "multi_step" : (h->chains > 1 ? "chained_steps" : (graph ? "graph_replay" : "launch_per_step"))) +
      "\",\"steps_per_launch\":" + num(multi ? h->multi_k : 1) + ",\"graph\":" + b(!multi && graph) +
      ",\"lanes_per_env\":" + num(std::max(1, multi ? ((h->chains > 1 && h->chains_forced) ? h->chain_rollout_split : h->rollout_split) : (rec ? (h->chains > 1 ? h->chain_split : h->split) : 0))) +
      ",\"chains\":" + num(multi && !h->chains_forced ? 1 : h->chains) +
      ",\"absorbs_reset\":" + b(h->defer_reset && (multi || chained_rollout_absorbs_reset(h))) + ",\"writes_snapshot\":" + b(h->snap_in_rollout && (multi || (h->chains > 1 && !graph && !h->trace))) + "}" +
      ",\"fused\":{\"usable\":" + b(fusable(h)) + ",\"lanes_per_env\":" + num(h->rollout_split ? h->rollout_split : 1) + "}" +
      ",\"reset\":{\"lanes_per_env\":" + num(h->reset_split ? 4 : 1) + "}" +
      ",\"overrides\":\"" + h->overrides + "\"" +
      ",\"policy\":{\"step_split4_max\":" + num(P.step_split4_max) + ",\"step_split2_max\":" + num(P.step_split2_max) +
      ",\"prefetch_other_max\":" + num(P.prefetch_other_max) + ",\"fused_split4_max\":" + num(P.fused_split4_max) +
      ",\"fused_split2_max\":" + num(P.fused_split2_max) + ",\"reset_split_max\":" + num(P.reset_split_max) +
      ",\"graph_max\":" + num(P.graph_max) + ",\"chains_min\":" + num(P.chains_min) + ",\"chains_max\":" + num(P.chains_max) +
      ",\"chain_split4_max\":" + num(P.chain_split4_max) + ",\"chain_split2_max\":" + num(P.chain_split2_max) +
      ",\"chain_prefetch_other_max\":" + num(P.chain_prefetch_other_max) + ",\"flat_from\":" + num(P.flat_from) +
      ",\"block_caps\":" + caps + ",\"multi_step_max\":" + num(P.multi_step_max) + ",\"multi_step_k\":" + num(P.multi_step_k) +
      ",\"max_recurrence_rotations\":" + num(P.max_recurrence_rotations) + ",\"ld_pad_above\":" + num(P.ld_pad_above) +
      ",\"ld_pad_floats\":" + num(P.ld_pad_floats) + "}" +
      ",\"arena_bytes\":" + num((long long)h->arena_bytes) + "}";
  return d.c_str();
}

int mt_set_stream(mt_handle h, void* hip_stream) {
  MT_REQUIRE(nullptr, h != nullptr, "handle is NULL");
  MT_ENTER(h);
  MT_HIP(h, hipStreamSynchronize(h->stream));   // everything queued so far is finished before the ordering changes
  h->stream = (hipStream_t)hip_stream;          // NULL is a stream too: the legacy default stream
  return MT_OK;
}

int mt_use_own_stream(mt_handle h) {
  MT_REQUIRE(nullptr, h != nullptr, "handle is NULL");
  MT_ENTER(h);
  MT_HIP(h, hipStreamSynchronize(h->stream));
  h->stream = h->own_stream;
  return MT_OK;
}

int mt_sync(mt_handle h) {
  MT_REQUIRE(nullptr, h != nullptr, "handle is NULL");
  MT_ENTER(h);  // (also: the NULL stream names the CURRENT device's default stream)
  MT_HIP(h, hipStreamSynchronize(h->stream));
  if (h->gather_pending) {  // a gather begun on the side stream is part of "everything queued on this handle"
    MT_HIP(h, hipStreamSynchronize(h->side_stream));
    MT_HIP(h, hipEventElapsedTime(&h->last_gather_ms, h->ev_g0, h->ev_g1));
    h->gather_pending = false;
    h->snap_free_known[0] = h->snap_free_known[1] = true;  // every exchange has finished
  }
  return MT_OK;
}

// ---- reset ------------------------------------------------------------------------------------
int mt_reset(mt_handle h, const float* points, int layout, int is_device) {
  MT_REQUIRE(nullptr, h != nullptr, "handle is NULL");
  MT_REQUIRE(h, points != nullptr, "points is NULL (use mt_reset_random for device-drawn targets)");
  MT_REQUIRE(h, layout == MT_ENV_MAJOR || layout == MT_SOA, "bad layout");
  if (!is_device && layout == MT_ENV_MAJOR) {
    const int64_t bad = first_unusable(points, h->n * 3 * h->K, kMaxFiniteBits);
    if (bad >= 0) return fail(h, MT_ERR_INVALID_ARG, "mt_reset: points element " + std::to_string(bad) + " is NaN or infinite");
  }
  if (!is_device && layout == MT_SOA)
    for (int r = 0; r < 3 * h->K; ++r) {  // the n live columns of every row (the pad is the caller's garbage)
      const int64_t bad = first_unusable(points + (int64_t)r * h->ld, h->n, kMaxFiniteBits);
      if (bad >= 0)
        return fail(h, MT_ERR_INVALID_ARG, "mt_reset: points row " + std::to_string(r) + " element " + std::to_string(bad) + " is NaN or infinite");
    }
  // (targets handed over in DEVICE memory cannot be screened here: reset_kernel drops an unusable one -- dead from the
  // start, coordinates zeroed -- and counts it, see mt_bad_action_count)
  MT_ENTER(h);
  const int rows = 3 * h->K;
  if (layout == MT_SOA) {
    const size_t bytes = (size_t)rows * h->ld * 4;
    MT_HIP(h, hipMemcpyAsync(h->args.points, points, bytes, is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                             h->stream));
  } else {
    const size_t bytes = (size_t)rows * h->n * 4;
    const float* src = points;
    if (!is_device) {
      int rc = ensure_staging(h, bytes);
      if (rc) return rc;
      MT_HIP(h, hipMemcpyAsync(h->staging, points, bytes, hipMemcpyHostToDevice, h->stream));
      src = (const float*)h->staging;
    }
    hipLaunchKernelGGL((env_major_to_soa<float>), grid_for(h->n), dim3(kBlock), 0, h->stream, src, rows, h->n,
                       h->args.points, h->ld);
    int rc = check_launch(h, "env_major_to_soa");
    if (rc) return rc;
  }
  {
    int rcg = order_behind_inplace_gather(h, h->stream);
    if (rcg) return rcg;
  }
  h->args.major = 0;  // caller-supplied targets start episode 0 of every env
  h->args.episode0 = 0;
  if (!h->goals_exposed) h->args.flags |= kFlagWholeGoals;  // every env is at the zero pose after this launch
  MT_DISPATCH_D(h->D, launch_reset_d, h, h->args, 0);
  int rc = check_launch(h, "reset_kernel");
  if (rc) return rc;
  if (!is_device) MT_HIP(h, hipStreamSynchronize(h->stream));  // the caller may free `points` on return
  h->is_reset = true;
  return MT_OK;
}

// mode 1: full random reset; 2: re-arm finished envs only.  The launches themselves (no deferral).
static int launch_reset_random(mt_handle h, uint64_t seed, uint32_t episode, int mode) {
  // A full random reset of a multi-chain handle is issued per chain, behind that chain's own last step, and leaves the
  // chains forked: one range's reset runs beside the other range's last step instead of behind a join.
  // (only WHILE they are forked: behind a joined call -- the multi-step mt_rollout of small shards, a getter -- the reset
  // is one launch on the handle's stream, and whoever forks next forks behind it)
  const bool per_chain = mode == 1 && h->chains > 1 && h->lazy_chains && h->forked;
  if (!per_chain) {
    int rc = join_chains(h);
    if (rc) return rc;
  }
  h->args.seed_lo = (uint32_t)seed;
  h->args.seed_hi = (uint32_t)(seed >> 32);
  h->args.major = episode;
  if (mode == 1) {
    h->args.episode0 = episode;  // full reset: finished-episode counts restart from here
    if (!h->goals_exposed) h->args.flags |= kFlagWholeGoals;  // every env is at the zero pose after this launch
  }
  if (per_chain) {
    int rc = fork_chains(h, h->chains);
    if (rc) return rc;
    const int64_t span = chain_span(h, h->chains);
    hipStream_t root = h->stream;
    for (int c = 0; c < h->chains && rc == MT_OK; ++c) {
      const int64_t off = (int64_t)c * span;
      if (off >= h->n) continue;
      const StepArgs ar = args_for_range(h, h->args, off, std::min(span, h->n - off));
      h->stream = c == 0 ? root : h->chain_streams[c];
      rc = order_behind_inplace_gather(h, h->stream);
      if (rc == MT_OK) MT_DISPATCH_D(h->D, launch_reset_d, h, ar, 1);
    }
    h->stream = root;
    if (rc == MT_OK) rc = check_launch(h, "reset_kernel (per chain)");
    if (rc) return rc;
    h->is_reset = true;
    return settle_chains(h);
  }
  int rc = order_behind_inplace_gather(h, h->stream);
  if (rc) return rc;
  MT_DISPATCH_D(h->D, launch_reset_d, h, h->args, mode);
  rc = check_launch(h, "reset_kernel");
  if (rc) return rc;
  h->is_reset = true;
  return MT_OK;
}

}  // extern "C" (interrupted: the next function has C++ linkage, engine_internal.h)

namespace mt {
int flush_pending_reset(mt_handle h) {
  if (!h->reset_pending) return MT_OK;
  h->reset_pending = false;
  return launch_reset_random(h, h->pend_seed, h->pend_episode, 1);
}
}  // namespace mt

extern "C" {

static int reset_random_impl(mt_handle h, uint64_t seed, uint32_t episode, int mode) {
  MT_REQUIRE(nullptr, h != nullptr, "handle is NULL");
  MT_ON_DEVICE(h, h->cfg.device);
  h->snap_valid = false;
  int rc = flush_pending_reset(h);  // an earlier deferred reset comes first (its last_return is what this one overwrites)
  if (rc) return rc;
  // A full reset of a handle whose mt_rollout runs k steps per launch is DEFERRED into the first launch of the next
  // mt_rollout (RolloutArgs::reset_first: the same state, bit for bit, without the reset's launch and without re-fetching what
  // it would have written: an episode end of a 131 072-env shard goes from 20 to 6 us on the device, tools/episode_end_cost.py).
  // Any other entry point launches it first (MT_ENTER).  Not while the chains are forked (the per-chain reset above), and
  // only on the handle's OWN stream: there the contract already is "mt_sync before anything else looks at the state" (and
  // mt_sync launches it); on a caller's stream the contract is stream order, and whatever the caller queues next without
  // calling the library -- a replay of a graph captured from mt_step, a torch kernel on a view it holds -- has to find
  // the reset done.
  if (mode == 1 && h->defer_reset && h->stream == h->own_stream &&
      ((rollout_is_multi_step(h) && !h->forked) || chained_rollout_absorbs_reset(h))) {
    h->args.seed_lo = (uint32_t)seed;
    h->args.seed_hi = (uint32_t)(seed >> 32);
    h->args.major = episode;
    h->args.episode0 = episode;
    if (!h->goals_exposed) h->args.flags |= kFlagWholeGoals;
    h->reset_pending = true;
    h->pend_seed = seed;
    h->pend_episode = episode;
    h->is_reset = true;
    return MT_OK;
  }
  return launch_reset_random(h, seed, episode, mode);
}

int mt_reset_random(mt_handle h, uint64_t seed, uint32_t episode) { return reset_random_impl(h, seed, episode, 1); }

int mt_reset_done(mt_handle h, uint64_t seed) {
  MT_REQUIRE(nullptr, h != nullptr, "handle is NULL");
  if (!h->is_reset) return fail(h, MT_ERR_STATE, "mt_reset_done before the first reset");
  return reset_random_impl(h, seed, 0, 2);
}

// ---- actions ----------------------------------------------------------------------------------
int mt_set_actions(mt_handle h, const void* actions, int dtype, int layout, int is_device) {
  MT_REQUIRE(nullptr, h != nullptr, "handle is NULL");
  MT_REQUIRE(h, actions != nullptr, "actions is NULL");
  MT_REQUIRE(h, layout == MT_ENV_MAJOR || layout == MT_SOA, "bad layout");
  MT_REQUIRE(h, dtype == MT_F32 || dtype == MT_F64 || dtype == MT_I32 || dtype == MT_I64, "bad action dtype");
  const size_t es = (dtype == MT_F32 || dtype == MT_I32) ? 4 : 8;
  // Device sources of a multi-chain handle are staged per chain -- each range's rows on its chain's stream, right ahead of
  // that range's mt_step -- so that a policy loop (mt_set_actions(device) / mt_step, ...) keeps the chains forked.
  const int chains = is_device ? usable_chains(h, true) : 1;
  if (chains > 1) {
    MT_ON_DEVICE(h, h->cfg.device);
    const int D = h->D;
    const int64_t ld = h->ld;
    hipError_t copy_err = hipSuccess;
    int rc = per_chain(h, chains, "set_actions (per chain)", [&](int, int64_t off, int64_t cnt) {
      const dim3 g = grid_for(cnt), b(kBlock);
      float* dst = h->args.actions + off;
      if (layout == MT_SOA && dtype == MT_F32) {
        hipError_t e = hipMemcpy2DAsync(dst, (size_t)ld * 4, (const float*)actions + off, (size_t)ld * 4, (size_t)cnt * 4, (size_t)D,
                                        hipMemcpyDeviceToDevice, h->stream);
        if (e != hipSuccess) copy_err = e;
      } else if (layout == MT_ENV_MAJOR) {
        const char* src = (const char*)actions + (size_t)off * D * es;
        switch (dtype) {
          case MT_F32: hipLaunchKernelGGL((env_major_to_soa<float>), g, b, 0, h->stream, (const float*)src, D, cnt, dst, ld); break;
          case MT_F64: hipLaunchKernelGGL((env_major_to_soa<double>), g, b, 0, h->stream, (const double*)src, D, cnt, dst, ld); break;
          case MT_I32: hipLaunchKernelGGL((env_major_to_soa<int32_t>), g, b, 0, h->stream, (const int32_t*)src, D, cnt, dst, ld); break;
          default: hipLaunchKernelGGL((env_major_to_soa<int64_t>), g, b, 0, h->stream, (const int64_t*)src, D, cnt, dst, ld); break;
        }
      } else {
        const char* src = (const char*)actions + (size_t)off * es;
        switch (dtype) {
          case MT_F64: hipLaunchKernelGGL((soa_to_soa_f32<double>), g, b, 0, h->stream, (const double*)src, D, cnt, ld, dst); break;
          case MT_I32: hipLaunchKernelGGL((soa_to_soa_f32<int32_t>), g, b, 0, h->stream, (const int32_t*)src, D, cnt, ld, dst); break;
          default: hipLaunchKernelGGL((soa_to_soa_f32<int64_t>), g, b, 0, h->stream, (const int64_t*)src, D, cnt, ld, dst); break;
        }
      }
    });
    if (rc) return rc;
    if (copy_err != hipSuccess) return fail(h, MT_ERR_HIP, std::string("hipMemcpy2DAsync (actions): ") + hipGetErrorString(copy_err));
    return MT_OK;
  }
  MT_ENTER(h);
  const int64_t cols = (layout == MT_SOA) ? h->ld : h->n;
  const size_t bytes = (size_t)h->D * cols * es;
  const void* src = actions;
  if (layout == MT_SOA && dtype == MT_F32) {
    MT_HIP(h, hipMemcpyAsync(h->args.actions, actions, bytes,
                             is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, h->stream));
    if (!is_device) MT_HIP(h, hipStreamSynchronize(h->stream));
    return MT_OK;
  }
  if (!is_device) {
    int rc = ensure_staging(h, bytes);
    if (rc) return rc;
    MT_HIP(h, hipMemcpyAsync(h->staging, actions, bytes, hipMemcpyHostToDevice, h->stream));
    src = h->staging;
  }
  const dim3 g = grid_for(h->n), b(kBlock);
  if (layout == MT_ENV_MAJOR) {
    switch (dtype) {
      case MT_F32: hipLaunchKernelGGL((env_major_to_soa<float>), g, b, 0, h->stream, (const float*)src, h->D, h->n, h->args.actions, h->ld); break;
      case MT_F64: hipLaunchKernelGGL((env_major_to_soa<double>), g, b, 0, h->stream, (const double*)src, h->D, h->n, h->args.actions, h->ld); break;
      case MT_I32: hipLaunchKernelGGL((env_major_to_soa<int32_t>), g, b, 0, h->stream, (const int32_t*)src, h->D, h->n, h->args.actions, h->ld); break;
      default: hipLaunchKernelGGL((env_major_to_soa<int64_t>), g, b, 0, h->stream, (const int64_t*)src, h->D, h->n, h->args.actions, h->ld); break;
    }
  } else {
    switch (dtype) {
      case MT_F64: hipLaunchKernelGGL((soa_to_soa_f32<double>), g, b, 0, h->stream, (const double*)src, h->D, h->n, h->ld, h->args.actions); break;
      case MT_I32: hipLaunchKernelGGL((soa_to_soa_f32<int32_t>), g, b, 0, h->stream, (const int32_t*)src, h->D, h->n, h->ld, h->args.actions); break;
      default: hipLaunchKernelGGL((soa_to_soa_f32<int64_t>), g, b, 0, h->stream, (const int64_t*)src, h->D, h->n, h->ld, h->args.actions); break;
    }
  }
  int rc = check_launch(h, "set_actions");
  if (rc) return rc;
  if (!is_device) MT_HIP(h, hipStreamSynchronize(h->stream));
  return MT_OK;
}

int mt_sample_actions(mt_handle h, uint64_t seed, uint32_t step_idx) {
  MT_REQUIRE(nullptr, h != nullptr, "handle is NULL");
  const int chains = usable_chains(h, true);
  if (chains > 1) {  // per chain: each range's actions are written on its chain's stream, ahead of that range's mt_step
    MT_ON_DEVICE(h, h->cfg.device);
    return per_chain(h, chains, "sample_actions_kernel (per chain)", [&](int, int64_t off, int64_t cnt) {
      hipLaunchKernelGGL(sample_actions_kernel, grid_for(cnt), dim3(kBlock), 0, h->stream, h->args.actions + off, cnt, h->ld, h->D,
                         h->args.env_base + off, seed, step_idx);
    });
  }
  MT_ENTER(h);
  hipLaunchKernelGGL(sample_actions_kernel, grid_for(h->n), dim3(kBlock), 0, h->stream, h->args.actions, h->n, h->ld,
                     h->D, h->args.env_base, seed, step_idx);
  return check_launch(h, "sample_actions_kernel");
}

// ---- step -------------------------------------------------------------------------------------
int mt_step(mt_handle h) {
  MT_REQUIRE(nullptr, h != nullptr, "handle is NULL");
  if (!h->is_reset) return fail(h, MT_ERR_STATE, "mt_step before mt_reset / mt_reset_random");
  // The policy-in-the-loop step: on a multi-chain handle each half of the env range is one launch on its own stream, as in
  // mt_rollout (a step of env i depends only on env i: same bits, one half's kernel boundary behind the other's kernel).
  // On the handle's own stream the chains stay forked across mt_set_actions(device) / mt_sample_actions / mt_step calls;
  // on a caller's stream (a torch policy on the same stream) the step is one launch (see usable_chains).
  const int chains = usable_chains(h, true);
  if (chains > 1) {
    MT_ON_DEVICE(h, h->cfg.device);
    h->snap_valid = false;
    {
      int rcf = flush_pending_reset(h);
      if (rcf) return rcf;
    }
    const StepArgs a = h->args;
    int rc = per_chain(h, chains, "step_kernel (per chain)", [&](int c, int64_t, int64_t) { launch_chain(h, a, a.major, 1, chains, c, false); });
    h->args.flags &= ~kFlagWholeGoals;  // staged actions are anybody's floats
    return rc;
  }
  MT_ENTER(h);
  launch_step(h, false);
  h->args.flags &= ~kFlagWholeGoals;  // staged actions are anybody's floats
  return check_launch(h, "step_kernel");
}

int mt_step_host(mt_handle h, const void* actions, int dtype, float* obs, int32_t* reward, uint8_t* done) {
  MT_REQUIRE(nullptr, h != nullptr, "handle is NULL");
  MT_REQUIRE(h, actions && obs && reward && done, "NULL argument");
  MT_REQUIRE(h, dtype == MT_F32 || dtype == MT_F64 || dtype == MT_I32 || dtype == MT_I64, "bad action dtype");
  if (!h->is_reset) return fail(h, MT_ERR_STATE, "mt_step_host before mt_reset / mt_reset_random");
  MT_ENTER(h);
  const size_t n = (size_t)h->n, es = (dtype == MT_F32 || dtype == MT_I32) ? 4 : 8;
  const size_t in_b = n * h->D * es, obs_b = n * 3 * h->K * 4, rew_b = n * 4, done_b = n;
  const size_t out_b = obs_b + rew_b + done_b, total = align_up(in_b, 256) + out_b;
  if (h->pinned_bytes < total) {
    if (h->pinned) {
      MT_HIP(h, hipStreamSynchronize(h->stream));
      (void)hipHostFree(h->pinned);
      h->pinned = nullptr;
      h->pinned_bytes = 0;
    }
    if (hipHostMalloc(&h->pinned, total, hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError();
      return fail(h, MT_ERR_ALLOC, "hipHostMalloc failed");
    }
    h->pinned_bytes = total;
  }
  int rc = ensure_staging(h, total);
  if (rc) return rc;
  char* hp = (char*)h->pinned;
  char* dp = (char*)h->staging;
  char* hp_out = hp + align_up(in_b, 256);
  char* dp_out = dp + align_up(in_b, 256);
  std::memcpy(hp, actions, in_b);
  MT_HIP(h, hipMemcpyAsync(dp, hp, in_b, hipMemcpyHostToDevice, h->stream));
  const dim3 g = grid_for(h->n), b(kBlock);
  switch (dtype) {
    case MT_F32: hipLaunchKernelGGL((env_major_to_soa<float>), g, b, 0, h->stream, (const float*)dp, h->D, h->n, h->args.actions, h->ld); break;
    case MT_F64: hipLaunchKernelGGL((env_major_to_soa<double>), g, b, 0, h->stream, (const double*)dp, h->D, h->n, h->args.actions, h->ld); break;
    case MT_I32: hipLaunchKernelGGL((env_major_to_soa<int32_t>), g, b, 0, h->stream, (const int32_t*)dp, h->D, h->n, h->args.actions, h->ld); break;
    default: hipLaunchKernelGGL((env_major_to_soa<int64_t>), g, b, 0, h->stream, (const int64_t*)dp, h->D, h->n, h->args.actions, h->ld); break;
  }
  launch_step(h, false);
  h->args.flags &= ~kFlagWholeGoals;
  hipLaunchKernelGGL((soa_to_env_major<float>), g, b, 0, h->stream, h->args.obs, h->ld, 3 * h->K, h->n, (float*)dp_out);
  rc = check_launch(h, "mt_step_host");
  if (rc) return rc;
  MT_HIP(h, hipMemcpyAsync(dp_out + obs_b, h->args.reward, rew_b, hipMemcpyDeviceToDevice, h->stream));
  MT_HIP(h, hipMemcpyAsync(dp_out + obs_b + rew_b, h->args.done, done_b, hipMemcpyDeviceToDevice, h->stream));
  MT_HIP(h, hipMemcpyAsync(hp_out, dp_out, out_b, hipMemcpyDeviceToHost, h->stream));
  MT_HIP(h, hipStreamSynchronize(h->stream));  // the only host wait of the call
  std::memcpy(obs, hp_out, obs_b);
  std::memcpy(reward, hp_out + obs_b, rew_b);
  std::memcpy(done, hp_out + obs_b + rew_b, done_b);
  return MT_OK;
}

int mt_step_random(mt_handle h, uint64_t seed, uint32_t step_idx) {
  MT_REQUIRE(nullptr, h != nullptr, "handle is NULL");
  if (!h->is_reset) return fail(h, MT_ERR_STATE, "mt_step_random before mt_reset / mt_reset_random");
  MT_ENTER(h);
  h->args.seed_lo = (uint32_t)seed;
  h->args.seed_hi = (uint32_t)(seed >> 32);
  h->args.major = step_idx;
  launch_step(h, true);
  return check_launch(h, "step_kernel");
}

// ---- one env of the batch (multienv.environment[i].step / .reset, manytor.py:82,118) ----------------------------
static int rebuild_done_word(mt_handle h, int64_t env) {
  hipLaunchKernelGGL(done_bits_word_kernel, dim3(1), dim3(64), 0, h->stream, h->args.done, h->n, env >> 6,
                     h->args.done_bits);
  return check_launch(h, "done_bits_word_kernel");
}

int mt_env_step(mt_handle h, int64_t env, const float* action, float* obs, int32_t* reward, uint8_t* done) {
  MT_REQUIRE(nullptr, h != nullptr, "handle is NULL");
  MT_REQUIRE(h, action && obs && reward && done, "NULL argument");
  MT_REQUIRE(h, env >= 0 && env < h->n, "env index out of range");
  if (!h->is_reset) return fail(h, MT_ERR_STATE, "mt_env_step before mt_reset / mt_reset_random");
  MT_ENTER(h);
  const StepArgs a = args_for_env(h, env);
  const size_t pitch = (size_t)h->ld * 4;
  // column `env` of the D action rows <- D host floats
  MT_HIP(h, hipMemcpy2DAsync(a.actions, pitch, action, 4, 4, (size_t)h->D, hipMemcpyHostToDevice, h->stream));
  launch_step(h, a, h->trace ? h->trace + env : nullptr, false, false);
  h->args.flags &= ~kFlagWholeGoals;
  int rc = check_launch(h, "step_kernel (one env)");
  if (rc) return rc;
  rc = rebuild_done_word(h, env);
  if (rc) return rc;
  MT_HIP(h, hipMemcpy2DAsync(obs, 4, a.obs, pitch, 4, (size_t)(3 * h->K), hipMemcpyDeviceToHost, h->stream));
  MT_HIP(h, hipMemcpyAsync(reward, a.reward, 4, hipMemcpyDeviceToHost, h->stream));
  MT_HIP(h, hipMemcpyAsync(done, a.done, 1, hipMemcpyDeviceToHost, h->stream));
  MT_HIP(h, hipStreamSynchronize(h->stream));
  return MT_OK;
}

int mt_env_reset(mt_handle h, int64_t env, const float* points, uint64_t seed, uint32_t episode) {
  MT_REQUIRE(nullptr, h != nullptr, "handle is NULL");
  MT_REQUIRE(h, env >= 0 && env < h->n, "env index out of range");
  if (!h->is_reset) return fail(h, MT_ERR_STATE, "mt_env_reset before the first reset of the batch");
  if (points && first_unusable(points, 3 * (int64_t)h->K, kMaxFiniteBits) >= 0)
    return fail(h, MT_ERR_INVALID_ARG, "mt_env_reset: a target coordinate is NaN or infinite");
  MT_ENTER(h);
  StepArgs a = args_for_env(h, env);
  if (points)  // (K, 3) host floats: flat index 3k + axis = row index
    MT_HIP(h, hipMemcpy2DAsync(a.points, (size_t)h->ld * 4, points, 4, 4, (size_t)(3 * h->K), hipMemcpyHostToDevice,
                               h->stream));
  a.seed_lo = (uint32_t)seed;
  a.seed_hi = (uint32_t)(seed >> 32);
  a.major = episode;
  {
    int rcg = order_behind_inplace_gather(h, h->stream);
    if (rcg) return rcg;
  }
  MT_DISPATCH_D(h->D, launch_reset_d, h, a, points ? 0 : 1);
  int rc = check_launch(h, "reset_kernel (one env)");
  if (rc) return rc;
  rc = rebuild_done_word(h, env);
  if (rc) return rc;
  MT_HIP(h, hipStreamSynchronize(h->stream));  // the caller may free `points` on return
  return MT_OK;
}

int mt_bad_action_count(mt_handle h, uint64_t* count) {
  MT_REQUIRE(nullptr, h != nullptr, "handle is NULL");
  MT_REQUIRE(h, count != nullptr, "count is NULL");
  MT_ENTER(h);
  uint32_t c = 0;
  MT_HIP(h, hipMemcpyAsync(&c, h->args.bad_actions, 4, hipMemcpyDeviceToHost, h->stream));
  MT_HIP(h, hipStreamSynchronize(h->stream));
  *count = c;
  return MT_OK;
}

// mt_rollout on small batches is bound by the kernel boundary, not by the kernels (profiles/r02_variants.md section 3):
// the T launches of a segment are captured once into a HIP graph (stream capture of the very same launches; each node's
// step index = its offset + a device word) and replayed -- 5-15 % less per step up to 131 072 arms, nothing above.
static bool rollout_uses_graph(mt_handle h, int n_steps) {
  if (h->graph_mode == 0 || h->trace || n_steps < 4) return false;
  if (h->graph_mode < 0 && h->n > h->graph_max) return false;

  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(h->stream, &st) != hipSuccess || st != hipStreamCaptureStatusNone) {
    (void)hipGetLastError();
    return false;  // the caller is capturing this stream into a graph of their own: plain launches
  }
  return true;
}

// The cached graph(s) of a T-step segment: one graph per chain (chain c's T launches), to be replayed on chain c's stream.
static int rollout_graph(mt_handle h, int T, uint64_t seed, int chains, const mt_engine::RolloutGraph** out) {
  // one word per chain: chain c's nodes read word c, which is rewritten on chain c's OWN stream right before its replay
  if (!h->graph_step0) MT_HIP(h, hipMalloc(&h->graph_step0, sizeof(uint32_t) * mt_engine::kMaxChains));
  StepArgs a = h->args;
  a.seed_lo = (uint32_t)seed;
  a.seed_hi = (uint32_t)(seed >> 32);
  a.major = 0;
  a.episode0 = 0;  // not read by the step kernels
  a.major_base = h->graph_step0;
  for (auto& g : h->graphs)
    if (g.T == T && g.chains == chains && std::memcmp(&g.args, &a, sizeof a) == 0) {
      *out = &g;
      return MT_OK;
    }
  // captured on the handle's private stream whatever stream the handle currently launches on (the caller's may be the
  // null stream, which cannot capture); the instantiated graphs are launched on the current one / the chain streams
  mt_engine::RolloutGraph rg{};
  rg.T = T;
  rg.chains = chains;
  rg.args = a;
  hipStream_t launch_stream = h->stream;
  auto drop = [&]() {
    for (int c = 0; c < chains; ++c)
      if (rg.exec[c]) (void)hipGraphExecDestroy(rg.exec[c]);
  };
  for (int c = 0; c < chains; ++c) {
    if (c > 0 && (int64_t)c * chain_span(h, chains) >= h->n) break;
    hipGraph_t graph = nullptr;
    MT_HIP(h, hipStreamBeginCapture(h->own_stream, hipStreamCaptureModeThreadLocal));
    h->stream = h->own_stream;
    if (chains > 1) {
      StepArgs ac = a;
      ac.major_base = h->graph_step0 + c;
      launch_chain(h, ac, 0u, T, chains, c);
    } else {
      for (int s = 0; s < T; ++s) {
        StepArgs as = a;
        as.major = (uint32_t)s;
        launch_step(h, as, nullptr, true, true);
      }
    }
    h->stream = launch_stream;
    hipError_t e = hipStreamEndCapture(h->own_stream, &graph);
    if (e != hipSuccess || !graph) {
      (void)hipGetLastError();
      drop();
      return fail(h, MT_ERR_HIP, std::string("mt_rollout: stream capture failed: ") + hipGetErrorString(e));
    }
    e = hipGraphInstantiate(&rg.exec[c], graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) {
      rg.exec[c] = nullptr;
      drop();
      return fail(h, MT_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
    }
  }
  if (h->graphs.size() >= 8) {  // a handful of segment lengths at most; drop the oldest beyond that
    // its per-chain graphs may still be executing on the chain streams (lazily forked chains): fold them back and drain
    // every stream a replay can be on before the executables go
    int rcj = join_chains(h);
    if (rcj) return rcj;
    MT_HIP(h, hipStreamSynchronize(h->stream));
    for (int c = 1; c < mt_engine::kMaxChains; ++c)
      if (h->chain_streams[c]) MT_HIP(h, hipStreamSynchronize(h->chain_streams[c]));
    for (hipGraphExec_t x : h->graphs.front().exec)
      if (x) (void)hipGraphExecDestroy(x);
    h->graphs.erase(h->graphs.begin());
  }
  h->graphs.push_back(rg);
  *out = &h->graphs.back();
  return MT_OK;
}

int mt_rollout(mt_handle h, int n_steps, uint64_t seed, uint32_t step_idx0) {
  MT_REQUIRE(nullptr, h != nullptr, "handle is NULL");
  MT_REQUIRE(h, n_steps >= 0, "n_steps must be >= 0");
  if (n_steps > 0 && !h->is_reset) return fail(h, MT_ERR_STATE, "mt_rollout before mt_reset / mt_reset_random");
  if (n_steps == 0) return MT_OK;
  MT_ON_DEVICE(h, h->cfg.device);  // per-chain call: joins only where it has to (below)
  // several independent chains of launches (env ranges on separate streams) where that pays; a caller who is capturing
  // the handle's stream gets the plain single-stream sequence
  // (a call that finds a deferred reset takes the form that absorbs it even for a single step)
  int chains = (n_steps >= 2 || h->reset_pending) ? usable_chains(h) : 1;
  h->snap_valid = false;  // the returns are about to change
  const bool multi = (n_steps >= 2 || h->reset_pending) && rollout_is_multi_step(h);
  const bool chained_fresh = h->reset_pending && !multi && chains > 1 && chained_rollout_absorbs_reset(h) && !rollout_uses_graph(h, n_steps);
  if (n_steps < 2 && !multi && !chained_fresh) chains = 1;
  if (h->reset_pending && !multi && !chained_fresh) {  // a deferred reset and a form that cannot absorb it: launch it now
    int rc = flush_pending_reset(h);
    if (rc) return rc;
  }
  // Small shards: k steps per launch through the rollout kernels (kPolicy.multi_step_*).  The call exposes the state after
  // n_steps steps and the outputs of the last one either way; every step still writes its outputs.  One chain (see
  // kPolicy) unless MT_CHAINS asks for more.  The episode boundary rides along: a deferred full reset becomes the prologue
  // of the first launch, and the last launch also stores the returns to the overlapped gather's snapshot row.
  if (multi) {
    if (!h->chains_forced) chains = 1;
    StepArgs a = h->args;
    a.seed_lo = (uint32_t)seed;
    a.seed_hi = (uint32_t)(seed >> 32);
    const bool fresh = h->reset_pending;
    h->reset_pending = false;
    // the snapshot row of the next exchange (double-buffered: engine_internal.h) may be written once the exchange that last
    // read it is known to have finished: asked, never waited for (mt_gather_returns_begin keeps the host informed)
    bool snap = h->snap_in_rollout && h->snap != nullptr;
    const int snap_p = h->snap_next;
    if (snap && !h->snap_free_known[snap_p]) {
      if (hipEventQuery(h->ev_gdone[snap_p]) == hipSuccess) {
        h->snap_free_known[snap_p] = true;
      } else {
        (void)hipGetLastError();
        snap = false;
      }
    }
    auto args_of = [&](int s0) {
      RolloutArgs r{std::min(h->multi_k, n_steps - s0), step_idx0 + (uint32_t)s0, 0u, h->cfg.radius, 0u, 0u, 0u, 0u, nullptr};
      if (fresh && s0 == 0) {
        r.reset_first = 1u;
        r.reset_episode = h->pend_episode;
        r.reset_seed_lo = (uint32_t)h->pend_seed;
        r.reset_seed_hi = (uint32_t)(h->pend_seed >> 32);
      }
      if (snap && s0 + h->multi_k >= n_steps) r.snap = h->snap_row(snap_p);
      return r;
    };
    int rc = MT_OK;
    if (fresh) {  // the in-kernel reset writes MT_F_LAST_RETURN: behind an exchange that still reads that row in place
      rc = order_behind_inplace_gather(h, h->stream);
      if (rc) return rc;
    }
    if (chains > 1) {
      rc = fork_chains(h, chains);
      if (rc) return rc;
      const int64_t span = chain_span(h, chains);
      hipStream_t root = h->stream;
      for (int s0 = 0; s0 < n_steps; s0 += h->multi_k) {
        for (int c = 0; c < chains; ++c) {
          const int64_t off = (int64_t)c * span;
          if (off >= h->n) continue;
          h->stream = c == 0 ? root : h->chain_streams[c];
          RolloutArgs r = args_of(s0);
          if (r.snap) r.snap += off;
          if (fresh && s0 == 0 && c > 0 && rc == MT_OK) rc = order_behind_inplace_gather(h, h->stream);
          if (rc == MT_OK) launch_rollout(h, args_for_range(h, a, off, std::min(span, h->n - off)), h->chain_rollout_split, r, h->rollout_early);
        }
      }
      h->stream = root;
      if (rc == MT_OK) rc = check_launch(h, "rollout_kernel (mt_rollout, per chain)");
      if (rc) return rc;
      rc = settle_chains(h);
    } else {
      rc = join_chains(h);
      if (rc) return rc;
      for (int s0 = 0; s0 < n_steps; s0 += h->multi_k) launch_rollout(h, a, h->rollout_split, args_of(s0), h->rollout_early);
      rc = check_launch(h, "rollout_kernel (mt_rollout)");
    }
    if (rc) return rc;
    h->snap_valid = snap;
    h->args.seed_lo = (uint32_t)seed;
    h->args.seed_hi = (uint32_t)(seed >> 32);
    h->args.major = step_idx0 + (uint32_t)(n_steps - 1);
    return MT_OK;
  }
  bool graph = rollout_uses_graph(h, n_steps);
  if (graph && h->graph_mode < 0) {
    // capture + instantiation cost about a dozen plain segments: only for a segment length that comes back
    bool cached = false, seen = false;
    for (auto& g : h->graphs) cached |= g.T == n_steps;
    for (int t : h->graph_seen) seen |= t == n_steps;
    if (!cached && !seen) {
      if (h->graph_seen.size() >= 32) h->graph_seen.erase(h->graph_seen.begin());
      h->graph_seen.push_back(n_steps);
      graph = false;
    }
  }
  if (chains == 1) {  // single-stream forms: the chains (if a per-chain reset left them forked) come back first
    int rc = join_chains(h);
    if (rc) return rc;
  }
  if (graph) {
    const mt_engine::RolloutGraph* rg = nullptr;
    int rc = rollout_graph(h, n_steps, seed, chains, &rg);
    if (rc) return rc;
    if (chains > 1) {
      rc = fork_chains(h, chains);
      if (rc) return rc;
      const int64_t span = chain_span(h, chains);
      for (int c = 0; c < chains; ++c) {
        if ((int64_t)c * span >= h->n) continue;
        // chain c's step-index word is set on chain c's stream, behind its previous replay and ahead of this one
        MT_HIP(h, hipMemsetD32Async((hipDeviceptr_t)(h->graph_step0 + c), (int)step_idx0, 1, chain_stream(h, c)));
        MT_HIP(h, hipGraphLaunch(rg->exec[c], chain_stream(h, c)));
      }
      rc = settle_chains(h);
      if (rc) return rc;
    } else {
      MT_HIP(h, hipMemsetD32Async((hipDeviceptr_t)h->graph_step0, (int)step_idx0, 1, h->stream));
      MT_HIP(h, hipGraphLaunch(rg->exec[0], h->stream));
    }
  } else if (chains > 1) {
    StepArgs a = h->args;
    a.seed_lo = (uint32_t)seed;
    a.seed_hi = (uint32_t)(seed >> 32);
    int rc = fork_chains(h, chains);  // no-op when a per-chain reset (or the previous segment) left them forked
    if (rc) return rc;
    // chains enqueued round-robin step by step, so that no stream runs dry while the host is busy with another one
    hipStream_t root = h->stream;
    const int64_t span = chain_span(h, chains);
    h->reset_pending = false;
    // the last step launch of every chain also stores its returns to the overlapped gather's next snapshot row (double-
    // buffered: engine_internal.h), once the exchange that last read that row is known to have finished
    StepArgs a_last = a;
    {
      const int sp = h->snap_next;
      bool snap = h->snap_in_rollout && h->snap != nullptr && !h->trace;
      if (snap && !h->snap_free_known[sp]) {
        if (hipEventQuery(h->ev_gdone[sp]) == hipSuccess) {
          h->snap_free_known[sp] = true;
        } else {
          (void)hipGetLastError();
          snap = false;
        }
      }
      if (snap) a_last.snap = h->snap_row(sp);
    }
    bool fresh_by_rollout_kernel = false;
    for (int st = 0; st < n_steps; ++st)
      for (int c = 0; c < chains; ++c) {
        h->stream = c == 0 ? root : h->chain_streams[c];
        const int64_t off = (int64_t)c * span;
        if (chained_fresh && st == 0 && off < h->n) {
          // the episode's first step carries the deferred reset as its prologue (the state of reset_kernel + step_kernel, bit
          // for bit: no reset launch, no stores of what the step overwrites, no re-fetch of what the reset would have
          // written): the step kernel's own FRESH form where the schedule has one, else ONE step of the rollout kernel
          if (rc == MT_OK) rc = order_behind_inplace_gather(h, h->stream);  // (the in-kernel reset writes MT_F_LAST_RETURN)
          if (rc != MT_OK) continue;
          StepArgs af = (n_steps == 1) ? a_last : a;
          af.reset_seed_lo = (uint32_t)h->pend_seed;
          af.reset_seed_hi = (uint32_t)(h->pend_seed >> 32);
          af.reset_episode = h->pend_episode;
          af.radius = h->cfg.radius;
          if (launch_chain_fresh(h, af, step_idx0, chains, c)) continue;
          fresh_by_rollout_kernel = true;
          const RolloutArgs r{1, step_idx0, 0u, h->cfg.radius, 1u, h->pend_episode, (uint32_t)h->pend_seed, (uint32_t)(h->pend_seed >> 32), nullptr};
          launch_rollout(h, args_for_range(h, a, off, std::min(span, h->n - off)), 1, r, h->rollout_early);
          continue;
        }
        if (rc == MT_OK) launch_chain(h, st == n_steps - 1 ? a_last : a, step_idx0 + (uint32_t)st, 1, chains, c);
      }
    h->stream = root;
    if (rc == MT_OK) rc = check_launch(h, "step_kernel (chained)");
    if (rc) return rc;
    rc = settle_chains(h);
    if (rc) return rc;
    h->snap_valid = a_last.snap != nullptr && !(fresh_by_rollout_kernel && n_steps == 1);
  } else {
    for (int s = 0; s < n_steps; ++s) {
      int rc = mt_step_random(h, seed, step_idx0 + (uint32_t)s);
      if (rc) return rc;
    }
    return MT_OK;
  }
  h->args.seed_lo = (uint32_t)seed;
  h->args.seed_hi = (uint32_t)(seed >> 32);
  h->args.major = step_idx0 + (uint32_t)(n_steps - 1);
  return MT_OK;
}

int mt_rollout_fused(mt_handle h, int n_steps, uint64_t seed, uint32_t step_idx0, int auto_reset) {
  MT_REQUIRE(nullptr, h != nullptr, "handle is NULL");
  MT_REQUIRE(h, n_steps >= 0, "n_steps must be >= 0");
  if (!h->is_reset) return fail(h, MT_ERR_STATE, "mt_rollout_fused before mt_reset / mt_reset_random");
  if (n_steps == 0) return MT_OK;
  // The fused kernel implements the default trigonometry only; the measured alternatives run the same thing as a
  // sequence of launches.
  if (!fusable(h)) {
    for (int s = 0; s < n_steps; ++s) {
      int rc = mt_step_random(h, seed, step_idx0 + (uint32_t)s);
      if (rc) return rc;
      if (auto_reset) {
        rc = mt_reset_done(h, seed);
        if (rc) return rc;
      }
    }
    return MT_OK;
  }
  MT_ENTER(h);
  h->args.seed_lo = (uint32_t)seed;
  h->args.seed_hi = (uint32_t)(seed >> 32);
  if (auto_reset) {  // the in-kernel re-arm writes MT_F_LAST_RETURN
    int rcg = order_behind_inplace_gather(h, h->stream);
    if (rcg) return rcg;
  }
  RolloutArgs r{n_steps, step_idx0, auto_reset ? 1u : 0u, h->cfg.radius, 0u, 0u, 0u, 0u, nullptr};
  launch_rollout(h, h->args, h->rollout_split, r);
  return check_launch(h, "rollout_kernel");
}

int mt_observe(mt_handle h) {
  MT_REQUIRE(nullptr, h != nullptr, "handle is NULL");
  if (!h->is_reset) return fail(h, MT_ERR_STATE, "mt_observe before reset");
  MT_ENTER(h);
  MT_DISPATCH_D(h->D, launch_observe_d, h);
  return check_launch(h, "observe_kernel");
}

int mt_check_done(mt_handle h) {
  MT_REQUIRE(nullptr, h != nullptr, "handle is NULL");
  if (!h->is_reset) return fail(h, MT_ERR_STATE, "mt_check_done before reset");
  MT_ENTER(h);
  MT_DISPATCH_D(h->D, launch_check_done_d, h);
  return check_launch(h, "check_done_kernel");
}

// ---- getters / setters ------------------------------------------------------------------------
static int64_t env_major_bytes(mt_handle h, int field) {
  const int64_t n = h->n;
  switch (field) {
    case MT_F_ACTIONS:
    case MT_F_GOALS: return n * h->D * 4;
    case MT_F_POINTS:
    case MT_F_OBS: return n * 3 * h->K * 4;
    case MT_F_ALIVE: return n * h->K;
    case MT_F_REWARD:
    case MT_F_TOTAL_REWARD:
    case MT_F_EPISODES:
    case MT_F_LAST_RETURN: return n * 4;
    case MT_F_DONE: return n;
    case MT_F_DONE_BITS: return (n + 63) / 64 * 8;
    case MT_F_EE: return n * 3 * 4;
    case MT_F_JOINTS: return n * h->D * 3 * 4;
    case MT_F_RETURN_RING: return h->args.ring ? n * (int64_t)h->args.ring_slots * 4 : -1;
    case MT_F_TRACE: return h->trace ? n * (int64_t)h->cfg.substeps * 3 * 4 : -1;
    case MT_F_ZMIN: return h->args.zmin ? n * 4 : -1;
    default: return -1;
  }
}

int mt_get(mt_handle h, int field, void* dst, int64_t dst_bytes, int is_device) {
  MT_REQUIRE(nullptr, h != nullptr, "handle is NULL");
  MT_REQUIRE(h, dst != nullptr, "dst is NULL");
  const int64_t need = env_major_bytes(h, field);
  MT_REQUIRE(h, need > 0, "unknown field (or a field this handle was created without)");
  MT_REQUIRE(h, dst_bytes == need, "dst_bytes does not match the field's env-major size");
  MT_ENTER(h);
  const dim3 g = grid_for(h->n), b(kBlock);
  void* out = dst;
  if (!is_device) {
    int rc = ensure_staging(h, (size_t)need);
    if (rc) return rc;
    out = h->staging;
  }
  const StepArgs& a = h->args;
  bool direct = false;  // single-row fields need no transpose
  const void* direct_src = nullptr;
  switch (field) {
    case MT_F_ACTIONS: hipLaunchKernelGGL((soa_to_env_major<float>), g, b, 0, h->stream, a.actions, h->ld, h->D, h->n, (float*)out); break;
    case MT_F_GOALS: hipLaunchKernelGGL((soa_to_env_major<float>), g, b, 0, h->stream, a.goals, h->ld, h->D, h->n, (float*)out); break;
    case MT_F_POINTS: hipLaunchKernelGGL((soa_to_env_major<float>), g, b, 0, h->stream, a.points, h->ld, 3 * h->K, h->n, (float*)out); break;
    case MT_F_OBS: hipLaunchKernelGGL((soa_to_env_major<float>), g, b, 0, h->stream, a.obs, h->ld, 3 * h->K, h->n, (float*)out); break;
    case MT_F_EE: hipLaunchKernelGGL((soa_to_env_major<float>), g, b, 0, h->stream, a.ee, h->ld, 3, h->n, (float*)out); break;
    case MT_F_RETURN_RING: hipLaunchKernelGGL((soa_to_env_major<float>), g, b, 0, h->stream, a.ring, h->ld, (int)a.ring_slots, h->n, (float*)out); break;
    case MT_F_TRACE: hipLaunchKernelGGL((soa_to_env_major<float>), g, b, 0, h->stream, h->trace, h->ld, 3 * h->cfg.substeps, h->n, (float*)out); break;
    case MT_F_ALIVE: hipLaunchKernelGGL(alive_unpack, g, b, 0, h->stream, a.alive, h->K, h->n, (uint8_t*)out); break;
    case MT_F_JOINTS: MT_DISPATCH_D(h->D, launch_joints_d, h, (float*)out); break;
    case MT_F_REWARD: direct = true; direct_src = a.reward; break;
    case MT_F_TOTAL_REWARD: direct = true; direct_src = a.total_reward; break;
    case MT_F_EPISODES: direct = true; direct_src = a.episodes; break;
    case MT_F_LAST_RETURN: direct = true; direct_src = a.last_return; break;
    case MT_F_DONE: direct = true; direct_src = a.done; break;
    case MT_F_DONE_BITS: direct = true; direct_src = a.done_bits; break;
    case MT_F_ZMIN: direct = true; direct_src = a.zmin; break;
    default: return fail(h, MT_ERR_INVALID_ARG, "unknown field");
  }
  if (direct) {
    MT_HIP(h, hipMemcpyAsync(dst, direct_src, (size_t)need, is_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost,
                             h->stream));
  } else {
    int rc = check_launch(h, "mt_get transpose");
    if (rc) return rc;
    if (!is_device) MT_HIP(h, hipMemcpyAsync(dst, h->staging, (size_t)need, hipMemcpyDeviceToHost, h->stream));
  }
  if (!is_device) MT_HIP(h, hipStreamSynchronize(h->stream));
  return MT_OK;
}

int mt_set(mt_handle h, int field, const void* src, int64_t src_bytes) {
  MT_REQUIRE(nullptr, h != nullptr, "handle is NULL");
  MT_REQUIRE(h, src != nullptr, "src is NULL");
  MT_REQUIRE(h, field == MT_F_GOALS || field == MT_F_POINTS || field == MT_F_ALIVE || field == MT_F_TOTAL_REWARD ||
                    field == MT_F_DONE || field == MT_F_EPISODES || field == MT_F_LAST_RETURN || field == MT_F_RETURN_RING,
             "field is not settable");
  const int64_t need = env_major_bytes(h, field);
  MT_REQUIRE(h, need > 0, "the handle was created without this field");
  MT_REQUIRE(h, src_bytes == need, "src_bytes does not match the field's env-major size");
  if (field == MT_F_GOALS || field == MT_F_POINTS || field == MT_F_TOTAL_REWARD || field == MT_F_LAST_RETURN ||
      field == MT_F_RETURN_RING) {
    const int64_t bad = first_unusable((const float*)src, need / 4, field == MT_F_GOALS ? kMaxAngleBits : kMaxFiniteBits);
    if (bad >= 0)
      return fail(h, MT_ERR_INVALID_ARG, "mt_set: element " + std::to_string(bad) + " is NaN, infinite or (joint angles) beyond "
                                         "+-32768 degrees");
  }
  MT_ENTER(h);
  const dim3 g = grid_for(h->n), b(kBlock);
  const StepArgs& a = h->args;
  void* row = field == MT_F_TOTAL_REWARD ? (void*)a.total_reward : field == MT_F_DONE ? (void*)a.done :
              field == MT_F_EPISODES ? (void*)a.episodes : field == MT_F_LAST_RETURN ? (void*)a.last_return : nullptr;
  if (field == MT_F_LAST_RETURN || field == MT_F_RETURN_RING) {  // rows an in-place exchange may still be reading
    int rcg = order_behind_inplace_gather(h, h->stream);
    if (rcg) return rcg;
  }
  if (row) {  // single-row fields: env-major == SoA
    MT_HIP(h, hipMemcpyAsync(row, src, (size_t)need, hipMemcpyHostToDevice, h->stream));
    if (field == MT_F_DONE) {
      hipLaunchKernelGGL(done_bits_rebuild_kernel, g, b, 0, h->stream, a.done, h->n, a.done_bits);
      int rc = check_launch(h, "done_bits_rebuild_kernel");
      if (rc) return rc;
    }
  } else {
    int rc = ensure_staging(h, (size_t)need);
    if (rc) return rc;
    MT_HIP(h, hipMemcpyAsync(h->staging, src, (size_t)need, hipMemcpyHostToDevice, h->stream));
    if (field == MT_F_GOALS) {
      hipLaunchKernelGGL((env_major_to_soa<float>), g, b, 0, h->stream, (const float*)h->staging, h->D, h->n, a.goals, h->ld);
      h->args.flags &= ~kFlagWholeGoals;
    } else if (field == MT_F_POINTS)
      hipLaunchKernelGGL((env_major_to_soa<float>), g, b, 0, h->stream, (const float*)h->staging, 3 * h->K, h->n, a.points, h->ld);
    else if (field == MT_F_RETURN_RING)
      hipLaunchKernelGGL((env_major_to_soa<float>), g, b, 0, h->stream, (const float*)h->staging, (int)a.ring_slots, h->n, a.ring, h->ld);
    else
      hipLaunchKernelGGL(alive_pack, g, b, 0, h->stream, (const uint8_t*)h->staging, h->K, h->n, a.alive);
    rc = check_launch(h, "mt_set");
    if (rc) return rc;
  }
  MT_HIP(h, hipStreamSynchronize(h->stream));
  return MT_OK;
}

int mt_set_episode_base(mt_handle h, uint32_t episode0) {
  MT_REQUIRE(nullptr, h != nullptr, "handle is NULL");
  h->args.episode0 = episode0;
  return MT_OK;
}

int mt_device_ptr(mt_handle h, int field, void** ptr, int64_t* rows, int64_t* ld, int* dtype) {
  MT_REQUIRE(nullptr, h != nullptr, "handle is NULL");
  MT_REQUIRE(h, ptr != nullptr, "ptr is NULL");
  FieldInfo fi{};
  int rc = field_info(h, field, &fi);
  if (rc) return rc;
  if (h->reset_pending) {  // the caller is about to look at (or write) the rows themselves
    MT_ON_DEVICE(h, h->cfg.device);
    rc = flush_pending_reset(h);
    if (rc) return rc;
  }
  h->snap_valid = false;
  if (field == MT_F_GOALS) {
    // the caller can now write joint angles behind the library's back at any time: the host's knowledge that every
    // angle is a whole degree (kFlagWholeGoals: table look-ups for the pose a step starts from) ends here, for good
    h->goals_exposed = true;
    h->args.flags &= ~kFlagWholeGoals;
  }
  *ptr = fi.ptr;
  if (rows) *rows = fi.rows;
  if (ld) *ld = (field == MT_F_DONE_BITS) ? h->ld / 64 : h->ld;
  if (dtype) *dtype = fi.dtype;
  return MT_OK;
}

// ---- timing -----------------------------------------------------------------------------------
int mt_timer_start(mt_handle h) {
  MT_REQUIRE(nullptr, h != nullptr, "handle is NULL");
  MT_ENTER_TIMER(h);
  MT_HIP(h, hipEventRecord(h->ev0, h->stream));
  return MT_OK;
}

int mt_timer_stop(mt_handle h, float* elapsed_ms) {
  MT_REQUIRE(nullptr, h != nullptr, "handle is NULL");
  MT_REQUIRE(h, elapsed_ms != nullptr, "elapsed_ms is NULL");
  MT_ENTER_TIMER(h);
  MT_HIP(h, hipEventRecord(h->ev1, h->stream));
  MT_HIP(h, hipEventSynchronize(h->ev1));
  MT_HIP(h, hipEventElapsedTime(elapsed_ms, h->ev0, h->ev1));
  return MT_OK;
}

int mt_timer_stop_async(mt_handle h) {
  MT_REQUIRE(nullptr, h != nullptr, "handle is NULL");
  MT_ON_DEVICE(h, h->cfg.device);  // NOT a join: one end event per stream that may still carry work of the handle
  MT_HIP(h, hipEventRecord(h->ev1, h->stream));
  h->timer_ends = 1;
  if (h->forked)
    for (int c = 1; c < h->chains; ++c) {
      if (!h->chain_streams[c] || (int64_t)c * chain_span(h, h->chains) >= h->n) continue;
      if (!h->ev1c[c]) MT_HIP(h, hipEventCreateWithFlags(&h->ev1c[c], event_flags(true)));
      MT_HIP(h, hipEventRecord(h->ev1c[c], h->chain_streams[c]));
      h->timer_ends = c + 1;
    }
  h->timer_with_gather = h->gather_pending;  // an exchange begun on the side stream ends at ev_g1
  return MT_OK;
}

int mt_timer_read(mt_handle h, float* elapsed_ms) {
  MT_REQUIRE(nullptr, h != nullptr, "handle is NULL");
  MT_REQUIRE(h, elapsed_ms != nullptr, "elapsed_ms is NULL");
  if (h->timer_ends < 1) return fail(h, MT_ERR_STATE, "mt_timer_read without mt_timer_stop_async");
  MT_ON_DEVICE(h, h->cfg.device);
  float best = 0.f, v = 0.f;
  MT_HIP(h, hipEventSynchronize(h->ev1));
  MT_HIP(h, hipEventElapsedTime(&best, h->ev0, h->ev1));
  for (int c = 1; c < h->timer_ends; ++c) {
    if (!h->ev1c[c]) continue;
    MT_HIP(h, hipEventSynchronize(h->ev1c[c]));
    MT_HIP(h, hipEventElapsedTime(&v, h->ev0, h->ev1c[c]));
    best = v > best ? v : best;
  }
  if (h->timer_with_gather && h->ev_g1) {
    MT_HIP(h, hipEventSynchronize(h->ev_g1));
    if (hipEventElapsedTime(&v, h->ev0, h->ev_g1) == hipSuccess)
      best = v > best ? v : best;  // (an exchange that ended before the timer started reads negative: ignored)
    else
      (void)hipGetLastError();
  }
  *elapsed_ms = best;
  h->timer_ends = 0;
  return MT_OK;
}

// The lap pool holds at least `count` unused events.  Called at the BEGIN of a lap for the whole lap (its begin event and one
// end event per stream), ahead of the begin event's record: creating an event can stall the host for milliseconds (80 ms
// observed, once per process, inside the HIP runtime), and a host stall between a lap's last launch and the record of its
// end event would read as device time of the lap -- an event recorded on an idle stream completes at once.
static int lap_reserve(mt_handle h, size_t count) {
  while (h->lap_events.size() < h->lap_events_used + count) {
    hipEvent_t e = nullptr;
    MT_HIP(h, hipEventCreateWithFlags(&e, event_flags(true)));
    h->lap_events.push_back(e);
  }
  return MT_OK;
}

// one event of the lap pool, recorded on `stream`; its index
static int lap_event(mt_handle h, hipStream_t stream, uint32_t* index) {
  int rcr = lap_reserve(h, 1);
  if (rcr) return rcr;
  MT_HIP(h, hipEventRecord(h->lap_events[h->lap_events_used], stream));
  *index = (uint32_t)h->lap_events_used++;
  return MT_OK;
}

int mt_timer_lap_begin(mt_handle h) {
  MT_REQUIRE(nullptr, h != nullptr, "handle is NULL");
  if (h->lap_open) return fail(h, MT_ERR_STATE, "mt_timer_lap_begin: a lap is already open");
  // A lap starts behind everything queued before it.  While the chains are forked on the handle's own stream they stay
  // forked: one begin event per chain, the lap runs from the earliest of them to the latest end event (a join here would
  // put two cross-stream dependencies into the timed work for the sake of the stopwatch: 20 us per episode at 1 M arms).
  const bool per_chain = h->forked && h->lazy_chains && h->stream == h->own_stream;
  if (per_chain) {
    MT_ON_DEVICE(h, h->cfg.device);
    int rcf = flush_pending_reset(h);  // (per chain as well while forked)
    if (rcf) return rcf;
  } else {
    MT_ENTER(h);
  }
  mt_engine::LapRec rec{0, 1, 0, 0};
  int rc = lap_reserve(h, 2 * (size_t)mt_engine::kMaxChains);  // the whole lap's events exist before it begins
  if (rc) return rc;
  rc = lap_event(h, h->stream, &rec.begin);
  if (rc) return rc;
  if (per_chain && h->forked)
    for (int c = 1; c < h->chains; ++c) {
      if (!h->chain_streams[c] || (int64_t)c * chain_span(h, h->chains) >= h->n) continue;
      uint32_t idx;
      rc = lap_event(h, h->chain_streams[c], &idx);  // consecutive indices: begin, begin + 1, ...
      if (rc) return rc;
      ++rec.n_begin;
    }
  h->lap_recs.push_back(rec);
  h->lap_open = true;
  return MT_OK;
}

int mt_timer_lap_end(mt_handle h) {
  MT_REQUIRE(nullptr, h != nullptr, "handle is NULL");
  if (!h->lap_open) return fail(h, MT_ERR_STATE, "mt_timer_lap_end without mt_timer_lap_begin");
  MT_ON_DEVICE(h, h->cfg.device);  // NOT a join: while the chains are forked the lap ends when the last of them is done
  mt_engine::LapRec& rec = h->lap_recs.back();
  int rc = lap_event(h, h->stream, &rec.end0);
  if (rc) return rc;
  rec.n_end = 1;
  if (h->forked)
    for (int c = 1; c < h->chains; ++c) {
      if (!h->chain_streams[c] || (int64_t)c * chain_span(h, h->chains) >= h->n) continue;
      uint32_t idx;
      rc = lap_event(h, h->chain_streams[c], &idx);  // consecutive indices: end0, end0 + 1, ...
      if (rc) return rc;
      ++rec.n_end;
    }
  h->lap_open = false;
  return MT_OK;
}

// milliseconds of lap `rec`: from the earliest of its begin events to the latest of its end events
static int lap_ms(mt_handle h, const mt_engine::LapRec& rec, float* ms) {
  float best = 0.f;
  for (uint32_t k = 0; k < rec.n_end; ++k) {
    MT_HIP(h, hipEventSynchronize(h->lap_events[rec.end0 + k]));
    for (uint32_t b = 0; b < rec.n_begin; ++b) {
      float v = 0.f;
      MT_HIP(h, hipEventElapsedTime(&v, h->lap_events[rec.begin + b], h->lap_events[rec.end0 + k]));
      best = v > best ? v : best;
    }
  }
  *ms = best;
  return MT_OK;
}

int mt_timer_lap_times(mt_handle h, float* ms, int capacity, int* n_laps) {
  MT_REQUIRE(nullptr, h != nullptr, "handle is NULL");
  MT_REQUIRE(h, n_laps != nullptr && (ms != nullptr || capacity == 0), "NULL argument");
  if (h->lap_open) return fail(h, MT_ERR_STATE, "mt_timer_lap_times: a lap is still open");
  const size_t laps = h->lap_recs.size();
  *n_laps = (int)laps;
  if ((size_t)capacity < laps) return fail(h, MT_ERR_INVALID_ARG, "mt_timer_lap_times: capacity is smaller than the number of laps");
  MT_ON_DEVICE(h, h->cfg.device);
  for (size_t i = 0; i < laps; ++i) {
    int rc = lap_ms(h, h->lap_recs[i], &ms[i]);
    if (rc) return rc;
  }
  h->lap_recs.clear();
  h->lap_events_used = 0;
  return MT_OK;
}

int mt_timer_laps_total(mt_handle h, float* total_ms, int* n_laps) {
  MT_REQUIRE(nullptr, h != nullptr, "handle is NULL");
  MT_REQUIRE(h, total_ms != nullptr, "total_ms is NULL");
  if (h->lap_open) return fail(h, MT_ERR_STATE, "mt_timer_laps_total: a lap is still open");
  MT_ON_DEVICE(h, h->cfg.device);
  double sum = 0.0;
  for (const auto& rec : h->lap_recs) {
    float v = 0.f;
    int rc = lap_ms(h, rec, &v);
    if (rc) return rc;
    sum += v;
  }
  *total_ms = (float)sum;
  if (n_laps) *n_laps = (int)h->lap_recs.size();
  h->lap_recs.clear();
  h->lap_events_used = 0;
  return MT_OK;
}

// ---- stateless helpers ------------------------------------------------------------------------
// They have no handle to keep buffers on, so the device scratch they need is cached per device and only ever
// grows; one lock serialises them (they are small synchronous calls on the null stream).
namespace {
struct Scratch {
  void* p = nullptr;
  size_t bytes = 0;
};
std::mutex g_scratch_mu;
Scratch g_scratch[64];

int scratch_reserve(int device, size_t bytes, void** out) {  // g_scratch_mu held, `device` current
  if (device < 0 || device >= 64) return fail(nullptr, MT_ERR_INVALID_ARG, "device ordinal out of range");
  Scratch& sc = g_scratch[device];
  if (sc.bytes < bytes) {
    if (sc.p) (void)hipFree(sc.p);
    sc.p = nullptr;
    sc.bytes = 0;
    const size_t want = align_up(bytes, (size_t)1 << 16);
    if (hipMalloc(&sc.p, want) != hipSuccess) {
      (void)hipGetLastError();
      sc.p = nullptr;
      return fail(nullptr, MT_ERR_ALLOC, "hipMalloc of the helper scratch failed");
    }
    sc.bytes = want;
  }
  *out = sc.p;
  return MT_OK;
}
}  // namespace

int mt_fk_batch(int device, const float* dh_table, int dof, int mode, const float* angles, int angles_in_radians,
                int64_t n, float* out_mat16) {
  MT_REQUIRE(nullptr, dh_table && angles && out_mat16, "NULL argument");
  MT_REQUIRE(nullptr, dof >= 1 && dof <= MT_MAX_DOF && mode >= 0 && mode <= dof, "dof/mode out of range");
  MT_REQUIRE(nullptr, n >= 1, "n must be >= 1");
  MT_REQUIRE(nullptr, first_unusable(dh_table, (int64_t)dof * 4, kMaxFiniteBits) < 0 && first_unusable(angles, n * dof, kMaxFiniteBits) < 0,
             "mt_fk_batch: NaN or infinity in dh_table / angles");
  MT_ON_DEVICE(nullptr, device);
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  const size_t in_b = align_up((size_t)n * dof * 4, 256);
  void* base = nullptr;
  int rc = scratch_reserve(device, in_b + (size_t)n * 64, &base);
  if (rc) return rc;
  float* d_in = (float*)base;
  float* d_out = (float*)((char*)base + in_b);
  FkArgs a{d_in, d_out, n, dof, mode, angles_in_radians, make_dh(dh_table, dof)};
  hipError_t e = hipMemcpy(d_in, angles, (size_t)n * dof * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(fk_kernel, grid_for(n), dim3(kBlock), 0, 0, a);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(out_mat16, d_out, (size_t)n * 64, hipMemcpyDeviceToHost);
  if (e != hipSuccess) rc = fail(nullptr, MT_ERR_HIP, std::string("mt_fk_batch: ") + hipGetErrorString(e));
  return rc;
}

int mt_route_trace(int device, const float* dh_table, int dof, int substeps, const float* prev, const float* action,
                   int64_t n, float* out) {
  MT_REQUIRE(nullptr, dh_table && prev && action && out, "NULL argument");
  MT_REQUIRE(nullptr, dof >= 2 && dof <= MT_MAX_DOF && substeps >= 2, "dof/substeps out of range");
  MT_REQUIRE(nullptr, n >= 1 && n * substeps < ((int64_t)1 << 31), "n out of range");
  MT_REQUIRE(nullptr, first_unusable(dh_table, (int64_t)dof * 4, kMaxFiniteBits) < 0 && first_unusable(prev, n * dof, kMaxFiniteBits) < 0 &&
                          first_unusable(action, n * dof, kMaxFiniteBits) < 0,
             "mt_route_trace: NaN or infinity in dh_table / prev / action");
  MT_ON_DEVICE(nullptr, device);
  const size_t in_b = align_up((size_t)n * dof * 4, 256), out_b = (size_t)n * substeps * dof * 3 * 4;
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  void* base = nullptr;
  int rc = scratch_reserve(device, 2 * in_b + out_b, &base);
  if (rc) return rc;
  char* d = (char*)base;
  TraceArgs a{(const float*)d, (const float*)(d + in_b), (float*)(d + 2 * in_b), n, dof, substeps, make_dh(dh_table, dof)};
  hipError_t e = hipMemcpy(d, prev, (size_t)n * dof * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d + in_b, action, (size_t)n * dof * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(route_trace_kernel, grid_for(n * substeps), dim3(kBlock), 0, 0, a);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(out, d + 2 * in_b, out_b, hipMemcpyDeviceToHost);
  if (e != hipSuccess) rc = fail(nullptr, MT_ERR_HIP, std::string("mt_route_trace: ") + hipGetErrorString(e));
  return rc;
}

int mt_r_theta_batch(int device, const float* v1, const float* v2, int64_t n, float* out_r_theta) {
  MT_REQUIRE(nullptr, v1 && v2 && out_r_theta, "NULL argument");
  MT_REQUIRE(nullptr, n >= 1, "n must be >= 1");
  MT_REQUIRE(nullptr, first_unusable(v1, 3 * n, kMaxFiniteBits) < 0 && first_unusable(v2, 3 * n, kMaxFiniteBits) < 0,
             "mt_r_theta_batch: NaN or infinity in v1 / v2");
  MT_ON_DEVICE(nullptr, device);
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  void* base = nullptr;
  int rc = scratch_reserve(device, (size_t)n * 8 * 4, &base);
  if (rc) return rc;
  float* d = (float*)base;
  float *d1 = d, *d2 = d + 3 * n, *dout = d + 6 * n;
  hipError_t e = hipMemcpy(d1, v1, (size_t)n * 12, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d2, v2, (size_t)n * 12, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(r_theta_kernel, grid_for(n), dim3(kBlock), 0, 0, d1, d2, n, dout);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(out_r_theta, dout, (size_t)n * 8, hipMemcpyDeviceToHost);
  if (e != hipSuccess) rc = fail(nullptr, MT_ERR_HIP, std::string("mt_r_theta_batch: ") + hipGetErrorString(e));
  return rc;
}

// The streaming yardstick of the step's own access shape (kernels.h: stream_probe_kernel): `reps` passes over n_envs envs
// after 3 untimed ones, on a private stream, HIP events around them.  Allocates and frees its own buffers.
int mt_stream_probe(int device, int dof, int n_targets, int64_t n_envs, int reps, float* us_per_pass, int64_t* bytes_per_pass) {
  MT_REQUIRE(nullptr, us_per_pass != nullptr && bytes_per_pass != nullptr, "NULL argument");
  MT_REQUIRE(nullptr, n_envs >= 256 && n_envs <= ((int64_t)1 << 30) - 256 && reps >= 1 && reps <= 100000, "n_envs / reps out of range");
  const bool d4k7 = dof == 4 && n_targets == 7, d7k7 = dof == 7 && n_targets == 7, d4k10 = dof == 4 && n_targets == 10;
  if (!(d4k7 || d7k7 || d4k10))
    return fail(nullptr, MT_ERR_UNSUPPORTED, "mt_stream_probe: built for (dof, targets) = (4, 7), (7, 7), (4, 10)");
  MT_ON_DEVICE(nullptr, device);
  const int64_t ld = (int64_t)align_up((size_t)n_envs, 256) + 256;  // the arena's row pitch (mt_create)
  const int RD = dof + 3 * n_targets + 2, WN = 3 * n_targets + 4;
  float *state = nullptr, *out = nullptr;
  uint8_t* bytes = nullptr;
  hipStream_t st = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipError_t e = hipMalloc(&state, (size_t)RD * ld * 4);
  if (e == hipSuccess) e = hipMalloc(&out, (size_t)WN * ld * 4);
  if (e == hipSuccess) e = hipMalloc(&bytes, (size_t)ld);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreate(&e0);
  if (e == hipSuccess) e = hipEventCreate(&e1);
  if (e == hipSuccess) e = hipMemsetAsync(state, 0, (size_t)RD * ld * 4, st);
  float ms = 0.f;
  if (e == hipSuccess) {
    auto pass = [&]() {
      if (d4k7)
        hipLaunchKernelGGL((stream_probe_kernel<4, 7>), grid_for(n_envs), dim3(kBlock), 0, st, state, out, bytes, n_envs, ld);
      else if (d7k7)
        hipLaunchKernelGGL((stream_probe_kernel<7, 7>), grid_for(n_envs), dim3(kBlock), 0, st, state, out, bytes, n_envs, ld);
      else
        hipLaunchKernelGGL((stream_probe_kernel<4, 10>), grid_for(n_envs), dim3(kBlock), 0, st, state, out, bytes, n_envs, ld);
    };
    for (int w = 0; w < 3; ++w) pass();
    e = hipEventRecord(e0, st);
    for (int r = 0; r < reps; ++r) pass();
    if (e == hipSuccess) e = hipEventRecord(e1, st);
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (st) {
    (void)hipStreamSynchronize(st);
    (void)hipStreamDestroy(st);
  }
  if (state) (void)hipFree(state);
  if (out) (void)hipFree(out);
  if (bytes) (void)hipFree(bytes);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail(nullptr, e == hipErrorOutOfMemory ? MT_ERR_ALLOC : MT_ERR_HIP, std::string("mt_stream_probe: ") + hipGetErrorString(e));
  }
  *us_per_pass = ms * 1e3f / (float)reps;
  *bytes_per_pass = (int64_t)(8 * dof + 24 * n_targets + 33) * n_envs;
  return MT_OK;
}

}  // extern "C"

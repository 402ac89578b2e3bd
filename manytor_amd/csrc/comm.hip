// The one exchange of the multi-GPU path: all-gather of per-env returns over RCCL / xGMI.
//
// The reference has no distributed layer (its "multi-env" is a serial Python loop, manytor.py:115-122); what is
// gathered here is what test_multi.py:32 prints per env.  One process per GPU; the communicator belongs to the
// handle and runs on the handle's stream, reading the arena row directly (no staging for equal shards).
// librccl.so is resolved at run time with dlopen, so the library itself links only libamdhip64.
#include <dlfcn.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "engine_internal.h"

using namespace mt;

namespace {

// The handful of RCCL entry points used, with the types of <rccl/rccl.h> (ncclResult_t and the enums are ints;
// ncclUniqueId is a 128-byte struct passed BY VALUE to ncclCommInitRank).
struct UniqueId {
  char internal[MT_UNIQUE_ID_BYTES];
};
using Comm = void*;
constexpr int kNcclSuccess = 0;
constexpr int kNcclInt64 = 4;    // ncclInt64
constexpr int kNcclFloat32 = 7;  // ncclFloat32

struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
  int (*CommDestroy)(Comm) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, Comm, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  std::string where, error;
};

std::mutex g_rccl_mu;
Rccl g_rccl;

template <typename F>
bool sym(void* lib, const char* name, F& fn) {
  fn = reinterpret_cast<F>(dlsym(lib, name));
  return fn != nullptr;
}

// Resolution order: $MT_RCCL_LIB (the Python host points it at the copy that shares torch's HIP runtime), a copy the
// process has already loaded, then the system one.
Rccl* rccl() {
  std::lock_guard<std::mutex> lock(g_rccl_mu);
  if (g_rccl.lib) return &g_rccl;
  std::vector<std::pair<std::string, int>> tries;
  if (const char* env = std::getenv("MT_RCCL_LIB"))
    if (*env) tries.push_back({env, RTLD_NOW | RTLD_LOCAL});
  tries.push_back({"librccl.so.1", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD});
  tries.push_back({"librccl.so.1", RTLD_NOW | RTLD_LOCAL});
  tries.push_back({"librccl.so", RTLD_NOW | RTLD_LOCAL});
  std::string errs;
  for (auto& t : tries) {
    void* lib = dlopen(t.first.c_str(), t.second);
    if (!lib) {
      const char* e = dlerror();
      errs += t.first + ": " + (e ? e : "not loaded") + "; ";
      continue;
    }
    Rccl r;
    if (sym(lib, "ncclGetUniqueId", r.GetUniqueId) && sym(lib, "ncclCommInitRank", r.CommInitRank) &&
        sym(lib, "ncclCommDestroy", r.CommDestroy) && sym(lib, "ncclAllGather", r.AllGather) &&
        sym(lib, "ncclGetErrorString", r.GetErrorString)) {
      r.lib = lib;
      r.where = t.first;
      g_rccl = r;
      return &g_rccl;
    }
    errs += t.first + ": missing nccl* symbols; ";
    dlclose(lib);
  }
  g_rccl.error = errs;
  return nullptr;
}

}  // namespace

struct mt_comm {
  Comm comm = nullptr;
  int rank = 0, world = 1;
  std::vector<int64_t> counts;  // envs of every rank
  int64_t total = 0, cmax = 0;
  bool equal = true;
  float* stage = nullptr;       // ragged shards only: [world + 1][cmax] (slot `world` = padded send buffer)
};

#define MT_NCCL(h, r, call)                                                                              \
  do {                                                                                                   \
    int e__ = (call);                                                                                    \
    if (e__ != kNcclSuccess)                                                                             \
      return fail(h, MT_ERR_HIP, std::string(#call) + ": " + ((r)->GetErrorString ? (r)->GetErrorString(e__) : "?")); \
  } while (0)

void mt_gather_release(mt_handle h) {
  if (!h) return;
  if (h->side_stream) {
    (void)hipStreamSynchronize(h->side_stream);
    (void)hipStreamDestroy(h->side_stream);
  }
  if (h->ev_snap) (void)hipEventDestroy(h->ev_snap);
  if (h->ev_g0) (void)hipEventDestroy(h->ev_g0);
  if (h->ev_g1) (void)hipEventDestroy(h->ev_g1);
  for (int p = 0; p < 2; ++p) {
    if (h->ev_gdone[p]) (void)hipEventDestroy(h->ev_gdone[p]);
    h->ev_gdone[p] = nullptr;
    h->snap_free_known[p] = true;
  }
  h->snap_next = 0;
  h->snap_valid = false;
  if (h->snap) (void)hipFree(h->snap);
  if (h->reduce_scratch) (void)hipFree(h->reduce_scratch);
  h->reduce_scratch = nullptr;
  h->reduce_scratch_bytes = 0;
  h->side_stream = nullptr;
  h->ev_snap = h->ev_g0 = h->ev_g1 = nullptr;
  h->snap = nullptr;
  h->gather_pending = false;
}

void mt_comm_release(mt_handle h) {
  if (!h || !h->comm) return;
  mt_comm* c = h->comm;
  h->comm = nullptr;
  if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
  if (c->stage) (void)hipFree(c->stage);
  delete c;
}

extern "C" {

int mt_comm_unique_id(void* id_out) {
  MT_REQUIRE(nullptr, id_out != nullptr, "id_out is NULL");
  Rccl* r = rccl();
  if (!r) return fail(nullptr, MT_ERR_UNSUPPORTED, "RCCL is not available: " + g_rccl.error);
  UniqueId id;
  MT_NCCL(nullptr, r, r->GetUniqueId(&id));
  std::memcpy(id_out, id.internal, MT_UNIQUE_ID_BYTES);
  return MT_OK;
}

int mt_comm_init(mt_handle h, const void* unique_id, int rank, int world_size) {
  MT_REQUIRE(nullptr, h != nullptr, "handle is NULL");
  MT_REQUIRE(h, unique_id != nullptr, "unique_id is NULL");
  MT_REQUIRE(h, world_size >= 1 && rank >= 0 && rank < world_size, "rank / world_size out of range");
  if (h->comm) return fail(h, MT_ERR_STATE, "mt_comm_init: the handle already has a communicator");
  Rccl* r = rccl();
  if (!r) return fail(h, MT_ERR_UNSUPPORTED, "RCCL is not available: " + g_rccl.error);
  MT_ENTER(h);
  mt_comm* c = new (std::nothrow) mt_comm();
  if (!c) return fail(h, MT_ERR_ALLOC, "out of host memory");
  c->rank = rank;
  c->world = world_size;
  UniqueId id;
  std::memcpy(id.internal, unique_id, MT_UNIQUE_ID_BYTES);
  h->comm = c;  // from here on mt_comm_release cleans up
#define MT_FAIL_INIT(stmt) \
  do {                     \
    int rc__ = (stmt);     \
    if (rc__ != MT_OK) {   \
      mt_comm_release(h);  \
      return rc__;         \
    }                      \
  } while (0)
  MT_FAIL_INIT([&]() -> int {
    MT_NCCL(h, r, r->CommInitRank(&c->comm, world_size, id, rank));
    // shard sizes of all ranks: one tiny all-gather through the staging buffer, once
    int rc = MT_OK;
    int64_t* d_counts = nullptr;
    MT_HIP(h, hipMalloc(&d_counts, sizeof(int64_t) * (size_t)(world_size + 1)));
    const int64_t mine = h->n;
    hipError_t e = hipMemcpyAsync(d_counts + world_size, &mine, sizeof mine, hipMemcpyHostToDevice, h->stream);
    int ne = kNcclSuccess;
    if (e == hipSuccess) ne = r->AllGather(d_counts + world_size, d_counts, 1, kNcclInt64, c->comm, h->stream);
    c->counts.assign((size_t)world_size, 0);
    if (e == hipSuccess && ne == kNcclSuccess)
      e = hipMemcpyAsync(c->counts.data(), d_counts, sizeof(int64_t) * (size_t)world_size, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess && ne == kNcclSuccess) e = hipStreamSynchronize(h->stream);
    (void)hipFree(d_counts);
    if (ne != kNcclSuccess) rc = fail(h, MT_ERR_HIP, std::string("ncclAllGather(counts): ") + r->GetErrorString(ne));
    else if (e != hipSuccess) rc = fail(h, MT_ERR_HIP, std::string("mt_comm_init: ") + hipGetErrorString(e));
    return rc;
  }());
#undef MT_FAIL_INIT
  c->total = 0;
  c->cmax = 0;
  for (int64_t v : c->counts) {
    c->total += v;
    if (v > c->cmax) c->cmax = v;
  }
  c->equal = true;
  for (int64_t v : c->counts) c->equal &= (v == c->cmax);
  if (c->counts[(size_t)rank] != h->n) {
    mt_comm_release(h);
    return fail(h, MT_ERR_HIP, "mt_comm_init: the gathered shard sizes do not contain this rank's own");
  }
  if (!c->equal) {
    if (hipMalloc(&c->stage, sizeof(float) * (size_t)(world_size + 1) * (size_t)c->cmax) != hipSuccess) {
      (void)hipGetLastError();
      mt_comm_release(h);
      return fail(h, MT_ERR_ALLOC, "hipMalloc of the ragged-gather staging buffer failed");
    }
  }
  return MT_OK;
}

int mt_comm_destroy(mt_handle h) {
  MT_REQUIRE(nullptr, h != nullptr, "handle is NULL");
  MT_ENTER(h);
  MT_HIP(h, hipStreamSynchronize(h->stream));
  if (h->gather_pending) {
    MT_HIP(h, hipStreamSynchronize(h->side_stream));
    h->gather_pending = false;
  }
  mt_comm_release(h);
  return MT_OK;
}

int mt_comm_total_envs(mt_handle h, int64_t* total) {
  MT_REQUIRE(nullptr, h != nullptr, "handle is NULL");
  MT_REQUIRE(h, total != nullptr, "total is NULL");
  *total = h->comm ? h->comm->total : h->n;
  return MT_OK;
}

// Which device row does (field, row) name?  NULL + error on a bad request.
static const float* gather_source(mt_handle h, int field, int row, int* rc) {
  const float* src = nullptr;
  int rows = 1;
  *rc = MT_OK;
  switch (field) {
    case MT_F_TOTAL_REWARD: src = h->args.total_reward; break;
    case MT_F_LAST_RETURN: src = h->args.last_return; break;
    case MT_F_RETURN_RING:
      src = h->args.ring;
      rows = (int)h->args.ring_slots;
      if (!src) {
        *rc = fail(h, MT_ERR_STATE, "MT_F_RETURN_RING: the handle was created with return_ring = 0");
        return nullptr;
      }
      break;
    default:
      *rc = fail(h, MT_ERR_INVALID_ARG, "mt_gather_returns: field must be TOTAL_REWARD, LAST_RETURN or RETURN_RING");
      return nullptr;
  }
  if (row < 0 || row >= rows) {
    *rc = fail(h, MT_ERR_INVALID_ARG, "row out of range");
    return nullptr;
  }
  return src + (int64_t)row * h->ld;
}

// The snapshot of a return row as a kernel: a device-to-device hipMemcpyAsync of 0.5-4 MB is a blit kernel between two
// barriers of the runtime's own (12 us of idle queue ahead of it and 7 behind it in the trace of a 131 072-env shard), a
// launch of this kernel follows the step kernels like any other launch.
__global__ __launch_bounds__(kBlock) void row_copy_kernel(float* __restrict__ dst, const float* __restrict__ src, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) dst[i] = src[i];
}
static int row_copy(mt_handle h, float* dst, const float* src, int64_t n, hipStream_t stream) {
  hipLaunchKernelGGL(row_copy_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, dst, src, n);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(h, MT_ERR_HIP, std::string("row_copy_kernel launch: ") + hipGetErrorString(e));
  return MT_OK;
}

// The exchange itself: `src` = this rank's n floats (an arena row or its snapshot), on `stream`.
static int gather_on_stream(mt_handle h, const float* src, float* dst, int64_t dst_elems, hipStream_t stream) {
  mt_comm* c = h->comm;
  if (!c || c->world == 1) {
    MT_REQUIRE(h, dst_elems == h->n, "dst_elems must be the total number of envs");
    return row_copy(h, dst, src, h->n, stream);  // (one rank: the "exchange" is a device copy; as a kernel, see row_copy)
  }
  MT_REQUIRE(h, dst_elems == c->total, "dst_elems must be the total number of envs over all ranks");
  Rccl* r = rccl();
  if (!r) return fail(h, MT_ERR_UNSUPPORTED, "RCCL is not available: " + g_rccl.error);
  if (c->equal) {  // straight from the source row into the caller's buffer
    MT_NCCL(h, r, r->AllGather(src, dst, (size_t)h->n, kNcclFloat32, c->comm, stream));
    return MT_OK;
  }
  // ragged shards: pad to the largest shard, gather, then close the gaps
  float* send = c->stage + (size_t)c->world * (size_t)c->cmax;
  MT_HIP(h, hipMemsetAsync(send, 0, sizeof(float) * (size_t)c->cmax, stream));
  MT_HIP(h, hipMemcpyAsync(send, src, sizeof(float) * (size_t)h->n, hipMemcpyDeviceToDevice, stream));
  MT_NCCL(h, r, r->AllGather(send, c->stage, (size_t)c->cmax, kNcclFloat32, c->comm, stream));
  int64_t off = 0;
  for (int k = 0; k < c->world; ++k) {
    MT_HIP(h, hipMemcpyAsync(dst + off, c->stage + (size_t)k * (size_t)c->cmax, sizeof(float) * (size_t)c->counts[(size_t)k],
                             hipMemcpyDeviceToDevice, stream));
    off += c->counts[(size_t)k];
  }
  return MT_OK;
}

// Block partials of {sum, min, max, done count} over one return row: a fixed assignment of envs to threads and a fixed
// tree inside the block, double accumulation -- the same bits whenever the same numbers come in.
constexpr int kReduceBlocks = 256;
struct ReducePartial {
  double sum, min, max, done;
};
__global__ __launch_bounds__(kBlock) void reduce_returns_kernel(const float* __restrict__ src,
                                                                const uint8_t* __restrict__ done, int64_t n,
                                                                ReducePartial* __restrict__ out) {
  __shared__ ReducePartial sh[kBlock];
  ReducePartial p{0.0, 1.0e300, -1.0e300, 0.0};
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const double v = (double)src[i];
    p.sum += v;
    p.min = v < p.min ? v : p.min;
    p.max = v > p.max ? v : p.max;
    p.done += done[i] != 0 ? 1.0 : 0.0;
  }
  sh[threadIdx.x] = p;
  __syncthreads();
  for (int w = kBlock / 2; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) {
      const ReducePartial o = sh[threadIdx.x + w];
      ReducePartial& m = sh[threadIdx.x];
      m.sum += o.sum;
      m.min = o.min < m.min ? o.min : m.min;
      m.max = o.max > m.max ? o.max : m.max;
      m.done += o.done;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = sh[0];
}

// A gather begun with mt_gather_returns_begin and not yet waited for: order `stream` behind it.
// (An exchange that has already FINISHED -- the normal case one episode later -- needs no barrier packet in the queue: the
// event is asked first.  A stream wait costs a few microseconds of device time even on a signalled event, and the episode
// end of a 131 072-env shard is ~40 us long: tools/region_timeline.py, profiles/r04_variants.md section 4.)
static int order_behind_pending_gather(mt_handle h, hipStream_t stream) {
  if (!h->gather_pending) return MT_OK;
  if (hipEventQuery(h->ev_g1) == hipSuccess) return MT_OK;
  (void)hipGetLastError();  // hipErrorNotReady is not an error of ours
  MT_HIP(h, hipStreamWaitEvent(stream, h->ev_g1, 0));
  return MT_OK;
}

int mt_gather_returns(mt_handle h, int field, int row, float* dst, int64_t dst_elems) {
  MT_REQUIRE(nullptr, h != nullptr, "handle is NULL");
  MT_REQUIRE(h, dst != nullptr, "dst is NULL");
  int rc;
  const float* src = gather_source(h, field, row, &rc);
  if (!src) return rc;
  MT_ENTER(h);
  rc = order_behind_pending_gather(h, h->stream);  // the two forms share the communicator and its staging buffer
  if (rc) return rc;
  return gather_on_stream(h, src, dst, dst_elems, h->stream);
}

int mt_reduce_returns(mt_handle h, int field, int row, mt_return_stats* out) {
  MT_REQUIRE(nullptr, h != nullptr, "handle is NULL");
  MT_REQUIRE(h, out != nullptr, "out is NULL");
  int rc;
  const float* src = gather_source(h, field, row, &rc);
  if (!src) return rc;
  MT_ENTER(h);
  mt_comm* c = h->comm;
  const int world = (c && c->world > 1) ? c->world : 1;
  // device scratch: [kReduceBlocks] block partials, then [world + 1] per-rank records (slot `world` = this rank's)
  const size_t bytes = sizeof(ReducePartial) * (size_t)(kReduceBlocks + world + 1);
  if (h->reduce_scratch_bytes < bytes) {  // kept on the handle: no hipMalloc / hipFree (= device-wide sync) per call
    if (h->reduce_scratch) {
      MT_HIP(h, hipStreamSynchronize(h->stream));
      (void)hipFree(h->reduce_scratch);
      h->reduce_scratch = nullptr;
      h->reduce_scratch_bytes = 0;
    }
    if (hipMalloc(&h->reduce_scratch, bytes) != hipSuccess) {
      (void)hipGetLastError();
      return fail(h, MT_ERR_ALLOC, "hipMalloc of the reduce scratch failed");
    }
    h->reduce_scratch_bytes = bytes;
  }
  ReducePartial* d = (ReducePartial*)h->reduce_scratch;
  std::vector<ReducePartial> host((size_t)kReduceBlocks + (size_t)world);
  auto body = [&]() -> int {
    hipLaunchKernelGGL(reduce_returns_kernel, dim3(kReduceBlocks), dim3(kBlock), 0, h->stream, src, h->args.done, h->n, d);
    MT_HIP(h, hipGetLastError());
    MT_HIP(h, hipMemcpyAsync(host.data(), d, sizeof(ReducePartial) * kReduceBlocks, hipMemcpyDeviceToHost, h->stream));
    MT_HIP(h, hipStreamSynchronize(h->stream));
    ReducePartial mine{0.0, 1.0e300, -1.0e300, 0.0};
    for (int b = 0; b < kReduceBlocks; ++b) {  // fixed order
      const ReducePartial& p = host[(size_t)b];
      mine.sum += p.sum;
      mine.min = p.min < mine.min ? p.min : mine.min;
      mine.max = p.max > mine.max ? p.max : mine.max;
      mine.done += p.done;
    }
    ReducePartial all = mine;
    int64_t count = h->n;
    if (world > 1) {
      Rccl* r = rccl();
      if (!r) return fail(h, MT_ERR_UNSUPPORTED, "RCCL is not available: " + g_rccl.error);
      rc = order_behind_pending_gather(h, h->stream);  // one communicator: never two exchanges in flight
      if (rc) return rc;
      ReducePartial* recs = d + kReduceBlocks;
      MT_HIP(h, hipMemcpyAsync(recs + world, &mine, sizeof mine, hipMemcpyHostToDevice, h->stream));
      static_assert(sizeof(ReducePartial) == 4 * sizeof(int64_t), "exchanged as four 64-bit words per rank");
      MT_NCCL(h, r, r->AllGather(recs + world, recs, 4, kNcclInt64, c->comm, h->stream));
      MT_HIP(h, hipMemcpyAsync(host.data() + kReduceBlocks, recs, sizeof(ReducePartial) * (size_t)world,
                               hipMemcpyDeviceToHost, h->stream));
      MT_HIP(h, hipStreamSynchronize(h->stream));
      all = ReducePartial{0.0, 1.0e300, -1.0e300, 0.0};
      for (int k = 0; k < world; ++k) {  // rank order: the same result on every rank
        const ReducePartial& p = host[(size_t)kReduceBlocks + (size_t)k];
        all.sum += p.sum;
        all.min = p.min < all.min ? p.min : all.min;
        all.max = p.max > all.max ? p.max : all.max;
        all.done += p.done;
      }
      count = c->total;
    }
    out->sum = all.sum;
    out->min = all.min;
    out->max = all.max;
    out->count = count;
    out->done = (int64_t)all.done;
    return MT_OK;
  };
  return body();
}

static int gather_begin_impl(mt_handle h, int field, int row, float* dst, int64_t dst_elems, bool inplace) {
  MT_REQUIRE(nullptr, h != nullptr, "handle is NULL");
  MT_REQUIRE(h, dst != nullptr, "dst is NULL");
  int rc;
  const float* src = gather_source(h, field, row, &rc);
  if (!src) return rc;
  // A snapshot gather begun while mt_rollout's chains are forked takes the snapshot PER CHAIN: every chain copies its own
  // env range behind its own last step, the exchange waits for all of them, and the chains stay forked -- a per-chain
  // reset queued next then runs beside the exchange instead of behind a join.  Everything else joins first.
  const bool per_chain = !inplace && h->forked && h->lazy_chains && h->stream == h->own_stream;
  MT_ON_DEVICE(h, h->cfg.device);
  rc = mt::flush_pending_reset(h);  // (a reset deferred into the next mt_rollout changes the rows this call reads)
  if (rc) return rc;
  // mt_rollout's last launch already stored the returns to the snapshot row and nothing has touched them since: no copy
  const bool have_snap = !inplace && h->snap_valid && src == h->args.total_reward && h->snap != nullptr;
  h->snap_valid = false;
  if (!per_chain) {
    rc = mt::join_chains(h);
    if (rc) return rc;
  }
  if (!h->snap) {  // first use: the side stream, its events, and the snapshot row (`snap` is set last: all or nothing)
    hipError_t e = hipStreamCreateWithFlags(&h->side_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_snap, mt::event_flags(false));
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_g0, mt::event_flags(true));
    if (e == hipSuccess) e = hipEventCreate(&h->ev_g1);  // (default: mt_gather_returns_wait(host) hands dst to the caller behind it)
    for (int p = 0; p < 2 && e == hipSuccess; ++p) e = hipEventCreateWithFlags(&h->ev_gdone[p], mt::event_flags(false));
    if (e == hipSuccess) e = hipMalloc(&h->snap, sizeof(float) * 2 * (size_t)h->n);  // two rows: see engine_internal.h
    if (e != hipSuccess) {
      (void)hipGetLastError();
      mt_gather_release(h);  // whatever was created goes again: the next call starts from scratch
      return fail(h, e == hipErrorOutOfMemory ? MT_ERR_ALLOC : MT_ERR_HIP,
                  std::string("mt_gather_returns_begin: set-up of the side stream failed: ") + hipGetErrorString(e));
    }
  }
  // Exchanges run one after the other on the side stream (one communicator, one staging buffer): nothing on the handle's
  // stream has to wait for the previous one -- unless it reads an arena row in place (the writers of that row are ordered
  // behind `the pending in-place exchange`, of which there is one) -- and the snapshot row of THIS exchange is free once the
  // exchange before the previous one, which read it, has finished.
  if (h->gather_pending && h->gather_inplace) {
    rc = order_behind_pending_gather(h, h->stream);
    if (rc) return rc;
  }
  const int p = h->snap_next;
  float* const snap = h->snap_row(p);
  auto row_is_free = [&](hipStream_t s) -> int {  // before a copy into the row on stream `s`
    if (!h->snap_free_known[p]) {
      if (hipEventQuery(h->ev_gdone[p]) == hipSuccess) {
        h->snap_free_known[p] = true;
      } else {
        (void)hipGetLastError();
        MT_HIP(h, hipStreamWaitEvent(s, h->ev_gdone[p], 0));
      }
    }
    return MT_OK;
  };
  if (per_chain) {
    const int64_t per = (h->n + h->chains - 1) / h->chains, span = (per + 255) / 256 * 256;  // engine.hip: chain_span
    for (int c = 0; c < h->chains; ++c) {
      const int64_t off = (int64_t)c * span;
      if (off >= h->n) continue;
      hipStream_t sc = c == 0 ? h->stream : h->chain_streams[c];
      const int64_t cnt = std::min(span, h->n - off);
      if (!have_snap) {
        rc = row_is_free(sc);
        if (rc == MT_OK) rc = row_copy(h, snap + off, src + off, cnt, sc);
        if (rc) return rc;
      }
      if (c > 0) {
        if (!h->ev_join[c]) return fail(h, MT_ERR_STATE, "mt_gather_returns_begin: chain without its event");
        // (ev_join[c] is free while the chains are forked: a join records it afresh)
        MT_HIP(h, hipEventRecord(h->ev_join[c], sc));
        MT_HIP(h, hipStreamWaitEvent(h->side_stream, h->ev_join[c], 0));
      }
    }
    src = snap;
  } else if (!inplace) {
    if (!have_snap) {
      rc = row_is_free(h->stream);
      if (rc == MT_OK) rc = row_copy(h, snap, src, h->n, h->stream);
      if (rc) return rc;
    }
    src = snap;
  }
  MT_HIP(h, hipEventRecord(h->ev_snap, h->stream));
  MT_HIP(h, hipStreamWaitEvent(h->side_stream, h->ev_snap, 0));
  MT_HIP(h, hipEventRecord(h->ev_g0, h->side_stream));
  rc = gather_on_stream(h, src, dst, dst_elems, h->side_stream);
  if (rc) return rc;
  MT_HIP(h, hipEventRecord(h->ev_g1, h->side_stream));
  h->gather_pending = true;
  h->gather_inplace = inplace;
  if (!inplace) {
    MT_HIP(h, hipEventRecord(h->ev_gdone[p], h->side_stream));
    h->snap_free_known[p] = false;
    h->snap_next = p ^ 1;
    // The row the next mt_rollout's last launch will write was read by the exchange BEFORE this one: let the host learn
    // that it has finished -- by waiting for it if the host has run that far ahead of the device (an unfenced episode loop
    // enqueues an episode in a third of the time the device needs for it).  The device is not idle meanwhile: it still has
    // the whole episode that ends in this exchange in its queue.  Without this a snapshot launch of its own and a stream
    // wait sit between the episodes (21 us of a 130 us episode at 131 072 envs: tools/shard_timeline.py).
    const int q = h->snap_next;
    if (!h->snap_free_known[q] && h->gather_throttle && h->snap_in_rollout && h->stream == h->own_stream) {
      for (int spins = 0;; ++spins) {
        const hipError_t e = hipEventQuery(h->ev_gdone[q]);
        if (e == hipSuccess) break;
        (void)hipGetLastError();
        if (e != hipErrorNotReady) return fail(h, MT_ERR_HIP, std::string("hipEventQuery: ") + hipGetErrorString(e));
        if (spins > 20000) {  // (a long exchange: sleep instead of spinning on)
          MT_HIP(h, hipEventSynchronize(h->ev_gdone[q]));
          break;
        }
        __builtin_ia32_pause();
      }
      h->snap_free_known[q] = true;
    }
  }
  return MT_OK;
}

int mt_gather_returns_begin(mt_handle h, int field, int row, float* dst, int64_t dst_elems) {
  return gather_begin_impl(h, field, row, dst, dst_elems, false);
}

int mt_gather_returns_begin_inplace(mt_handle h, int field, int row, float* dst, int64_t dst_elems) {
  return gather_begin_impl(h, field, row, dst, dst_elems, true);
}

int mt_gather_returns_wait(mt_handle h, int host_wait, float* elapsed_ms) {
  MT_REQUIRE(nullptr, h != nullptr, "handle is NULL");
  if (elapsed_ms) *elapsed_ms = host_wait ? h->last_gather_ms : 0.f;  // nothing pending: the last completed exchange
  if (!h->gather_pending) return MT_OK;
  MT_ON_DEVICE(h, h->cfg.device);  // no join: only the handle's own stream is ordered behind the exchange
  if (hipEventQuery(h->ev_g1) != hipSuccess) {  // (already finished: no barrier packet needed, see order_behind_pending_gather)
    (void)hipGetLastError();
    MT_HIP(h, hipStreamWaitEvent(h->stream, h->ev_g1, 0));
  }
  if (host_wait) {
    MT_HIP(h, hipEventSynchronize(h->ev_g1));
    MT_HIP(h, hipEventElapsedTime(&h->last_gather_ms, h->ev_g0, h->ev_g1));
    if (elapsed_ms) *elapsed_ms = h->last_gather_ms;
    h->gather_pending = false;
    h->snap_free_known[0] = h->snap_free_known[1] = true;  // the latest exchange has finished, and so have the ones before it
  }
  return MT_OK;
}

}  // extern "C"

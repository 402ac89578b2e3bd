// TEST INFRASTRUCTURE ONLY -- a stand-in for librccl.so with which comm.hip's multi-rank code path (unique id ->
// ncclCommInitRank -> shard-size exchange -> equal / ragged all-gather, in line and on the side stream) can run with
// several ranks on ONE GPU, where the real RCCL refuses ("duplicate GPU").  It implements the five entry points
// comm.hip resolves, over a POSIX shared-memory segment named by the unique id: every collective is
//   wait for the stream -> device-to-host copy of this rank's slice into the segment -> barrier ->
//   host-to-device copy of all slices into the receive buffer -> barrier.
// Host-synchronous, slow, and only as ordered as a test needs.  Selected with MT_RCCL_LIB=<this .so>; never built or
// loaded outside tests/test_gpu_multirank_fake.py.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>

namespace {
constexpr size_t kSlot = 8u << 20;  // bytes per rank in the segment
constexpr int kMaxRanks = 8;
struct Header {
  std::atomic<int> arrived;
  std::atomic<int> generation;
  std::atomic<int> attached;
};
struct Comm {
  char name[64];
  int rank, nranks;
  Header* hdr;
  char* data;
  size_t bytes;
};
void barrier(Comm* c) {
  const int gen = c->hdr->generation.load();
  if (c->hdr->arrived.fetch_add(1) + 1 == c->nranks) {
    c->hdr->arrived.store(0);
    c->hdr->generation.fetch_add(1);
  } else {
    while (c->hdr->generation.load() == gen) usleep(50);
  }
}
size_t dtype_size(int dt) { return dt == 4 || dt == 5 || dt == 8 ? 8 : (dt == 0 || dt == 1 ? 1 : (dt == 6 ? 2 : 4)); }
}  // namespace

extern "C" {

struct ncclUniqueId {
  char internal[128];
};

int ncclGetUniqueId(ncclUniqueId* id) {
  std::memset(id->internal, 0, sizeof id->internal);
  std::snprintf(id->internal, sizeof id->internal, "/mtfake_%d_%ld", (int)getpid(), (long)time(nullptr));
  return 0;
}

int ncclCommInitRank(void** comm, int nranks, ncclUniqueId id, int rank) {
  if (nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return 4;  // ncclInvalidArgument
  Comm* c = new Comm();
  std::strncpy(c->name, id.internal, sizeof c->name - 1);
  c->rank = rank;
  c->nranks = nranks;
  c->bytes = sizeof(Header) + kSlot * (size_t)nranks;
  int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
  if (fd < 0) return 2;  // ncclSystemError
  if (ftruncate(fd, (off_t)c->bytes) != 0) return 2;
  void* p = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return 2;
  c->hdr = reinterpret_cast<Header*>(p);  // a fresh segment is zero-filled: counters start at 0
  c->data = reinterpret_cast<char*>(p) + sizeof(Header);
  c->hdr->attached.fetch_add(1);
  while (c->hdr->attached.load() < nranks) usleep(100);  // like the real call: returns once every rank has joined
  *comm = c;
  return 0;
}

int ncclAllGather(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t stream) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  const size_t bytes = count * dtype_size(dtype);
  if (bytes > kSlot) return 4;
  if (hipStreamSynchronize(stream) != hipSuccess) return 1;
  if (hipMemcpy(c->data + kSlot * (size_t)c->rank, send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  barrier(c);
  for (int r = 0; r < c->nranks; ++r)
    if (hipMemcpy(reinterpret_cast<char*>(recv) + bytes * (size_t)r, c->data + kSlot * (size_t)r, bytes,
                  hipMemcpyHostToDevice) != hipSuccess)
      return 1;
  barrier(c);
  return 0;
}

int ncclCommDestroy(void* comm) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  if (c->hdr->attached.fetch_sub(1) == 1) shm_unlink(c->name);
  munmap(reinterpret_cast<void*>(c->hdr), c->bytes);
  delete c;
  return 0;
}

const char* ncclGetErrorString(int code) { return code == 0 ? "no error" : "fake rccl error"; }

}  // extern "C"

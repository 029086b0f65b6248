// Direct-RCCL telemetry exchange: every rank contributes one 128-byte snapshot
// (16 doubles) and receives the [world, 16] table with a single ncclAllGather on a
// dedicated communicator and a dedicated non-blocking HIP stream, so the exchange never
// queues behind (or in front of) the model's own collectives and kernels.
//
// xGMI is point-to-point: a ring all-gather of world*128 bytes is latency-bound, not
// bandwidth-bound, so the payload is kept to one cache line per rank and ticks are meant to be
// batched by the caller (kvt_gather_many exchanges n_ticks snapshots in one collective).
//
// Bootstrap: rank 0 creates the ncclUniqueId (kvt_unique_id) and the caller ships its 128
// bytes to the other ranks over whatever channel it already has (torch.distributed
// broadcast, a file, an env var); then every rank calls kvt_init.

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdint>
#include <cstring>

namespace {

struct State {
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;
  double* d_local = nullptr;
  double* d_table = nullptr;
  int world = 0, rank = 0, device = 0;
  long long cap_ticks = 0;
  bool ready = false;
};
State g;

constexpr int SNAP = 16;

int ensure_capacity(long long ticks) {
  if (ticks <= g.cap_ticks) return 0;
  if (g.d_local) (void)hipFree(g.d_local);
  if (g.d_table) (void)hipFree(g.d_table);
  g.d_local = g.d_table = nullptr;
  if (hipMalloc((void**)&g.d_local, sizeof(double) * SNAP * ticks) != hipSuccess) return -10;
  if (hipMalloc((void**)&g.d_table, sizeof(double) * SNAP * ticks * g.world) != hipSuccess) return -10;
  g.cap_ticks = ticks;
  return 0;
}

}  // namespace

extern "C" {

int kvt_unique_id_bytes() { return (int)sizeof(ncclUniqueId); }

// Rank 0: fill `out` (kvt_unique_id_bytes() bytes).
int kvt_unique_id(void* out) {
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) return -1;
  std::memcpy(out, &id, sizeof id);
  return 0;
}

int kvt_init(int rank, int world, const void* unique_id, int device) {
  if (g.ready) return -1;
  if (world < 1 || rank < 0 || rank >= world) return -2;
  if (hipSetDevice(device) != hipSuccess) return -3;
  g.rank = rank; g.world = world; g.device = device;
  if (hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking) != hipSuccess) return -4;
  ncclUniqueId id;
  std::memcpy(&id, unique_id, sizeof id);
  if (ncclCommInitRank(&g.comm, world, id, rank) != ncclSuccess) return -5;
  const int rc = ensure_capacity(64);
  if (rc) return rc;
  g.ready = true;
  return 0;
}

// local: n_ticks x 16 doubles (host).  table: world x n_ticks x 16 doubles (host).
int kvt_gather_many(const double* local, double* table, long long n_ticks) {
  if (!g.ready) return -1;
  if (n_ticks < 1) return -2;
  if (hipSetDevice(g.device) != hipSuccess) return -3;
  int rc = ensure_capacity(n_ticks);
  if (rc) return rc;
  const size_t count = (size_t)SNAP * (size_t)n_ticks;
  if (hipMemcpyAsync(g.d_local, local, count * sizeof(double), hipMemcpyHostToDevice, g.stream) != hipSuccess) return -6;
  if (ncclAllGather(g.d_local, g.d_table, count, ncclDouble, g.comm, g.stream) != ncclSuccess) return -7;
  if (hipMemcpyAsync(table, g.d_table, count * sizeof(double) * g.world, hipMemcpyDeviceToHost, g.stream) != hipSuccess) return -8;
  if (hipStreamSynchronize(g.stream) != hipSuccess) return -9;
  return 0;
}

int kvt_gather(const double* local16, double* table) { return kvt_gather_many(local16, table, 1); }

int kvt_world() { return g.ready ? g.world : 0; }

int kvt_destroy() {
  if (!g.ready) return -1;
  (void)hipStreamSynchronize(g.stream);
  ncclCommDestroy(g.comm);
  (void)hipStreamDestroy(g.stream);
  if (g.d_local) (void)hipFree(g.d_local);
  if (g.d_table) (void)hipFree(g.d_table);
  g = State{};
  return 0;
}

}  // extern "C"

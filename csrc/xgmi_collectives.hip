// Hand-written collectives over xGMI peer memory for latency-bound messages on an 8 x MI355X
// node: one-shot all-gather (telemetry tables), one-shot all-reduce (<= 256 KiB: every rank
// stores its whole message into every peer's slot, all 7 links of a GPU busy at once, one hop),
// two-shot all-reduce (reduce-scatter + all-gather, <= 4 MiB: each link carries 1/W of the
// message twice) and the tensor-parallel epilogue fused around the one-shot form:
//     residual' = residual + bf16(sum over ranks of bf16(sum of this rank's split-K slabs))
// with the per-row sums of squares of residual' for the folded RMSNorm — the all-reduce that
// follows every row-parallel projection, without a trip through a library ring (a ring moves
// 2 (W-1)/W of the message over ONE ~153 GB/s link path in 2 (W-1) latency-bound hops).
//
// Memory: every rank owns ONE buffer of device memory allocated uncached (no L2 copy can go
// stale when a peer writes it) and exported with hipIpcGetMemHandle; peers map it with
// hipIpcOpenMemHandle.  Layout: [error words | flags | slots | gather area].
//
// Protocol (per workgroup, no host involvement, no grid-wide barrier):
//   * PUSH  every lane stores its 16 bytes into slot[parity][my rank] of every peer (own included);
//   * SIGNAL  __threadfence_system() by every lane, workgroup barrier, then lane r stores 1 into
//     flag[parity][phase][my rank][workgroup] ON PEER r (system-scope release store);
//   * WAIT  lane r spins (bounded, s_sleep between polls) on its OWN flag[parity][phase][r][workgroup]
//     until it reads 1, then stores 0 back (the flag is ready for the call after next);
//     workgroup barrier; system-scope acquire fence; then every lane reads the slots.
//   Workgroup b of rank A only ever waits for workgroup b of the other ranks: all ranks launch
//   the same grid for the same call.
//   * Consecutive calls alternate `parity` (two sets of slots and flags).  Why two suffice: rank
//     A can pass the WAIT of call n+1 only after every peer has pushed for call n+1, i.e. after
//     every peer's kernel of call n has completed (kernels of one rank run in stream order), so
//     nobody still reads the slots of call n when A overwrites them in call n+2.  The flag reset
//     of call n is ordered before A's pushes of call n+1 by the release fence of that call's
//     SIGNAL, hence before any peer's call-n+2 signal can arrive.
//   * Flags carry the constant 1 (not a sequence number) and parity is an argument: a captured
//     HIP graph replays correctly as long as it holds an EVEN number of calls (two per layer).
//   * Every spin is bounded (about 3 s of s_memrealtime); a timeout counts up error word 0 and
//     lets the workgroup finish, so a missing peer can never hang the GPU.

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

constexpr int MAX_WORLD = 8;

struct Comm {
  unsigned char* peer[MAX_WORLD];  // peer[rank] is this rank's own buffer
  int world, rank;
  long long flags_off, slots_off, gather_off, slot_bytes;
  int max_wg;
};

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) unsigned short bf16x8;

__device__ inline float bf2f(unsigned short v) { return __uint_as_float((unsigned int)v << 16); }
__device__ inline unsigned short f2bf(float f) {
  unsigned int u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

__device__ inline unsigned int* flag_ptr(const Comm& c, int on_rank, int parity, int phase, int from_rank, int wg) {
  return reinterpret_cast<unsigned int*>(c.peer[on_rank] + c.flags_off) + ((long long)((parity * 2 + phase) * MAX_WORLD + from_rank) * c.max_wg + wg);
}
__device__ inline unsigned char* slot_ptr(const Comm& c, int on_rank, int parity, int from_rank) {
  return c.peer[on_rank] + c.slots_off + (long long)(parity * MAX_WORLD + from_rank) * c.slot_bytes;
}
__device__ inline unsigned char* gather_ptr(const Comm& c, int on_rank, int parity) { return c.peer[on_rank] + c.gather_off + (long long)parity * c.slot_bytes; }

// SIGNAL + WAIT of one phase for the calling workgroup (all threads must call it).
__device__ inline void signal_and_wait(const Comm& c, int parity, int phase, int wg) {
  __threadfence_system();  // this lane's pushes are performed at their destinations
  __syncthreads();
  const int tid = threadIdx.x;
  if (tid < c.world) {
    __hip_atomic_store(flag_ptr(c, tid, parity, phase, c.rank, wg), 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    unsigned int* mine = flag_ptr(c, c.rank, parity, phase, tid, wg);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
    bool ok = true;
    while (__hip_atomic_load(mine, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != 1u) {
      if (__builtin_amdgcn_s_memrealtime() - t0 > 300000000ull) { ok = false; break; }
      __builtin_amdgcn_s_sleep(4);
    }
    if (!ok) atomicAdd(reinterpret_cast<unsigned int*>(c.peer[c.rank]), 1u);  // error word 0: timeouts
    __hip_atomic_store(mine, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // system scope: nothing read below may come from a stale cache line
}

// ---- one-shot all-gather: out[r][nbytes] = rank r's x[nbytes]; 256 threads x 16 B per workgroup
__global__ __launch_bounds__(256) void xgmi_allgather_kernel(const Comm c, const unsigned char* __restrict__ x, unsigned char* __restrict__ out, long long nbytes,
                                                             int parity) {
  const long long off = ((long long)blockIdx.x * 256 + threadIdx.x) * 16;
  const bool live = off < nbytes;
  if (live) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(x + off);
    for (int k = 0; k < c.world; ++k) {
      const int j = (c.rank + 1 + k) % c.world;  // start with the next rank: the W - 1 links fill in different orders
      *reinterpret_cast<u32x4*>(slot_ptr(c, j, parity, c.rank) + off) = v;
    }
  }
  signal_and_wait(c, parity, 0, blockIdx.x);
  if (live)
    for (int r = 0; r < c.world; ++r) *reinterpret_cast<u32x4*>(out + (long long)r * nbytes + off) = *reinterpret_cast<const u32x4*>(slot_ptr(c, c.rank, parity, r) + off);
}

// ---- one-shot all-reduce of bf16 x[n] (n % 8 == 0): out = bf16(sum over ranks, fp32, rank order)
__global__ __launch_bounds__(256) void xgmi_allreduce1_kernel(const Comm c, const unsigned short* __restrict__ x, unsigned short* __restrict__ out, long long n,
                                                              int parity) {
  const long long e0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 8;
  const bool live = e0 < n;
  if (live) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + e0);
    for (int k = 0; k < c.world; ++k) {
      const int j = (c.rank + 1 + k) % c.world;
      *reinterpret_cast<bf16x8*>(slot_ptr(c, j, parity, c.rank) + e0 * 2) = v;
    }
  }
  signal_and_wait(c, parity, 0, blockIdx.x);
  if (live) {
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < c.world; ++r) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(slot_ptr(c, c.rank, parity, r) + e0 * 2);
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] += bf2f(v[j]);
    }
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = f2bf(a[j]);
    *reinterpret_cast<bf16x8*>(out + e0) = o;
  }
}

// ---- two-shot all-reduce: x[n] as W chunks of cn elements (cn % 8 == 0, n == W * cn);
// workgroup b owns elements [b * 2048, (b + 1) * 2048) of EVERY chunk
__global__ __launch_bounds__(256) void xgmi_allreduce2_kernel(const Comm c, const unsigned short* __restrict__ x, unsigned short* __restrict__ out, long long cn,
                                                              int parity) {
  const long long e0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 8;
  const bool live = e0 < cn;
  if (live)  // reduce-scatter: chunk j of my data goes to rank j
    for (int k = 0; k < c.world; ++k) {
      const int j = (c.rank + 1 + k) % c.world;
      *reinterpret_cast<bf16x8*>(slot_ptr(c, j, parity, c.rank) + e0 * 2) = *reinterpret_cast<const bf16x8*>(x + (long long)j * cn + e0);
    }
  signal_and_wait(c, parity, 0, blockIdx.x);
  if (live) {
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < c.world; ++r) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(slot_ptr(c, c.rank, parity, r) + e0 * 2);
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] += bf2f(v[j]);
    }
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = f2bf(a[j]);
    for (int k = 0; k < c.world; ++k) {  // all-gather: my reduced chunk goes to everybody
      const int j = (c.rank + 1 + k) % c.world;
      *reinterpret_cast<bf16x8*>(gather_ptr(c, j, parity) + ((long long)c.rank * cn + e0) * 2) = o;
    }
  }
  signal_and_wait(c, parity, 1, blockIdx.x);
  if (live)
    for (int r = 0; r < c.world; ++r)
      *reinterpret_cast<bf16x8*>(out + (long long)r * cn + e0) = *reinterpret_cast<const bf16x8*>(gather_ptr(c, c.rank, parity) + ((long long)r * cn + e0) * 2);
}

// ---- tensor-parallel epilogue: slab sum -> one-shot all-reduce -> residual add -> sums of squares.
// grid = (ceil(H / 512), T), one wavefront per workgroup, 8 columns per lane (the shape of
// residual_reduce_kernel in csrc/engine_kernels.hip, whose single-GPU result this equals for W = 1).
template <bool SLAB>
__global__ __launch_bounds__(64) void xgmi_residual_allreduce_kernel(const Comm c, const float* __restrict__ slabs, int n_slabs, long long slab_stride,
                                                                     const unsigned short* __restrict__ x, unsigned short* __restrict__ residual, int has_residual,
                                                                     float* __restrict__ ss_part, int H, int parity) {
  const int t = blockIdx.y, p = blockIdx.x, lane = threadIdx.x;
  const int col = (p * 64 + lane) * 8;
  const bool live = col < H;
  const long long off = (long long)t * H + col;
  const int wg = blockIdx.y * gridDim.x + blockIdx.x;
  if (live) {
    bf16x8 mine;
    if (SLAB) {
      float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int s = 0; s < n_slabs; ++s) {
        const f32x4* q = reinterpret_cast<const f32x4*>(slabs + (long long)s * slab_stride + off);
        const f32x4 lo = q[0], hi = q[1];
#pragma unroll
        for (int j = 0; j < 4; ++j) { a[j] += lo[j]; a[4 + j] += hi[j]; }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) mine[j] = f2bf(a[j]);
    } else {
      mine = *reinterpret_cast<const bf16x8*>(x + off);
    }
    for (int k = 0; k < c.world; ++k) {
      const int j = (c.rank + 1 + k) % c.world;
      *reinterpret_cast<bf16x8*>(slot_ptr(c, j, parity, c.rank) + off * 2) = mine;
    }
  }
  signal_and_wait(c, parity, 0, wg);
  float ss = 0.f;
  if (live) {
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < c.world; ++r) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(slot_ptr(c, c.rank, parity, r) + off * 2);
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] += bf2f(v[j]);
    }
    bf16x8 o;
    if (has_residual) {
      const bf16x8 b = *reinterpret_cast<const bf16x8*>(residual + off);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = f2bf(bf2f(f2bf(a[j])) + bf2f(b[j]));
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = f2bf(a[j]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float f = bf2f(o[j]); ss += f * f; }
    *reinterpret_cast<bf16x8*>(residual + off) = o;
  }
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
  if (lane == 0) ss_part[(long long)t * gridDim.x + p] = ss;
}

struct Host {
  Comm c{};
  void* buf = nullptr;
  long long total = 0;
  bool opened[MAX_WORLD] = {false, false, false, false, false, false, false, false};
  int parity = 0;
  long long calls = 0;
};

}  // namespace

extern "C" {

// Layout constants a caller needs to size messages: max workgroups (flags) and bytes per slot.
int kvx_max_wg() { return 4096; }

// Create this rank's communicator state and buffer.  Returns an opaque handle (nullptr on failure).
void* kvx_create(int world, int rank, long long slot_bytes) {
  if (world < 1 || world > MAX_WORLD || rank < 0 || rank >= world || slot_bytes < 4096 || (slot_bytes & 4095)) return nullptr;
  Host* h = new Host();
  Comm& c = h->c;
  c.world = world; c.rank = rank; c.max_wg = kvx_max_wg(); c.slot_bytes = slot_bytes;
  c.flags_off = 4096;
  const long long flag_bytes = 2LL * 2 * MAX_WORLD * c.max_wg * 4;
  c.slots_off = c.flags_off + flag_bytes;
  c.gather_off = c.slots_off + 2LL * MAX_WORLD * slot_bytes;
  h->total = c.gather_off + 2LL * slot_bytes;
  if (hipExtMallocWithFlags(&h->buf, (size_t)h->total, hipDeviceMallocUncached) != hipSuccess) { delete h; return nullptr; }
  if (hipMemset(h->buf, 0, (size_t)c.slots_off) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { (void)hipFree(h->buf); delete h; return nullptr; }
  for (int r = 0; r < MAX_WORLD; ++r) c.peer[r] = nullptr;
  c.peer[rank] = (unsigned char*)h->buf;
  h->opened[rank] = true;
  return h;
}

int kvx_handle_bytes() { return (int)sizeof(hipIpcMemHandle_t); }

int kvx_get_handle(void* hp, void* out) {
  Host* h = (Host*)hp;
  hipIpcMemHandle_t ih;
  if (hipIpcGetMemHandle(&ih, h->buf) != hipSuccess) return -1;
  __builtin_memcpy(out, &ih, sizeof ih);
  return 0;
}

// Map peer `r`'s buffer from the handle it exported (another process; any device of the node).
int kvx_open_peer(void* hp, int r, const void* handle) {
  Host* h = (Host*)hp;
  if (r < 0 || r >= h->c.world || r == h->c.rank || h->opened[r]) return -1;
  hipIpcMemHandle_t ih;
  __builtin_memcpy(&ih, handle, sizeof ih);
  void* p = nullptr;
  if (hipIpcOpenMemHandle(&p, ih, hipIpcMemLazyEnablePeerAccess) != hipSuccess) return -2;
  h->c.peer[r] = (unsigned char*)p;
  h->opened[r] = true;
  return 0;
}

int kvx_ready(void* hp) {
  Host* h = (Host*)hp;
  for (int r = 0; r < h->c.world; ++r)
    if (!h->opened[r]) return 0;
  return 1;
}

long long kvx_slot_bytes(void* hp) { return ((Host*)hp)->c.slot_bytes; }
long long kvx_calls(void* hp) { return ((Host*)hp)->calls; }
int kvx_parity(void* hp) { return ((Host*)hp)->parity; }

// Timeouts counted by the kernels so far (reads the error word: synchronises the device).
int kvx_errors(void* hp) {
  Host* h = (Host*)hp;
  unsigned int e = 0;
  if (hipMemcpy(&e, h->buf, 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return (int)e;
}

static int next_parity(Host* h) {
  const int p = h->parity;
  h->parity ^= 1;
  ++h->calls;
  return p;
}

int kvx_allgather(void* hp, const void* x, void* out, long long nbytes, hipStream_t stream) {
  Host* h = (Host*)hp;
  if (!kvx_ready(hp) || nbytes <= 0 || (nbytes & 15) || nbytes > h->c.slot_bytes) return -1;
  const long long wgs = (nbytes + 4095) / 4096;
  if (wgs > h->c.max_wg) return -1;
  hipLaunchKernelGGL(xgmi_allgather_kernel, dim3((unsigned)wgs), dim3(256), 0, stream, h->c, (const unsigned char*)x, (unsigned char*)out, nbytes, next_parity(h));
  return (int)hipGetLastError();
}

// bf16 x[n] -> out[n] (may alias x).  two_shot: n must be a multiple of 8 * world.
int kvx_allreduce(void* hp, const void* x, void* out, long long n, int two_shot, hipStream_t stream) {
  Host* h = (Host*)hp;
  if (!kvx_ready(hp) || n <= 0 || (n & 7)) return -1;
  if (two_shot) {
    if (n % (8LL * h->c.world) || n * 2 > h->c.slot_bytes) return -1;
    const long long cn = n / h->c.world;
    const long long wgs = (cn + 2047) / 2048;
    if (wgs > h->c.max_wg) return -1;
    hipLaunchKernelGGL(xgmi_allreduce2_kernel, dim3((unsigned)wgs), dim3(256), 0, stream, h->c, (const unsigned short*)x, (unsigned short*)out, cn, next_parity(h));
  } else {
    if (n * 2 > h->c.slot_bytes) return -1;
    const long long wgs = (n + 2047) / 2048;
    if (wgs > h->c.max_wg) return -1;
    hipLaunchKernelGGL(xgmi_allreduce1_kernel, dim3((unsigned)wgs), dim3(256), 0, stream, h->c, (const unsigned short*)x, (unsigned short*)out, n, next_parity(h));
  }
  return (int)hipGetLastError();
}

// x: bf16 [T, H] (n_slabs == 0) or fp32 slabs [n_slabs][T][H]; residual bf16 [T, H] in/out; ss_part float [T][ceil(H/512)]
int kvx_residual_allreduce(void* hp, const float* slabs, int n_slabs, long long slab_stride, const void* x, void* residual, int has_residual, float* ss_part, int T,
                           int H, hipStream_t stream) {
  Host* h = (Host*)hp;
  if (T <= 0) return 0;
  if (!kvx_ready(hp) || (H & 7)) return -1;
  const int segs = (H + 511) / 512;
  if ((long long)T * H * 2 > h->c.slot_bytes || (long long)T * segs > h->c.max_wg) return -1;
  dim3 grid(segs, T);
  if (n_slabs > 0)
    hipLaunchKernelGGL(xgmi_residual_allreduce_kernel<true>, grid, dim3(64), 0, stream, h->c, slabs, n_slabs, slab_stride, (const unsigned short*)nullptr,
                       (unsigned short*)residual, has_residual, ss_part, H, next_parity(h));
  else
    hipLaunchKernelGGL(xgmi_residual_allreduce_kernel<false>, grid, dim3(64), 0, stream, h->c, (const float*)nullptr, 0, 0LL, (const unsigned short*)x,
                       (unsigned short*)residual, has_residual, ss_part, H, next_parity(h));
  return (int)hipGetLastError();
}

int kvx_destroy(void* hp) {
  Host* h = (Host*)hp;
  (void)hipDeviceSynchronize();
  for (int r = 0; r < h->c.world; ++r)
    if (r != h->c.rank && h->opened[r] && h->c.peer[r]) (void)hipIpcCloseMemHandle(h->c.peer[r]);
  if (h->buf) (void)hipFree(h->buf);
  delete h;
  return 0;
}

}  // extern "C"

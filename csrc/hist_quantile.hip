// Latency statistics kernels for gfx950 (MI355X): LDS-binned histograms and exact quantiles.
//
// What this replaces: the reference reduces request/token latencies on the CPU by
// sorting a Python list once per percentile (analyze.py:59-81, ~25 sorts per run) and
// by a Python loop for its 20-bin histograms (analyze.py:84-122).  At per-token
// granularity (10^6..10^8 inter-token gaps per run) that is the harness's only
// compute-bound path, so it lives on the GPU:
//
//   kv_stream_hist      streaming histogram with 2048 log-spaced bins per metric, plus
//                        count / sum / min / max, accumulated into persistent device
//                        state (launch it on every new slice of token events; the state
//                        is never re-read from the samples).  Per-wavefront private
//                        sub-histograms live in LDS (4 waves x 2048 bins x 4 B = 32 KiB of
//                        the CU's 160 KiB) and are updated with LDS atomics; one global
//                        atomic per non-empty bin per workgroup flushes them.
//   kv_radix_hist/scan  exact order statistics by MSD radix select on order-preserving
//                        integer keys: 11-bit digits, i.e. the same 2048-bin LDS histogram,
//                        restricted to samples whose higher digits match the prefix of a
//                        target rank; a single-workgroup scan kernel turns the bin counts
//                        into the next prefix.  Up to 8 ranks are tracked at once (floor and
//                        ceil rank of p50/p95/p99), so three quantiles of 10^8 doubles cost 6
//                        streaming passes and no sort.  The result is bit-exact: the CPU
//                        interpolates between the two selected samples exactly like
//                        analyze.py does.
//   kv_equal_hist       the 20-bin equal-width histogram over [min, max] (last bin closed),
//                        LDS-binned the same way, with min/max read from device memory.
//
// Batching: samples are a row-major [n_metrics, row_stride] matrix; NaN entries are
// skipped (analyze.py drops NaNs), so ragged metrics are NaN-padded and one launch with
// grid.y = metric handles latency / TTFT / TLLT / generation / per-token x all / cold /
// warm together.
//
// Wavefront width is 64 on CDNA; workgroups are 256 threads = 4 waves, one per SIMD.

#include <hip/hip_runtime.h>
#include <stdint.h>

#define KV_BINS 2048
#define KV_WAVES 4
#define KV_THREADS 256
#define KV_MAX_RANKS 8

namespace {

// ---------------------------------------------------------------------------------
// order-preserving keys
template <typename T> struct KeyOf;
template <> struct KeyOf<float> {
  using type = uint32_t;
  static constexpr int bits = 32;
  __device__ static inline uint32_t key(float v) {
    uint32_t b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
  }
};
template <> struct KeyOf<double> {
  using type = uint64_t;
  static constexpr int bits = 64;
  __device__ static inline uint64_t key(double v) {
    uint64_t b = (uint64_t)__double_as_longlong(v);
    return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
  }
};

// log-spaced bin of a positive value: 2048 bins over [2^lo_log2, 2^hi_log2)
__device__ inline int log_bin(float v, float lo_log2, float inv_step) {
  if (!(v > 0.f)) return 0;
  float f = (__log2f(v) - lo_log2) * inv_step;
  int b = (int)f;
  return b < 0 ? 0 : (b >= KV_BINS ? KV_BINS - 1 : b);
}

// Histogram increment with wave-level aggregation of the common bin.  Latency samples
// are heavily clustered (most of a wave's 64 lanes hit the same bin), which would
// serialise 64 LDS atomics on one address; instead the lanes that share the first
// active lane's bin elect one lane to add their population count.  Must be called by
// exactly the lanes that hold a valid sample (the active set is the participant set).
__device__ inline void hist_add(unsigned int* h, unsigned int d) {
  unsigned int d0 = __builtin_amdgcn_readfirstlane(d);
  bool same = d == d0;
  unsigned long long mask = __ballot(same);
  if (same) {
    if ((int)__lane_id() == __ffsll((long long)mask) - 1) atomicAdd(&h[d0], (unsigned int)__popcll(mask));
  } else {
    atomicAdd(&h[d], 1u);
  }
}

__device__ inline double wave_sum(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ inline unsigned long long wave_sum_u64(unsigned long long v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <typename K> __device__ inline K wave_min(K v) {
  for (int o = 32; o > 0; o >>= 1) { K w = __shfl_xor(v, o, 64); v = w < v ? w : v; }
  return v;
}
template <typename K> __device__ inline K wave_max(K v) {
  for (int o = 32; o > 0; o >>= 1) { K w = __shfl_xor(v, o, 64); v = w > v ? w : v; }
  return v;
}

// Per-metric persistent state of the streaming histogram.
struct StreamState {
  unsigned long long count;
  double sum;
  unsigned long long min_key;  // order-preserving key of the minimum (init: ~0)
  unsigned long long max_key;  // order-preserving key of the maximum (init: 0)
};

// ---------------------------------------------------------------------------------
// kv_stream_hist: x[n_metrics][row_stride] (first n valid columns), hist[n_metrics][2048]
template <typename T>
__global__ __launch_bounds__(KV_THREADS) void stream_hist_kernel(const T* __restrict__ x, long long n, long long row_stride,
                                                                 unsigned int* __restrict__ hist, StreamState* __restrict__ state,
                                                                 float lo_log2, float inv_step) {
  __shared__ unsigned int lds[KV_WAVES * KV_BINS];
  const int tid = threadIdx.x, wave = tid >> 6;
  const int m = blockIdx.y;
  for (int i = tid; i < KV_WAVES * KV_BINS; i += KV_THREADS) lds[i] = 0;
  __syncthreads();

  const T* row = x + (long long)m * row_stride;
  unsigned int* my = lds + wave * KV_BINS;
  using KT = typename KeyOf<T>::type;
  unsigned long long cnt = 0;
  double sum = 0.0;
  KT kmin = ~(KT)0, kmax = 0;

  constexpr int VEC = 16 / sizeof(T);  // 16-byte loads: 4 floats or 2 doubles per lane
  const long long nvec = n / VEC;
  const long long stride = (long long)gridDim.x * KV_THREADS;
  struct alignas(16) Pack { T v[VEC]; };
  const bool aligned = (((uintptr_t)row) & 15) == 0;
  if (aligned) {
    const Pack* pv = reinterpret_cast<const Pack*>(row);
    auto consume = [&](const Pack& p) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        T v = p.v[j];
        if (v == v) {
          hist_add(my, (unsigned int)log_bin((float)v, lo_log2, inv_step));
          ++cnt; sum += (double)v;
          KT k = KeyOf<T>::key(v);
          kmin = k < kmin ? k : kmin; kmax = k > kmax ? k : kmax;
        }
      }
    };
    // four 16-byte loads in flight per lane: the LDS atomics of one pack hide the latency of
    // the next loads instead of serialising load -> bin -> load
    long long i = (long long)blockIdx.x * KV_THREADS + tid;
    for (; i + 3 * stride < nvec; i += 4 * stride) {
      const Pack p0 = pv[i], p1 = pv[i + stride], p2 = pv[i + 2 * stride], p3 = pv[i + 3 * stride];
      consume(p0); consume(p1); consume(p2); consume(p3);
    }
    for (; i < nvec; i += stride) consume(pv[i]);
  }
  // tail (or everything, when the row is not 16-byte aligned)
  for (long long i = (aligned ? nvec * VEC : 0) + (long long)blockIdx.x * KV_THREADS + tid; i < n; i += stride) {
    T v = row[i];
    if (v == v) {
      hist_add(my, (unsigned int)log_bin((float)v, lo_log2, inv_step));
      ++cnt; sum += (double)v;
      KT k = KeyOf<T>::key(v);
      kmin = k < kmin ? k : kmin; kmax = k > kmax ? k : kmax;
    }
  }
  __syncthreads();
  unsigned int* gh = hist + (long long)m * KV_BINS;
  for (int b = tid; b < KV_BINS; b += KV_THREADS) {
    unsigned int c = lds[b] + lds[KV_BINS + b] + lds[2 * KV_BINS + b] + lds[3 * KV_BINS + b];
    if (c) atomicAdd(&gh[b], c);
  }
  // scalars: wave reduction, then one atomic per wave
  cnt = wave_sum_u64(cnt);
  sum = wave_sum(sum);
  kmin = wave_min<KT>(kmin);
  kmax = wave_max<KT>(kmax);
  if ((tid & 63) == 0 && cnt) {
    StreamState* s = state + m;
    atomicAdd(&s->count, cnt);
    atomicAdd(&s->sum, sum);
    // widen 32-bit keys so one state layout serves both dtypes
    unsigned long long k0 = (unsigned long long)kmin << (64 - KeyOf<T>::bits);
    unsigned long long k1 = (unsigned long long)kmax << (64 - KeyOf<T>::bits);
    atomicMin(&s->min_key, k0);
    atomicMax(&s->max_key, k1);
  }
}

// ---------------------------------------------------------------------------------
// Radix select.  Per metric: U distinct prefixes (U <= 8).  hist[m][u][2048].
struct RadixState {
  unsigned long long prefix[KV_MAX_RANKS];  // key bits decided so far (high bits), per rank
  unsigned long long k[KV_MAX_RANKS];       // rank remaining inside the prefix
  unsigned long long uniq[KV_MAX_RANKS];    // distinct prefixes for the next histogram pass
  int slot[KV_MAX_RANKS];                   // rank -> index into uniq
  int n_uniq;
  int n_ranks;
};

template <typename T>
__global__ __launch_bounds__(KV_THREADS) void radix_hist_kernel(const T* __restrict__ x, long long n, long long row_stride,
                                                                const RadixState* __restrict__ st, unsigned int* __restrict__ hist,
                                                                int shift, int digit_bits, int first_pass) {
  // first pass: one prefix (empty) -> per-wave private sub-histograms; later passes: one
  // workgroup-shared histogram per distinct prefix.
  __shared__ unsigned int lds[KV_MAX_RANKS * KV_BINS];  // 64 KiB
  using KT = typename KeyOf<T>::type;
  const int tid = threadIdx.x, wave = tid >> 6;
  const int m = blockIdx.y;
  const RadixState* s = st + m;
  const int U = first_pass ? 1 : s->n_uniq;
  const int nclear = first_pass ? KV_WAVES * KV_BINS : U * KV_BINS;
  for (int i = tid; i < nclear; i += KV_THREADS) lds[i] = 0;
  KT pre[KV_MAX_RANKS];
#pragma unroll
  for (int u = 0; u < KV_MAX_RANKS; ++u) pre[u] = (KT)(u < U && !first_pass ? (s->uniq[u] >> (64 - KeyOf<T>::bits)) : 0);
  __syncthreads();

  const T* row = x + (long long)m * row_stride;
  const KT dmask = (KT)((1u << digit_bits) - 1u);
  // bits above the current digit
  const int hi_shift = shift + digit_bits;
  constexpr int VEC = 16 / sizeof(T);
  struct alignas(16) Pack { T v[VEC]; };
  const long long nvec = n / VEC;
  const long long stride = (long long)gridDim.x * KV_THREADS;
  const bool aligned = (((uintptr_t)row) & 15) == 0;

  auto visit = [&](T v) {
    if (v == v) {
      KT key = KeyOf<T>::key(v);
      unsigned int d = (unsigned int)((key >> shift) & dmask);
      if (first_pass) {
        hist_add(lds + wave * KV_BINS, d);
      } else {
        KT hi = hi_shift >= KeyOf<T>::bits ? (KT)0 : (KT)(key >> hi_shift);
#pragma unroll
        for (int u = 0; u < KV_MAX_RANKS; ++u)
          if (u < U && hi == (hi_shift >= KeyOf<T>::bits ? (KT)0 : (KT)(pre[u] >> hi_shift))) hist_add(lds + u * KV_BINS, d);
      }
    }
  };
  if (aligned) {
    const Pack* pv = reinterpret_cast<const Pack*>(row);
    long long i = (long long)blockIdx.x * KV_THREADS + tid;
    for (; i + 3 * stride < nvec; i += 4 * stride) {  // four loads in flight per lane
      const Pack p0 = pv[i], p1 = pv[i + stride], p2 = pv[i + 2 * stride], p3 = pv[i + 3 * stride];
#pragma unroll
      for (int j = 0; j < VEC; ++j) visit(p0.v[j]);
#pragma unroll
      for (int j = 0; j < VEC; ++j) visit(p1.v[j]);
#pragma unroll
      for (int j = 0; j < VEC; ++j) visit(p2.v[j]);
#pragma unroll
      for (int j = 0; j < VEC; ++j) visit(p3.v[j]);
    }
    for (; i < nvec; i += stride) {
      Pack p = pv[i];
#pragma unroll
      for (int j = 0; j < VEC; ++j) visit(p.v[j]);
    }
  }
  for (long long i = (aligned ? nvec * VEC : 0) + (long long)blockIdx.x * KV_THREADS + tid; i < n; i += stride) visit(row[i]);
  __syncthreads();

  unsigned int* gh = hist + (long long)m * KV_MAX_RANKS * KV_BINS;
  if (first_pass) {
    for (int b = tid; b < KV_BINS; b += KV_THREADS) {
      unsigned int c = lds[b] + lds[KV_BINS + b] + lds[2 * KV_BINS + b] + lds[3 * KV_BINS + b];
      if (c) atomicAdd(&gh[b], c);
    }
  } else {
    for (int i = tid; i < U * KV_BINS; i += KV_THREADS) {
      unsigned int c = lds[i];
      if (c) atomicAdd(&gh[i], c);
    }
  }
}

// Ranks from quantile levels, on the device (no host round trip between the streaming
// pass that counts the non-NaN samples and the selection): pos = p*(n-1), ranks floor(pos)
// and ceil(pos) — the definition of analyze.py:75-81.
__global__ void radix_init_kernel(const StreamState* __restrict__ stream, RadixState* __restrict__ st, const double* __restrict__ ps, int n_ps) {
  const int m = blockIdx.x;
  if (threadIdx.x != 0) return;
  RadixState* s = st + m;
  unsigned long long n = stream[m].count;
  int r = 0;
  if (n > 0) {
    for (int i = 0; i < n_ps && r + 1 < KV_MAX_RANKS; ++i) {
      double p = ps[i];
      p = p < 0.0 ? 0.0 : (p > 1.0 ? 1.0 : p);
      double pos = p * (double)(n - 1);
      s->prefix[r] = 0; s->k[r] = (unsigned long long)floor(pos); ++r;
      s->prefix[r] = 0; s->k[r] = (unsigned long long)ceil(pos); ++r;
    }
  }
  for (int i = r; i < KV_MAX_RANKS; ++i) { s->prefix[i] = 0; s->k[i] = 0; }
  for (int i = 0; i < KV_MAX_RANKS; ++i) { s->uniq[i] = 0; s->slot[i] = 0; }
  s->n_uniq = r ? 1 : 0;
  s->n_ranks = r;
}

// One workgroup per metric: prefix-scan each rank's histogram, pick the bin that holds
// rank k, append the digit to the prefix, subtract the counts below it, rebuild the
// distinct-prefix list, and clear the histograms for the next pass.
__global__ __launch_bounds__(KV_THREADS) void radix_scan_kernel(RadixState* __restrict__ st, unsigned int* __restrict__ hist, int shift, int first_pass) {
  __shared__ unsigned long long part[KV_THREADS];
  __shared__ int sel_bin;
  __shared__ unsigned long long sel_below;
  const int tid = threadIdx.x;
  const int m = blockIdx.x;
  RadixState* s = st + m;
  unsigned int* gh = hist + (long long)m * KV_MAX_RANKS * KV_BINS;
  const int R = s->n_ranks;
  constexpr int PER = KV_BINS / KV_THREADS;  // 8 bins per thread

  for (int r = 0; r < R; ++r) {
    const unsigned int* h = gh + (first_pass ? 0 : s->slot[r]) * KV_BINS;
    unsigned long long k = s->k[r];
    unsigned int c[PER];
    unsigned long long local = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) { c[j] = h[tid * PER + j]; local += c[j]; }
    part[tid] = local;
    __syncthreads();
    // inclusive scan over the 256 partial sums (Hillis-Steele in LDS)
    for (int o = 1; o < KV_THREADS; o <<= 1) {
      unsigned long long v = tid >= o ? part[tid - o] : 0;
      __syncthreads();
      part[tid] += v;
      __syncthreads();
    }
    unsigned long long before = part[tid] - local;  // samples in bins below this thread's
    if (k >= before && k < before + local) {
      unsigned long long run = before;
#pragma unroll
      for (int j = 0; j < PER; ++j) {
        if (k >= run && k < run + c[j]) { sel_bin = tid * PER + j; sel_below = run; }
        run += c[j];
      }
    }
    __syncthreads();
    if (tid == 0) {
      s->prefix[r] |= ((unsigned long long)sel_bin) << shift;
      s->k[r] = k - sel_below;
    }
    __syncthreads();
  }
  if (tid == 0) {
    int nu = 0;
    for (int r = 0; r < R; ++r) {
      int f = -1;
      for (int u = 0; u < nu; ++u) if (s->uniq[u] == s->prefix[r]) f = u;
      if (f < 0) { s->uniq[nu] = s->prefix[r]; f = nu++; }
      s->slot[r] = f;
    }
    s->n_uniq = nu;
  }
  __syncthreads();
  for (int i = tid; i < KV_MAX_RANKS * KV_BINS; i += KV_THREADS) gh[i] = 0;
}

// ---------------------------------------------------------------------------------
// 20-bin (n_bins <= 64) equal-width histogram over [min, max] taken from StreamState.
template <typename T>
__global__ __launch_bounds__(KV_THREADS) void equal_hist_kernel(const T* __restrict__ x, long long n, long long row_stride,
                                                                const StreamState* __restrict__ state, unsigned long long* __restrict__ out, int n_bins) {
  __shared__ unsigned int lds[KV_WAVES * 64];
  const int tid = threadIdx.x, wave = tid >> 6;
  const int m = blockIdx.y;
  for (int i = tid; i < KV_WAVES * 64; i += KV_THREADS) lds[i] = 0;
  __syncthreads();
  const StreamState* s = state + m;
  // decode min/max keys back to values
  auto unkey = [](unsigned long long k) -> double {
    if (sizeof(T) == 4) {
      uint32_t b = (uint32_t)(k >> 32);
      b = (b & 0x80000000u) ? (b & 0x7fffffffu) : ~b;
      return (double)__uint_as_float(b);
    }
    unsigned long long b = (k & 0x8000000000000000ull) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
  };
  const double lo = unkey(s->min_key), hi = unkey(s->max_key);
  const double width = (hi - lo) / (double)n_bins;
  const T* row = x + (long long)m * row_stride;
  const long long stride = (long long)gridDim.x * KV_THREADS;
  if (s->count && hi > lo) {
    for (long long i = (long long)blockIdx.x * KV_THREADS + tid; i < n; i += stride) {
      T v = row[i];
      if (v == v) {
        int b = (int)(((double)v - lo) / width);
        b = b >= n_bins ? n_bins - 1 : (b < 0 ? 0 : b);
        hist_add(lds + wave * 64, (unsigned int)b);
      }
    }
  }
  __syncthreads();
  if (tid < n_bins) {
    unsigned int c = lds[tid] + lds[64 + tid] + lds[128 + tid] + lds[192 + tid];
    if (c) atomicAdd(&out[(long long)m * 64 + tid], (unsigned long long)c);
  }
}

inline int grid_x(long long n, int per_thread) {
  long long blocks = (n + (long long)KV_THREADS * per_thread - 1) / ((long long)KV_THREADS * per_thread);
  if (blocks < 1) blocks = 1;
  // 256 CUs x 2 blocks; grid-stride covers the rest.  Every block ends with one global atomic
  // per non-empty bin on the SAME few hundred addresses: with 2048 blocks that flush (about a
  // million contended atomics) cost more than streaming the samples.
  if (blocks > 512) blocks = 512;
  return (int)blocks;
}

// ---------------------------------------------------------------------------------------------
// Live inter-token gaps from raw token events.  events: int64 [n][3] = (request id, token index,
// t_ns) in arrival order, as the load generator's SSE parser appended them; token index < 0 marks
// the end of a request.  The gap of an event with index k > 0 is its time minus the time of the
// same request's previous event: event (req, k - 1) of this batch if it is here, else the time the
// request was last seen in an earlier batch (last_t[req], -1: never).  Three launches:
//   insert: (req, index) -> t of every event into a batch-local open-addressing table
//   emit  : gaps[i] (ms, fp32; NaN for events without a gap) — what stream_hist_kernel then bins
//   commit: last_t[req] = max(last_t[req], t) over the batch (timestamps of a request only grow)
// so the host does no per-event work at all: one copy of the ring's new slice, then launches.
__device__ inline unsigned long long event_key(long long req, long long idx) {
  return ((((unsigned long long)req) << 24) | ((unsigned long long)idx & 0xFFFFFFull)) + 1ull;  // 0 = empty slot
}
__device__ inline unsigned int event_hash(unsigned long long k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
  return (unsigned int)k;
}

__global__ __launch_bounds__(256) void gap_insert_kernel(const long long* __restrict__ ev, int n, unsigned long long* __restrict__ keys, long long* __restrict__ vals,
                                                         unsigned int mask) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const long long req = ev[3 * i], idx = ev[3 * i + 1], t = ev[3 * i + 2];
  if (idx < 0 || req < 0) return;
  const unsigned long long key = event_key(req, idx);
  unsigned int h = event_hash(key) & mask;
  for (unsigned int probe = 0; probe <= mask; ++probe) {  // the table holds at least 2 n slots: a free one exists
    const unsigned long long old = atomicCAS(&keys[h], 0ull, key);
    if (old == 0ull || old == key) { vals[h] = t; return; }
    h = (h + 1) & mask;
  }
}

__global__ __launch_bounds__(256) void gap_emit_kernel(const long long* __restrict__ ev, int n, const unsigned long long* __restrict__ keys,
                                                       const long long* __restrict__ vals, unsigned int mask, const long long* __restrict__ last_t, long long n_last,
                                                       float* __restrict__ gaps) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const long long req = ev[3 * i], idx = ev[3 * i + 1], t = ev[3 * i + 2];
  float gap = __builtin_nanf("");
  if (idx > 0 && req >= 0) {
    long long prev = -1;
    const unsigned long long key = event_key(req, idx - 1);
    unsigned int h = event_hash(key) & mask;
    for (unsigned int probe = 0; probe <= mask; ++probe) {
      const unsigned long long k = keys[h];
      if (k == key) { prev = vals[h]; break; }
      if (k == 0ull) break;
      h = (h + 1) & mask;
    }
    if (prev < 0 && req < n_last) prev = last_t[req];
    if (prev >= 0) gap = (float)((double)(t - prev) * 1e-6);
  }
  gaps[i] = gap;
}

__global__ __launch_bounds__(256) void gap_commit_kernel(const long long* __restrict__ ev, int n, long long* __restrict__ last_t, long long n_last) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const long long req = ev[3 * i], idx = ev[3 * i + 1], t = ev[3 * i + 2];
  if (idx < 0 || req < 0 || req >= n_last) return;
  atomicMax(&last_t[req], t);
}

}  // namespace

extern "C" {

// keys / vals: scratch tables of `slots` entries (a power of two >= 2 n); keys must be zero on entry
// (the caller clears it: hipMemsetAsync in the binding).  last_t: int64 [n_last], -1 = never seen.
int kv_token_gaps(const long long* ev, int n, unsigned long long* keys, long long* vals, int slots, long long* last_t, long long n_last, float* gaps, hipStream_t stream) {
  if (n <= 0) return 0;
  if (slots < 2 * n || (slots & (slots - 1)) != 0 || n_last < 0) return -1;
  const dim3 grid((n + 255) / 256), block(256);
  hipLaunchKernelGGL(gap_insert_kernel, grid, block, 0, stream, ev, n, keys, vals, (unsigned int)(slots - 1));
  hipLaunchKernelGGL(gap_emit_kernel, grid, block, 0, stream, ev, n, keys, vals, (unsigned int)(slots - 1), last_t, n_last, gaps);
  hipLaunchKernelGGL(gap_commit_kernel, grid, block, 0, stream, ev, n, last_t, n_last);
  return (int)hipGetLastError();
}

size_t kv_stream_state_bytes() { return sizeof(StreamState); }
size_t kv_radix_state_bytes() { return sizeof(RadixState); }

// dtype: 0 = float32, 1 = float64
int kv_stream_hist(const void* x, long long n, long long row_stride, int n_metrics, int dtype, unsigned int* hist, void* state,
                   float lo_log2, float hi_log2, hipStream_t stream) {
  if (n <= 0 || n_metrics <= 0) return 0;
  float inv_step = (float)KV_BINS / (hi_log2 - lo_log2);
  dim3 grid(grid_x(n, 16), n_metrics);
  if (dtype == 0)
    hipLaunchKernelGGL(stream_hist_kernel<float>, grid, dim3(KV_THREADS), 0, stream, (const float*)x, n, row_stride, hist, (StreamState*)state, lo_log2, inv_step);
  else
    hipLaunchKernelGGL(stream_hist_kernel<double>, grid, dim3(KV_THREADS), 0, stream, (const double*)x, n, row_stride, hist, (StreamState*)state, lo_log2, inv_step);
  return (int)hipGetLastError();
}

// Initialise the selection state from quantile levels ``ps`` (device array, n_ps <= 4)
// and the sample counts that kv_stream_hist accumulated.
int kv_radix_init(const void* stream_state, void* st, const double* ps, int n_ps, int n_metrics, hipStream_t stream) {
  hipLaunchKernelGGL(radix_init_kernel, dim3(n_metrics), dim3(64), 0, stream, (const StreamState*)stream_state, (RadixState*)st, ps, n_ps);
  return (int)hipGetLastError();
}

// Runs every radix pass.  ``st`` must be initialised by the caller (prefix = 0, k = target
// rank, n_ranks), ``hist`` is [n_metrics][8][2048] zeros.  On return st->prefix[r]
// holds the order-preserving key of rank r (left-aligned in 64 bits).
int kv_radix_select(const void* x, long long n, long long row_stride, int n_metrics, int dtype, void* st, unsigned int* hist, hipStream_t stream) {
  if (n <= 0 || n_metrics <= 0) return 0;
  const int key_bits = dtype == 0 ? 32 : 64;
  dim3 grid(grid_x(n, 16), n_metrics);
  int top = key_bits;
  int first = 1;
  while (top > 0) {
    int digit_bits = top >= 11 ? 11 : top;
    int shift = top - digit_bits;
    if (dtype == 0)
      hipLaunchKernelGGL(radix_hist_kernel<float>, grid, dim3(KV_THREADS), 0, stream, (const float*)x, n, row_stride, (const RadixState*)st, hist, shift, digit_bits, first);
    else
      hipLaunchKernelGGL(radix_hist_kernel<double>, grid, dim3(KV_THREADS), 0, stream, (const double*)x, n, row_stride, (const RadixState*)st, hist, shift, digit_bits, first);
    // prefixes are kept left-aligned in 64 bits for both dtypes
    hipLaunchKernelGGL(radix_scan_kernel, dim3(n_metrics), dim3(KV_THREADS), 0, stream, (RadixState*)st, hist, shift + (64 - key_bits), first);
    first = 0;
    top = shift;
  }
  return (int)hipGetLastError();
}

// The same selection one pass at a time, for samples spread over several ranks: every rank runs
// pass_hist on its own samples, the ranks SUM the digit histograms (one all-reduce of
// [n_metrics][8][2048] counts per pass), and every rank runs pass_scan on the summed counts — the
// selection state stays identical everywhere and ends at the exact order statistics of the union.
// The state must have been initialised from the GLOBAL sample count.
int kv_radix_num_passes(int dtype) { return dtype == 0 ? 3 : 6; }

static void radix_pass_geometry(int dtype, int pass, int* shift, int* digit_bits) {
  int top = dtype == 0 ? 32 : 64;
  for (int i = 0;; ++i) {
    const int db = top >= 11 ? 11 : top;
    if (i == pass) { *shift = top - db; *digit_bits = db; return; }
    top -= db;
  }
}

int kv_radix_pass_hist(const void* x, long long n, long long row_stride, int n_metrics, int dtype, const void* st, unsigned int* hist, int pass, hipStream_t stream) {
  if (n_metrics <= 0 || pass < 0 || pass >= kv_radix_num_passes(dtype)) return -1;
  if (n <= 0) return 0;  // a rank without samples contributes zeros
  int shift = 0, digit_bits = 0;
  radix_pass_geometry(dtype, pass, &shift, &digit_bits);
  dim3 grid(grid_x(n, 16), n_metrics);
  if (dtype == 0)
    hipLaunchKernelGGL(radix_hist_kernel<float>, grid, dim3(KV_THREADS), 0, stream, (const float*)x, n, row_stride, (const RadixState*)st, hist, shift, digit_bits, pass == 0 ? 1 : 0);
  else
    hipLaunchKernelGGL(radix_hist_kernel<double>, grid, dim3(KV_THREADS), 0, stream, (const double*)x, n, row_stride, (const RadixState*)st, hist, shift, digit_bits, pass == 0 ? 1 : 0);
  return (int)hipGetLastError();
}

int kv_radix_pass_scan(void* st, unsigned int* hist, int n_metrics, int dtype, int pass, hipStream_t stream) {
  if (n_metrics <= 0 || pass < 0 || pass >= kv_radix_num_passes(dtype)) return -1;
  int shift = 0, digit_bits = 0;
  radix_pass_geometry(dtype, pass, &shift, &digit_bits);
  const int key_bits = dtype == 0 ? 32 : 64;
  hipLaunchKernelGGL(radix_scan_kernel, dim3(n_metrics), dim3(KV_THREADS), 0, stream, (RadixState*)st, hist, shift + (64 - key_bits), pass == 0 ? 1 : 0);
  return (int)hipGetLastError();
}

int kv_equal_hist(const void* x, long long n, long long row_stride, int n_metrics, int dtype, const void* state, unsigned long long* out, int n_bins,
                  hipStream_t stream) {
  if (n <= 0 || n_metrics <= 0) return 0;
  if (n_bins > 64) return -1;
  dim3 grid(grid_x(n, 8), n_metrics);
  if (dtype == 0)
    hipLaunchKernelGGL(equal_hist_kernel<float>, grid, dim3(KV_THREADS), 0, stream, (const float*)x, n, row_stride, (const StreamState*)state, out, n_bins);
  else
    hipLaunchKernelGGL(equal_hist_kernel<double>, grid, dim3(KV_THREADS), 0, stream, (const double*)x, n, row_stride, (const StreamState*)state, out, n_bins);
  return (int)hipGetLastError();
}

}  // extern "C"

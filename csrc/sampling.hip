// Temperature / top-k / top-p sampling of logits rows on gfx950 — one 1024-thread workgroup per
// row, no sort.
//
// Sampling from the nucleus does not need the sorted order, only the THRESHOLD: the kept set is
// "every token whose logit is >= t".  bf16 logits have 65 536 distinct values, so with the
// order-preserving 16-bit key of a logit the threshold is found by a 16-ary search over the key
// space (four passes, 15 candidate thresholds per pass in per-thread accumulators, no atomics):
//   top-k : counts   — the largest key K with  #{key >= K} >= k
//   top-p : weights  — the largest key K with  W{key >= K} >= p * Z,  w = exp((x - max) / T),
//           taken over the tokens that survived top-k, Z their total weight
// (tokens that tie with the threshold value are all kept: the sort-based reference keeps an
// arbitrary subset of them).  The token is then drawn by inverting the running sum of the kept
// weights in index order at u * W_kept: a block scan over 1024 per-thread segment sums and a
// walk through one segment.  About a dozen passes over a row that sits in L2 (256 KB for a 128 k
// vocabulary) against a full sort + cumsum + scatter over fp32 copies of it.

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

constexpr int TPB = 1024;

__device__ inline float bf2f(unsigned short b) { return __uint_as_float(((unsigned int)b) << 16); }
// order-preserving key: larger logit <=> larger key (NaNs sort above +inf; never produced by a GEMM of finite inputs)
__device__ inline unsigned int key16(unsigned short b) { return (b & 0x8000u) ? (unsigned int)(~b & 0xffffu) : (unsigned int)(b | 0x8000u); }

typedef __attribute__((ext_vector_type(8))) unsigned short us8;

// f(raw bf16 bits, index) for every element of the row, each thread in a fixed order of its own:
// 16-byte loads (8 logits), several in flight — a 2-byte load per trip leaves the loop waiting
// an L2 round trip for every element.
template <typename F>
__device__ inline void for_each_logit(const unsigned short* __restrict__ row, int V, F f) {
  int done = 0;
  if ((((uintptr_t)row) & 15) == 0) {
    const us8* rv = reinterpret_cast<const us8*>(row);
    const int nv = V >> 3;
#pragma unroll 4
    for (int i = threadIdx.x; i < nv; i += TPB) {
      const us8 a = rv[i];
#pragma unroll
      for (int j = 0; j < 8; ++j) f(a[j], i * 8 + j);
    }
    done = nv * 8;
  }
  for (int i = done + threadIdx.x; i < V; i += TPB) f(row[i], i);
}

struct Sel {
  float red[TPB / 64];
  float part[TPB / 64][16];
  unsigned int res;
  float above;
};

__device__ inline float block_max(float v, float* red) {
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = red[0];
  for (int w = 1; w < TPB / 64; ++w) r = fmaxf(r, red[w]);
  __syncthreads();
  return r;
}

__device__ inline float block_sum(float v, float* red) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = 0.f;
  for (int w = 0; w < TPB / 64; ++w) r += red[w];
  __syncthreads();
  return r;
}

// The largest key K >= floor_key with  S{key >= K} >= target,  S = weight (BY_WEIGHT) or count of
// the keys >= floor_key.  16-ary search over the 16-bit key space: four passes over the row,
// each of which evaluates S at 15 candidate thresholds at once (per-thread accumulators, one
// block reduction) — no atomics, so neither contention on the few histogram bins that logits
// crowd into nor run-to-run variation.  If even the floor does not reach the target (rounding),
// the floor is returned.
template <bool BY_WEIGHT>
__device__ inline unsigned int select_threshold(const unsigned short* __restrict__ row, int V, float m, float inv_t, unsigned int floor_key, float target, Sel& s) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // invariant: S{key >= lo} >= target > S{key >= hi}; `above` = S{key >= hi}.  Only the keys in
  // [lo, hi) need the 15 comparisons; everything from hi up counts for every candidate alike.
  unsigned int lo = 0, hi = 0x10000u;
  float above = 0.f;
  for (int shift = 12; shift >= 0; shift -= 4) {
    float acc[15];
#pragma unroll
    for (int j = 0; j < 15; ++j) acc[j] = 0.f;
    for_each_logit(row, V, [&](unsigned short raw, int) {
      const unsigned int k = key16(raw);
      if (k >= floor_key && k >= lo && k < hi) {
        const float w = BY_WEIGHT ? __expf((bf2f(raw) - m) * inv_t) : 1.f;
        const unsigned int d = (k - lo) >> shift;  // candidates lo + (j << shift), j = 1..15, that k reaches
#pragma unroll
        for (int j = 0; j < 15; ++j) acc[j] += (d >= (unsigned int)(j + 1)) ? w : 0.f;
      }
    });
#pragma unroll
    for (int j = 0; j < 15; ++j) {
      float v = acc[j];
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      if (lane == 0) s.part[wave][j] = v;
    }
    __syncthreads();
    if (tid == 0) {
      unsigned int best = 0;
      float next_above = above;  // S at the candidate right above the chosen one
      for (int j = 14; j >= 0; --j) {  // from the top: the first candidate that reaches the target is the largest such
        float t = above;
        for (int w = 0; w < TPB / 64; ++w) t += s.part[w][j];
        if (t >= target) { best = (unsigned int)(j + 1); break; }
        next_above = t;
      }
      s.res = best;
      s.above = next_above;
    }
    __syncthreads();
    const unsigned int best = s.res;
    above = s.above;
    if (best < 15) hi = lo + ((best + 1) << shift);
    lo = lo + (best << shift);
    __syncthreads();
  }
  return lo > floor_key ? lo : floor_key;
}

// logits: bf16 [*, V] rows (row_stride elements apart); rows[r] = row to sample; out[rows[r]] = id.
__global__ __launch_bounds__(TPB) void sample_rows_kernel(const unsigned short* __restrict__ logits, long long row_stride, int V, const long long* __restrict__ rows,
                                                          const float* __restrict__ temperature, const int* __restrict__ top_k, const float* __restrict__ top_p,
                                                          const float* __restrict__ uniform, long long* __restrict__ out) {
  __shared__ Sel s;
  __shared__ float scan[TPB / 64];
  __shared__ int winner;
  const int r = blockIdx.x, tid = threadIdx.x;
  const long long row_id = rows[r];
  const unsigned short* row = logits + row_id * row_stride;
  const float inv_t = 1.f / fmaxf(temperature[r], 1e-5f);
  // ---- row maximum
  float mx = -INFINITY;
  for_each_logit(row, V, [&](unsigned short raw, int) { mx = fmaxf(mx, bf2f(raw)); });
  const float m = block_max(mx, s.red);
  // ---- top-k threshold (by count)
  unsigned int thr = 0;
  const int k = top_k[r];
  if (k > 0 && k < V) thr = select_threshold<false>(row, V, m, inv_t, 0u, (float)k, s);
  // ---- total weight of the survivors, then the top-p threshold (by weight) among them
  float z = 0.f;
  for_each_logit(row, V, [&](unsigned short raw, int) {
    if (key16(raw) >= thr) z += __expf((bf2f(raw) - m) * inv_t);
  });
  float kept = block_sum(z, s.red);
  const float p = top_p[r];
  if (p < 1.f) {
    const unsigned int thr_p = select_threshold<true>(row, V, m, inv_t, thr, p * kept, s);
    if (thr_p > thr) thr = thr_p;
  }
  // ---- draw: invert the running sum of the kept weights at u * W_kept.  Any fixed order of the
  // tokens gives the right distribution; the order here is thread-major (all of thread 0's
  // elements in its own traversal order, then thread 1's, ...): a block scan over the per-thread
  // sums finds the thread, which then walks its elements again.
  float mine = 0.f;
  for_each_logit(row, V, [&](unsigned short raw, int) {
    if (key16(raw) >= thr) mine += __expf((bf2f(raw) - m) * inv_t);
  });
  float inc = mine;
  const int lane = tid & 63, wave = tid >> 6;
  for (int o = 1; o < 64; o <<= 1) {
    const float t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) scan[wave] = inc;
  if (tid == 0) winner = -1;
  __syncthreads();
  float base = 0.f;
  for (int w = 0; w < wave; ++w) base += scan[w];
  float total = 0.f;
  for (int w = 0; w < TPB / 64; ++w) total += scan[w];
  inc += base;
  const float exc = inc - mine;
  const float target = fminf(uniform[r], 0.99999994f) * total;
  // one thread's half-open interval [exc, inc) holds the target (inc is monotone, the last one = total > target)
  if (mine > 0.f && exc <= target && target < inc) winner = tid;
  __syncthreads();
  // rounding at an interval edge can leave no winner: fall back to the last thread that kept
  // anything.  The decision is taken on a snapshot that every thread reads between the same two
  // barriers, so all waves take the same side of the branch (the barrier inside it is uniform).
  const int w0 = winner;
  __syncthreads();
  if (w0 < 0) {
    if (mine > 0.f) atomicMax(&winner, tid);
    __syncthreads();
  }
  if (tid == winner) {
    float acc = exc;
    long long pick = -1, last_kept = -1;
    for_each_logit(row, V, [&](unsigned short raw, int i) {
      if (pick < 0 && key16(raw) >= thr) {
        acc += __expf((bf2f(raw) - m) * inv_t);
        last_kept = i;
        if (acc > target) pick = i;
      }
    });
    out[row_id] = pick >= 0 ? pick : (last_kept >= 0 ? last_kept : 0);
  }
}

}  // namespace

extern "C" int kv_sample_rows(const void* logits, long long row_stride, int V, const long long* rows, int R, const float* temperature, const int* top_k,
                              const float* top_p, const float* uniform, long long* out, hipStream_t stream) {
  if (R <= 0) return 0;
  if (V <= 0) return -1;
  hipLaunchKernelGGL(sample_rows_kernel, dim3(R), dim3(TPB), 0, stream, (const unsigned short*)logits, row_stride, V, rows, temperature, top_k, top_p, uniform, out);
  return (int)hipGetLastError();
}

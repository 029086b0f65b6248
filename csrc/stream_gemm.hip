// Weight-streaming GEMM v2 for decode / small mixed batches on gfx950 (MI355X):
//     y[M, N] = x[M, K] @ W[N, K]^T     bf16 in, fp32 accumulate,  M in {16, 32, 48, 64}
//
// The op is bound by streaming W from HBM once; everything else has to stay out of the
// way.  Measured on the v1 kernel (csrc/skinny_gemm.hip), the time grew linearly with M
// because every wave re-read its x fragments from L2 in 64-byte row pieces.  Here:
//
//  * W is stored pre-packed in MFMA fragment order [N/16][K/32][64 lanes][8 bf16], so a
//    wave-level load of one A fragment tile is one contiguous 1 KiB, and a wave's k-steps
//    are consecutive KiBs.  Weights go global -> VGPR directly (each byte is used once),
//    register double-buffered one K-chunk ahead.
//  * A workgroup = 4 waves = 64 output features (one 16-row tile per wave) x one K-slice.
//    The x chunk [M x KC] is fetched ONCE per workgroup, written to LDS in B-fragment order
//    (linear, conflict-free ds_write_b128 / ds_read_b128) and shared by the 4 waves:
//    x traffic per weight byte drops from M/32 to M/64 and leaves the vector-memory path to W.
//  * Narrow outputs (N = 4096..6144) do not give 256 CUs enough workgroups, so K is split
//    across workgroups (grid.y); each split writes an fp32 slab [split][M][N] and the NEXT
//    kernel (residual-add + RMSNorm, or RoPE) sums the slabs in its prologue — the
//    launch-boundary reduce: no atomics, no extra kernel, bitwise reproducible.
//  * SwiGLU epilogue for the gate/up projection: rows are interleaved in groups of 16
//    (even tiles = gate, odd tiles = up), so wave pairs (0,1) and (2,3) exchange their
//    accumulators through LDS and write silu(gate) * up in bf16.
//
// mfma_f32_16x16x32_bf16 operand maps: A lane l = W[row l&15][k = 8*(l>>4) + j], B lane l =
// x[token l&15][k = 8*(l>>4) + j]; D lane l reg r = y[token l&15][feature 4*(l>>4) + r].

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 frag_ab;
typedef __attribute__((ext_vector_type(4))) float frag_cd;
typedef __attribute__((ext_vector_type(4))) unsigned short us4;

__device__ inline unsigned short f2bf(float f) {
  unsigned int u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

enum Epilogue { EPI_BF16 = 0, EPI_SLAB = 1, EPI_SWIGLU = 2 };

// KS = k-steps (of 32) per chunk.  grid = (N/64, splits).  chunks_per_split chunks per workgroup.
template <int MT, int KS, int EPI>
__global__ __launch_bounds__(256) void stream_gemm_kernel(const __bf16* __restrict__ Wp, const __bf16* __restrict__ x, void* __restrict__ out, int N, int K,
                                                          long long ldx, long long ldy, int chunks_per_split, long long slab_stride,
                                                          const float* __restrict__ row_ss, int ss_parts, float inv_k, float eps) {
  constexpr int XF = KS * MT;          // x fragment tiles (1 KiB each) per chunk
  constexpr int XL = (XF + 3) / 4;     // per-thread 16-byte x loads per chunk
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[2 * XF * 1024 > 4 * 64 * 4 * 4 * MT ? 2 * XF * 1024 : 4 * 64 * 4 * 4 * MT];
  frag_ab* xs = reinterpret_cast<frag_ab*>(lds_raw);  // [2][XF][64]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 15, q = lane >> 4;
  const int ksteps = K >> 5;
  const int nchunks = ksteps / KS;
  const int c0 = blockIdx.y * chunks_per_split;
  const int c1 = c0 + chunks_per_split < nchunks ? c0 + chunks_per_split : nchunks;
  const long long tile = (long long)blockIdx.x * 4 + wave;  // 16-row tile of W owned by this wave
  const __bf16* wp = Wp + (tile * ksteps * 64 + lane) * 8;

  // x fragment slot s = tid + 256*i  ->  tile f = s/64 (= wave + 4*i), ks = f / MT, j = f % MT
  const __bf16* xsrc[XL];
  bool xok[XL];
#pragma unroll
  for (int i = 0; i < XL; ++i) {
    const int f = wave + 4 * i;
    xok[i] = f < XF;
    const int ks = f / MT, j = f % MT;
    xsrc[i] = x + (long long)(j * 16 + r) * ldx + ks * 32 + q * 8;
  }

  frag_cd acc[MT];
#pragma unroll
  for (int j = 0; j < MT; ++j) acc[j] = (frag_cd){0.f, 0.f, 0.f, 0.f};

  frag_ab a_cur[KS], a_nxt[KS], xr[XL];

  // ---- prologue: chunk c0
  if (c0 < c1) {
#pragma unroll
    for (int i = 0; i < XL; ++i)
      if (xok[i]) xr[i] = *reinterpret_cast<const frag_ab*>(xsrc[i] + (long long)c0 * KS * 32);
#pragma unroll
    for (int u = 0; u < KS; ++u) a_cur[u] = *reinterpret_cast<const frag_ab*>(wp + ((long long)c0 * KS + u) * 512);
#pragma unroll
    for (int i = 0; i < XL; ++i)
      if (xok[i]) xs[(0 * XF + wave + 4 * i) * 64 + lane] = xr[i];
  }

  for (int c = c0; c < c1; ++c) {
    const int buf = (c - c0) & 1;
    const bool more = c + 1 < c1;
    if (more) {
#pragma unroll
      for (int i = 0; i < XL; ++i)
        if (xok[i]) xr[i] = *reinterpret_cast<const frag_ab*>(xsrc[i] + (long long)(c + 1) * KS * 32);
#pragma unroll
      for (int u = 0; u < KS; ++u) a_nxt[u] = *reinterpret_cast<const frag_ab*>(wp + ((long long)(c + 1) * KS + u) * 512);
    }
    __syncthreads();  // x chunk c is in LDS buffer `buf`; everyone is done reading buffer `buf^1`
#pragma unroll
    for (int u = 0; u < KS; ++u) {
      frag_ab b[MT];
#pragma unroll
      for (int j = 0; j < MT; ++j) b[j] = xs[(buf * XF + u * MT + j) * 64 + lane];
#pragma unroll
      for (int j = 0; j < MT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_cur[u], b[j], acc[j], 0, 0, 0);
    }
    if (more) {
#pragma unroll
      for (int i = 0; i < XL; ++i)
        if (xok[i]) xs[((buf ^ 1) * XF + wave + 4 * i) * 64 + lane] = xr[i];
#pragma unroll
      for (int u = 0; u < KS; ++u) a_cur[u] = a_nxt[u];
    }
  }

  // ---- epilogue.  Lane holds y[token j*16 + r][feature tile*16 + 4q + e], e = 0..3.
  const long long n_base = tile * 16 + q * 4;
  // Folded RMSNorm: x is the un-normalised residual stream and W carries the norm gain,
  // so rows are scaled by 1/rms(x[m]) here (slabs are scaled by their consumer).
  if (EPI != EPI_SLAB && row_ss != nullptr) {
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      float ssum = 0.f;
      for (int p = 0; p < ss_parts; ++p) ssum += row_ss[(long long)(j * 16 + r) * ss_parts + p];
      const float rs = rsqrtf(ssum * inv_k + eps);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[j][e] *= rs;
    }
  }
  if (EPI == EPI_SLAB) {
    float* o = reinterpret_cast<float*>(out) + (long long)blockIdx.y * slab_stride;
#pragma unroll
    for (int j = 0; j < MT; ++j) *reinterpret_cast<frag_cd*>(o + (long long)(j * 16 + r) * ldy + n_base) = acc[j];
  } else if (EPI == EPI_BF16) {
    unsigned short* o = reinterpret_cast<unsigned short*>(out);
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      us4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = f2bf(acc[j][e]);
      *reinterpret_cast<us4*>(o + (long long)(j * 16 + r) * ldy + n_base) = v;
    }
  } else {
    // SwiGLU: odd waves (up tiles) hand their accumulators to the even wave of the pair.
    __syncthreads();  // all waves are done with the x buffers that alias `ex`
    float* ex = reinterpret_cast<float*>(lds_raw);  // [2 pairs][MT][4][64]
    if (wave & 1) {
#pragma unroll
      for (int j = 0; j < MT; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) ex[(((wave >> 1) * MT + j) * 4 + e) * 64 + lane] = acc[j][e];
    }
    __syncthreads();
    if (!(wave & 1)) {
      unsigned short* o = reinterpret_cast<unsigned short*>(out);
      const long long i_base = (tile >> 1) * 16 + q * 4;  // intermediate feature index
#pragma unroll
      for (int j = 0; j < MT; ++j) {
        us4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float g = acc[j][e], up = ex[(((wave >> 1) * MT + j) * 4 + e) * 64 + lane];
          v[e] = f2bf(g / (1.f + __expf(-g)) * up);
        }
        *reinterpret_cast<us4*>(o + (long long)(j * 16 + r) * ldy + i_base) = v;
      }
    }
  }
}

template <int MT, int KS, int EPI>
int launch(const void* Wp, const void* x, void* out, int N, int K, long long ldx, long long ldy, int splits, int cps, long long slab_stride,
           const float* row_ss, int ss_parts, float eps, hipStream_t stream) {
  dim3 grid(N / 64, splits), block(256);
  hipLaunchKernelGGL((stream_gemm_kernel<MT, KS, EPI>), grid, block, 0, stream, (const __bf16*)Wp, (const __bf16*)x, out, N, K, ldx, ldy, cps, slab_stride,
                     row_ss, ss_parts, 1.f / (float)K, eps);
  return (int)hipGetLastError();
}

template <int MT, int KS>
int dispatch_epi(int epi, const void* Wp, const void* x, void* out, int N, int K, long long ldx, long long ldy, int splits, int cps, long long slab_stride,
                 const float* row_ss, int ss_parts, float eps, hipStream_t stream) {
  switch (epi) {
    case EPI_BF16: return launch<MT, KS, EPI_BF16>(Wp, x, out, N, K, ldx, ldy, splits, cps, slab_stride, row_ss, ss_parts, eps, stream);
    case EPI_SLAB: return launch<MT, KS, EPI_SLAB>(Wp, x, out, N, K, ldx, ldy, splits, cps, slab_stride, row_ss, ss_parts, eps, stream);
    case EPI_SWIGLU: return launch<MT, KS, EPI_SWIGLU>(Wp, x, out, N, K, ldx, ldy, splits, cps, slab_stride, row_ss, ss_parts, eps, stream);
  }
  return -1;
}

}  // namespace

extern "C" {

// Number of K-splits the kernel will use for a shape when up to `max_splits` are allowed
// (callers size the slab buffer with it).  Splits are chosen so that the grid has at
// least ~2 workgroups per CU and every split owns at least one chunk.
int kv_stream_gemm_plan(int N, int K, int ks_per_chunk, int max_splits, int* chunks_per_split) {
  const int nchunks = (K >> 5) / ks_per_chunk;
  const int tiles = N / 64;
  int want = (512 + tiles - 1) / tiles;
  if (want > max_splits) want = max_splits;
  if (want > nchunks) want = nchunks;
  if (want < 1) want = 1;
  int cps = (nchunks + want - 1) / want;
  if (cps < 1) cps = 1;
  const int splits = (nchunks + cps - 1) / cps;
  if (chunks_per_split) *chunks_per_split = cps;
  return splits;
}

// epi: 0 = bf16 y[M, N] (single split), 1 = fp32 slabs out[splits][M][N], 2 = SwiGLU bf16 y[M, N/2].
// ks_per_chunk: 4 or 8.  W must be packed (kv_pack_weight).  Requirements: M in {16,32,48,64},
// N % 64 == 0 (N % 128 for SwiGLU), K % (32*ks_per_chunk) == 0.
int kv_stream_gemm(const void* Wp, const void* x, void* out, int M, int N, int K, long long ldx, long long ldy, int epi, int ks_per_chunk, int splits,
                   int chunks_per_split, long long slab_stride, const float* row_ss, int ss_parts, float eps, hipStream_t stream) {
  if (M <= 0 || M > 64 || (M & 15) || (N & 63) || (ks_per_chunk != 4 && ks_per_chunk != 8) || (K % (32 * ks_per_chunk))) return -1;
  if (epi == EPI_SWIGLU && (N & 127)) return -1;
  if (epi != EPI_SLAB && splits != 1) return -1;
  if (splits < 1 || chunks_per_split < 1 || (long long)splits * chunks_per_split < (K >> 5) / ks_per_chunk) return -1;
  const int mt = M >> 4;
#define KV_CASE(MTV)                                                                                                                 \
  case MTV:                                                                                                                          \
    return ks_per_chunk == 4 ? dispatch_epi<MTV, 4>(epi, Wp, x, out, N, K, ldx, ldy, splits, chunks_per_split, slab_stride, row_ss, ss_parts, eps, stream)  \
                             : dispatch_epi<MTV, 8>(epi, Wp, x, out, N, K, ldx, ldy, splits, chunks_per_split, slab_stride, row_ss, ss_parts, eps, stream);
  switch (mt) {
    KV_CASE(1)
    KV_CASE(2)
    KV_CASE(3)
    KV_CASE(4)
  }
#undef KV_CASE
  return -1;
}

}  // extern "C"

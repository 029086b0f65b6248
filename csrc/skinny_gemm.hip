// Weight-streaming "skinny" GEMM for decode on gfx950 (MI355X):
//     y[M, N] = x[M, K] @ W[N, K]^T          bf16 in, fp32 accumulate, bf16 out,  M <= 64.
//
// A decode step of an 8B model reads every weight once (16 GB) for a few dozen token
// rows, so the op is bound by HBM streaming of W, not by math.  Design:
//
//  * MFMA 16x16x32 bf16 with W as the A operand (16 output features x 32 k per
//    instruction) and x^T as the B operand (16 token columns).  A lane's A fragment is
//    8 consecutive k of one W row = one 16-byte global load straight into VGPRs — no LDS
//    staging: each W byte is used exactly once (nothing to share between waves), and
//    loads stay asynchronous until their first use.  x (M x K, <= 2 MB) is re-read by every
//    workgroup from L2.
//  * One workgroup owns 16*NR output features; its WAVES wavefronts split K between them
//    (in-workgroup split-K) and each keeps U k-steps of loads in flight in a two-deep
//    register pipeline, so a CU holds >= 32 KiB of outstanding weight reads — what it
//    takes to cover HBM latency at ~6 TB/s over 256 CUs.  Partial sums meet in LDS.
//  * Optional SwiGLU epilogue: with gate/up rows interleaved in groups of 8 (NR = 2:
//    tile 0 = gate, tile 1 = up of the same 16 intermediate features) the workgroup writes
//    silu(gate) * up directly, so the [M, 2I] intermediate never reaches memory.
//  * PACKED weight layout (optional): W re-ordered offline into MFMA fragment order
//    [N/16][K/32][64 lanes][8], so every wave-level load instruction reads one contiguous
//    1 KiB instead of 16 separate 64-byte row pieces.
//
// D layout of mfma_f32_16x16x32: lane l, register r holds D[row = 4*(l>>4) + r][col = l&15]
// = y[token = l&15][feature = 4*(l>>4) + r]: four consecutive features of one token ->
// one 8-byte store.

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 frag_ab;  // 8 bf16 = 16 bytes = 4 VGPRs
typedef __attribute__((ext_vector_type(4))) float frag_cd;
typedef __attribute__((ext_vector_type(4))) unsigned short us4;

__device__ inline unsigned short f2bf(float f) {
  unsigned int u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

template <int MT, int NR, int WAVES, int U, bool SWIGLU, bool PACKED>
__global__ __launch_bounds__(WAVES * 64) void skinny_gemm_kernel(const __bf16* __restrict__ W, const __bf16* __restrict__ x, unsigned short* __restrict__ y,
                                                                 int N, int K, long long ldx, long long ldy) {
  constexpr int NT = NR * MT;
  __shared__ float red[WAVES * NT * 4 * 64];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 15, q = lane >> 4;
  const int n0 = blockIdx.x * 16 * NR;
  const int ksteps = K >> 5;
  const int per = (ksteps + WAVES - 1) / WAVES;
  const int ks0 = wave * per;
  const int ks1 = ks0 + per < ksteps ? ks0 + per : ksteps;

  // per-lane base pointers (k-step 0)
  const __bf16* wp[NR];
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    if (PACKED)
      wp[i] = W + ((long long)(blockIdx.x * NR + i) * ksteps * 64 + lane) * 8;
    else
      wp[i] = W + (long long)(n0 + i * 16 + r) * K + q * 8;
  }
  const __bf16* xp[MT];
#pragma unroll
  for (int j = 0; j < MT; ++j) xp[j] = x + (long long)(j * 16 + r) * ldx + q * 8;

  frag_cd acc[NR][MT];
#pragma unroll
  for (int i = 0; i < NR; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = (frag_cd){0.f, 0.f, 0.f, 0.f};

  frag_ab a0[U][NR], b0[U][MT], a1[U][NR], b1[U][MT];

#define KV_LOAD(A, B, KS)                                                                                     \
  {                                                                                                           \
    _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                           \
      const long long kk = (long long)((KS) + u);                                                             \
      _Pragma("unroll") for (int i = 0; i < NR; ++i) A[u][i] = *reinterpret_cast<const frag_ab*>(wp[i] + kk * (PACKED ? 512 : 32)); \
      _Pragma("unroll") for (int j = 0; j < MT; ++j) B[u][j] = *reinterpret_cast<const frag_ab*>(xp[j] + kk * 32);                   \
    }                                                                                                         \
  }
#define KV_MMA(A, B)                                                                                          \
  {                                                                                                           \
    _Pragma("unroll") for (int u = 0; u < U; ++u)                                                             \
      _Pragma("unroll") for (int i = 0; i < NR; ++i)                                                          \
        _Pragma("unroll") for (int j = 0; j < MT; ++j)                                                        \
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[u][i], B[u][j], acc[i][j], 0, 0, 0);           \
  }

  const int nchunks = ks1 > ks0 ? (ks1 - ks0) / U : 0;
  int ks = ks0;
  if (nchunks > 0) {
    KV_LOAD(a0, b0, ks);
    for (int c = 0; c < nchunks; c += 2) {
      if (c + 1 < nchunks) KV_LOAD(a1, b1, ks + U);
      KV_MMA(a0, b0);
      if (c + 2 < nchunks) KV_LOAD(a0, b0, ks + 2 * U);
      if (c + 1 < nchunks) KV_MMA(a1, b1);
      ks += 2 * U;
    }
  }
  // remainder k-steps (K/32 not a multiple of WAVES*U)
  for (int kk = ks0 + nchunks * U; kk < ks1; ++kk) {
    frag_ab ar[NR], br[MT];
#pragma unroll
    for (int i = 0; i < NR; ++i) ar[i] = *reinterpret_cast<const frag_ab*>(wp[i] + (long long)kk * (PACKED ? 512 : 32));
#pragma unroll
    for (int j = 0; j < MT; ++j) br[j] = *reinterpret_cast<const frag_ab*>(xp[j] + (long long)kk * 32);
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
      for (int j = 0; j < MT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ar[i], br[j], acc[i][j], 0, 0, 0);
  }
#undef KV_LOAD
#undef KV_MMA

  // ---- cross-wave reduction through LDS: red[wave][tile][reg][lane]
#pragma unroll
  for (int i = 0; i < NR; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) red[((wave * NT + i * MT + j) * 4 + e) * 64 + lane] = acc[i][j][e];
  __syncthreads();

  // token-tile j is finished by wave (j % WAVES)
  for (int j = wave; j < MT; j += WAVES) {
    float s[NR][4];
#pragma unroll
    for (int i = 0; i < NR; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) v += red[((w * NT + i * MT + j) * 4 + e) * 64 + lane];
        s[i][e] = v;
      }
    const long long row = (long long)(j * 16 + r) * ldy;
    if (SWIGLU) {
      // every 16-row tile holds 8 gate rows (lanes q < 2) and the 8 up rows of the same
      // features (lanes q >= 2): the up value sits 32 lanes further on
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        us4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float g = s[i][e], up = __shfl_xor(s[i][e], 32, 64);
          o[e] = f2bf(g / (1.f + __expf(-g)) * up);
        }
        if (q < 2) *reinterpret_cast<us4*>(y + row + ((long long)blockIdx.x * NR + i) * 8 + q * 4) = o;
      }
    } else {
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        us4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = f2bf(s[i][e]);
        *reinterpret_cast<us4*>(y + row + n0 + i * 16 + q * 4) = o;
      }
    }
  }
}

// Re-order W[N, K] (row-major) into MFMA fragment order [N/16][K/32][64][8].
__global__ __launch_bounds__(256) void pack_weight_kernel(const unsigned short* __restrict__ W, unsigned short* __restrict__ Wp, int N, int K) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;  // one 16-byte fragment per thread
  const int ksteps = K >> 5;
  const long long total = (long long)(N >> 4) * ksteps * 64;
  if (idx >= total) return;
  const int lane = (int)(idx & 63);
  const long long tile = idx >> 6;
  const int ks = (int)(tile % ksteps);
  const long long nt = tile / ksteps;
  const int r = lane & 15, q = lane >> 4;
  const us4* src = reinterpret_cast<const us4*>(W + (nt * 16 + r) * K + (long long)ks * 32 + q * 8);
  us4* dst = reinterpret_cast<us4*>(Wp + idx * 8);
  dst[0] = src[0];
  dst[1] = src[1];
}

template <int MT, int NR, int WAVES, int U, bool SWIGLU, bool PACKED>
int launch(const void* W, const void* x, void* y, int N, int K, long long ldx, long long ldy, hipStream_t stream) {
  dim3 grid(N / (16 * NR)), block(WAVES * 64);
  hipLaunchKernelGGL((skinny_gemm_kernel<MT, NR, WAVES, U, SWIGLU, PACKED>), grid, block, 0, stream, (const __bf16*)W, (const __bf16*)x, (unsigned short*)y,
                     N, K, ldx, ldy);
  return (int)hipGetLastError();
}

template <int MT, bool SWIGLU, bool PACKED>
int dispatch_shape(const void* W, const void* x, void* y, int N, int K, long long ldx, long long ldy, int waves, hipStream_t stream) {
  // NR = 2 halves the x re-reads per weight byte; NR = 1 doubles the workgroup count for
  // narrow outputs (N/32 < 2 workgroups per CU).  SwiGLU needs gate+up in one workgroup.
  const bool nr2 = SWIGLU || (N % 32 == 0 && N / 32 >= 512);
  if (nr2) {
    if (waves == 8) return launch<MT, 2, 8, (MT >= 3 ? 2 : 4), SWIGLU, PACKED>(W, x, y, N, K, ldx, ldy, stream);
    return launch<MT, 2, 4, (MT >= 3 ? 2 : 4), SWIGLU, PACKED>(W, x, y, N, K, ldx, ldy, stream);
  }
  if (SWIGLU) return -1;
  if (waves == 8) return launch<MT, 1, 8, 4, false, PACKED>(W, x, y, N, K, ldx, ldy, stream);
  return launch<MT, 1, 4, 4, false, PACKED>(W, x, y, N, K, ldx, ldy, stream);
}

}  // namespace

extern "C" {

// M must be 16, 32, 48 or 64 (the caller pads token buckets); N % 32 == 0 (N % 16 for
// non-SwiGLU NR=1 shapes), K % 32 == 0.  For SwiGLU, W holds 2*I interleaved rows
// (N = 2I) and y is [M, I].  waves: 4 or 8 wavefronts per workgroup (0 = pick).
int kv_skinny_gemm(const void* W, const void* x, void* y, int M, int N, int K, long long ldx, long long ldy, int swiglu, int packed, int waves,
                   hipStream_t stream) {
  if (M <= 0 || M > 64 || (M & 15) || (K & 31) || (N & 15)) return -1;
  if (swiglu && (N & 31)) return -1;
  if (waves == 0) waves = (K >= 8192) ? 8 : 4;
  const int mt = M >> 4;
#define KV_CASE(MTV)                                                                                                        \
  case MTV:                                                                                                                 \
    if (swiglu) return packed ? dispatch_shape<MTV, true, true>(W, x, y, N, K, ldx, ldy, waves, stream)                     \
                              : dispatch_shape<MTV, true, false>(W, x, y, N, K, ldx, ldy, waves, stream);                   \
    return packed ? dispatch_shape<MTV, false, true>(W, x, y, N, K, ldx, ldy, waves, stream)                               \
                  : dispatch_shape<MTV, false, false>(W, x, y, N, K, ldx, ldy, waves, stream);
  switch (mt) {
    KV_CASE(1)
    KV_CASE(2)
    KV_CASE(3)
    KV_CASE(4)
  }
#undef KV_CASE
  return -1;
}

int kv_pack_weight(const void* W, void* Wp, int N, int K, hipStream_t stream) {
  if ((N & 15) || (K & 31)) return -1;
  const long long total = (long long)(N >> 4) * (K >> 5) * 64;
  hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const unsigned short*)W, (unsigned short*)Wp, N, K);
  return (int)hipGetLastError();
}

}  // extern "C"

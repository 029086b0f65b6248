// Weight-streaming GEMM v2 for decode / small mixed batches on gfx950 (MI355X):
//     y[M, N] = x[M, K] @ W[N, K]^T     bf16 in, fp32 accumulate,  M in {16, 32, 48, 64}
//
// The op is bound by streaming W from HBM once; everything else has to stay out of the
// way.  Measured on the first kernel of this kind (no LDS staging of x), the time grew linearly with M
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
//  * SwiGLU epilogue for the gate/up projection: rows are interleaved in groups of 8, so every
//    16-row MFMA tile carries 8 gate rows and the 8 up rows of the same features; a wave
//    finishes silu(gate) * up on its own with one cross-lane read per accumulator register.
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

typedef __attribute__((ext_vector_type(4))) unsigned int raw16;  // one 16-byte weight load
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

// EPI_RESID: split-K slabs as EPI_SLAB, then the LAST workgroup of a feature group to arrive folds
// them into the residual stream (residual' = bf16(residual + bf16(sum of slabs)), sums of squares
// of residual' per row) — the residual add + RMSNorm statistics without a kernel of their own.
enum Epilogue { EPI_BF16 = 0, EPI_SLAB = 1, EPI_SWIGLU = 2, EPI_RESID = 3 };

// What EPI_RESID needs besides the slabs: the residual stream (bf16 [M, N], updated in place;
// has_residual = 0: treated as zero), the per-row partial sums of squares it leaves for the next
// projection ([M][feature groups]) and one arrival counter per feature group (zero at rest).
struct ResidArgs { unsigned short* residual; int has_residual; float* ss_out; int* counters; };

// A fragment for k-step u out of the raw weight registers.  bf16: one 16-byte load per
// k-step.  FP8 (OCP e4m3, weight-only): one 16-byte load carries TWO k-steps (8 bytes each);
// the bytes are widened to bf16 in registers (exact: e4m3 fits in bf16) with the packed
// gfx950 conversion, and the per-output-row scale is applied once in the epilogue.
template <bool W8>
__device__ inline frag_ab a_frag(const raw16* raw, int u) {
  if constexpr (!W8) {
    return __builtin_bit_cast(frag_ab, raw[u]);
  } else {
    const raw16 v = raw[u >> 1];
    const unsigned int lo = (u & 1) ? v[2] : v[0], hi = (u & 1) ? v[3] : v[1];
    const bf16x2 p0 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(lo, 1.0f, false);
    const bf16x2 p1 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(lo, 1.0f, true);
    const bf16x2 p2 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(hi, 1.0f, false);
    const bf16x2 p3 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(hi, 1.0f, true);
    frag_ab a;
    a[0] = p0[0]; a[1] = p0[1]; a[2] = p1[0]; a[3] = p1[1]; a[4] = p2[0]; a[5] = p2[1]; a[6] = p3[0]; a[7] = p3[1];
    return a;
  }
}

// MXFP4 weights (OCP microscaling: e2m1 values, one E8M0 scale per 32 elements along K, i.e.
// per (row, k-step)): a lane's 8 values of a k-step are 4 bytes, so one 16-byte load carries
// FOUR k-steps, and `sc` holds their four scale bytes (byte u = k-step u).  Each pair of
// nibbles is widened to bf16 with the scale applied by the packed gfx950 conversion (exact:
// e2m1 times a power of two fits in bf16).
__device__ inline frag_ab a_frag4(const raw16 raw, unsigned int sc, int u) {
  const unsigned int w = raw[u];
  const float scale = __uint_as_float(((sc >> (8 * u)) & 0xffu) << 23);  // 2^(e - 127)
  const bf16x2 p0 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, scale, 0);
  const bf16x2 p1 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, scale, 1);
  const bf16x2 p2 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, scale, 2);
  const bf16x2 p3 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp4(w, scale, 3);
  frag_ab a;
  a[0] = p0[0]; a[1] = p0[1]; a[2] = p1[0]; a[3] = p1[1]; a[4] = p2[0]; a[5] = p2[1]; a[6] = p3[0]; a[7] = p3[1];
  return a;
}

// One thread per (16-feature tile, 128-column group, lane): its 16 bytes are four k-steps of 8
// values; each becomes the lane's 16-byte bf16 fragment of that k-step (1 KiB per wave and store).
__global__ __launch_bounds__(256) void unpack_mxfp4_kernel(const raw16* __restrict__ wp4, const unsigned int* __restrict__ scales, raw16* __restrict__ wp,
                                                           long long total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int lane = (int)(i & 63);
  const long long tg = i >> 6;  // tile * groups + group
  const raw16 raw = wp4[i];
  const unsigned int sc = scales[tg * 16 + (lane & 15)];
  raw16* dst = wp + (tg * 4) * 64 + lane;  // k-step 4 * group + u of the same tile: 64 fragments further per u
#pragma unroll
  for (int u = 0; u < 4; ++u) dst[u * 64] = __builtin_bit_cast(raw16, a_frag4(raw, sc, u));
}

// Shared epilogue.  Lane holds y[token j*16 + r][feature tile*16 + 4q + e], e = 0..3, in acc[j][e].
// 1/rms of the token rows a workgroup's epilogue scales, left in LDS: row_ss holds `ss_parts` partial
// sums of squares per row — 8 from the residual kernel (512-column segments), hidden/16 from the
// row-parallel projections that update the residual themselves (one per feature tile) — so a row
// is summed by a whole wave (lane-strided loads, butterfly), not by one lane in a serial loop.
template <int ROWS, int NW>
__device__ inline void row_scales_to_lds(float* rs_lds, const float* __restrict__ row_ss, int ss_parts, float inv_k, float eps, int wave, int lane) {
  for (int row = wave; row < ROWS; row += NW) {
    const float* p = row_ss + (long long)row * ss_parts;
    float ssum = 0.f;
    for (int i = lane; i < ss_parts; i += 64) ssum += p[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ssum += __shfl_xor(ssum, o, 64);
    if (lane == 0) rs_lds[row] = rsqrtf(ssum * inv_k + eps);
  }
}

template <int MT, int EPI, bool W8>
__device__ inline void gemm_epilogue(frag_cd (&acc)[MT], unsigned char* lds_raw, void* __restrict__ out, long long tile, int wave, int lane, long long ldy,
                                     long long slab_stride, const float* __restrict__ row_ss, int ss_parts, float inv_k, float eps,
                                     const float* __restrict__ w_scale, const float* rs_lds = nullptr) {
  const int r = lane & 15, q = lane >> 4;
  const long long n_base = tile * 16 + q * 4;
  if constexpr (W8) {
    // dequantise: one scale per output feature (weight row)
    const frag_cd ws = *reinterpret_cast<const frag_cd*>(w_scale + n_base);
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[j][e] *= ws[e];
  }
  // Folded RMSNorm: x is the un-normalised residual stream and W carries the norm gain,
  // so rows are scaled by 1/rms(x[m]) here (slabs are scaled by their consumer).
  if (EPI != EPI_SLAB && EPI != EPI_RESID && row_ss != nullptr) {
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      float rs;
      if (rs_lds != nullptr) {
        rs = rs_lds[j * 16 + r];
      } else {
        float ssum = 0.f;
        for (int p = 0; p < ss_parts; ++p) ssum += row_ss[(long long)(j * 16 + r) * ss_parts + p];
        rs = rsqrtf(ssum * inv_k + eps);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[j][e] *= rs;
    }
  }
  if (EPI == EPI_SLAB) {
    float* o = reinterpret_cast<float*>(out) + (long long)blockIdx.y * slab_stride;
#pragma unroll
    for (int j = 0; j < MT; ++j) *reinterpret_cast<frag_cd*>(o + (long long)(j * 16 + r) * ldy + n_base) = acc[j];
  } else if (EPI == EPI_RESID) {
    // These slabs are read back by ANOTHER workgroup of this launch, possibly on another XCD (own
    // L2): device-scope stores write through, so no cache write-back is needed before the arrival
    // is counted (a release fence would walk the whole L2 once per workgroup: measured +60 us per call).
    float* o = reinterpret_cast<float*>(out) + (long long)blockIdx.y * slab_stride;
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) __hip_atomic_store(o + (long long)(j * 16 + r) * ldy + n_base + e, acc[j][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
    // SwiGLU: a 16-row weight tile holds 8 gate rows and the 8 up rows of the same features, so
    // lane (q, r) with q < 2 has gate[4q + e] and lane (q + 2, r) the matching up value: one
    // cross-lane read per register, no LDS round trip, no coupling between waves.
    unsigned short* o = reinterpret_cast<unsigned short*>(out);
    const long long i_base = tile * 8 + q * 4;  // intermediate feature index
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      us4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float g = acc[j][e], up = __shfl_xor(acc[j][e], 32, 64);
        v[e] = f2bf(g / (1.f + __expf(-g)) * up);
      }
      if (q < 2) *reinterpret_cast<us4*>(o + (long long)(j * 16 + r) * ldy + i_base) = v;
    }
  }
}

// KS = k-steps (of 32) per chunk.  grid = (N/64, splits).  chunks_per_split chunks per workgroup.
template <int MT, int KS, int EPI, bool W8>
__global__ __launch_bounds__(256) void stream_gemm_kernel(const void* __restrict__ Wp, const __bf16* __restrict__ x, void* __restrict__ out, int N, int K,
                                                          long long ldx, long long ldy, int chunks_per_split, long long slab_stride,
                                                          const float* __restrict__ row_ss, int ss_parts, float inv_k, float eps,
                                                          const float* __restrict__ w_scale) {
  constexpr int AL = W8 ? KS / 2 : KS;  // 16-byte weight loads per chunk per lane
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
  // 16-byte units: bf16 [tile][kstep][lane], fp8 [tile][kstep pair][lane]
  const raw16* wp = reinterpret_cast<const raw16*>(Wp) + tile * (W8 ? ksteps / 2 : ksteps) * 64 + lane;

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

  raw16 a_cur[AL], a_nxt[AL];
  frag_ab xr[XL];

  // ---- prologue: chunk c0
  if (c0 < c1) {
#pragma unroll
    for (int i = 0; i < XL; ++i)
      if (xok[i]) xr[i] = *reinterpret_cast<const frag_ab*>(xsrc[i] + (long long)c0 * KS * 32);
#pragma unroll
    for (int u = 0; u < AL; ++u) a_cur[u] = wp[((long long)c0 * AL + u) * 64];
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
      for (int u = 0; u < AL; ++u) a_nxt[u] = wp[((long long)(c + 1) * AL + u) * 64];
    }
    __syncthreads();  // x chunk c is in LDS buffer `buf`; everyone is done reading buffer `buf^1`
#pragma unroll
    for (int u = 0; u < KS; ++u) {
      frag_ab b[MT];
#pragma unroll
      for (int j = 0; j < MT; ++j) b[j] = xs[(buf * XF + u * MT + j) * 64 + lane];
      const frag_ab a = a_frag<W8>(a_cur, u);
#pragma unroll
      for (int j = 0; j < MT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[j], acc[j], 0, 0, 0);
    }
    if (more) {
#pragma unroll
      for (int i = 0; i < XL; ++i)
        if (xok[i]) xs[((buf ^ 1) * XF + wave + 4 * i) * 64 + lane] = xr[i];
#pragma unroll
      for (int u = 0; u < AL; ++u) a_cur[u] = a_nxt[u];
    }
  }

  gemm_epilogue<MT, EPI, W8>(acc, lds_raw, out, tile, wave, lane, ldy, slab_stride, row_ss, ss_parts, inv_k, eps, w_scale);
}

// ---------------------------------------------------------------------------------------
// v3: deep register pipeline.  The v2 loop keeps ONE chunk (KS KiB per wave) in flight and
// has to drain it every iteration, so a wave moves KS KiB per memory round trip and a CU
// with ~7 resident waves holds < 32 KiB of HBM loads in flight — latency-bound at ~5 TB/s.
// Here every wave keeps PD = NB-1 chunks in flight in a ring of NB register buffers (weights
// AND its share of the x chunk, issued together so the in-order load counter never makes a
// cache-hit x load wait behind younger HBM loads), i.e. 3 x 4 KiB per wave at NB = 4.
// The chunk loop is unrolled by NB so ring slots are compile-time, and the K range of a
// workgroup is a whole number of NB-chunk groups: the steady-state body is branch-free
// (exact s_waitcnt counts) and the last group is a peeled copy that only drains.
// NT: weight loads carry the non-temporal hint (each byte is read once by one CU).
template <bool NT>
__device__ inline raw16 ldw16(const raw16* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}

// WB = weight bits: 16 (bf16), 8 (e4m3 + one fp32 scale per row, applied in the epilogue) or
// 4 (MXFP4: w_scale then points at the packed E8M0 block scales, uint32 [N/16][K/128][16 rows]).
// FT = 16-feature tiles per wave (consecutive tiles).  Every x fragment a wave reads from LDS
// feeds FT MFMAs instead of one: with many token rows (MT = 8, 16) or few weight bytes per MFMA
// (FP8, MXFP4) the x fragment reads, not HBM, bound the kernel at FT = 1 — a wave reads
// MT KiB of x per k-step whatever the weights weigh.
template <int MT, int KS, int NB, int EPI, int WB, bool NT, int NW, int FT = 1>
__global__ __launch_bounds__(NW * 64) void stream_gemm3_kernel(const void* __restrict__ Wp, const __bf16* __restrict__ x, void* __restrict__ out, int N, int K,
                                                           long long ldx, long long ldy, int groups_per_split, long long slab_stride,
                                                           const float* __restrict__ row_ss, int ss_parts, float inv_k, float eps,
                                                           const float* __restrict__ w_scale, ResidArgs ra) {
  constexpr bool W8 = WB == 8, W4 = WB == 4;
  static_assert(!W4 || KS % 4 == 0, "an MXFP4 load carries four k-steps");
  static_assert(FT == 1 || EPI != EPI_RESID, "the fused residual fix-up is built for one tile per wave");
  constexpr int AL = W4 ? KS / 4 : W8 ? KS / 2 : KS;
  constexpr int XF = KS * MT;
  constexpr int XL = (XF + NW - 1) / NW;  // x fragments of a chunk per wave
  constexpr int PD = NB - 1;
  // XCO (48 .. 128 token rows): the x chunk is fetched in whole 256-byte row segments — a wave-level load covers four
  // consecutive rows x 128 columns, two full cache lines per row — instead of one 64-byte piece of 16 different rows
  // per 16 lanes (the B-fragment order), and the pieces are scattered into fragment order on the way into LDS.
  // Measured on MI355X (gate/up projection, weights rotated): the time of this kernel follows the bytes that pass
  // the CUs' vector-memory path, x included (32 / 64 / 128 rows: 38.5 / 45.0 / 69.5 us = 7.2 - 8.2 TB/s of weights +
  // x whatever the split between them), and x in 64-byte pieces costs far more than its bytes: with wave-linear x
  // loads 64 rows take 39.7 us and 128 rows 57.0.  At 32 rows x does not show (39.1 either way): the old order stays.
  // LDS image: 16-byte slot of (k-step ks, row tile j, lane quarter q, row r) = ks * KSS + j * 64 + q * 16 + (r ^ q),
  // KSS = 64 MT + 4.  Reads (ds_read_b128: four groups of 16 lanes, e.g. q = 0 with r in {0-3, 12-15} and q = 1 with
  // r in {4-11}; banks by slot modulo 16): r ^ q only permutes rows inside aligned groups of four, so every group
  // still covers 16 different slots.  Writes (ds_write_b128: groups of 8 contiguous lanes, slot modulo 8): the 8
  // lanes are the pieces (ks, q) of ONE row for two values of ks — r ^ q gives four different slots inside one
  // half of the bank row, and the + 4 per k-step puts the second k-step into the other half.
  constexpr bool XCO = MT >= (WB == 16 ? 3 : 2) && MT <= 8 && KS == 4;  // (32 rows: only where the weight bytes are few — FP8 2.49 -> 2.42 ms per decode step, MXFP4 2.22 -> 2.19; bf16 3.33 either way)
  constexpr int FSZ = 64, KSS = MT * FSZ + 4;                 // slots per fragment / per k-step
  constexpr int XBUF = XCO ? KS * KSS : XF * 64;              // slots per buffer
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[2 * XBUF * 16];
  frag_ab* xs = reinterpret_cast<frag_ab*>(lds_raw);  // [2][XF][64] (XCO: [2][KS][KSS])
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 15, q = lane >> 4;
  const int ksteps = K >> 5;
  const int ngroups = ksteps / (KS * NB);
  const int g0 = blockIdx.y * groups_per_split;
  const int g1 = g0 + groups_per_split < ngroups ? g0 + groups_per_split : ngroups;
  // The last workgroup may be ragged (N / 16 not a multiple of NW: any workgroup size then serves any width, and the
  // planner picks the one that shares the tiles out most evenly over the CUs): a wave past the last tile streams the
  // last tile again — it takes part in the x staging and the barriers like the others — and stores nothing.
  const long long ntiles = N >> 4;
  const long long tile_raw = ((long long)blockIdx.x * NW + wave) * FT;
  const bool tile_ok = tile_raw + FT <= ntiles;
  const long long tile = tile_ok ? tile_raw : ntiles - FT;  // the wave's first tile
  const long long tile_units = (long long)(ksteps * AL / KS) * 64;    // 16-byte units of one tile's weights
  const raw16* wp = reinterpret_cast<const raw16*>(Wp) + tile * tile_units + lane;
  const long long sc_units = (long long)(ksteps / 4) * 16;            // scale words of one tile (W4)
  const unsigned int* scp = reinterpret_cast<const unsigned int*>(w_scale) + (W4 ? tile * sc_units + r : 0);  // W4: one word per chunk
  unsigned int sc[NB][FT][W4 ? AL : 1];

  const __bf16* xsrc[XL];
  int xdst[XL];  // slot of the load's 16 bytes inside a buffer
#pragma unroll
  for (int i = 0; i < XL; ++i) {
    const int f = wave + NW * i;
    if constexpr (XCO) {
      const int g = (f < XF ? f : 0) * 64 + lane;  // piece g of the chunk: row g / 16, 16-byte piece g % 16 of its 128 columns
      const int row = g >> 4, piece = g & 15;
      xsrc[i] = x + (long long)row * ldx + piece * 8;
      xdst[i] = (piece >> 2) * KSS + (row >> 4) * FSZ + (piece & 3) * 16 + ((row & 15) ^ (piece & 3));
    } else {
      const int ks = (f < XF ? f : 0) / MT, j = (f < XF ? f : 0) % MT;
      xsrc[i] = x + (long long)(j * 16 + r) * ldx + ks * 32 + q * 8;
      xdst[i] = (f < XF ? f : 0) * 64 + lane;
    }
  }
  const int xrd = XCO ? q * 16 + (r ^ q) : lane;  // a lane's slot inside a fragment

  frag_cd acc[FT][MT];
#pragma unroll
  for (int f = 0; f < FT; ++f)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[f][j] = (frag_cd){0.f, 0.f, 0.f, 0.f};

  raw16 a[NB][FT][AL];
  frag_ab xr[NB][XL];

#define KV_ISSUE(slot, chunk)                                                                                   \
  do {                                                                                                          \
    _Pragma("unroll") for (int i = 0; i < XL; ++i) if (wave + NW * i < XF)                                      \
        xr[slot][i] = *reinterpret_cast<const frag_ab*>(xsrc[i] + (long long)(chunk) * KS * 32);                \
    _Pragma("unroll") for (int f = 0; f < FT; ++f) {                                                            \
      _Pragma("unroll") for (int u = 0; u < AL; ++u) a[slot][f][u] = ldw16<NT>(wp + f * tile_units + ((long long)(chunk) * AL + u) * 64); \
      if constexpr (W4) {                                                                                       \
        _Pragma("unroll") for (int u = 0; u < AL; ++u) sc[slot][f][u] = scp[f * sc_units + ((long long)(chunk) * AL + u) * 16]; \
      }                                                                                                         \
    }                                                                                                           \
  } while (0)

#define KV_CONSUME(slot, chunk)                                                                                 \
  do {                                                                                                          \
    const int buf = (chunk) & 1;                                                                                \
    _Pragma("unroll") for (int i = 0; i < XL; ++i) if (wave + NW * i < XF)                                      \
        xs[buf * XBUF + xdst[i]] = xr[slot][i];                                                                 \
    __syncthreads();                                                                                            \
    _Pragma("unroll") for (int u = 0; u < KS; ++u) {                                                            \
      frag_ab b[MT];                                                                                            \
      _Pragma("unroll") for (int j = 0; j < MT; ++j) b[j] = xs[buf * XBUF + (XCO ? u * KSS + j * FSZ : (u * MT + j) * 64) + xrd]; \
      _Pragma("unroll") for (int f = 0; f < FT; ++f) {                                                          \
        frag_ab af;                                                                                             \
        if constexpr (W4) af = a_frag4(a[slot][f][u >> 2], sc[slot][f][u >> 2], u & 3);                         \
        else af = a_frag<W8>(a[slot][f], u);                                                                    \
        _Pragma("unroll") for (int j = 0; j < MT; ++j) acc[f][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, b[j], acc[f][j], 0, 0, 0); \
      }                                                                                                         \
    }                                                                                                           \
  } while (0)

  // prologue: the first PD chunks of the first group
  const int cbase = g0 * NB;
#pragma unroll
  for (int s = 0; s < PD; ++s) KV_ISSUE(s, cbase + s);

  // 1/rms of the token rows, for the epilogue: summed while the first weight chunks are on their
  // way (the loads queue behind them; the barriers of the chunk loop publish the LDS writes)
  __shared__ float rs_lds[16 * MT];
  const bool scale_rows = EPI != EPI_SLAB && EPI != EPI_RESID && row_ss != nullptr;
  if (scale_rows) row_scales_to_lds<16 * MT, NW>(rs_lds, row_ss, ss_parts, inv_k, eps, wave, lane);

  // steady state: every sub-step issues chunk c+PD and consumes chunk c
  for (int g = g0; g + 1 < g1; ++g) {
    const int c = g * NB;
#pragma unroll
    for (int s = 0; s < NB; ++s) {
      KV_ISSUE((s + PD) % NB, c + s + PD);
      KV_CONSUME(s, c + s);
    }
  }
  // last group: only its own final chunk is still to be issued
  {
    const int c = (g1 - 1) * NB;
    KV_ISSUE(PD, c + PD);
#pragma unroll
    for (int s = 0; s < NB; ++s) KV_CONSUME(s, c + s);
  }
#undef KV_ISSUE
#undef KV_CONSUME

  if (tile_ok) {
#pragma unroll
    for (int f = 0; f < FT; ++f) gemm_epilogue<MT, EPI, W8>(acc[f], lds_raw, out, tile + f, wave, lane, ldy, slab_stride, row_ss, ss_parts, inv_k, eps, w_scale, scale_rows ? rs_lds : nullptr);
  }

  if constexpr (EPI == EPI_RESID) {
    // ---- fix-up by the last split of this feature group (16 * NW features x 16 * MT rows)
    constexpr int GC = 16 * NW, ROWS = 16 * MT, NTH = 64 * NW;
    __shared__ int is_last;
    float* ss_lds = reinterpret_cast<float*>(lds_raw);  // the x staging area is free once every wave is past its last chunk
    // write-through stores are asynchronous: wait until this wave's have reached memory (vmcnt(0)) —
    // a workgroup-scope release alone does not — then the barrier, then the arrival is counted
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // also a compiler barrier: no store may sink below it
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (tid == 0) {
      const int arrived = __hip_atomic_fetch_add(ra.counters + blockIdx.x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      is_last = arrived == (int)gridDim.y - 1;
      if (is_last) __hip_atomic_store(ra.counters + blockIdx.x, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
    }
    if (tid < ROWS) ss_lds[tid] = 0.f;
    __syncthreads();
    if (!is_last) return;
    const float* slab0 = reinterpret_cast<const float*>(out);  // read with device-scope loads: past this XCD's L2
    const long long col0 = (long long)blockIdx.x * GC;
    const int n_splits = (int)gridDim.y;
    for (int idx = tid; idx < ROWS * (GC / 4); idx += NTH) {
      const int t = idx / (GC / 4), c4 = (idx % (GC / 4)) * 4;
      const long long off = (long long)t * ldy + col0 + c4;
      frag_cd a = (frag_cd){0.f, 0.f, 0.f, 0.f};
      for (int sidx = 0; sidx < n_splits; ++sidx) {  // slab order: the sum does not depend on who arrived last
        const float* sp = slab0 + (long long)sidx * slab_stride + off;
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] += __hip_atomic_load(sp + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      us4 c;
      float ssq = 0.f;
      us4 b = (us4){0, 0, 0, 0};
      if (ra.has_residual) b = *reinterpret_cast<const us4*>(ra.residual + off);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = __uint_as_float(((unsigned int)f2bf(a[e])) << 16) + __uint_as_float(((unsigned int)b[e]) << 16);
        c[e] = f2bf(v);
        const float f = __uint_as_float(((unsigned int)c[e]) << 16);
        ssq += f * f;
      }
      *reinterpret_cast<us4*>(ra.residual + off) = c;
      atomicAdd(&ss_lds[t], ssq);
    }
    __syncthreads();
    if (tid < ROWS) ra.ss_out[(long long)tid * gridDim.x + blockIdx.x] = ss_lds[tid];
  }
}

// ---------------------------------------------------------------------------------------
// Row-parallel projection + residual update in ONE launch, without slabs (output and down
// projections of a decode step):
//     residual' = bf16(residual + bf16(x @ W^T)),   ss[t][tile] = sum over the tile's 16 features of residual'[t]^2
// One workgroup owns ONE 16-feature tile over the whole of K; its NW waves split K among
// themselves (wave w: columns [w K/NW, (w+1) K/NW)), meet in LDS and the first MT waves add the
// residual.  With 256 tiles (hidden 4096) that is one workgroup per CU and no cross-workgroup
// dependency at all: no fp32 slabs written and read back, no arrival counter, and the separate
// residual kernel (a launch boundary plus a chain of dependent loads, ~5 us per call) disappears.
// The price is x traffic: every wave fetches its own B fragments of x straight from L2 into
// registers (no sharing through LDS: the waves of a workgroup read different columns), i.e. the
// whole of x once per CU.  x is a few hundred KB that every CU reads — it lives in each XCD's L2.
// The waves' partial sums are added in wave order, which is the slab order of the split-K path
// with NW equal splits: the new residual is bit-identical to stream_gemm3 + residual_reduce.
// The main loop has no barrier: a ring of NB register slots per wave (weights and x of a chunk of
// 4 k-steps issued together), steady state unrolled by NB with exact s_waitcnt counts, then a
// guarded drain (chunks per wave need not be a multiple of NB).
template <int MT, int NB, bool W8, int NW>
__global__ __launch_bounds__(NW * 64) void stream_gemm_rowsum_kernel(const void* __restrict__ Wp, const __bf16* __restrict__ x, int N, int K, long long ldx, int cpw,
                                                                     unsigned short* __restrict__ residual, int has_residual, float* __restrict__ ss_out,
                                                                     const float* __restrict__ w_scale) {
  constexpr int KS = 4, AL = W8 ? KS / 2 : KS, PD = NB - 1;
  __shared__ __attribute__((aligned(16))) float red[NW][MT][64][4];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int r = lane & 15, q = lane >> 4;
  const int ksteps = K >> 5;
  const long long tile = blockIdx.x;
  const raw16* wp = reinterpret_cast<const raw16*>(Wp) + tile * (long long)(ksteps * AL / KS) * 64 + (long long)wave * cpw * AL * 64 + lane;
  const __bf16* xp = x + (long long)r * ldx + (long long)wave * cpw * (KS * 32) + q * 8;

  frag_cd acc[MT];
#pragma unroll
  for (int j = 0; j < MT; ++j) acc[j] = (frag_cd){0.f, 0.f, 0.f, 0.f};
  raw16 a[NB][AL];
  frag_ab xr[NB][KS][MT];

#define KV_ISSUE(slot, chunk)                                                                                              \
  do {                                                                                                                     \
    _Pragma("unroll") for (int u = 0; u < KS; ++u) {                                                                       \
      _Pragma("unroll") for (int j = 0; j < MT; ++j)                                                                       \
          xr[slot][u][j] = *reinterpret_cast<const frag_ab*>(xp + (long long)j * 16 * ldx + ((long long)(chunk) * KS + u) * 32); \
    }                                                                                                                      \
    _Pragma("unroll") for (int u = 0; u < AL; ++u) a[slot][u] = ldw16<true>(wp + ((long long)(chunk) * AL + u) * 64);      \
  } while (0)

#define KV_CONSUME(slot)                                                                                                   \
  do {                                                                                                                     \
    _Pragma("unroll") for (int u = 0; u < KS; ++u) {                                                                       \
      const frag_ab af = a_frag<W8>(a[slot], u);                                                                           \
      _Pragma("unroll") for (int j = 0; j < MT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, xr[slot][u][j], acc[j], 0, 0, 0); \
    }                                                                                                                      \
  } while (0)

#pragma unroll
  for (int s = 0; s < PD; ++s)
    if (s < cpw) KV_ISSUE(s, s);
  int c = 0;
  for (; c + NB + PD <= cpw; c += NB) {  // every issue of this group is inside the wave's range: branch-free
#pragma unroll
    for (int s = 0; s < NB; ++s) {
      KV_ISSUE((s + PD) % NB, c + s + PD);
      KV_CONSUME(s);
    }
  }
  // drain: at most NB + PD - 1 chunks are left (c is a multiple of NB, so slot = step % NB)
#pragma unroll
  for (int i = 0; i < NB + PD - 1; ++i) {
    if (c + i < cpw) {
      if (c + i + PD < cpw) KV_ISSUE((i + PD) % NB, c + i + PD);
      KV_CONSUME(i % NB);
    }
  }
#undef KV_ISSUE
#undef KV_CONSUME

  if constexpr (W8) {  // one scale per output feature, applied per K slice as the split-K path does per slab
    const frag_cd ws = *reinterpret_cast<const frag_cd*>(w_scale + tile * 16 + q * 4);
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[j][e] *= ws[e];
  }
#pragma unroll
  for (int j = 0; j < MT; ++j) *reinterpret_cast<frag_cd*>(&red[wave][j][lane][0]) = acc[j];
  __syncthreads();
  if (wave < MT) {
    // wave j finishes token rows 16j .. 16j+15: lane (q, r) holds features 4q .. 4q+3 of row 16j + r
    const int j = wave;
    frag_cd s = (frag_cd){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < NW; ++w) s += *reinterpret_cast<const frag_cd*>(&red[w][j][lane][0]);
    const long long off = (long long)(j * 16 + r) * N + tile * 16 + q * 4;
    us4 b = (us4){0, 0, 0, 0};
    if (has_residual) b = *reinterpret_cast<const us4*>(residual + off);
    us4 cv;
    float ssq = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float v = __uint_as_float(((unsigned int)f2bf(s[e])) << 16) + __uint_as_float(((unsigned int)b[e]) << 16);
      cv[e] = f2bf(v);
      const float f = __uint_as_float(((unsigned int)cv[e]) << 16);
      ssq += f * f;
    }
    *reinterpret_cast<us4*>(residual + off) = cv;
    ssq += __shfl_xor(ssq, 16, 64);
    ssq += __shfl_xor(ssq, 32, 64);
    if (q == 0) ss_out[(long long)(j * 16 + r) * (N >> 4) + tile] = ssq;
  }
}

template <int MT, bool W8>
int launch_rowsum(const void* Wp, const void* x, int N, int K, long long ldx, int nb, void* residual, int has_residual, float* ss_out, const float* w_scale,
                  hipStream_t stream) {
  constexpr int NW = 8;
  const int cpw = K / (128 * NW);
  dim3 grid(N / 16), block(NW * 64);
#define KV_RS(NBv)                                                                                                                                         \
  hipLaunchKernelGGL((stream_gemm_rowsum_kernel<MT, NBv, W8, NW>), grid, block, 0, stream, Wp, (const __bf16*)x, N, K, ldx, cpw, (unsigned short*)residual, \
                     has_residual, ss_out, w_scale)
  // ring depth: 16 MT + 16 registers per slot (x fragments + weights), so deep rings only for few rows
  if (nb == 2) KV_RS(2);
  else if (nb == 3 && MT <= 2) { if constexpr (MT <= 2) KV_RS(3); }
  else if (nb == 4 && MT <= 2) { if constexpr (MT <= 2) KV_RS(4); }
  else return -1;
#undef KV_RS
  return (int)hipGetLastError();
}

struct Args {
  const void* Wp; const void* x; void* out; int N, K; long long ldx, ldy; int splits, cps; long long slab_stride;
  const float* row_ss; int ss_parts; float eps; const float* w_scale; hipStream_t stream;
  ResidArgs ra{nullptr, 0, nullptr, nullptr};
};

template <int MT, int KS, int EPI, bool W8>
int launch(const Args& a) {
  dim3 grid(a.N / 64, a.splits), block(256);
  hipLaunchKernelGGL((stream_gemm_kernel<MT, KS, EPI, W8>), grid, block, 0, a.stream, a.Wp, (const __bf16*)a.x, a.out, a.N, a.K, a.ldx, a.ldy, a.cps, a.slab_stride,
                     a.row_ss, a.ss_parts, 1.f / (float)a.K, a.eps, a.w_scale);
  return (int)hipGetLastError();
}

template <int MT, int KS, bool W8>
int dispatch_epi(int epi, const Args& a) {
  switch (epi) {
    case EPI_BF16: return launch<MT, KS, EPI_BF16, W8>(a);
    case EPI_SLAB: return launch<MT, KS, EPI_SLAB, W8>(a);
    case EPI_SWIGLU: return launch<MT, KS, EPI_SWIGLU, W8>(a);
  }
  return -1;
}

template <int MT>
int dispatch_ks(int epi, int ks, bool w8, const Args& a) {
  if (w8) return ks == 4 ? dispatch_epi<MT, 4, true>(epi, a) : dispatch_epi<MT, 8, true>(epi, a);
  return ks == 4 ? dispatch_epi<MT, 4, false>(epi, a) : dispatch_epi<MT, 8, false>(epi, a);
}

// k-steps per chunk.  Measured with MXFP4 weights on the Llama-3-8B gate/up projection (32 rows):
// 16-k-step chunks (one barrier per 4 KB of weights per wave) 24.0 us, 4-k-step chunks in a ring of
// 8 21.8 us — the barrier is not what holds 4-bit weights back; the x fragments are (every wave
// reads 2 KB of them from LDS per 256 B of weights).  So every format uses 4-k-step chunks.
constexpr int chunk_ksteps(int mt, int wb) { return 4; }

template <int MT, int NB, int EPI, int WB, bool NT, int NW, int FT = 1>
int launch3(const Args& a) {
  dim3 grid((a.N + 16 * NW * FT - 1) / (16 * NW * FT), a.splits), block(NW * 64);
  hipLaunchKernelGGL((stream_gemm3_kernel<MT, chunk_ksteps(MT, WB), NB, EPI, WB, NT, NW, FT>), grid, block, 0, a.stream, a.Wp, (const __bf16*)a.x, a.out, a.N, a.K, a.ldx, a.ldy, a.cps,
                     a.slab_stride, a.row_ss, a.ss_parts, 1.f / (float)a.K, a.eps, a.w_scale, a.ra);
  return (int)hipGetLastError();
}

template <int MT, int NB, int WB, bool NT, int NW, int FT = 1>
int dispatch3_epi(int epi, const Args& a) {
  switch (epi) {
    case EPI_BF16: return launch3<MT, NB, EPI_BF16, WB, NT, NW, FT>(a);
    case EPI_SLAB: return launch3<MT, NB, EPI_SLAB, WB, NT, NW, FT>(a);
    case EPI_SWIGLU: return launch3<MT, NB, EPI_SWIGLU, WB, NT, NW, FT>(a);
    case EPI_RESID:
      if constexpr (MT <= 4 && FT == 1) return launch3<MT, NB, EPI_RESID, WB, NT, NW>(a);
      return -1;
  }
  return -1;
}

// Two feature tiles per wave (FT = 2), built where the x fragment reads bound the kernel: MXFP4
// weights at up to 64 rows (every ring depth), and 128 / 256-row steps of every weight format
// (2-chunk rings); workgroups of 4 or 8 waves (128 or 256 features).
template <int MT>
int dispatch3_ft2_w4(int epi, int nb, int nw, const Args& a) {
#define KV_FT2(NBv, NWv) \
  if (nb == NBv && nw == NWv) return dispatch3_epi<MT, NBv, 4, true, NWv, 2>(epi, a)
  KV_FT2(2, 4); KV_FT2(2, 8); KV_FT2(4, 4); KV_FT2(4, 8);
  if constexpr (MT <= 2) { KV_FT2(8, 4); KV_FT2(8, 8); }
#undef KV_FT2
  return -1;
}

template <int MT, int WB>
int dispatch3_ft2_wide(int epi, int nw, const Args& a) {
  if (nw == 4) return dispatch3_epi<MT, 2, WB, true, 4, 2>(epi, a);
  if constexpr (MT <= 8) {  // 16 row tiles x 2 feature tiles need 512 registers per lane: 4-wave workgroups only
    if (nw == 8) return dispatch3_epi<MT, 2, WB, true, 8, 2>(epi, a);
  }
  return -1;
}

template <int MT, int WB, int NW>
int dispatch3_nb(int epi, int nb, const Args& a) {
  // weight loads are always non-temporal here: measured 8-12 % faster than the default
  // policy on every decode shape (each weight byte is read once, by one CU)
  switch (nb) {
    case 2: return dispatch3_epi<MT, 2, WB, true, NW>(epi, a);
    case 4: return dispatch3_epi<MT, 4, WB, true, NW>(epi, a);
    case 8:
      // quantised weights only: a chunk of e4m3 / MXFP4 weights is 2 KB / 1 KB per wave, so the
      // ring has to be deeper for the same bytes in flight
      if constexpr (WB != 16 && MT <= 2) return dispatch3_epi<MT, 8, WB, true, NW>(epi, a);
      return -1;
  }
  return -1;
}

// nw = waves (16-feature tiles) per workgroup.  4 is the default; 7 makes 1792 tiles (the
// Llama-3-8B gate/up projection) exactly one workgroup per CU of a 256-CU part, 6 does the
// same for 384 tiles x 4 K-splits (its QKV projection); 8 halves the x traffic per weight
// byte for wide token batches.
template <int MT, int WB>
int dispatch3_nw(int epi, int nb, int nw, const Args& a) {
  switch (nw) {
    case 4: return dispatch3_nb<MT, WB, 4>(epi, nb, a);
    case 6: return dispatch3_nb<MT, WB, 6>(epi, nb, a);
    case 7: return dispatch3_nb<MT, WB, 7>(epi, nb, a);
    case 8: return dispatch3_nb<MT, WB, 8>(epi, nb, a);
  }
  return -1;
}

// 128 / 256-row steps (prompt chunks): every weight fragment feeds 8 / 16 MFMAs, the x chunk in
// LDS is 32 / 64 KiB per buffer, one workgroup per CU.
template <int MT, int WB>
int dispatch3_wide(int epi, int nw, const Args& a) {
  switch (nw) {
    case 4: return dispatch3_epi<MT, 2, WB, true, 4>(epi, a);
    case 6: return dispatch3_epi<MT, 2, WB, true, 6>(epi, a);
    case 7: return dispatch3_epi<MT, 2, WB, true, 7>(epi, a);
    case 8: return dispatch3_epi<MT, 2, WB, true, 8>(epi, a);
  }
  return -1;
}

template <int MT>
int dispatch3_wb(int epi, int nb, int nw, int wb, const Args& a) {
  return wb == 4 ? dispatch3_nw<MT, 4>(epi, nb, nw, a) : wb == 8 ? dispatch3_nw<MT, 8>(epi, nb, nw, a) : dispatch3_nw<MT, 16>(epi, nb, nw, a);
}

template <int MT>
int dispatch3_wide_wb(int epi, int nw, int wb, const Args& a) {
  return wb == 4 ? dispatch3_wide<MT, 4>(epi, nw, a) : wb == 8 ? dispatch3_wide<MT, 8>(epi, nw, a) : dispatch3_wide<MT, 16>(epi, nw, a);
}

// W[N, K] fp8 bytes (row-major) -> [N/16][K/64][64 lanes][16 bytes]: lane l of a tile holds
// W[row l&15][k = 32*(2p+h) + 8*(l>>4) .. +8) for h = 0, 1 in its two 8-byte halves.
__global__ __launch_bounds__(256) void pack_weight_fp8_kernel(const unsigned char* __restrict__ W, unsigned char* __restrict__ Wp, int N, int K) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;  // one 16-byte unit per thread
  const int kpairs = K >> 6;
  const long long total = (long long)(N >> 4) * kpairs * 64;
  if (idx >= total) return;
  const int lane = (int)(idx & 63);
  const long long t = idx >> 6;
  const int p = (int)(t % kpairs);
  const long long nt = t / kpairs;
  const int r = lane & 15, q = lane >> 4;
  const unsigned char* row = W + (nt * 16 + r) * (long long)K + (long long)p * 64 + q * 8;
  const unsigned long long lo = *reinterpret_cast<const unsigned long long*>(row);
  const unsigned long long hi = *reinterpret_cast<const unsigned long long*>(row + 32);
  unsigned long long* dst = reinterpret_cast<unsigned long long*>(Wp + idx * 16);
  dst[0] = lo;
  dst[1] = hi;
}

// Re-order W[N, K] (row-major bf16) into MFMA fragment order [N/16][K/32][64 lanes][8]: every
// wave-level load of the streaming kernels then reads one contiguous 1 KiB.
__global__ __launch_bounds__(256) void pack_weight_kernel(const unsigned short* __restrict__ W, unsigned short* __restrict__ Wp, int N, int K) {
  typedef __attribute__((ext_vector_type(4))) unsigned short pk_us4;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;  // one 16-byte fragment per thread
  const int ksteps = K >> 5;
  const long long total = (long long)(N >> 4) * ksteps * 64;
  if (idx >= total) return;
  const int lane = (int)(idx & 63);
  const long long tile = idx >> 6;
  const int ks = (int)(tile % ksteps);
  const long long nt = tile / ksteps;
  const int r = lane & 15, q = lane >> 4;
  const pk_us4* src = reinterpret_cast<const pk_us4*>(W + (nt * 16 + r) * K + (long long)ks * 32 + q * 8);
  pk_us4* dst = reinterpret_cast<pk_us4*>(Wp + idx * 8);
  dst[0] = src[0];
  dst[1] = src[1];
}

}  // namespace

extern "C" {

int kv_pack_weight(const void* W, void* Wp, int N, int K, hipStream_t stream) {
  if ((N & 15) || (K & 31)) return -1;
  const long long total = (long long)(N >> 4) * (K >> 5) * 64;
  hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const unsigned short*)W, (unsigned short*)Wp, N, K);
  return (int)hipGetLastError();
}

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
// w_scale != nullptr selects the FP8 weight path: Wp holds OCP e4m3 bytes packed by
// kv_pack_weight_fp8 and w_scale[N] the per-row dequantisation scales.
int kv_stream_gemm(const void* Wp, const void* x, void* out, int M, int N, int K, long long ldx, long long ldy, int epi, int ks_per_chunk, int splits,
                   int chunks_per_split, long long slab_stride, const float* row_ss, int ss_parts, float eps, const float* w_scale, hipStream_t stream) {
  if (M <= 0 || M > 64 || (M & 15) || (N & 63) || (ks_per_chunk != 4 && ks_per_chunk != 8) || (K % (32 * ks_per_chunk))) return -1;
  if (epi == EPI_SWIGLU && (N & 15)) return -1;
  if (epi != EPI_SLAB && splits != 1) return -1;
  if (splits < 1 || chunks_per_split < 1 || (long long)splits * chunks_per_split < (K >> 5) / ks_per_chunk) return -1;
  const Args a{Wp, x, out, N, K, ldx, ldy, splits, chunks_per_split, slab_stride, row_ss, ss_parts, eps, w_scale, stream};
  const bool w8 = w_scale != nullptr;
  switch (M >> 4) {
    case 1: return dispatch_ks<1>(epi, ks_per_chunk, w8, a);
    case 2: return dispatch_ks<2>(epi, ks_per_chunk, w8, a);
    case 3: return dispatch_ks<3>(epi, ks_per_chunk, w8, a);
    case 4: return dispatch_ks<4>(epi, ks_per_chunk, w8, a);
  }
  return -1;
}

// v3 (deep register pipeline): chunks are 4 k-steps (128 columns of K); a workgroup owns
// `groups_per_split` groups of `nb` chunks.  Requirements as kv_stream_gemm plus
// K % (128 * nb) == 0, nb in {2, 4}.  w_scale != nullptr selects FP8 weights.  The plan aims
// at `target_wgs` workgroups (tiles x splits).
int kv_stream_gemm3_plan(int N, int K, int nb, int nw, int max_splits, int target_wgs, int* groups_per_split) {
  if (nb < 1 || K % (128 * nb) || nw < 1 || N % (16 * nw)) return -1;
  if (target_wgs < 1) target_wgs = 512;
  const int ngroups = K / (128 * nb);
  const int tiles = N / (16 * nw);
  int want = (target_wgs + tiles - 1) / tiles;
  if (want > max_splits) want = max_splits;
  if (want > ngroups) want = ngroups;
  if (want < 1) want = 1;
  const int gps = (ngroups + want - 1) / want;
  if (groups_per_split) *groups_per_split = gps;
  return (ngroups + gps - 1) / gps;
}

int kv_stream_gemm3(const void* Wp, const void* x, void* out, int M, int N, int K, long long ldx, long long ldy, int epi, int nb, int nw, int splits,
                    int groups_per_split, long long slab_stride, const float* row_ss, int ss_parts, float eps, const float* w_scale, int weight_bits,
                    void* residual, int has_residual, float* ss_out, int* counters, int ft, hipStream_t stream) {
  // ft: 16-feature tiles per wave, 1 or 2 (2: see dispatch3_ft2_w4 / dispatch3_ft2_wide for the shapes built)
  if (ft != 1 && ft != 2) return -1;
  // weight_bits: 16, 8 (w_scale = fp32 per-row scales), 4 (MXFP4; w_scale = packed E8M0 block scales) or 0 = 8 if w_scale else 16
  const int wb = weight_bits ? weight_bits : (w_scale != nullptr ? 8 : 16);
  if ((wb != 16 && wb != 8 && wb != 4) || ((wb != 16) != (w_scale != nullptr))) return -1;
  const bool wide = M == 96 || M == 128 || M == 256;  // 6 / 8 / 16 row tiles per wave: 2-chunk groups only (registers)
  const int chunk_cols = 32 * chunk_ksteps(M >> 4, wb);
  if (M <= 0 || (M > 64 && !wide) || (M & 15) || (nw != 4 && nw != 6 && nw != 7 && nw != 8) || (N % (16 * ft)) || ((N % (16 * nw * ft)) && (ft != 1 || epi == EPI_RESID)) || (nb != 2 && nb != 4 && nb != 8) || (K % (chunk_cols * nb))) return -1;
  if (wide && nb != 2) return -1;
  if (epi == EPI_SWIGLU && (N & 15)) return -1;
  if (epi != EPI_SLAB && epi != EPI_RESID && splits != 1) return -1;
  if (epi == EPI_RESID && (!residual || !ss_out || !counters || M > 64 || ldy != N)) return -1;
  const int ngroups = K / (chunk_cols * nb);
  if (splits < 1 || groups_per_split < 1 || (long long)splits * groups_per_split < ngroups || (long long)(splits - 1) * groups_per_split >= ngroups) return -1;
  Args a{Wp, x, out, N, K, ldx, ldy, splits, groups_per_split, slab_stride, row_ss, ss_parts, eps, w_scale, stream};
  a.ra = ResidArgs{(unsigned short*)residual, has_residual, ss_out, counters};
  if (ft == 2) {
    if (epi == EPI_RESID) return -1;
    switch (M >> 4) {
      case 1: return wb == 4 ? dispatch3_ft2_w4<1>(epi, nb, nw, a) : -1;
      case 2: return wb == 4 ? dispatch3_ft2_w4<2>(epi, nb, nw, a) : -1;
      case 3: return wb == 4 ? dispatch3_ft2_w4<3>(epi, nb, nw, a) : -1;
      case 4: return wb == 4 ? dispatch3_ft2_w4<4>(epi, nb, nw, a) : -1;
      case 8: return wb == 4 ? dispatch3_ft2_wide<8, 4>(epi, nw, a) : wb == 8 ? dispatch3_ft2_wide<8, 8>(epi, nw, a) : dispatch3_ft2_wide<8, 16>(epi, nw, a);
      case 16: return wb == 4 ? -1 : wb == 8 ? dispatch3_ft2_wide<16, 8>(epi, nw, a) : dispatch3_ft2_wide<16, 16>(epi, nw, a);
    }
    return -1;
  }
  switch (M >> 4) {
    case 1: return dispatch3_wb<1>(epi, nb, nw, wb, a);
    case 2: return dispatch3_wb<2>(epi, nb, nw, wb, a);
    case 3: return dispatch3_wb<3>(epi, nb, nw, wb, a);
    case 4: return dispatch3_wb<4>(epi, nb, nw, wb, a);
    case 6: return dispatch3_wide_wb<6>(epi, nw, wb, a);
    case 8: return dispatch3_wide_wb<8>(epi, nw, wb, a);
    case 16: return wb == 4 ? -1 : (wb == 8 ? dispatch3_wide<16, 8>(epi, nw, a) : dispatch3_wide<16, 16>(epi, nw, a));  // (256-row tiles are not built for MXFP4: callers use 128-row passes)
  }
  return -1;
}

// residual[M, N] (bf16, in place; has_residual = 0: taken as zero) += x[M, K] @ W^T, and
// ss_out[M][N/16] = per-row sums of squares of the new residual over each 16-feature tile, in one
// launch without slabs (stream_gemm_rowsum_kernel).  M in {16, 32, 48, 64}, N % 16 == 0,
// K % 1024 == 0 (8 waves x chunks of 128 columns), nb = ring depth (2; 3 or 4 up to 32 rows).
// w_scale != nullptr: FP8 (e4m3) weights with one scale per output row.
int kv_stream_gemm_rowsum(const void* Wp, const void* x, int M, int N, int K, long long ldx, int nb, void* residual, int has_residual, float* ss_out,
                          const float* w_scale, hipStream_t stream) {
  if (M <= 0 || M > 64 || (M & 15) || N <= 0 || (N & 15) || K <= 0 || (K % 1024) || !residual || !ss_out || !Wp || !x) return -1;
  const bool w8 = w_scale != nullptr;
  switch (M >> 4) {
    case 1: return w8 ? launch_rowsum<1, true>(Wp, x, N, K, ldx, nb, residual, has_residual, ss_out, w_scale, stream) : launch_rowsum<1, false>(Wp, x, N, K, ldx, nb, residual, has_residual, ss_out, w_scale, stream);
    case 2: return w8 ? launch_rowsum<2, true>(Wp, x, N, K, ldx, nb, residual, has_residual, ss_out, w_scale, stream) : launch_rowsum<2, false>(Wp, x, N, K, ldx, nb, residual, has_residual, ss_out, w_scale, stream);
    case 3: return w8 ? launch_rowsum<3, true>(Wp, x, N, K, ldx, nb, residual, has_residual, ss_out, w_scale, stream) : launch_rowsum<3, false>(Wp, x, N, K, ldx, nb, residual, has_residual, ss_out, w_scale, stream);
    case 4: return w8 ? launch_rowsum<4, true>(Wp, x, N, K, ldx, nb, residual, has_residual, ss_out, w_scale, stream) : launch_rowsum<4, false>(Wp, x, N, K, ldx, nb, residual, has_residual, ss_out, w_scale, stream);
  }
  return -1;
}

// MXFP4 packed weights (pack_mxfp4 order: [N/16][K/128][64 lanes][16 B] + E8M0 scale words
// [N/16][K/128][16 rows]) -> the bf16 fragment order of kv_pack_weight ([N/16][K/32][64][8]),
// exact (e2m1 times a power of two fits in bf16).  Wide steps expand a matrix into a scratch
// buffer with this and run the bf16 tiled GEMM on it: 0.5 bytes read and 2 written per weight,
// once per use, instead of one 128-row pass of the streaming kernel per 128 token rows.
int kv_unpack_mxfp4(const void* Wp4, const void* scales, void* Wp, int N, int K, hipStream_t stream) {
  if (N <= 0 || K <= 0 || (N & 15) || (K & 127)) return -1;
  const long long total = (long long)(N >> 4) * (K >> 7) * 64;
  hipLaunchKernelGGL(unpack_mxfp4_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const raw16*)Wp4, (const unsigned int*)scales, (raw16*)Wp, total);
  return (int)hipGetLastError();
}

int kv_pack_weight_fp8(const void* W, void* Wp, int N, int K, hipStream_t stream) {
  if ((N & 15) || (K & 63)) return -1;
  const long long total = (long long)(N >> 4) * (K >> 6) * 64;
  hipLaunchKernelGGL(pack_weight_fp8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const unsigned char*)W, (unsigned char*)Wp, N, K);
  return (int)hipGetLastError();
}

}  // extern "C"

// Tiled MFMA GEMM for wide token batches (prompt chunks: 129 .. 2048+ token rows) on gfx950:
//     y[M, N] = x[M, K] @ W[N, K]^T      bf16 (or e4m3 weights) in, fp32 accumulate
//
// Above ~128 token rows the projection stops being a weight-streaming problem: every weight
// byte feeds hundreds of MACs, so both operands have to be re-used from LDS by many waves.
// This kernel reads the SAME packed weights as the streaming kernels (csrc/stream_gemm.hip:
// MFMA fragment order [N/16][K/32][64 lanes][8 bf16], FP8 [N/16][K/64][64][16 B]), so a model
// keeps ONE copy of its weights for every batch width.
//
//  * Workgroup tile BM tokens x BN features, BK = 64 per step, WM x WN waves; a wave owns
//    (BM/WM) x (BN/WN) outputs = FB x FA tiles of mfma_f32_16x16x32_bf16 (W is the A operand,
//    x the B operand: an accumulator register holds 4 consecutive features of one token).
//  * Both operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave
//    instruction, no VGPR round trip) into a ring of NS stages whose loads stay in flight
//    ACROSS the barriers (counted s_waitcnt vmcnt(N) + raw s_barrier, one barrier per stage).
//  * Software pipeline in registers: while the MFMAs of stage i run from one register set, the
//    fragments of stage i+1 are read from LDS into the other set and the DMA of stage i+NS is
//    issued, all interleaved between the MFMAs (see the kernel's header comment).
//  * Weights: a (16-feature tile, k-step) fragment is 1 KiB contiguous in memory AND in LDS;
//    a wave's ds_read_b128 of it is lane-linear, hence bank-conflict-free with no swizzle.
//  * x is row-major, so it is staged in full 128-byte lines (8 rows x 128 B per LDS-DMA piece)
//    into a [BM][64] image whose 16-byte chunks are XOR-swizzled with (row >> 1) & 7: the
//    B-fragment read (16 rows x one chunk per quarter wave) then touches 16 distinct 16-byte
//    slots of the 256-byte bank row in each of ds_read_b128's four 16-lane groups.  LDS-DMA
//    writes lane-linear, so the swizzle is applied to the per-lane SOURCE address.
//  * Narrow outputs at moderate M do not fill 256 CUs with tiles alone: K can be split
//    across workgroups into fp32 slabs that the consumer kernels (residual add, RoPE) sum in
//    their prologue, exactly like the streaming kernels' slabs.
//  * The blockIdx -> tile map first regroups ids so that each XCD (blockIdx % 8) owns a
//    contiguous range of tiles, then runs token tiles fastest: the workgroups of one XCD
//    share W panels through that XCD's L2.
//  * Epilogues: bf16 (optional 1/rms row scale of the folded RMSNorm, FP8 per-feature
//    dequantisation scale), fp32 slabs, SwiGLU on the 8-row interleaved gate/up layout.

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 frag_ab;
typedef __attribute__((ext_vector_type(4))) float frag_cd;
typedef __attribute__((ext_vector_type(4))) unsigned short us4;
typedef __attribute__((ext_vector_type(4))) unsigned int raw16;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(8))) int v8i;
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

// one LDS-DMA piece of 4-byte units: lane l's dword at `src` lands at lds_base + 4 * l
__device__ inline void dma4(const void* src, void* lds_base) {
  __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)lds_base, 4, 0, 0);
}

// MXFP4 fragment as packed (lane (r, q), dword u = the 8 values k = 32 u + 8 q .. of row r) -> the FP4
// operand of the scaled MFMA (lane (r, q) holds k = 32 q .. 32 q + 31 in order): a 4 x 4 transpose of
// dwords across the four lane quarters, two swap levels (rows of 16 lanes, then halves of 32).
__device__ inline v8i fp4_operand(raw16 w) {
  typedef __attribute__((ext_vector_type(2))) unsigned int u2;
  u2 a = __builtin_amdgcn_permlane16_swap(w[0], w[1], false, false);
  u2 b = __builtin_amdgcn_permlane16_swap(w[2], w[3], false, false);
  u2 c = __builtin_amdgcn_permlane32_swap(a[0], b[0], false, false);
  u2 d = __builtin_amdgcn_permlane32_swap(a[1], b[1], false, false);
  v8i v;
  v[0] = (int)c[0]; v[1] = (int)d[0]; v[2] = (int)c[1]; v[3] = (int)d[1];
  v[4] = 0; v[5] = 0; v[6] = 0; v[7] = 0;
  return v;
}

// the 32-byte operand of the scaled MFMA from two 16-byte pieces
__device__ inline v8i join16(raw16 a, raw16 b) {
  v8i v;
  v[0] = (int)a[0]; v[1] = (int)a[1]; v[2] = (int)a[2]; v[3] = (int)a[3];
  v[4] = (int)b[0]; v[5] = (int)b[1]; v[6] = (int)b[2]; v[7] = (int)b[3];
  return v;
}

enum Epilogue { EPI_BF16 = 0, EPI_SLAB = 1, EPI_SWIGLU = 2 };

__device__ inline unsigned short f2bf(float f) {
  unsigned int u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

// 8 e4m3 bytes (two dwords) -> 8 bf16, exact
__device__ inline frag_ab fp8x8_to_bf16(unsigned int lo, unsigned int hi) {
  const bf16x2 p0 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(lo, 1.0f, false);
  const bf16x2 p1 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(lo, 1.0f, true);
  const bf16x2 p2 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(hi, 1.0f, false);
  const bf16x2 p3 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(hi, 1.0f, true);
  frag_ab a;
  a[0] = p0[0]; a[1] = p0[1]; a[2] = p1[0]; a[3] = p1[1]; a[4] = p2[0]; a[5] = p2[1]; a[6] = p3[0]; a[7] = p3[1];
  return a;
}


// one LDS-DMA piece: lane l's 16 bytes at `src` land at lds_base + 16 * l
__device__ inline void dma16(const void* src, void* lds_base) {
  __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)lds_base, 16, 0, 0);
}

struct Params {
  const unsigned char* Wp; const __bf16* x; void* out;
  int M, N, K; long long ldx, ldy;
  int stages_per_split, splits; long long slab_stride;
  const float* row_ss; int ss_parts; float inv_k, eps; const float* w_scale;
  int n_mtiles, n_ntiles, m_fastest;
  const int* w_blk;      // MXFP4: packed E8M0 block scales, int32 [N/16][K/128][16 rows] (byte u = block u of the 128 columns)
  const float* a_scale;  // FP8 activations: per-token scale (x ~= x8 * a_scale[t]); p.x then points at e4m3 bytes and K, ldx count PAIRS of them
};

// Epilogue shared by the kernels below:
// acc[i][j][e] = y[token m0 + wm TM + 16 j + r][feature n0 + wn TN + 16 i + 4 q + e]
template <int FA, int FB, int TM, int TN, int EPI, bool W8, bool A8 = false>
__device__ inline void wide_epilogue(frag_cd (&acc)[FA][FB], const Params& p, int m0, int n0, int wm, int wn, int r, int q, int split) {
  const int tok0 = m0 + wm * TM + r;
  const int feat0 = n0 + wn * TN + 4 * q;
  float rs[FB];
#pragma unroll
  for (int j = 0; j < FB; ++j) {
    rs[j] = 1.f;
    if (EPI != EPI_SLAB && p.row_ss != nullptr) {
      const int t = tok0 + 16 * j < p.M ? tok0 + 16 * j : p.M - 1;
      float ssum = 0.f;
      for (int k = 0; k < p.ss_parts; ++k) ssum += p.row_ss[(long long)t * p.ss_parts + k];
      rs[j] = rsqrtf(ssum * p.inv_k + p.eps);
    }
    if constexpr (A8) {  // per-token scale of the FP8 activations (fp32 slabs carry it too: it differs from row to row, not from split to split)
      const int t = tok0 + 16 * j < p.M ? tok0 + 16 * j : p.M - 1;
      rs[j] *= p.a_scale[t];
    }
  }
#pragma unroll
  for (int ii = 0; ii < FA; ++ii) {
    const int f = feat0 + 16 * ii;
    frag_cd ws = (frag_cd){1.f, 1.f, 1.f, 1.f};
    if constexpr (W8) ws = *reinterpret_cast<const frag_cd*>(p.w_scale + f);
#pragma unroll
    for (int j = 0; j < FB; ++j) {
      const int t = tok0 + 16 * j;
      frag_cd v = acc[ii][j];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= ws[e] * rs[j];
      if constexpr (EPI == EPI_SLAB) {
        if (t < p.M) *reinterpret_cast<frag_cd*>(reinterpret_cast<float*>(p.out) + (long long)split * p.slab_stride + (long long)t * p.ldy + f) = v;
      } else if constexpr (EPI == EPI_BF16) {
        us4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = f2bf(v[e]);
        if (t < p.M) *reinterpret_cast<us4*>(reinterpret_cast<unsigned short*>(p.out) + (long long)t * p.ldy + f) = o;
      } else {
        // a 16-row weight tile = 8 gate rows + the 8 up rows of the same features: lanes q < 2
        // hold gate[4q + e], lanes q + 2 the matching up values
        us4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float g = v[e], up = __shfl_xor(v[e], 32, 64);
          o[e] = f2bf(g / (1.f + __expf(-g)) * up);
        }
        const int fi = ((n0 + wn * TN + 16 * ii) >> 1) + 4 * q;
        if (q < 2 && t < p.M) *reinterpret_cast<us4*>(reinterpret_cast<unsigned short*>(p.out) + (long long)t * p.ldy + fi) = o;
      }
    }
  }
}

// KS = k-steps (of 32) per LDS stage: 2 -> 128-byte x rows (full cache lines), 1 -> 64-byte rows
// NS = LDS stages in the ring
//
// Software pipeline (one barrier per stage, every wave's stage is the same mix of work):
//   top of step i:  s_waitcnt vmcnt((NS - 2) G)   this wave's pieces of stage i+1 have landed
//                   s_barrier                     ... and everybody's; all reads of stage i are done
//                   LDS-DMA of stage i+NS   -> buffer i % NS (free: stage i sits in registers)
//                   fragment reads of stage i+1 -> register set (i+1) & 1   } interleaved
//                   MFMAs on register set i & 1 (stage i)                   }
// so neither the DMA issue (~10 cycles of the CU's vector-memory path per KiB), nor LDS read
// latency, nor the wait for a stage stands between two MFMAs of a wave: measured on the first
// version of this kernel (all DMAs, then all reads, then all MFMAs per stage) those took 23 %,
// 13 % and 22 % (barrier skew) of a step next to 38 % for the MFMAs.
// Stages past the end of the K range are re-reads of the last one (the counts stay exact).
template <int BM, int BN, int WM, int WN, int KS, int NS, int EPI, int WQ>
__global__ __launch_bounds__(WM * WN * 64) void wide_gemm_kernel(const Params p) {
  // WQ: 0 bf16 | 1 FP8 weights widened to bf16 in registers | 2 FP8 weights x FP8 activations | 3 MXFP4 weights x FP8
  // activations (2, 3: on the block-scaled MFMA; x is e4m3 bytes, a stage covers 128 K-elements)
  constexpr bool W4 = WQ == 3;
  constexpr bool W8 = WQ == 1 || W4;  // one 16-byte unit per (tile, stage) and lane: 1 KiB of W per tile and stage
  constexpr bool A8 = WQ == 2 || W4;
  static_assert(!A8 || KS == 2, "the scaled MFMA takes a whole 64-byte-wide stage pair");
  constexpr int NW = WM * WN;
  constexpr int TM = BM / WM, TN = BN / WN;  // per-wave tokens x features
  constexpr int FB = TM / 16, FA = TN / 16;  // B (x) and A (W) fragment tiles per wave and k-step
  static_assert(!W8 || KS == 2, "FP8 weights are packed two k-steps per 16-byte unit");
  static_assert(NS >= 2, "ring of at least two stages");
  constexpr int A_TILE = W8 ? 1024 : KS * 1024;  // bytes of one 16-feature tile per stage
  constexpr int A_BYTES = (BN / 16) * A_TILE;
  constexpr int XROW = KS * 64;                  // bytes of an x row per stage
  constexpr int X_BYTES = BM * XROW;
  constexpr int SC_BYTES = W4 ? (BN / 16) * 64 : 0;  // MXFP4: the E8M0 scale words of the stage, one per (tile, row)
  constexpr int STAGE = A_BYTES + X_BYTES + SC_BYTES;
  constexpr int PA = A_BYTES / 1024, PX = X_BYTES / 1024;  // 1 KiB LDS-DMA pieces per stage
  constexpr int PS = SC_BYTES / 256, PS_W = W4 ? 1 : 0;    // 256-byte pieces of scale words (four tiles each); one slot per wave (waves beyond PS repeat a piece)
  constexpr int SR = W4 ? (BN / WN) / 16 : 0;              // scale words a lane reads per stage
  // Every wave issues the same number of LDS-DMA pieces per stage (the vmcnt counts are exact).
  // When the W pieces do not divide by the waves (224- and 192-feature tiles), the surplus slots
  // copy a piece a second time: same source, same destination, same bytes.
  static_assert(PX % NW == 0, "x pieces divide by the waves");
  constexpr int PA_W = (PA + NW - 1) / NW, PX_W = PX / NW;
  static_assert(PA_W * NW - PA <= PA, "a surplus slot repeats a real piece");
  constexpr int G = PA_W + PX_W + PS_W;  // LDS-DMA instructions per wave and stage
  constexpr int AR = W8 ? FA : KS * FA;  // 16-byte LDS reads of W per wave and stage
  constexpr int NREAD = AR + SR + KS * FB;  // LDS reads per wave and stage
  constexpr int NGRP = KS * FA;          // MFMA groups (FB MFMAs each) per stage
  constexpr int MEM = NREAD + G;         // mem ops to spread over the groups
  constexpr int PER = (MEM + NGRP - 1) / NGRP;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave % WM, wn = wave / WM;
  const int r = lane & 15, q = lane >> 4;

  // ---- blockIdx -> (token tile, feature tile, K split): XCD-contiguous ranges of tiles
  const int nwg = gridDim.x;
  int id = blockIdx.x;
  {
    const int xcd = id & 7, qd = nwg >> 3, rm = nwg & 7;
    id = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (id >> 3);
  }
  int mt, nt, split;
  if (p.m_fastest) {
    mt = id % p.n_mtiles; id /= p.n_mtiles;
    split = id % p.splits; nt = id / p.splits;
  } else {
    nt = id % p.n_ntiles; id /= p.n_ntiles;
    split = id % p.splits; mt = id / p.splits;
  }
  const int m0 = mt * BM, n0 = nt * BN;
  const int nst = p.K / (32 * KS);  // stages over the whole K
  const int s0 = split * p.stages_per_split;
  const int s1 = s0 + p.stages_per_split < nst ? s0 + p.stages_per_split : nst;
  const int n = s1 - s0;

  // ---- per-lane LDS-DMA source pointers of stage s0 (a stage further = + A_TILE / + XROW bytes)
  const unsigned char* asrc[PA_W];
  const unsigned char* xsrc[PX_W];
  int apc[PA_W];  // W piece of slot i of this wave
#pragma unroll
  for (int i = 0; i < PA_W; ++i) {
    const int pc = wave + NW * i < PA ? wave + NW * i : wave + NW * i - PA;
    apc[i] = pc;
    if constexpr (W8) {
      asrc[i] = p.Wp + ((((long long)(n0 >> 4) + pc) * nst + s0) * 64 + lane) * 16;
    } else {
      const int ti = pc / KS, sl = pc % KS;
      asrc[i] = p.Wp + ((((long long)(n0 >> 4) + ti) * (nst * KS) + (long long)s0 * KS + sl) * 64 + lane) * 16;
    }
  }
#pragma unroll
  for (int i = 0; i < PX_W; ++i) {
    const int pc = wave + NW * i;
    int row, chunk;
    if constexpr (KS == 2) {  // 8 rows x 128 B per piece; chunk ^= (row >> 1) & 7
      row = 8 * pc + (lane >> 3);
      chunk = (lane & 7) ^ ((row >> 1) & 7);
    } else {  // 16 rows x 64 B per piece; chunk ^= (4 - (row >> 2)) & 3
      row = 16 * pc + (lane >> 2);
      chunk = (lane & 3) ^ ((4 - ((row >> 2) & 3)) & 3);
    }
    row += m0;
    row = row < p.M ? row : p.M - 1;  // rows past the end read the last row; their outputs are not stored
    xsrc[i] = reinterpret_cast<const unsigned char*>(p.x + (long long)row * p.ldx + (long long)s0 * (32 * KS)) + chunk * 16;
  }

  // MXFP4 block scales: int32 [N/16][K/128][16 rows]; this wave's 256-byte piece = four tiles of the stage
  const int spc = PS > 0 ? wave % (PS > 0 ? PS : 1) : 0;
  const unsigned char* ssrc = W4 ? reinterpret_cast<const unsigned char*>(p.w_blk + ((((long long)(n0 >> 4) + 4 * spc + (lane >> 4)) * nst + s0) * 16 + (lane & 15))) : nullptr;

  frag_cd acc[FA][FB];
#pragma unroll
  for (int i = 0; i < FA; ++i)
#pragma unroll
    for (int j = 0; j < FB; ++j) acc[i][j] = (frag_cd){0.f, 0.f, 0.f, 0.f};

  // lane-constant LDS read offsets
  const int a_off = wn * (FA * A_TILE) + lane * 16;
  const int s_off = A_BYTES + X_BYTES + wn * (FA * 64) + r * 4;
  int x_off[KS];
  if constexpr (KS == 2) {
    x_off[0] = A_BYTES + (wm * TM + r) * 128 + (((0 + q) ^ (r >> 1)) << 4);
    x_off[1] = A_BYTES + (wm * TM + r) * 128 + (((4 + q) ^ (r >> 1)) << 4);
  } else {
    x_off[0] = A_BYTES + (wm * TM + r) * 64 + ((q ^ ((4 - (r >> 2)) & 3)) << 4);
  }

  // two register sets of fragments: W as read (bf16: one k-step per read, laid out [tile][k-step];
  // FP8: both k-steps of a tile in one read), x per k-step and token tile
  raw16 ra[2][AR];
  frag_ab fb[2][KS][FB];
  int sreg[2][SR > 0 ? SR : 1];  // MXFP4: the lane's scale byte (block q of row r) per W tile

  // mem op m of a step: m < G -> LDS-DMA piece m of stage `st` into `dst`; else fragment read m - G
  // of the stage at `src` into register set SET
#define WG_MEMOP(m, SET, dst, src, st)                                                                                                   \
  do {                                                                                                                                  \
    if ((m) < PA_W) dma16(asrc[(m) < PA_W ? (m) : 0] + (long long)(st) * A_TILE, (dst) + apc[(m) < PA_W ? (m) : 0] * 1024);             \
    else if ((m) < PA_W + PX_W) dma16(xsrc[(m) >= PA_W && (m) < PA_W + PX_W ? (m) - PA_W : 0] + (long long)(st) * XROW, (dst) + A_BYTES + (wave + NW * ((m) - PA_W)) * 1024); \
    else if ((m) < G) dma4(ssrc + (long long)(st) * 64, (dst) + A_BYTES + X_BYTES + spc * 256);                                         \
    else if ((m) - G < AR) ra[SET][(m) >= G && (m) - G < AR ? (m) - G : 0] = *reinterpret_cast<const raw16*>((src) + a_off + ((m) - G) * 1024); \
    else if ((m) - G - AR < SR) sreg[SET][(m) - G - AR >= 0 && (m) - G - AR < SR ? (m) - G - AR : 0] = *reinterpret_cast<const int*>((src) + s_off + ((m) - G - AR) * 64) >> (8 * q); \
    else {                                                                                                                              \
      const int kk_ = (m) - G - AR - SR;                                                                                                \
      const int ss_ = kk_ >= 0 && kk_ < KS * FB ? kk_ / FB : 0, jj_ = kk_ >= 0 ? kk_ % FB : 0;                                          \
      fb[SET][ss_][jj_] = *reinterpret_cast<const frag_ab*>((src) + x_off[ss_] + jj_ * (16 * XROW));                                    \
    }                                                                                                                                   \
  } while (0)

  // ---- prologue: the whole ring in flight, then stage 0 into register set 0
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    unsigned char* dst = smem + s * STAGE;
    const int st = s < n ? s : n - 1;
#pragma unroll
    for (int m = 0; m < G; ++m) WG_MEMOP(m, 0, dst, smem, st);
  }
  if constexpr ((NS - 1) * G == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 1) * G) : "memory");
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int m = G; m < MEM; ++m) WG_MEMOP(m, 0, smem, smem, 0);

  // one step: MFMAs on register set CUR; the DMA of stage I + NS and the reads of stage I + 1
  // (into set NXT) are spread between the MFMA groups
#define WG_STEP(CUR, NXT, I)                                                                                                 \
  do {                                                                                                                      \
    const int i_ = (I);                                                                                                     \
    if constexpr ((NS - 2) * G == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                       \
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * G) : "memory");                                                \
    __builtin_amdgcn_s_barrier();                                                                                           \
    unsigned char* dst_ = smem + (i_ % NS) * STAGE;                                                                         \
    const unsigned char* src_ = smem + ((i_ + 1) % NS) * STAGE;                                                             \
    const int st_ = i_ + NS < n ? i_ + NS : n - 1;                                                                          \
    _Pragma("unroll") for (int g = 0; g < NGRP; ++g) {                                                                      \
      const int s = g / FA, ii = g % FA;                                                                                    \
      if constexpr (W4) {                                                                                                   \
        /* MXFP4 x FP8: the packed fragment is transposed across the lane quarters into the FP4 operand; the lane's */     \
        /* scale byte goes with it (block q of its row), the activations carry unit block scales                     */     \
        const v8i av = fp4_operand(ra[CUR][ii < AR ? ii : 0]);                                                              \
        _Pragma("unroll") for (int j = s * ((FB + 1) / 2); j < (s == 0 ? (FB + 1) / 2 : FB); ++j)                           \
            acc[ii][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, join16(__builtin_bit_cast(raw16, fb[CUR][0][j]), __builtin_bit_cast(raw16, fb[CUR][KS - 1][j])), \
                                                                          acc[ii][j], 4, 0, 0, sreg[CUR][ii < SR ? ii : 0], 0, 0x7f7f7f7f); \
      } else if constexpr (A8) {                                                                                            \
        /* one 16x16x128 scaled MFMA per (W tile, token tile): both 16-byte pieces of the stage on either side; the   */  \
        /* token tiles of a W tile are shared out between its two groups so that the mem ops stay evenly interleaved */  \
        const v8i av = join16(ra[CUR][(ii * 2) < AR ? ii * 2 : 0], ra[CUR][(ii * 2 + 1) < AR ? ii * 2 + 1 : 0]);            \
        _Pragma("unroll") for (int j = s * ((FB + 1) / 2); j < (s == 0 ? (FB + 1) / 2 : FB); ++j)                           \
            acc[ii][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, join16(__builtin_bit_cast(raw16, fb[CUR][0][j]), __builtin_bit_cast(raw16, fb[CUR][KS - 1][j])), \
                                                                          acc[ii][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);  \
      } else {                                                                                                              \
      frag_ab af;                                                                                                           \
      if constexpr (W8) af = s == 0 ? fp8x8_to_bf16(ra[CUR][ii][0], ra[CUR][ii][1]) : fp8x8_to_bf16(ra[CUR][ii][2], ra[CUR][ii][3]); \
      else af = __builtin_bit_cast(frag_ab, ra[CUR][(ii * KS + s) < AR ? ii * KS + s : 0]);                                  \
      _Pragma("unroll") for (int j = 0; j < FB; ++j)                                                                        \
          acc[ii][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, fb[CUR][s][j], acc[ii][j], 0, 0, 0);                      \
      }                                                                                                                     \
      _Pragma("unroll") for (int m = g * PER; m < (g + 1) * PER; ++m)                                                       \
          if (m < MEM) WG_MEMOP(m, NXT, dst_, src_, st_);                                                                   \
      __builtin_amdgcn_sched_barrier(0); /* keep the hand-placed interleave: the scheduler would sink the reads to their use */ \
    }                                                                                                                       \
  } while (0)

  int i = 0;
  for (; i + 1 < n; i += 2) {
    WG_STEP(0, 1, i);
    WG_STEP(1, 0, i + 1);
  }
  if (i < n) WG_STEP(0, 1, i);
#undef WG_STEP
#undef WG_MEMOP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the re-read stages of the tail

  wide_epilogue<FA, FB, TM, TN, EPI, (WQ == 1 || WQ == 2), A8>(acc, p, m0, n0, wm, wn, r, q, split);
}

// ---------------------------------------------------------------------------------------
// Role-split rings (129 .. ~512 token rows).  With few token tiles the weights of a tile column
// are used once or twice: W comes from HBM, x from L2, and the time is set by the W bytes a CU
// keeps in flight.  In the kernel above both operands share one ring, and x takes most of
// every stage (256 x 128 tile: 32 KiB of x next to 16 KiB of W), so only 32 KiB of W are ever
// in flight per CU (measured on the Llama-3-8B gate/up projection at 256 rows: 82 us with one
// stage in flight, 69 us with two, 54 us when the weights are already in cache).  The rings
// cannot simply be given different depths: a wave's s_waitcnt vmcnt retires its loads in issue
// order, so waiting for a shallow x stage would also wait for every W stage issued before it.
// Here the waves split the copying: the first half issue ONLY W pieces into a deep ring of NSW
// stages, the second half ONLY x pieces into a ring of NSX stages; each wave's counter then
// covers one ring, and all waves multiply as before.  64-wide stages (KS = 2).
template <int BM, int BN, int WM, int WN, int NSX, int NSW, int EPI, int WQ>
__global__ __launch_bounds__(WM * WN * 64) void wide_gemm_rs_kernel(const Params p) {
  static_assert(WQ != 3, "MXFP4 weights run in the single-ring kernel");
  constexpr bool W8 = WQ == 1, A8 = WQ == 2;  // FP8 weights widened to bf16 in registers / FP8 weights x FP8 activations on the scaled MFMA
  constexpr int KS = 2;
  constexpr int NW = WM * WN, NL = NW / 2;  // waves, loader waves per operand
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int FB = TM / 16, FA = TN / 16;
  static_assert(NSX >= 2 && NSW >= 2 && (NW % 2) == 0, "two roles, rings of at least two stages");
  constexpr int A_TILE = W8 ? 1024 : KS * 1024;
  constexpr int A_BYTES = (BN / 16) * A_TILE;
  constexpr int XROW = KS * 64;
  constexpr int X_BYTES = BM * XROW;
  constexpr int PA = A_BYTES / 1024, PX = X_BYTES / 1024;
  static_assert(PA % NL == 0 && PX % NL == 0, "pieces divide by the loader waves of their operand");
  constexpr int GA = PA / NL, GX = PX / NL;  // LDS-DMA instructions per stage of a W / an x loader wave
  constexpr int GMAX = GA > GX ? GA : GX;
  constexpr int AR = W8 ? FA : KS * FA;
  constexpr int NREAD = AR + KS * FB;
  constexpr int NGRP = KS * FA;
  constexpr int MEM = NREAD + GMAX;
  constexpr int PER = (MEM + NGRP - 1) / NGRP;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const a_ring = smem;                  // NSW stages of A_BYTES
  unsigned char* const x_ring = smem + NSW * A_BYTES;  // NSX stages of X_BYTES

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave % WM, wn = wave / WM;
  const int r = lane & 15, q = lane >> 4;
  const bool w_role = wave < NL;  // wave-uniform
  const int lw = w_role ? wave : wave - NL;

  const int nwg = gridDim.x;
  int id = blockIdx.x;
  {
    const int xcd = id & 7, qd = nwg >> 3, rm = nwg & 7;
    id = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (id >> 3);
  }
  int mt, nt, split;
  if (p.m_fastest) {
    mt = id % p.n_mtiles; id /= p.n_mtiles;
    split = id % p.splits; nt = id / p.splits;
  } else {
    nt = id % p.n_ntiles; id /= p.n_ntiles;
    split = id % p.splits; mt = id / p.splits;
  }
  const int m0 = mt * BM, n0 = nt * BN;
  const int nst = p.K / (32 * KS);
  const int s0 = split * p.stages_per_split;
  const int s1 = s0 + p.stages_per_split < nst ? s0 + p.stages_per_split : nst;
  const int n = s1 - s0;

  // ---- this wave's LDS-DMA pieces: source pointer at stage s0, byte offset inside a ring stage
  const unsigned char* src[GMAX];
  int doff[GMAX];
#pragma unroll
  for (int i = 0; i < GMAX; ++i) {
    src[i] = p.Wp;
    doff[i] = 0;
    if (w_role) {
      if (i < GA) {
        const int pc = lw + NL * i;
        if constexpr (W8) {
          src[i] = p.Wp + ((((long long)(n0 >> 4) + pc) * nst + s0) * 64 + lane) * 16;
        } else {
          const int ti = pc / KS, sl = pc % KS;
          src[i] = p.Wp + ((((long long)(n0 >> 4) + ti) * (nst * KS) + (long long)s0 * KS + sl) * 64 + lane) * 16;
        }
        doff[i] = pc * 1024;
      }
    } else if (i < GX) {
      const int pc = lw + NL * i;
      int row = 8 * pc + (lane >> 3);  // 8 rows x 128 B per piece; chunk ^= (row >> 1) & 7
      const int chunk = (lane & 7) ^ ((row >> 1) & 7);
      row += m0;
      row = row < p.M ? row : p.M - 1;
      src[i] = reinterpret_cast<const unsigned char*>(p.x + (long long)row * p.ldx + (long long)s0 * (32 * KS)) + chunk * 16;
      doff[i] = pc * 1024;
    }
  }
  const int n_mine = w_role ? GA : GX;           // pieces per stage of this wave
  const int sstride = w_role ? A_TILE : XROW;    // source bytes from one stage to the next
  const int depth = w_role ? NSW : NSX;
  unsigned char* const ring = w_role ? a_ring : x_ring;
  const int ring_stage = w_role ? A_BYTES : X_BYTES;

  frag_cd acc[FA][FB];
#pragma unroll
  for (int i = 0; i < FA; ++i)
#pragma unroll
    for (int j = 0; j < FB; ++j) acc[i][j] = (frag_cd){0.f, 0.f, 0.f, 0.f};

  const int a_off = wn * (FA * A_TILE) + lane * 16;
  int x_off[KS];
  x_off[0] = (wm * TM + r) * 128 + (((0 + q) ^ (r >> 1)) << 4);
  x_off[1] = (wm * TM + r) * 128 + (((4 + q) ^ (r >> 1)) << 4);

  raw16 ra[2][AR];
  frag_ab fb[2][KS][FB];

  // mem op m of a step: m < GMAX -> this wave's LDS-DMA piece m of stage `st` into ring buffer
  // `lb`; else fragment read m - GMAX of the stage in buffers (sa, sx) into register set SET
#define RS_MEMOP(m, SET, lb, sa, sx, st)                                                                                          \
  do {                                                                                                                           \
    if ((m) < GMAX) {                                                                                                            \
      if ((m) < n_mine) dma16(src[(m) < GMAX ? (m) : 0] + (long long)(st) * sstride, ring + (lb) * ring_stage + doff[(m) < GMAX ? (m) : 0]); \
    } else if ((m) - GMAX < AR) {                                                                                                \
      ra[SET][(m) >= GMAX && (m) - GMAX < AR ? (m) - GMAX : 0] = *reinterpret_cast<const raw16*>((sa) + a_off + ((m) - GMAX) * 1024); \
    } else {                                                                                                                     \
      const int kk_ = (m) - GMAX - AR;                                                                                           \
      const int ss_ = kk_ >= 0 && kk_ < KS * FB ? kk_ / FB : 0, jj_ = kk_ >= 0 ? kk_ % FB : 0;                                   \
      fb[SET][ss_][jj_] = *reinterpret_cast<const frag_ab*>((sx) + x_off[ss_] + jj_ * (16 * XROW));                              \
    }                                                                                                                            \
  } while (0)

  // ---- prologue: every ring in flight, then stage 0 into register set 0
  for (int s = 0; s < depth; ++s) {
    const int st = s < n ? s : n - 1;
#pragma unroll
    for (int m = 0; m < GMAX; ++m) RS_MEMOP(m, 0, s, a_ring, x_ring, st);
  }
  if (w_role) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSW - 1) * GA) : "memory");
  else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSX - 1) * GX) : "memory");
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int m = GMAX; m < MEM; ++m) RS_MEMOP(m, 0, 0, a_ring, x_ring, 0);

  int lb = 0;          // ring buffer of stage i in this wave's own ring: refilled with stage i + depth
  int ba = 1 % NSW, bx = 1 % NSX;  // buffers of stage i + 1 in the W and x rings
#define RS_STEP(CUR, NXT, I)                                                                                          \
  do {                                                                                                               \
    const int i_ = (I);                                                                                              \
    if (w_role) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSW - 2) * GA) : "memory");                                \
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSX - 2) * GX) : "memory");                                       \
    __builtin_amdgcn_s_barrier();                                                                                    \
    const unsigned char* sa_ = a_ring + ba * A_BYTES;                                                                \
    const unsigned char* sx_ = x_ring + bx * X_BYTES;                                                                \
    const int st_ = i_ + depth < n ? i_ + depth : n - 1;                                                             \
    _Pragma("unroll") for (int g = 0; g < NGRP; ++g) {                                                               \
      const int s = g / FA, ii = g % FA;                                                                             \
      if constexpr (A8) {                                                                                            \
        const v8i av = join16(ra[CUR][(ii * 2) < AR ? ii * 2 : 0], ra[CUR][(ii * 2 + 1) < AR ? ii * 2 + 1 : 0]);     \
        _Pragma("unroll") for (int j = s * ((FB + 1) / 2); j < (s == 0 ? (FB + 1) / 2 : FB); ++j)                    \
            acc[ii][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, join16(__builtin_bit_cast(raw16, fb[CUR][0][j]), __builtin_bit_cast(raw16, fb[CUR][1][j])), \
                                                                          acc[ii][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f); \
      } else {                                                                                                       \
      frag_ab af;                                                                                                    \
      if constexpr (W8) af = s == 0 ? fp8x8_to_bf16(ra[CUR][ii][0], ra[CUR][ii][1]) : fp8x8_to_bf16(ra[CUR][ii][2], ra[CUR][ii][3]); \
      else af = __builtin_bit_cast(frag_ab, ra[CUR][(ii * KS + s) < AR ? ii * KS + s : 0]);                           \
      _Pragma("unroll") for (int j = 0; j < FB; ++j)                                                                 \
          acc[ii][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, fb[CUR][s][j], acc[ii][j], 0, 0, 0);               \
      }                                                                                                              \
      _Pragma("unroll") for (int m = g * PER; m < (g + 1) * PER; ++m)                                                \
          if (m < MEM) RS_MEMOP(m, NXT, lb, sa_, sx_, st_);                                                          \
      __builtin_amdgcn_sched_barrier(0);                                                                             \
    }                                                                                                                \
    lb = lb + 1 == depth ? 0 : lb + 1;                                                                               \
    ba = ba + 1 == NSW ? 0 : ba + 1;                                                                                 \
    bx = bx + 1 == NSX ? 0 : bx + 1;                                                                                 \
  } while (0)

  int i = 0;
  for (; i + 1 < n; i += 2) {
    RS_STEP(0, 1, i);
    RS_STEP(1, 0, i + 1);
  }
  if (i < n) RS_STEP(0, 1, i);
#undef RS_STEP
#undef RS_MEMOP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  wide_epilogue<FA, FB, TM, TN, EPI, (WQ != 0), A8>(acc, p, m0, n0, wm, wn, r, q, split);
}

template <int BM, int BN, int WM, int WN, int NSX, int NSW, int EPI, int WQ>
int launch_rs(const Params& p, hipStream_t stream) {
  constexpr int A_BYTES = (BN / 16) * (WQ == 1 ? 1024 : 2048);
  constexpr int LDS = NSW * A_BYTES + NSX * BM * 128;
  static_assert(LDS <= 160 * 1024, "LDS budget of a CU");
  auto kern = wide_gemm_rs_kernel<BM, BN, WM, WN, NSX, NSW, EPI, WQ>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return -2;
    attr_set = true;
  }
  const int grid = p.n_mtiles * p.n_ntiles * p.splits;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(WM * WN * 64), LDS, stream, p);
  return (int)hipGetLastError();
}

template <int BM, int BN, int WM, int WN, int NSX, int NSW, int WQ>
int dispatch_epi_rs(int epi, const Params& p, hipStream_t s) {
  switch (epi) {
    case EPI_BF16: return launch_rs<BM, BN, WM, WN, NSX, NSW, EPI_BF16, WQ>(p, s);
    case EPI_SLAB: return launch_rs<BM, BN, WM, WN, NSX, NSW, EPI_SLAB, WQ>(p, s);
    case EPI_SWIGLU: return launch_rs<BM, BN, WM, WN, NSX, NSW, EPI_SWIGLU, WQ>(p, s);
  }
  return -1;
}

template <int BM, int BN, int WM, int WN, int KS, int NS, int EPI, int WQ>
int launch(const Params& p, hipStream_t stream) {
  constexpr int A_BYTES = (BN / 16) * ((WQ == 1 || WQ == 3) ? 1024 : KS * 1024);
  constexpr int LDS = NS * (A_BYTES + BM * KS * 64 + (WQ == 3 ? (BN / 16) * 64 : 0));
  static_assert(LDS <= 160 * 1024, "LDS budget of a CU");
  auto kern = wide_gemm_kernel<BM, BN, WM, WN, KS, NS, EPI, WQ>;
  static bool attr_set = false;  // per instantiation; LDS above the 64 KiB default has to be requested once
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return -2;
    attr_set = true;
  }
  const int grid = p.n_mtiles * p.n_ntiles * p.splits;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(WM * WN * 64), LDS, stream, p);
  return (int)hipGetLastError();
}

template <int BM, int BN, int WM, int WN, int KS, int NS, int WQ>
int dispatch_epi(int epi, const Params& p, hipStream_t s) {
  switch (epi) {
    case EPI_BF16: return launch<BM, BN, WM, WN, KS, NS, EPI_BF16, WQ>(p, s);
    case EPI_SLAB: return launch<BM, BN, WM, WN, KS, NS, EPI_SLAB, WQ>(p, s);
    case EPI_SWIGLU: return launch<BM, BN, WM, WN, KS, NS, EPI_SWIGLU, WQ>(p, s);
  }
  return -1;
}

// variant = pipeline shape of a tile: 0 = 64-wide stages in the deepest ring the LDS holds,
// 1 / 3 = 32-wide stages (bf16 weights only), 2 = 64-wide stages in a two-stage ring,
// 4 / 5 = role-split rings (64-wide stages; W and x in rings of different depth)
template <int WQ>
int dispatch_tile(int bm, int bn, int variant, int epi, const Params& p, hipStream_t s) {
  const int key = bm * 1000 + bn;
  switch (key * 10 + variant) {
    // 256 x 256, 8 waves (wave: 64 tokens x 128 features)
    case 2562560: return dispatch_epi<256, 256, 4, 2, 2, 2, WQ>(epi, p, s);
    // 256 x 224 and 256 x 192, 8 waves (wave: 64 x 112 / 64 x 96): feature counts that are a
    // multiple of 7 x 32 or 3 x 64 but leave a ragged last round of 256-wide tiles over the CUs
    case 2562240: return dispatch_epi<256, 224, 4, 2, 2, 2, WQ>(epi, p, s);
    case 2561920: return dispatch_epi<256, 192, 4, 2, 2, 2, WQ>(epi, p, s);
    // 256 x 128, 8 waves (wave: 64 x 64)
    case 2561280: return dispatch_epi<256, 128, 4, 2, 2, 3, WQ>(epi, p, s);
    case 2561282: return dispatch_epi<256, 128, 4, 2, 2, 2, WQ>(epi, p, s);
    // 128 x 256, 8 waves (wave: 64 x 64)
    case 1282560: return dispatch_epi<128, 256, 2, 4, 2, 3, WQ>(epi, p, s);
    case 1282562: return dispatch_epi<128, 256, 2, 4, 2, 2, WQ>(epi, p, s);
    // 128 x 128, 4 waves (wave: 64 x 64); two workgroups per CU with two stages
    case 1281280: return dispatch_epi<128, 128, 2, 2, 2, 4, WQ>(epi, p, s);
    case 1281282: return dispatch_epi<128, 128, 2, 2, 2, 2, WQ>(epi, p, s);
  }
  if constexpr (WQ != 3) switch (key * 10 + variant) {
    // role-split rings (wide_gemm_rs_kernel): 4 = x ring of 3 + the deepest W ring the LDS holds,
    // 5 = x ring of 2 and a deeper W ring still (128 x 128: small enough for two workgroups per CU)
    case 2561284: return dispatch_epi_rs<256, 128, 4, 2, 3, 4, WQ>(epi, p, s);
    case 2561285: return dispatch_epi_rs<256, 128, 4, 2, 2, 6, WQ>(epi, p, s);
    case 1281284: return dispatch_epi_rs<128, 128, 2, 2, 3, 6, WQ>(epi, p, s);
    case 1281285: return dispatch_epi_rs<128, 128, 2, 2, 2, 3, WQ>(epi, p, s);
    case 1282564: return dispatch_epi_rs<128, 256, 2, 4, 2, 4, WQ>(epi, p, s);
  }
  if constexpr (WQ == 0) {
    switch (key * 10 + variant) {
      // 32-wide stages: deeper rings in the same LDS
      case 2562561: return dispatch_epi<256, 256, 4, 2, 1, 4, 0>(epi, p, s);
      case 2562563: return dispatch_epi<256, 256, 4, 2, 1, 5, 0>(epi, p, s);
      // 256 x 256, 4 waves (wave: 128 tokens x 128 features, one wave per SIMD): a k-step takes 16 fragment reads for 64
      // MFMAs per wave where the 8-wave form takes 12 for 32 (64 KiB of LDS reads per CU and k-step instead of 96).  Not in
      // the planner's table — measured on MI355X, 2048 x 28672 x 4096: 729 us against 444 us for the 8-wave form in the same
      // run (library: 386): with one wave per SIMD nothing runs while that wave sits at the stage barrier, and the
      // 64-wide-stage form of it spills (fragment sets of 256 registers next to 256 accumulators; 1142 us).
      case 2562567: return dispatch_epi<256, 256, 2, 2, 1, 4, 0>(epi, p, s);
      case 2561281: return dispatch_epi<256, 128, 4, 2, 1, 6, 0>(epi, p, s);
      case 2562241: return dispatch_epi<256, 224, 4, 2, 1, 4, 0>(epi, p, s);
      case 2562243: return dispatch_epi<256, 224, 4, 2, 1, 5, 0>(epi, p, s);
      case 2561921: return dispatch_epi<256, 192, 4, 2, 1, 4, 0>(epi, p, s);
      case 2561923: return dispatch_epi<256, 192, 4, 2, 1, 5, 0>(epi, p, s);
      // 256 x 64, 8 waves (wave: 32 x 64); an FP8 stage of it would be 4 pieces for 8 waves
      case 2560640: return dispatch_epi<256, 64, 8, 1, 2, 3, 0>(epi, p, s);
      case 2560642: return dispatch_epi<256, 64, 8, 1, 2, 2, 0>(epi, p, s);
    }
  }
  return -1;
}

}  // namespace

extern "C" {

// k-steps (of 32) per pipeline stage of a (tile, variant): K splits are handed out in stages
int kv_wide_gemm_stage_ksteps(int bm, int bn, int variant) { (void)bm; (void)bn; return (variant == 1 || variant == 3 || variant == 7) ? 1 : 2; }

// epi: 0 = bf16 y[M, N], 1 = fp32 slabs out[splits][M][N], 2 = SwiGLU bf16 y[M, N/2].
// (bm, bn): workgroup tile, one of (256,256) (256,224) (256,192) (256,128) (256,64) (128,128) (128,256); variant: see
// dispatch_tile.  N % bn == 0, K % 64 == 0; with S = K / (32 * stage k-steps) stages in all,
// splits * stages_per_split >= S > (splits - 1) * stages_per_split.
// w_scale != nullptr: Wp holds e4m3 bytes packed by kv_pack_weight_fp8, w_scale[N] the row scales.
int kv_wide_gemm(const void* Wp, const void* x, void* out, int M, int N, int K, long long ldx, long long ldy, int epi, int bm, int bn, int variant, int splits,
                 int stages_per_split, long long slab_stride, const float* row_ss, int ss_parts, float eps, const float* w_scale, int m_fastest,
                 hipStream_t stream) {
  if (M <= 0 || N <= 0 || K <= 0 || (K & 63) || bn <= 0 || (N % bn)) return -1;
  if (epi != EPI_SLAB && splits != 1) return -1;
  const int nst = K / (32 * kv_wide_gemm_stage_ksteps(bm, bn, variant));
  if (splits < 1 || stages_per_split < 1 || (long long)splits * stages_per_split < nst || (long long)(splits - 1) * stages_per_split >= nst) return -1;
  if ((ldx & 7) || (ldy & 3)) return -1;
  Params p;
  p.Wp = (const unsigned char*)Wp; p.x = (const __bf16*)x; p.out = out;
  p.M = M; p.N = N; p.K = K; p.ldx = ldx; p.ldy = ldy;
  p.stages_per_split = stages_per_split; p.splits = splits; p.slab_stride = slab_stride;
  p.row_ss = row_ss; p.ss_parts = ss_parts; p.inv_k = 1.f / (float)K; p.eps = eps; p.w_scale = w_scale;
  p.n_mtiles = (M + bm - 1) / bm; p.n_ntiles = N / bn; p.m_fastest = m_fastest; p.a_scale = nullptr; p.w_blk = nullptr;
  return w_scale ? dispatch_tile<1>(bm, bn, variant, epi, p, stream) : dispatch_tile<0>(bm, bn, variant, epi, p, stream);
}

// FP8 weights x FP8 activations on the block-scaled matrix instruction (v_mfma_scale_f32_16x16x128_f8f6f4,
// e4m3 operands, unit block scales: twice the bf16 MFMA rate).  x8: e4m3 [M][ldx8 bytes] as kv_quantize_rows_fp8
// writes it (the k order of the packed weights inside every 64 elements), a_scale[M] its per-token scales, w_scale[N]
// the weight row scales.  Byte for byte the data movement of the bf16 kernel on K / 2 columns: the same tiles and
// variants (64-wide stages only: variants 0, 2, 4, 5), stages counted in 128 K-elements.  K % 128 == 0, ldx8 % 16 == 0.
int kv_wide_gemm_fp8a(const void* Wp, const void* x8, void* out, int M, int N, int K, long long ldx8, long long ldy, int epi, int bm, int bn, int variant, int splits,
                      int stages_per_split, long long slab_stride, const float* row_ss, int ss_parts, float eps, const float* w_scale, const float* a_scale,
                      int m_fastest, hipStream_t stream) {
  if (M <= 0 || N <= 0 || K <= 0 || (K & 127) || bn <= 0 || (N % bn) || w_scale == nullptr || a_scale == nullptr) return -1;
  if (variant == 1 || variant == 3 || variant == 7) return -1;
  if (epi != EPI_SLAB && splits != 1) return -1;
  const int nst = K / 128;
  if (splits < 1 || stages_per_split < 1 || (long long)splits * stages_per_split < nst || (long long)(splits - 1) * stages_per_split >= nst) return -1;
  if ((ldx8 & 15) || (ldy & 3)) return -1;
  Params p;
  p.Wp = (const unsigned char*)Wp; p.x = (const __bf16*)x8; p.out = out;
  p.M = M; p.N = N; p.K = K / 2; p.ldx = ldx8 / 2; p.ldy = ldy;  // in pairs of bytes: what the bf16 data movement counts in
  p.stages_per_split = stages_per_split; p.splits = splits; p.slab_stride = slab_stride;
  p.row_ss = row_ss; p.ss_parts = ss_parts; p.inv_k = 1.f / (float)K; p.eps = eps; p.w_scale = w_scale; p.a_scale = a_scale;
  p.n_mtiles = (M + bm - 1) / bm; p.n_ntiles = N / bn; p.m_fastest = m_fastest; p.w_blk = nullptr;
  return dispatch_tile<2>(bm, bn, variant, epi, p, stream);
}

// MXFP4 weights x FP8 activations on the block-scaled MFMA: the weights as kv_pack_mxfp4 lays them out (the streaming
// kernels' layout: [N/16][K/128][64 lanes][16 B]) with their E8M0 block scales w_blk (int32 [N/16][K/128][16]) applied by
// the instruction itself; x8: e4m3 [M][ldx8] in NATURAL k order (kv_quantize_rows_fp8 with permute = 0), a_scale[M].
// Tiles (256,128), (128,256), (128,128), variants 0 and 2.  K % 128 == 0, N % 64 == 0.
int kv_wide_gemm_w4a8(const void* Wp4, const int* w_blk, const void* x8, void* out, int M, int N, int K, long long ldx8, long long ldy, int epi, int bm, int bn, int variant,
                      int splits, int stages_per_split, long long slab_stride, const float* row_ss, int ss_parts, float eps, const float* a_scale, int m_fastest,
                      hipStream_t stream) {
  if (M <= 0 || N <= 0 || K <= 0 || (K & 127) || bn <= 0 || (N % bn) || (bn & 63) || w_blk == nullptr || a_scale == nullptr) return -1;
  if (variant != 0 && variant != 2) return -1;
  if (epi != EPI_SLAB && splits != 1) return -1;
  const int nst = K / 128;
  if (splits < 1 || stages_per_split < 1 || (long long)splits * stages_per_split < nst || (long long)(splits - 1) * stages_per_split >= nst) return -1;
  if ((ldx8 & 15) || (ldy & 3)) return -1;
  Params p;
  p.Wp = (const unsigned char*)Wp4; p.x = (const __bf16*)x8; p.out = out;
  p.M = M; p.N = N; p.K = K / 2; p.ldx = ldx8 / 2; p.ldy = ldy;
  p.stages_per_split = stages_per_split; p.splits = splits; p.slab_stride = slab_stride;
  p.row_ss = row_ss; p.ss_parts = ss_parts; p.inv_k = 1.f / (float)K; p.eps = eps; p.w_scale = nullptr; p.a_scale = a_scale; p.w_blk = w_blk;
  p.n_mtiles = (M + bm - 1) / bm; p.n_ntiles = N / bn; p.m_fastest = m_fastest;
  if (bm == 256 && bn == 256) return -1;
  return dispatch_tile<3>(bm, bn, variant, epi, p, stream);
}

}  // extern "C"

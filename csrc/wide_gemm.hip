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
//    instruction, no VGPR round trip) into a ring of NS stages; NS - 1 stages of loads stay in
//    flight ACROSS the barriers (counted s_waitcnt vmcnt(N) + raw s_barrier, one barrier per
//    stage): with 32 workgroups of an XCD marching through K in lock-step every stage's
//    first touch misses the L2, and one stage of lookahead does not cover that round trip.
//    Optionally each wave also touches the lines of a stage further ahead with 4-byte LDS-DMAs
//    into a scratch area (an L2 prefetch that needs no registers and no LDS stage).
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

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

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
  unsigned long long* dbg;  // diagnostic builds only: per-wave cycle sums of the loop's segments
};

// KS  = k-steps (of 32) per LDS stage: 2 -> 128-byte x rows (full cache lines), 1 -> 64-byte rows
// NS  = LDS stages in the ring; NS - 1 stages of loads are in flight while one is computed on
// PFD = L2 prefetch distance in stages beyond the LDS ring (0 = off): one 4-byte LDS-DMA per
//       cache line of a future stage into a scratch area, so that the real loads of that stage
//       find their lines in the XCD's L2 instead of waiting ~2 us for the Infinity Cache / HBM
// One s_memtime stamp (diagnostic builds): its own wait keeps it ordered against LDS reads.
__device__ inline unsigned long long stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}

template <int BM, int BN, int WM, int WN, int KS, int NS, int PFD, int EPI, bool W8, bool STAMP = false>
__global__ __launch_bounds__(WM * WN * 64) void wide_gemm_kernel(const Params p) {
  constexpr int NW = WM * WN;
  constexpr int TM = BM / WM, TN = BN / WN;  // per-wave tokens x features
  constexpr int FB = TM / 16, FA = TN / 16;  // B (x) and A (W) fragment tiles per wave and k-step
  static_assert(!W8 || KS == 2, "FP8 weights are packed two k-steps per 16-byte unit");
  constexpr int A_TILE = W8 ? 1024 : KS * 1024;  // bytes of one 16-feature tile per stage
  constexpr int A_BYTES = (BN / 16) * A_TILE;
  constexpr int XROW = KS * 64;                  // bytes of an x row per stage
  constexpr int X_BYTES = BM * XROW;
  constexpr int STAGE = A_BYTES + X_BYTES;
  constexpr int PA = A_BYTES / 1024, PX = X_BYTES / 1024;  // 1 KiB LDS-DMA pieces per stage
  static_assert(PA % NW == 0 && PX % NW == 0, "every wave issues the same number of LDS-DMA pieces (exact vmcnt counts)");
  constexpr int PA_W = PA / NW, PX_W = PX / NW;
  constexpr int G = PA_W + PX_W;                 // LDS-DMA instructions per wave and stage
  constexpr int A_LINES = A_BYTES / 128;         // cache lines of a stage: W, then one per x row
  constexpr int PF_OPS = PFD > 0 ? (A_LINES + BM + NW * 64 - 1) / (NW * 64) : 0;
  constexpr int OPS = G + PF_OPS;                // vector-memory operations per wave and step
  constexpr int SCRATCH = NS * STAGE;            // [NW][256] landing area of the prefetch DMAs
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
#pragma unroll
  for (int i = 0; i < PA_W; ++i) {
    const int pc = wave + NW * i;
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
  // L2 prefetch: lane slot g of a stage = cache line g of its weights, or x row g - A_LINES
  const unsigned char* pfsrc[PF_OPS > 0 ? PF_OPS : 1];
  long long pfstep[PF_OPS > 0 ? PF_OPS : 1];
#pragma unroll
  for (int i = 0; i < PF_OPS; ++i) {
    int g = (i * NW + wave) * 64 + lane;
    g = g < A_LINES + BM ? g : A_LINES + BM - 1;
    if (g < A_LINES) {
      const int ti = g / (A_TILE / 128), ln = g % (A_TILE / 128);
      pfsrc[i] = p.Wp + (((long long)(n0 >> 4) + ti) * nst + s0) * A_TILE + ln * 128;
      pfstep[i] = A_TILE;
    } else {
      int row = m0 + g - A_LINES;
      row = row < p.M ? row : p.M - 1;
      pfsrc[i] = reinterpret_cast<const unsigned char*>(p.x + (long long)row * p.ldx + (long long)s0 * (32 * KS));
      pfstep[i] = XROW;
    }
  }

  auto stage = [&](int buf, int st) {  // st = stage index relative to s0
    unsigned char* base = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < PA_W; ++i) dma16(asrc[i] + (long long)st * A_TILE, base + (wave + NW * i) * 1024);
#pragma unroll
    for (int i = 0; i < PX_W; ++i) dma16(xsrc[i] + (long long)st * XROW, base + A_BYTES + (wave + NW * i) * 1024);
  };
  auto prefetch = [&](int st) {
    const int stc = st < n ? st : n - 1;  // past the end of this split: touch its last stage again (same count of operations)
#pragma unroll
    for (int i = 0; i < PF_OPS; ++i)
      __builtin_amdgcn_global_load_lds((glb_void*)(pfsrc[i] + stc * pfstep[i]), (lds_void*)(smem + SCRATCH + wave * 256), 4, 0, 0);
  };

  frag_cd acc[FA][FB];
#pragma unroll
  for (int i = 0; i < FA; ++i)
#pragma unroll
    for (int j = 0; j < FB; ++j) acc[i][j] = (frag_cd){0.f, 0.f, 0.f, 0.f};

  // lane-constant LDS read offsets
  const int a_off = wn * (FA * A_TILE) + lane * 16;
  int x_off[KS];
  if constexpr (KS == 2) {
    x_off[0] = A_BYTES + (wm * TM + r) * 128 + (((0 + q) ^ (r >> 1)) << 4);
    x_off[1] = A_BYTES + (wm * TM + r) * 128 + (((4 + q) ^ (r >> 1)) << 4);
  } else {
    x_off[0] = A_BYTES + (wm * TM + r) * 64 + ((q ^ ((4 - (r >> 2)) & 3)) << 4);
  }

  // ---- prologue: NS - 1 stages in flight (stages past the end are re-reads of the last one
  // into their own buffers: the operation count per step stays exact)
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) {
    stage(s, s < n ? s : n - 1);
    prefetch(s + NS - 1 + PFD);
  }
  unsigned long long seg[5] = {0, 0, 0, 0, 0}, t0 = 0, t1 = 0;
  for (int i = 0; i < n; ++i) {
    const int buf = i % NS;
    if constexpr (STAMP) t0 = stamp();
    // Stage i has landed once all but the (NS - 2) younger steps' operations of this wave are
    // done; the barrier then orders every wave's pieces before anybody's reads, and says that
    // every wave has finished reading the buffer that is re-filled next.
    if constexpr ((NS - 2) * OPS == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * OPS) : "memory");
    if constexpr (STAMP) { t1 = stamp(); seg[0] += t1 - t0; t0 = t1; }
    __builtin_amdgcn_s_barrier();
    if constexpr (STAMP) { t1 = stamp(); seg[1] += t1 - t0; t0 = t1; }
    {
      const int st = i + NS - 1;
      stage(st % NS, st < n ? st : n - 1);
      prefetch(st + PFD);
    }
    if constexpr (STAMP) { t1 = stamp(); seg[2] += t1 - t0; t0 = t1; }
    const unsigned char* sb = smem + buf * STAGE;
    // all fragment reads of the stage are issued up front (registers permitting), so that the
    // MFMAs wait for LDS latency once per stage, not once per group of four
    frag_ab a[KS][FA], b[KS][FB];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if constexpr (W8) {
        if (s == 0) {
#pragma unroll
          for (int ii = 0; ii < FA; ++ii) {
            const raw16 v = *reinterpret_cast<const raw16*>(sb + a_off + ii * 1024);
            a[0][ii] = fp8x8_to_bf16(v[0], v[1]);
            a[KS - 1][ii] = fp8x8_to_bf16(v[2], v[3]);
          }
        }
      } else {
#pragma unroll
        for (int ii = 0; ii < FA; ++ii) a[s][ii] = *reinterpret_cast<const frag_ab*>(sb + a_off + (ii * KS + s) * 1024);
      }
#pragma unroll
      for (int j = 0; j < FB; ++j) b[s][j] = *reinterpret_cast<const frag_ab*>(sb + x_off[s] + j * (16 * XROW));
    }
    if constexpr (STAMP) { t1 = stamp(); seg[3] += t1 - t0; t0 = t1; }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ii = 0; ii < FA; ++ii)
#pragma unroll
        for (int j = 0; j < FB; ++j) acc[ii][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s][ii], b[s][j], acc[ii][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
    if constexpr (STAMP) { t1 = stamp(); seg[4] += t1 - t0; }
  }
  if constexpr (STAMP) {
    if (lane == 0 && p.dbg != nullptr) {
      unsigned long long* d = p.dbg + ((long long)blockIdx.x * NW + wave) * 8;
      for (int k = 0; k < 5; ++k) d[k] = seg[k];
      d[5] = n;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the re-read stages of the tail

  // ---- epilogue: acc[i][j][e] = y[token m0 + wm TM + 16 j + r][feature n0 + wn TN + 16 i + 4 q + e]
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
  }
#pragma unroll
  for (int i = 0; i < FA; ++i) {
    const int f = feat0 + 16 * i;
    frag_cd ws = (frag_cd){1.f, 1.f, 1.f, 1.f};
    if constexpr (W8) ws = *reinterpret_cast<const frag_cd*>(p.w_scale + f);
#pragma unroll
    for (int j = 0; j < FB; ++j) {
      const int t = tok0 + 16 * j;
      frag_cd v = acc[i][j];
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
        const int fi = ((n0 + wn * TN + 16 * i) >> 1) + 4 * q;
        if (q < 2 && t < p.M) *reinterpret_cast<us4*>(reinterpret_cast<unsigned short*>(p.out) + (long long)t * p.ldy + fi) = o;
      }
    }
  }
}

template <int BM, int BN, int WM, int WN, int KS, int NS, int PFD, int EPI, bool W8, bool STAMP = false>
int launch(const Params& p, hipStream_t stream) {
  constexpr int A_BYTES = (BN / 16) * (W8 ? 1024 : KS * 1024);
  constexpr int LDS = NS * (A_BYTES + BM * KS * 64) + (PFD > 0 ? WM * WN * 256 : 0);
  static_assert(LDS <= 160 * 1024, "LDS budget of a CU");
  auto kern = wide_gemm_kernel<BM, BN, WM, WN, KS, NS, PFD, EPI, W8, STAMP>;
  static bool attr_set = false;  // per instantiation; LDS above the 64 KiB default has to be requested once
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return -2;
    attr_set = true;
  }
  const int grid = p.n_mtiles * p.n_ntiles * p.splits;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(WM * WN * 64), LDS, stream, p);
  return (int)hipGetLastError();
}

template <int BM, int BN, int WM, int WN, int KS, int NS, int PFD, bool W8>
int dispatch_epi(int epi, const Params& p, hipStream_t s) {
  switch (epi) {
    case EPI_BF16: return launch<BM, BN, WM, WN, KS, NS, PFD, EPI_BF16, W8>(p, s);
    case EPI_SLAB: return launch<BM, BN, WM, WN, KS, NS, PFD, EPI_SLAB, W8>(p, s);
    case EPI_SWIGLU: return launch<BM, BN, WM, WN, KS, NS, PFD, EPI_SWIGLU, W8>(p, s);
  }
  return -1;
}

// variant = pipeline shape of a tile: 0 = two 64-wide stages, 1 = the same + L2 prefetch,
// 2 = deep ring (as many stages as the LDS holds), 3 = deep ring + L2 prefetch
template <bool W8>
int dispatch_tile(int bm, int bn, int variant, int epi, const Params& p, hipStream_t s) {
  const int key = bm * 1000 + bn;
  switch (key * 10 + variant) {
    // 256 x 256, 8 waves (wave: 64 tokens x 128 features)
    case 2562560: return dispatch_epi<256, 256, 4, 2, 2, 2, 0, W8>(epi, p, s);
    case 2562561: return dispatch_epi<256, 256, 4, 2, 2, 2, 3, W8>(epi, p, s);
    case 2562565: return dispatch_epi<256, 256, 2, 2, 2, 2, 0, W8>(epi, p, s);  // 4 waves (wave: 128 x 128, one per SIMD)
    // 256 x 128, 8 waves (wave: 64 x 64)
    case 2561280: return dispatch_epi<256, 128, 4, 2, 2, 2, 0, W8>(epi, p, s);
    case 2561281: return dispatch_epi<256, 128, 4, 2, 2, 2, 3, W8>(epi, p, s);
    case 2561282: return dispatch_epi<256, 128, 4, 2, 2, 3, 0, W8>(epi, p, s);
    case 2561283: return dispatch_epi<256, 128, 4, 2, 2, 3, 2, W8>(epi, p, s);
    // 128 x 256, 8 waves (wave: 64 x 64)
    case 1282560: return dispatch_epi<128, 256, 2, 4, 2, 2, 0, W8>(epi, p, s);
    case 1282562: return dispatch_epi<128, 256, 2, 4, 2, 3, 0, W8>(epi, p, s);
    case 1282563: return dispatch_epi<128, 256, 2, 4, 2, 3, 2, W8>(epi, p, s);
    // 128 x 128, 4 waves (wave: 64 x 64); two workgroups per CU with two stages
    case 1281280: return dispatch_epi<128, 128, 2, 2, 2, 2, 0, W8>(epi, p, s);
    case 1281282: return dispatch_epi<128, 128, 2, 2, 2, 4, 0, W8>(epi, p, s);
    case 1281283: return dispatch_epi<128, 128, 2, 2, 2, 4, 2, W8>(epi, p, s);
  }
  if constexpr (!W8) {
    // diagnostic builds (segment stamps, p.dbg): variants 90 + v of the 256 x 256 tile, bf16 output
    if (key == 256256 && epi == EPI_BF16 && p.dbg != nullptr) {
      if (variant == 90) return launch<256, 256, 4, 2, 2, 2, 0, EPI_BF16, false, true>(p, s);
      if (variant == 92) return launch<256, 256, 4, 2, 1, 4, 0, EPI_BF16, false, true>(p, s);
      if (variant == 95) return launch<256, 256, 2, 2, 2, 2, 0, EPI_BF16, false, true>(p, s);
    }
  }
  if constexpr (!W8) {  // bf16 weights only: 256 x 64 (an FP8 stage of it is 4 pieces for 8 waves) and the 32-wide stages
    switch (key * 10 + variant) {
      // 256 x 64, 8 waves (wave: 32 x 64)
      case 2560640: return dispatch_epi<256, 64, 8, 1, 2, 2, 0, false>(epi, p, s);
      case 2560642: return dispatch_epi<256, 64, 8, 1, 2, 3, 0, false>(epi, p, s);
      case 2560643: return dispatch_epi<256, 64, 8, 1, 2, 3, 2, false>(epi, p, s);
      // 256 x 256 with a 4 / 5-deep ring of 32-wide stages
      case 2562562: return dispatch_epi<256, 256, 4, 2, 1, 4, 0, false>(epi, p, s);
      case 2562563: return dispatch_epi<256, 256, 4, 2, 1, 4, 4, false>(epi, p, s);
      case 2562564: return dispatch_epi<256, 256, 4, 2, 1, 5, 0, false>(epi, p, s);
      case 2562566: return dispatch_epi<256, 256, 2, 2, 1, 4, 0, false>(epi, p, s);  // 4 waves, 4-deep ring
    }
  }
  return -1;
}

}  // namespace

extern "C" {

// k-steps (of 32) per pipeline stage of a (tile, variant): K splits are handed out in stages
int kv_wide_gemm_stage_ksteps(int bm, int bn, int variant) {
  variant %= 90;
  return (bm == 256 && bn == 256 && (variant == 2 || variant == 3 || variant == 4 || variant == 6)) ? 1 : 2;
}

// epi: 0 = bf16 y[M, N], 1 = fp32 slabs out[splits][M][N], 2 = SwiGLU bf16 y[M, N/2].
// (bm, bn): workgroup tile, one of (256,256) (256,128) (256,64) (128,128) (128,256); variant: see
// dispatch_tile.  N % bn == 0, K % 64 == 0; with S = K / (32 * stage k-steps) stages in all,
// splits * stages_per_split >= S > (splits - 1) * stages_per_split.
// w_scale != nullptr: Wp holds e4m3 bytes packed by kv_pack_weight_fp8, w_scale[N] the row scales.
int kv_wide_gemm(const void* Wp, const void* x, void* out, int M, int N, int K, long long ldx, long long ldy, int epi, int bm, int bn, int variant, int splits,
                 int stages_per_split, long long slab_stride, const float* row_ss, int ss_parts, float eps, const float* w_scale, int m_fastest,
                 hipStream_t stream, unsigned long long* dbg) {
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
  p.n_mtiles = (M + bm - 1) / bm; p.n_ntiles = N / bn; p.m_fastest = m_fastest; p.dbg = dbg;
  return w_scale ? dispatch_tile<true>(bm, bn, variant, epi, p, stream) : dispatch_tile<false>(bm, bn, variant, epi, p, stream);
}

}  // extern "C"

// Memory-bound decoder kernels of the native engine for gfx950 (MI355X):
// fused residual-add + RMSNorm, RoPE + paged-KV write, ragged paged attention, argmax.
//
// All activations are token-major bf16 ([T, hidden]); T mixes decode tokens and
// prefill-chunk tokens of many sequences.  Loads/stores are 16 bytes per lane
// (8 bf16), math is fp32, wavefronts are 64 lanes, workgroups are 256 threads.

#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>
#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(8))) unsigned short bf16x8;  // 16 bytes

__device__ inline float bf2f(unsigned short b) { return __uint_as_float(((unsigned int)b) << 16); }
__device__ inline unsigned short f2bf(float f) {  // round-to-nearest-even, NaN kept quiet
  unsigned int u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
// value of lane dpp_ctrl(lane) within the lane's row of 16 (data-parallel primitive: one VALU op)
template <int CTRL>
__device__ inline float dpp_f32(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
// packed bf16 pairs, the operand type of v_dot2_f32_bf16 and the result of the packed FP8 conversions
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
struct bf16x2x4 { bf16x2 p[4]; };
struct bf16x2x2 { bf16x2 p[2]; };
typedef __attribute__((ext_vector_type(4))) unsigned short bf16x4;  // 8 bytes

// ---- paged KV cache elements.  KV8 = false: bf16.  KV8 = true: OCP e4m3 bytes holding x / scale
// (one scale per kv head and per K / V); the bytes are widened to bf16 in registers (exact) with
// the packed gfx950 conversion, and the scale is applied once per score / output instead of once
// per element.  Offsets are in ELEMENTS in both layouts.
template <bool KV8>
__device__ inline bf16x8 kv_load8(const void* base, long long off) {
  if constexpr (!KV8) {
    return *reinterpret_cast<const bf16x8*>(reinterpret_cast<const unsigned short*>(base) + off);
  } else {
    const uint2 r = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned char*>(base) + off);
    bf16x2x4 o;
    o.p[0] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(r.x, 1.0f, false);
    o.p[1] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(r.x, 1.0f, true);
    o.p[2] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(r.y, 1.0f, false);
    o.p[3] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(r.y, 1.0f, true);
    return __builtin_bit_cast(bf16x8, o);
  }
}
// the same 8 elements addressed as a wave-uniform base (scalar registers) plus a 32-bit byte
// offset per lane: the load takes the base from SGPRs and needs no 64-bit vector arithmetic
template <bool KV8>
__device__ inline bf16x8 kv_load8_at(const char* base, unsigned byte_off) {
  if constexpr (!KV8) {
    return *reinterpret_cast<const bf16x8*>(base + byte_off);
  } else {
    const uint2 r = *reinterpret_cast<const uint2*>(base + byte_off);
    bf16x2x4 o;
    o.p[0] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(r.x, 1.0f, false);
    o.p[1] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(r.x, 1.0f, true);
    o.p[2] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(r.y, 1.0f, false);
    o.p[3] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(r.y, 1.0f, true);
    return __builtin_bit_cast(bf16x8, o);
  }
}
template <bool KV8>
__device__ inline bf16x4 kv_load4(const void* base, long long off) {
  if constexpr (!KV8) {
    return *reinterpret_cast<const bf16x4*>(reinterpret_cast<const unsigned short*>(base) + off);
  } else {
    const unsigned int r = *reinterpret_cast<const unsigned int*>(reinterpret_cast<const unsigned char*>(base) + off);
    bf16x2x2 o;
    o.p[0] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(r, 1.0f, false);
    o.p[1] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(r, 1.0f, true);
    return __builtin_bit_cast(bf16x4, o);
  }
}
// 8 e4m3 bytes (two dwords) -> 8 bf16, exact
__device__ inline bf16x8 fp8x8_widen(unsigned int lo, unsigned int hi) {
  bf16x2x4 o;
  o.p[0] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(lo, 1.0f, false);
  o.p[1] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(lo, 1.0f, true);
  o.p[2] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(hi, 1.0f, false);
  o.p[3] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(hi, 1.0f, true);
  return __builtin_bit_cast(bf16x8, o);
}
// One K/V block (16 tokens) of a wave of the fused decode attention kernel, as the lanes hold it
// between the load and the use.  Lane roles (head_dim 128): K — lane (key = lane >> 2, chunk =
// lane & 3) owns 32 consecutive dims of a key; V — lane (tg = lane >> 4, dv = lane & 15) owns dims
// [8 dv, 8 dv + 8) of the tokens tg + 4 i, i = 0..3.
//   bf16 cache: four 16-byte loads each, kept as they are.
//   e4m3 cache: the stage keeps the RAW bytes of four 16-byte loads (half the registers of the
//   widened form, and half the load instructions of reading 8 bytes at a time — at long contexts
//   the kernel is bound by the loads it issues, so an FP8 cache read 8 bytes at a time halved
//   the memory held but not the time).  K: the lane's 32 bytes are two loads.  V: the lanes dv
//   and dv ^ 1 (neighbours in a quad) need the two halves of the same 16 bytes of every token, so
//   each loads 16 bytes of HALF the tokens (even lane: i = 0, 2; odd lane: i = 1, 3) and, when
//   the block is used, keeps its half and hands the other to its neighbour with one DPP quad
//   permute per dword.  Widening and exchange happen in unpack(), i.e. when the block becomes
//   the current one: nothing waits for a load at the point where it is issued.
template <bool KV8>
struct KVStage;
template <>
struct KVStage<false> {
  bf16x8 k[4], v[4];
  __device__ inline void load(const void* k_cache, const void* v_cache, long long blk, int key, int chunk, int tg, int dv) {
    constexpr int HD = 128;
#pragma unroll
    for (int i = 0; i < 4; ++i) k[i] = kv_load8<false>(k_cache, blk + key * HD + chunk * 32 + i * 8);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = kv_load8<false>(v_cache, blk + dv * 8 + (tg + 4 * i) * HD);
  }
  __device__ inline void unpack(bf16x8 (&kk)[4], bf16x8 (&vv)[4], int dv) const {
    (void)dv;
#pragma unroll
    for (int i = 0; i < 4; ++i) { kk[i] = k[i]; vv[i] = v[i]; }
  }
};
template <>
struct KVStage<true> {
  uint4 k0, k1, v0, v1;
  __device__ inline void load(const void* k_cache, const void* v_cache, long long blk, int key, int chunk, int tg, int dv) {
    constexpr int HD = 128;
    const unsigned char* kp = reinterpret_cast<const unsigned char*>(k_cache) + blk + key * HD + chunk * 32;
    k0 = *reinterpret_cast<const uint4*>(kp);
    k1 = *reinterpret_cast<const uint4*>(kp + 16);
    const unsigned char* vp = reinterpret_cast<const unsigned char*>(v_cache) + blk + (dv >> 1) * 16 + (tg + 4 * (dv & 1)) * HD;
    v0 = *reinterpret_cast<const uint4*>(vp);           // token tg + 4 (0 + odd), dims [16 (dv >> 1), + 16)
    v1 = *reinterpret_cast<const uint4*>(vp + 8 * HD);  // token tg + 4 (2 + odd)
  }
  // the same loads from wave-uniform block bases (scalar registers) plus per-lane byte offsets that do not change
  // from block to block: koff = key * 128 + chunk * 32, voff = (dv >> 1) * 16 + (tg + 4 (dv & 1)) * 128
  __device__ inline void load_at(const char* kbase, const char* vbase, unsigned koff, unsigned voff) {
    k0 = *reinterpret_cast<const uint4*>(kbase + koff);
    k1 = *reinterpret_cast<const uint4*>(kbase + koff + 16);
    v0 = *reinterpret_cast<const uint4*>(vbase + voff);
    v1 = *reinterpret_cast<const uint4*>(vbase + voff + 8 * 128);
  }
  static __device__ inline unsigned k_off(int key, int chunk) { return (unsigned)(key * 128 + chunk * 32); }
  static __device__ inline unsigned v_off(int tg, int dv) { return (unsigned)((dv >> 1) * 16 + (tg + 4 * (dv & 1)) * 128); }
  __device__ inline void unpack(bf16x8 (&kk)[4], bf16x8 (&vv)[4], int dv) const {
    kk[0] = fp8x8_widen(k0.x, k0.y); kk[1] = fp8x8_widen(k0.z, k0.w);
    kk[2] = fp8x8_widen(k1.x, k1.y); kk[3] = fp8x8_widen(k1.z, k1.w);
    const bool odd = dv & 1;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const uint4 r = j ? v1 : v0;
      // my token is i = 2 j + odd: I keep the half with my dims and give my neighbour the other
      const unsigned int mine0 = odd ? r.z : r.x, mine1 = odd ? r.w : r.y;
      const unsigned int give0 = odd ? r.x : r.z, give1 = odd ? r.y : r.w;
      const unsigned int got0 = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)give0, 0xB1, 0xF, 0xF, true);
      const unsigned int got1 = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)give1, 0xB1, 0xF, 0xF, true);
      const bf16x8 own = fp8x8_widen(mine0, mine1), other = fp8x8_widen(got0, got1);
      vv[2 * j] = odd ? other : own;
      vv[2 * j + 1] = odd ? own : other;
    }
  }
  // K as above; V as fp32 straight from the bytes (packed conversion, two values per instruction) and in the
  // lane's OWN order: vf[i] is token tg + 4 (i ^ (dv & 1)) — the lane's loaded token first, its neighbour's second —
  // so nothing is selected by lane parity after the exchange (the consumer reads P at that token instead).
  __device__ inline void unpack_f32(bf16x8 (&kk)[4], float (&vf)[4][8], int dv) const {
    kk[0] = fp8x8_widen(k0.x, k0.y); kk[1] = fp8x8_widen(k0.z, k0.w);
    kk[2] = fp8x8_widen(k1.x, k1.y); kk[3] = fp8x8_widen(k1.z, k1.w);
    const bool odd = dv & 1;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const uint4 r = j ? v1 : v0;
      const unsigned int w[4] = {odd ? r.z : r.x, odd ? r.w : r.y,  // my dims of my token
                                 (unsigned int)__builtin_amdgcn_update_dpp(0, (int)(odd ? r.x : r.z), 0xB1, 0xF, 0xF, true),
                                 (unsigned int)__builtin_amdgcn_update_dpp(0, (int)(odd ? r.y : r.w), 0xB1, 0xF, 0xF, true)};  // my dims of the neighbour's token
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        typedef __attribute__((ext_vector_type(2))) float f32x2;
        const f32x2 lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)w[d], false), hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)w[d], true);
        float* o = vf[2 * j + (d >> 1)] + 4 * (d & 1);
        o[0] = lo[0]; o[1] = lo[1]; o[2] = hi[0]; o[3] = hi[1];
      }
    }
  }
};
template <bool KV8, int N>
__device__ inline auto kv_loadn(const void* base, long long off) {
  if constexpr (N == 8) return kv_load8<KV8>(base, off);
  else return kv_load4<KV8>(base, off);
}
// A K/V block of a wave of paged_attention_kernel between load and use (same lane roles as KVStage).
// Generic form: widened to bf16 where it is loaded.  Head width 128 on an e4m3 cache: KVStage<true> — raw
// 16-byte loads, widened when the block becomes the current one (half the load instructions and half the
// registers per block in flight; with 8-byte loads the kernel moved 3.3 TB/s of e4m3 against 5.8 of bf16).
template <int HD, bool KV8>
struct PAStage {
  static constexpr int DQ = HD / 4, NV = DQ / 8, DP = HD / 16;
  typedef __attribute__((ext_vector_type(DP))) unsigned short vrow_t;
  bf16x8 k[NV];
  vrow_t v[4];
  __device__ inline void load(const void* k_cache, const void* v_cache, long long blk, int key, int chunk, int tg, int dv) {
#pragma unroll
    for (int i = 0; i < NV; ++i) k[i] = kv_load8<KV8>(k_cache, blk + key * HD + chunk * DQ + i * 8);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = kv_loadn<KV8, DP>(v_cache, blk + dv * DP + (tg + 4 * i) * HD);
  }
  static constexpr int ESZ = KV8 ? 1 : 2;
  __device__ inline void load_at(const char* kbase, const char* vbase, unsigned koff, unsigned voff) {
#pragma unroll
    for (int i = 0; i < NV; ++i) k[i] = kv_load8_at<KV8>(kbase, koff + i * 8 * ESZ);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if constexpr (DP == 8) v[i] = kv_load8_at<KV8>(vbase, voff + 4 * i * HD * ESZ);
      else v[i] = kv_load4<KV8>(vbase, (long long)((voff + 4 * i * HD * ESZ) / ESZ));
    }
  }
  static __device__ inline unsigned k_off(int key, int chunk) { return (unsigned)((key * HD + chunk * DQ) * ESZ); }
  static __device__ inline unsigned v_off(int tg, int dv) { return (unsigned)((dv * DP + tg * HD) * ESZ); }
  __device__ inline void unpack(bf16x8 (&kk)[NV], vrow_t (&vv)[4], int dv) const {
    (void)dv;
#pragma unroll
    for (int i = 0; i < NV; ++i) kk[i] = k[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) vv[i] = v[i];
  }
};
template <>
struct PAStage<128, true> : KVStage<true> {};
// 8 values -> e4m3 bytes of v * inv_scale, clamped to the format's range (+-448)
__device__ inline uint2 fp8_pack8(bf16x8 v, float inv_scale) {
  float f[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = fminf(fmaxf(__uint_as_float(((unsigned int)v[j]) << 16) * inv_scale, -448.f), 448.f);
  int lo = 0, hi = 0;
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], lo, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], hi, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
  return make_uint2((unsigned int)lo, (unsigned int)hi);
}
// store 8 elements; returns what a later kv_load8 of them yields (bf16: v itself; FP8: the rounded value in stored units)
template <bool KV8>
__device__ inline bf16x8 kv_store8(void* base, long long off, bf16x8 v, float inv_scale, bool do_store = true) {
  if constexpr (!KV8) {
    if (do_store) *reinterpret_cast<bf16x8*>(reinterpret_cast<unsigned short*>(base) + off) = v;
    return v;
  } else {
    const uint2 q = fp8_pack8(v, inv_scale);
    if (do_store) *reinterpret_cast<uint2*>(reinterpret_cast<unsigned char*>(base) + off) = q;
    bf16x2x4 o;
    o.p[0] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(q.x, 1.0f, false);
    o.p[1] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(q.x, 1.0f, true);
    o.p[2] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(q.y, 1.0f, false);
    o.p[3] = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(q.y, 1.0f, true);
    return __builtin_bit_cast(bf16x8, o);
  }
}

__device__ inline float wave_sum(float v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Sum of the `parts` partial sums of squares of one token row, by a whole wave (lane-strided loads
// + butterfly; every lane gets the total).  Rows carry 8 parts from the residual kernel and
// hidden/16 from the row-parallel projections that update the residual in their own epilogue.
__device__ inline float row_ss_total(const float* __restrict__ p, int parts, int lane) {
  float v = 0.f;
  for (int i = lane; i < parts; i += 64) v += p[i];
  return wave_sum(v);
}

// ---------------------------------------------------------------------------------
// Weight prefetch beside a latency-bound kernel.  The residual / RMSNorm-statistics kernel and
// the decode attention kernel at short contexts are chains of dependent loads: HBM idles while
// they run, and the weight-streaming GEMM that follows them is bound by HBM.  Such a launch can
// carry extra workgroups (after its own in dispatch order) that do nothing but read the START of
// the byte ranges the next GEMM's workgroups will stream, so the GEMM finds them in cache.
// Which cache decides the gain: every XCD has an L2 of its own, and the dispatcher deals
// workgroups to the 8 XCDs round-robin by linear id, in this launch and in the GEMM's alike — a
// prefetch workgroup whose linear id is x (mod 8) therefore reads only what the GEMM workgroups
// with id x (mod 8) will read, and the bytes land in the L2 those will look in (4 MiB per XCD:
// the caller sizes `kib` so that the prefix fits; what L2 drops is still in the Infinity Cache).
// Were the placement different, the result would be the same and only the speed-up smaller.
//
// The plan mirrors the grid of stream_gemm3_kernel (csrc/stream_gemm.hip): workgroup (bx, by) of
// a (gx, gy) grid streams, for each of its nw waves w, the bytes
//     base + (bx * nw + w) * tile_bytes + by * split_bytes + [0, split_bytes)
// of the packed weights; the first `kib` KiB of each such range are fetched here.
struct PrefetchPlan {
  const unsigned char* base;  // packed weights of the next GEMM (nullptr: no prefetch workgroups were launched)
  int gx, gy, nw, kib;
  long long tile_bytes, split_bytes;
  int wgs;                    // prefetch workgroups in this launch, a multiple of 8
  unsigned int* sink;         // never written in practice; keeps the loads alive
};

// j0: index of this workgroup among the launch's prefetch workgroups; lin: its linear id in the
// whole grid; a wave takes whole ranges (one per trip, all of a range's loads in flight at once).
__device__ inline void prefetch_role(const PrefetchPlan& pf, int j0, int lin, int wave, int nwaves, int lane) {
  if (pf.base == nullptr || j0 >= pf.wgs) return;
  const int x = lin & 7;
  const int workers = (pf.wgs >> 3) * nwaves;  // prefetch waves that share this XCD
  const int worker = (j0 >> 3) * nwaves + wave;
  const int G = pf.gx * pf.gy;
  const int ranges = ((G - x + 7) >> 3) * pf.nw;  // GEMM workgroups x, x + 8, ... < G, nw ranges each
  // The data is never used, so nothing waits for it: every load targets the same scratch
  // registers (returns are in order, the last one wins) and the wave issues all its ranges back
  // to back — one memory round trip per wave instead of one per range (measured: with a wait per
  // range, a range cost 2-3 us whatever its size, and a wave's ranges lie far apart).  The
  // hardware's own limit on outstanding loads per wave is the only throttle.
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  u32x4 scratch = {0u, 0u, 0u, 0u};
  for (int rg = worker; rg < ranges; rg += workers) {
    const int ii = rg / pf.nw, w = rg - ii * pf.nw;
    const int g = 8 * ii + x;
    const int by = g / pf.gx, bx = g - by * pf.gx;
    const uint4* p = reinterpret_cast<const uint4*>(pf.base + ((long long)bx * pf.nw + w) * pf.tile_bytes + (long long)by * pf.split_bytes) + lane;
    for (int k = 0; k < pf.kib; ++k) asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(scratch) : "v"(p + k * 64) : "memory");
  }
  // `scratch` is an in/out operand of every load and of this wait, so it is live from the first
  // load to here and the compiler keeps other values out of the registers the loads land in.
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(scratch) : : "memory");  // the wave ends with nothing in flight
  (void)pf.sink;
}

// ---------------------------------------------------------------------------------
// residual' = bf16(residual + x);  y = bf16(rmsnorm(residual') * w).  One workgroup per
// token row; H must be a multiple of 8.  x/residual/y may not alias each other except
// that residual is updated in place.
__global__ __launch_bounds__(256) void add_rmsnorm_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ residual,
                                                          const unsigned short* __restrict__ w, unsigned short* __restrict__ y, int H, float eps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* red = reinterpret_cast<float*>(smem);               // 4 floats
  unsigned short* row = reinterpret_cast<unsigned short*>(smem + 16);  // H bf16: the summed row
  const int t = blockIdx.x, tid = threadIdx.x;
  const bf16x8* xv = reinterpret_cast<const bf16x8*>(x + (long long)t * H);
  bf16x8* rv = reinterpret_cast<bf16x8*>(residual + (long long)t * H);
  bf16x8* rowv = reinterpret_cast<bf16x8*>(row);
  const int nv = H >> 3;
  float ss = 0.f;
  for (int i = tid; i < nv; i += 256) {
    bf16x8 a = xv[i], b = rv[i], c;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      unsigned short s = f2bf(bf2f(a[j]) + bf2f(b[j]));
      c[j] = s;
      float f = bf2f(s);
      ss += f * f;
    }
    rv[i] = c;
    rowv[i] = c;
  }
  ss = wave_sum(ss);
  if ((tid & 63) == 0) red[tid >> 6] = ss;
  __syncthreads();
  const float inv = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)H + eps);
  const bf16x8* wv = reinterpret_cast<const bf16x8*>(w);
  bf16x8* yv = reinterpret_cast<bf16x8*>(y + (long long)t * H);
  for (int i = tid; i < nv; i += 256) {
    bf16x8 c = rowv[i], g = wv[i], o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = f2bf(bf2f(c[j]) * inv * bf2f(g[j]));
    yv[i] = o;
  }
}

// Same, with x given as n_slabs fp32 partial sums [n_slabs][T][H] written by the split-K
// weight-streaming GEMM: the slabs are summed here, in the consumer's prologue.
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ __launch_bounds__(256) void add_rmsnorm_slab_kernel(const float* __restrict__ slabs, int n_slabs, long long slab_stride,
                                                               unsigned short* __restrict__ residual, const unsigned short* __restrict__ w,
                                                               unsigned short* __restrict__ y, int H, float eps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* red = reinterpret_cast<float*>(smem);
  unsigned short* row = reinterpret_cast<unsigned short*>(smem + 16);
  const int t = blockIdx.x, tid = threadIdx.x;
  bf16x8* rv = reinterpret_cast<bf16x8*>(residual + (long long)t * H);
  bf16x8* rowv = reinterpret_cast<bf16x8*>(row);
  const int nv = H >> 3;
  float ss = 0.f;
  for (int i = tid; i < nv; i += 256) {
    float a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = 0.f;
    for (int sidx = 0; sidx < n_slabs; ++sidx) {
      const f32x4* p = reinterpret_cast<const f32x4*>(slabs + (long long)sidx * slab_stride + (long long)t * H + i * 8);
      f32x4 lo = p[0], hi = p[1];
#pragma unroll
      for (int j = 0; j < 4; ++j) { a[j] += lo[j]; a[4 + j] += hi[j]; }
    }
    bf16x8 b = rv[i], c;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      // the GEMM output is rounded to bf16 first, like the unsplit kernel's store
      unsigned short s = f2bf(bf2f(f2bf(a[j])) + bf2f(b[j]));
      c[j] = s;
      float f = bf2f(s);
      ss += f * f;
    }
    rv[i] = c;
    rowv[i] = c;
  }
  ss = wave_sum(ss);
  if ((tid & 63) == 0) red[tid >> 6] = ss;
  __syncthreads();
  const float inv = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)H + eps);
  const bf16x8* wv = reinterpret_cast<const bf16x8*>(w);
  bf16x8* yv = reinterpret_cast<bf16x8*>(y + (long long)t * H);
  for (int i = tid; i < nv; i += 256) {
    bf16x8 c = rowv[i], g = wv[i], o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = f2bf(bf2f(c[j]) * inv * bf2f(g[j]));
    yv[i] = o;
  }
}

// ---------------------------------------------------------------------------------
// Parallel residual update for the norm-folded pipeline.  RMSNorm's gain is folded into
// the next projection's weights and its per-row scale 1/rms is applied in that
// projection's epilogue, so what is left of "residual add + RMSNorm" is
//     residual' = bf16(residual + bf16(sum of slabs))      (or + x for a bf16 input)
//     ss_part[t][p] = sum of residual'^2 over the p-th 512-column segment of row t.
// grid = (ceil(H/512), T + prefetch rows), one wavefront per workgroup, 8 columns per lane: a
// [32, 4096] activation is 256 independent waves instead of 32 serial row loops.
template <bool SLAB>
__global__ __launch_bounds__(64) void residual_reduce_kernel(const float* __restrict__ slabs, int n_slabs, long long slab_stride,
                                                             const unsigned short* __restrict__ x, unsigned short* __restrict__ residual, int has_residual,
                                                             float* __restrict__ ss_part, int H, int T, PrefetchPlan pf) {
  const int t = blockIdx.y, p = blockIdx.x, lane = threadIdx.x;
  if (t >= T) {  // rows past the last token: weight prefetch for the next GEMM (see prefetch_role)
    prefetch_role(pf, (t - T) * (int)gridDim.x + p, t * (int)gridDim.x + p, 0, 1, lane);
    return;
  }
  const int col = (p * 64 + lane) * 8;
  float ss = 0.f;
  if (col < H) {
    float a[8];
    const long long off = (long long)t * H + col;
    if (SLAB) {
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = 0.f;
      // all slab loads of the lane are issued before the first add (one memory round trip
      // instead of n_slabs / 4); the sum runs in slab order
      constexpr int MAX_S = 16;
      f32x4 lo[MAX_S], hi[MAX_S];
#pragma unroll
      for (int sidx = 0; sidx < MAX_S; ++sidx)
        if (sidx < n_slabs) {
          const f32x4* q = reinterpret_cast<const f32x4*>(slabs + (long long)sidx * slab_stride + off);
          lo[sidx] = q[0];
          hi[sidx] = q[1];
        }
#pragma unroll
      for (int sidx = 0; sidx < MAX_S; ++sidx)
        if (sidx < n_slabs) {
#pragma unroll
          for (int j = 0; j < 4; ++j) { a[j] += lo[sidx][j]; a[4 + j] += hi[sidx][j]; }
        }
      for (int sidx = MAX_S; sidx < n_slabs; ++sidx) {  // more than 16 splits: the rare tail
        const f32x4* q = reinterpret_cast<const f32x4*>(slabs + (long long)sidx * slab_stride + off);
        f32x4 l2 = q[0], h2 = q[1];
#pragma unroll
        for (int j = 0; j < 4; ++j) { a[j] += l2[j]; a[4 + j] += h2[j]; }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = bf2f(f2bf(a[j]));
    } else {
      bf16x8 v = *reinterpret_cast<const bf16x8*>(x + off);
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = bf2f(v[j]);
    }
    bf16x8 c;
    if (has_residual) {
      bf16x8 b = *reinterpret_cast<const bf16x8*>(residual + off);
#pragma unroll
      for (int j = 0; j < 8; ++j) c[j] = f2bf(a[j] + bf2f(b[j]));
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) c[j] = f2bf(a[j]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { float f = bf2f(c[j]); ss += f * f; }
    *reinterpret_cast<bf16x8*>(residual + off) = c;
  }
  ss = wave_sum(ss);
  if (lane == 0) ss_part[(long long)t * gridDim.x + p] = ss;
}

// ---------------------------------------------------------------------------------
// RoPE (rotate-half pairing) on the q and k parts of the fused qkv row, in place for q;
// rotated k and v are scattered into the paged cache
//   cache[block][kv_head][slot_in_block][head_dim],  slot = slot_mapping[t] (-1: skip).
// One workgroup per token; a thread task = 8 consecutive dims of the first half of one
// head together with the matching 8 dims of the second half.
// SLAB = true: qkv arrives as n_slabs fp32 partial sums [n_slabs][T][width] (split-K GEMM
// output) that are summed on load, and rotated q is written to q_out [T, nh*hd] (bf16).
// KV8: the cache holds e4m3 bytes of x / kv_scale[head] (kv_scale: [2][nkv], K scales then V scales).
template <bool SLAB, bool KV8>
__global__ __launch_bounds__(64) void rope_kv_write_kernel(unsigned short* __restrict__ qkv, const float* __restrict__ slabs, int n_slabs, long long slab_stride,
                                                            unsigned short* __restrict__ q_out, const int* __restrict__ positions,
                                                            const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                                                            void* __restrict__ k_cache, void* __restrict__ v_cache,
                                                            const int* __restrict__ slot_mapping, int nh, int nkv, int hd, int block_size,
                                                            const float* __restrict__ row_ss, int ss_parts, float inv_h, float eps, const float* __restrict__ kv_scale) {
  // grid = (T, ceil(tasks / 64)), one wavefront per workgroup: tasks of a token run on
  // several CUs instead of one 256-thread loop.
  const int t = blockIdx.x, tid = blockIdx.y * 64 + threadIdx.x;
  float rs = 1.f;  // 1/rms of the row when the preceding RMSNorm was folded into the projection
  if (row_ss != nullptr) rs = rsqrtf(row_ss_total(row_ss + (long long)t * ss_parts, ss_parts, threadIdx.x & 63) * inv_h + eps);
  const int half = hd >> 1;
  const int tasks_per_head = half >> 3;
  const int width = (nh + 2 * nkv) * hd;
  unsigned short* row = SLAB ? nullptr : qkv + (long long)t * width;
  auto load8 = [&](int col) -> bf16x8 {
    if (!SLAB) return *reinterpret_cast<const bf16x8*>(row + col);
    float a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = 0.f;
    // loads of up to 16 slabs in flight together, summed in slab order
    constexpr int MAX_S = 16;
    f32x4 lo[MAX_S], hi[MAX_S];
#pragma unroll
    for (int sidx = 0; sidx < MAX_S; ++sidx)
      if (sidx < n_slabs) {
        const f32x4* p = reinterpret_cast<const f32x4*>(slabs + (long long)sidx * slab_stride + (long long)t * width + col);
        lo[sidx] = p[0];
        hi[sidx] = p[1];
      }
#pragma unroll
    for (int sidx = 0; sidx < MAX_S; ++sidx)
      if (sidx < n_slabs) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { a[j] += lo[sidx][j]; a[4 + j] += hi[sidx][j]; }
      }
    for (int sidx = MAX_S; sidx < n_slabs; ++sidx) {
      const f32x4* p = reinterpret_cast<const f32x4*>(slabs + (long long)sidx * slab_stride + (long long)t * width + col);
      f32x4 l2 = p[0], h2 = p[1];
#pragma unroll
      for (int j = 0; j < 4; ++j) { a[j] += l2[j]; a[4 + j] += h2[j]; }
    }
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = f2bf(a[j] * rs);
    return o;
  };
  const int pos = positions[t];
  const int slot = slot_mapping[t];
  const float* cs = cos_t + (long long)pos * half;
  const float* sn = sin_t + (long long)pos * half;
  const int blk = slot >= 0 ? slot / block_size : 0, off = slot >= 0 ? slot % block_size : 0;

  const int rot_tasks = (nh + nkv) * tasks_per_head;
  if (tid < rot_tasks) {
    const int task = tid;
    const int head = task / tasks_per_head, d0 = (task % tasks_per_head) << 3;
    bf16x8 a = load8(head * hd + d0);
    bf16x8 b = load8(head * hd + half + d0);
    bf16x8 oa, ob;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float c = cs[d0 + j], s = sn[d0 + j];
      float x1 = bf2f(a[j]), x2 = bf2f(b[j]);
      oa[j] = f2bf(x1 * c - x2 * s);
      ob[j] = f2bf(x2 * c + x1 * s);
    }
    if (head < nh) {
      unsigned short* hp = SLAB ? q_out + (long long)t * nh * hd + head * hd : row + head * hd;
      *reinterpret_cast<bf16x8*>(hp + d0) = oa;
      *reinterpret_cast<bf16x8*>(hp + half + d0) = ob;
    } else if (slot >= 0) {
      const int g = head - nh;
      const long long kp = (((long long)blk * nkv + g) * block_size + off) * hd;
      const float inv_ks = KV8 ? 1.f / kv_scale[g] : 1.f;
      kv_store8<KV8>(k_cache, kp + d0, oa, inv_ks);
      kv_store8<KV8>(k_cache, kp + half + d0, ob, inv_ks);
    }
  }
  if (slot >= 0) {
    const int v_tasks = nkv * (hd >> 3);
    const int task = tid - rot_tasks;
    if (task >= 0 && task < v_tasks) {
      const int g = task / (hd >> 3), d0 = (task % (hd >> 3)) << 3;
      const long long vp = (((long long)blk * nkv + g) * block_size + off) * hd;
      kv_store8<KV8>(v_cache, vp + d0, load8((nh + nkv) * hd + g * hd + d0), KV8 ? 1.f / kv_scale[nkv + g] : 1.f);
    }
  }
}

// All-reduce over the 4 lanes {l, l^16, l^32, l^48} in the VALU: v_permlane16_swap exchanges the
// odd 16-lane rows of one register with the even rows of another, v_permlane32_swap the upper
// half of one with the lower half of the other; fed the same value twice they leave a pair
// (x[l], x[l^16]) resp. (x[l], x[l^32]) in every lane — no LDS crossbar round trip.
__device__ inline float quad_rows_max(float x) {
  unsigned u = __float_as_uint(x);
  auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  u = __float_as_uint(fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1])));
  auto b = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ inline float quad_rows_sum(float x) {
  unsigned u = __float_as_uint(x);
  auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  u = __float_as_uint(__uint_as_float(a[0]) + __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// lane `lane`'s value of v, in a scalar register (lane: wave-uniform)
__device__ inline float lane_value(float v, int lane) { return __uint_as_float((unsigned int)__builtin_amdgcn_readlane((int)__float_as_uint(v), lane)); }

// ---------------------------------------------------------------------------------
// Ragged paged attention.  One workgroup per (token, kv head); the REP = nh/nkv query
// heads of the group share every K/V load.  Each of the 4 waves walks KV blocks
// w, w+4, ... of the token's sequence with an online softmax, then the waves merge
// through LDS.  Specialised for head_dim 64 / 128, block_size 16, REP <= 8 (host checks).
//
// Lane roles inside a wave, per 16-token KV block:
//   QK: lane = 4*token + c; the lane dots a quarter of the head dims of the K row
//       against the REP query heads; two xor-shuffles finish the 128-dim dot.
//   PV: lane = 16*tg + dv; it owns 1/16 of the output dims and tokens {tg, tg+4, tg+8,
//       tg+12} of the block; the 4 token groups are summed once at the end.
template <int REP, int HD, int NWAVE, bool KV8>
__global__ __launch_bounds__(64 * NWAVE) void paged_attention_kernel(const unsigned short* __restrict__ q, long long q_row_stride,
                                                              const void* __restrict__ k_cache, const void* __restrict__ v_cache,
                                                              const int* __restrict__ block_tables, int max_blocks_per_seq,
                                                              const int* __restrict__ tok_seq, const int* __restrict__ ctx_lens,
                                                              unsigned short* __restrict__ out, int nh, int nkv, float scale_log2e,
                                                              const float* __restrict__ kv_scale) {
  constexpr int BS = 16, NT = 64 * NWAVE;
  constexpr int DQ = HD / 4;   // dims of the QK dot owned by a lane
  constexpr int NV = DQ / 8;   // 16-byte K loads per lane
  constexpr int DP = HD / 16;  // output dims owned by a lane in PV
  constexpr bool RAWV = KV8 && HD == 128;
  static_assert(HD == 64 || HD == 128, "head_dim 64 or 128");
  __shared__ float p_lds[NWAVE][REP][BS];    // per-wave probabilities of the current block
  __shared__ float m_lds[NWAVE][REP], l_lds[NWAVE][REP];
  __shared__ float o_lds[NWAVE][REP][HD];
  const int t = blockIdx.x, g = blockIdx.y;
  // wave index in a scalar register: the block-table reads below become scalar loads, which do
  // not queue behind (and so do not drain) the K/V loads in flight
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int ctx = ctx_lens[t];
  float v_scale = 1.f;
  if constexpr (KV8) {  // stored K / V are x / scale: the K scale folds into the score scale, the V scale into the output
    scale_log2e *= kv_scale[g];
    v_scale = kv_scale[nkv + g];
  }
  unsigned short* orow = out + (long long)t * nh * HD + (long long)g * REP * HD;
  if (ctx <= 0) {  // 0: padding token, zero output; negative: a prompt-chunk row that the tiled kernel writes
    if (ctx == 0)
      for (int i = tid; i < REP * HD; i += NT) orow[i] = 0;
    return;
  }
  const int* bt = block_tables + (long long)tok_seq[t] * max_blocks_per_seq;
  const int n_blocks = (ctx + BS - 1) / BS;

  // Each wave keeps two K/V blocks in flight beyond the one it computes on, in three register stages whose roles
  // rotate through a loop unrolled by three (no register copies, and no copy that would wait for the newest loads).
  // The loads are unconditional — past the sequence's end the last block is read again and never used — so the
  // number of loads in flight is the same on every path and a stage is waited for by count (two stages stay in
  // flight), not by draining the queue.
  // MF (e4m3 cache, head width 128): the scores come from the matrix cores — a K fragment lane is (key = lane & 15,
  // dims 32 (lane >> 4) ..), the A operand of v_mfma_f32_16x16x32_bf16 as it is loaded, against q as the B operand
  // (column = head); the vector ALU keeps the softmax and P V
  constexpr bool MF = HD == 128;
  const int qk_tok = MF ? (lane & 15) : (lane >> 2), qk_c = MF ? (lane >> 4) : (lane & 3);
  const int pv_tg = lane >> 4, pv_dv = lane & 15;
  typedef __attribute__((ext_vector_type(DP))) unsigned short vrow_t;  // 16 B (HD 128) or 8 B (HD 64)
  typedef PAStage<HD, KV8> stage_t;
  const int last_b = n_blocks - 1;
  auto entry = [&](int b) __attribute__((always_inline)) {
    return bt[b < last_b ? b : last_b];
  };
  // block bases in scalar registers (the block id is wave-uniform), per-lane offsets fixed for the whole walk: the
  // loads need no vector address arithmetic, whose temporaries would otherwise share registers with a stage in flight
  // and wait for it
  const unsigned koff = stage_t::k_off(qk_tok, qk_c), voff = stage_t::v_off(pv_tg, pv_dv);
  constexpr int ESZ = KV8 ? 1 : 2;
  auto issue = [&](stage_t& st, int blk) __attribute__((always_inline)) {
    const long long base = ((long long)__builtin_amdgcn_readfirstlane(blk) * nkv + g) * (BS * HD * ESZ);
    st.load_at(reinterpret_cast<const char*>(k_cache) + base, reinterpret_cast<const char*>(v_cache) + base, koff, voff);
  };
  stage_t s0, s1, s2;
  issue(s0, entry(wave));
  issue(s1, entry(wave + NWAVE));
  // this lane's DQ-dim slice of the REP query heads, kept as packed bf16 pairs for
  // v_dot2_f32_bf16 (half the registers and multiplies of an fp32 copy, no unpacking of K)
  bf16x2 qp[MF ? 1 : REP][DQ / 2];
  typedef __attribute__((ext_vector_type(8))) __bf16 mf_ab;
  typedef __attribute__((ext_vector_type(4))) float mf_cd;
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  mf_ab qb[4];             // MF: q of head (lane & 15), dims 32 qk_c + 8 v .. + 8 (zero for columns past the group's heads)
  const int col = lane & 15;
  if constexpr (MF) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      bf16x8 raw = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
      if (col < REP) raw = *reinterpret_cast<const bf16x8*>(q + (long long)t * q_row_stride + (long long)(g * REP + col) * HD + qk_c * 32 + v * 8);
      qb[v] = __builtin_bit_cast(mf_ab, raw);
    }
  } else {
    const unsigned short* qrow = q + (long long)t * q_row_stride + (long long)g * REP * HD + qk_c * DQ;
#pragma unroll
    for (int h = 0; h < REP; ++h)
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const bf16x2x4 a = __builtin_bit_cast(bf16x2x4, *reinterpret_cast<const bf16x8*>(qrow + h * HD + v * 8));
#pragma unroll
        for (int j = 0; j < 4; ++j) qp[h][v * 4 + j] = a.p[j];
      }
  }
  float m_own = -1e30f;    // MF: running maximum of this lane's own head (equals m_run[col])
  float acc[REP][DP];
  float m_run[REP], l_run[REP];
#pragma unroll
  for (int h = 0; h < REP; ++h) {
    m_run[h] = -1e30f; l_run[h] = 0.f;
#pragma unroll
    for (int j = 0; j < DP; ++j) acc[h][j] = 0.f;
  }

  // one block: scores, online softmax, P V on the stage that holds block b
  auto block = [&](const stage_t& st, int b) __attribute__((always_inline)) {
    bf16x8 kv[NV];
    vrow_t vv[4];
    float vf[RAWV ? 4 : 1][8];  // e4m3 cache, head width 128: V as fp32, in the lane's own token order (KVStage<true>::unpack_f32)
    if constexpr (RAWV) st.unpack_f32(kv, vf, pv_dv);
    else st.unpack(kv, vv, pv_dv);
    if constexpr (MF) {
      mf_cd sc = (mf_cd){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int v = 0; v < 4; ++v) sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mf_ab, kv[v]), qb[v], sc, 0, 0, 0);
      // lane (head col, quarter qk_c): sc[e] = q_col . k of key 4 qk_c + e of the block
      float mx = -1e30f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        sc[e] = (col < REP && b * BS + 4 * qk_c + e < ctx) ? sc[e] * scale_log2e : -1e30f;
        mx = fmaxf(mx, sc[e]);
      }
      mx = quad_rows_max(mx);  // over the four quarters: the block's maximum for head col
      const float m_own_new = fmaxf(m_own, mx);
      mf_cd pe;
      float ps = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        pe[e] = __builtin_amdgcn_exp2f(sc[e] - m_own_new);
        ps += pe[e];
      }
      m_own = m_own_new;
      ps = quad_rows_sum(ps);
      if (col < REP) *reinterpret_cast<mf_cd*>(&p_lds[wave][col][4 * qk_c]) = pe;
#pragma unroll
      for (int h = 0; h < REP; ++h) {  // (head h's maximum and sum sit in lane h: scalar reads)
        const float m_new = fmaxf(m_run[h], lane_value(mx, h));
        const float corr = __builtin_amdgcn_exp2f(m_run[h] - m_new);
        l_run[h] = l_run[h] * corr + lane_value(ps, h);
        m_run[h] = m_new;
#pragma unroll
        for (int j = 0; j < DP; ++j) acc[h][j] *= corr;
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): p_lds visible to the wave
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int tok = RAWV ? pv_tg + 4 * (i ^ (pv_dv & 1)) : pv_tg + 4 * i;
#pragma unroll
        for (int h = 0; h < REP; ++h) {
          const float p = p_lds[wave][h][tok];
#pragma unroll
          for (int j = 0; j < DP; ++j) {
            if constexpr (RAWV) acc[h][j] += p * vf[i][j];
            else acc[h][j] += p * bf2f(vv[i][j]);
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
      return;
    }
    // ---- QK^T
    float s[REP];
#pragma unroll
    for (int h = 0; h < REP; ++h) {
      float d = 0.f;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const bf16x2x4 kw = __builtin_bit_cast(bf16x2x4, kv[v]);
#pragma unroll
        for (int j = 0; j < 4; ++j) d = __builtin_amdgcn_fdot2_f32_bf16(qp[h][v * 4 + j], kw.p[j], d, false);
      }
      d += dpp_f32<0xB1>(d);  // the 4 dim chunks of a key sit in one quad: quad swaps in registers
      d += dpp_f32<0x4E>(d);
      s[h] = (b * BS + qk_tok < ctx) ? d * scale_log2e : -1e30f;
    }
    // ---- online softmax over the block's 16 tokens (token index lives in lane bits 2..5)
#pragma unroll
    for (int h = 0; h < REP; ++h) {
      // quads hold equal values: half-row and row mirror fold the 4 quads of a row in
      // registers, only the two cross-row steps go through the LDS crossbar
      float mx = s[h];
      mx = fmaxf(mx, dpp_f32<0x141>(mx));
      mx = fmaxf(mx, dpp_f32<0x140>(mx));
      mx = quad_rows_max(mx);
      const float m_new = fmaxf(m_run[h], mx);
      const float p = __builtin_amdgcn_exp2f(s[h] - m_new);
      float ps = p;
      ps += dpp_f32<0x141>(ps);
      ps += dpp_f32<0x140>(ps);
      ps = quad_rows_sum(ps);
      const float corr = __builtin_amdgcn_exp2f(m_run[h] - m_new);
      l_run[h] = l_run[h] * corr + ps;
      m_run[h] = m_new;
#pragma unroll
      for (int j = 0; j < DP; ++j) acc[h][j] *= corr;
      if (qk_c == 0) p_lds[wave][h][qk_tok] = p;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): p_lds visible to the wave (single wave: no barrier needed)
    __builtin_amdgcn_wave_barrier();
    // ---- PV
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int tok = RAWV ? pv_tg + 4 * (i ^ (pv_dv & 1)) : pv_tg + 4 * i;
#pragma unroll
      for (int h = 0; h < REP; ++h) {
        const float p = p_lds[wave][h][tok];
#pragma unroll
        for (int j = 0; j < DP; ++j) {
          if constexpr (RAWV) acc[h][j] += p * vf[i][j];
          else acc[h][j] += p * bf2f(vv[i][j]);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  };

  for (int b = wave; b < n_blocks; b += 3 * NWAVE) {
    // the round's three block-table entries: scalar loads issued together (nothing about them is carried around the
    // loop — a carried id ends up in a vector register that doubles as a load destination, and waiting for it drains the queue)
    const int e2 = entry(b + 2 * NWAVE), e3 = entry(b + 3 * NWAVE), e4 = entry(b + 4 * NWAVE);
    issue(s2, e2);
    block(s0, b);
    if (b + NWAVE >= n_blocks) break;
    issue(s0, e3);
    block(s1, b + NWAVE);
    if (b + 2 * NWAVE >= n_blocks) break;
    issue(s1, e4);
    block(s2, b + 2 * NWAVE);
  }
  // ---- sum the 4 token groups inside the wave (lanes l, l^16, l^32, ...)
#pragma unroll
  for (int h = 0; h < REP; ++h)
#pragma unroll
    for (int j = 0; j < DP; ++j) {
      float v = acc[h][j];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      acc[h][j] = v;
    }
  if (pv_tg == 0) {
#pragma unroll
    for (int h = 0; h < REP; ++h)
#pragma unroll
      for (int j = 0; j < DP; ++j) o_lds[wave][h][pv_dv * DP + j] = acc[h][j];
  }
  if (lane == 0) {
#pragma unroll
    for (int h = 0; h < REP; ++h) { m_lds[wave][h] = m_run[h]; l_lds[wave][h] = l_run[h]; }
  }
  __syncthreads();
  // ---- merge the waves; thread i handles output element i (REP*HD elements)
  for (int i = tid; i < REP * HD; i += NT) {
    const int h = i / HD, d = i % HD;
    float m = m_lds[0][h];
#pragma unroll
    for (int w = 1; w < NWAVE; ++w) m = fmaxf(m, m_lds[w][h]);
    float num = 0.f, den = 0.f;
#pragma unroll
    for (int w = 0; w < NWAVE; ++w) {
      const float c = __builtin_amdgcn_exp2f(m_lds[w][h] - m);
      num += o_lds[w][h][d] * c;
      den += l_lds[w][h] * c;
    }
    orow[i] = f2bf(den > 0.f ? num / den * v_scale : 0.f);
  }
}

// ---------------------------------------------------------------------------------
// Tiled (flash-style) causal attention for prompt chunks on the paged cache, on MFMA.
// paged_attention_kernel gives every query token its own workgroup and re-reads the whole
// prefix per token: fine for decode, quadratic L2 traffic for a prompt.  Here one workgroup
// serves tiles of 16 consecutive query tokens of ONE chunk for one kv head (all REP query heads
// of the group: wave w owns heads w, w + 4), walking the keys 32 at a time.  The 32 key rows and
// 32 value rows of a step are read from the paged cache once per workgroup (16 lanes per 256-byte
// row) and staged ROW-MAJOR into LDS with 16-byte writes; the four waves read K fragments with
// ds_read_b128 and the transposed V fragments with ds_read_b64_tr_b16, so V never needs a
// transposing write.  Two MFMA shapes:
//   prefill_attention_tiles16   one 16-query tile, one head per wave (head groups of at most 4),
//                               mfma_f32_16x16x32_bf16: prompts too short to fill the chip with
//                               pairs of tiles, and tile pairs that are not neighbours of one chunk;
//   prefill_attention_tiles32   32 columns per wave on mfma_f32_32x32x16_bf16: two neighbouring
//                               tiles for one head, or one tile for the two heads of a wave.
// Tiles are described by host-staged arrays (first token row, valid queries, first position,
// block-table row); tile_n == 0 marks an unused tile.  K/V of the chunk's own tokens must be
// in the cache already (rope_kv_write runs first).  head_dim 128, block_size 16.
typedef __attribute__((ext_vector_type(8))) __bf16 mfma_ab;
typedef __attribute__((ext_vector_type(4))) float mfma_cd;
typedef __attribute__((ext_vector_type(4))) short tr_b64;  // result of one ds_read_b64_tr_b16
typedef __attribute__((address_space(3))) tr_b64 lds_tr_b64;

// One pass over the keys for ONE 16-query tile (rows row0 .., positions pos0 .., n_q valid
// queries) of one sequence; wave w serves head w of the group (waves past REP only stage).
//
// Everything is computed TRANSPOSED, so that P never leaves the registers:
//   S^T = K Q^T   A = K rows (lane: key lr, dims 32*ks + 8*lq ..), B = Q (lane: query lr, same
//                 dims) -> accumulator lane (lr, lq), reg r = score(key 4*lq + r, query lr):
//                 a lane owns ONE query and 8 of a step's 32 keys;
//   softmax       the row maximum is 7 in-lane max + 2 cross-lane exchanges (the 4 lanes of a
//                 query), one running maximum / sum / rescale factor per lane;
//   O^T += V^T P^T  the exponentiated accumulators, packed to bf16, ARE the B operand
//                 (lane: query lr, k-slots 8*lq ..+8) once the k dimension is read as the
//                 permutation slot(8*q + 4*b + j) = key 16*b + 4*q + j; the A operand V^T comes
//                 from the row-major V image by transposed reads: the 16 lanes lq of a read take
//                 the 4 rows key 16*b + 4*lq .. +3 x 16 dims and lane lr receives its dim of
//                 those 4 keys, i.e. elements 4*b .. 4*b+3 of its fragment.
// The output accumulator lane holds O[query lr][dims 16*nt + 4*lq .. +4]: 8-byte stores.
// Both LDS images are XOR-swizzled so that every lane group of the reads and writes covers
// distinct banks (a ds_read_b128 group mixes two values of lq here, so padding alone does not):
//   K  16-byte chunk c of row r sits at slot c ^ (r & 15) of the row's 16;
//   V  32-byte piece n (dims 16n ..) of row r sits at piece n ^ (r & 7) of the row's 8.
template <int REP, bool KV8>
__device__ __forceinline__ void prefill_attention_tiles16(const unsigned short* __restrict__ q, long long q_row_stride, const void* __restrict__ k_cache,
                                                          const void* __restrict__ v_cache, const int* __restrict__ bt, int row0, int pos0, int n_q,
                                                          unsigned short* __restrict__ out, int nh, int nkv, float scale_log2e, float v_scale, unsigned short* v_lds,
                                                          unsigned short* k_lds) {
  static_assert(REP <= 4, "one query head per wave");
  constexpr int HD = 128, BS = 16, KSTEP = 32;
  constexpr int TILE = KSTEP * HD;  // elements of one K or V image (8 KiB)
  const int g = blockIdx.x;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int lr = lane & 15, lq = lane >> 4;
  const int last_key = pos0 + n_q - 1;          // highest key index any query of the tile may see
  const bool head_on = REP == 4 || wave < REP;  // wave-uniform

  // Q fragments: lane = Q[query lr][dims 32*ks + 8*lq .. +8]
  mfma_ab qa[4];
  if (head_on) {
    const int qrow = row0 + (lr < n_q ? lr : n_q - 1);
    const unsigned short* qp = q + (long long)qrow * q_row_stride + (long long)(g * REP + wave) * HD + lq * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qa[ks] = *reinterpret_cast<const mfma_ab*>(qp + ks * 32);
  }
  mfma_cd o[8];
#pragma unroll
  for (int nt = 0; nt < 8; ++nt) o[nt] = (mfma_cd){0.f, 0.f, 0.f, 0.f};
  float m_run = -1e30f, l_run = 0.f;
  // Staging role, the same for K and V: 16-byte chunk sv_c of the rows of keys sv_kr and
  // sv_kr + 16 of the step, so 16 consecutive lanes fetch one whole 256-byte row.  The rows
  // are read from the cache ONCE per workgroup (32 bytes of each of K and V per thread), one
  // step ahead, and written to the swizzled images (the two keys of a thread share r & 15).
  const int sv_c = tid & 15, sv_kr = tid >> 4;
  const int k_st = sv_kr * HD + ((sv_c ^ sv_kr) << 3);
  const int v_st = sv_kr * HD + ((((sv_c >> 1) ^ (sv_kr & 7)) << 4) | ((sv_c & 1) << 3));
  bf16x8 k0, k1, v0, v1;
  // A step's 32 keys are the rows of TWO cache blocks (kb is a multiple of 32): a thread's keys
  // sv_kr and sv_kr + 16 are row sv_kr of block kb/16 and of block kb/16 + 1.  The block-table
  // entries and the blocks' base addresses are therefore wave-uniform (scalar loads, SGPR
  // arithmetic) and a lane adds one of two precomputed 32-bit offsets: its row, or — in the last
  // block of the pass and past it — its row clamped to the last key (a valid row; masked in the
  // softmax).  The entries are fetched one step before the rows, the rows one step before the
  // LDS write, so the dependent lookup -> address -> cache read chain never stalls the loop.
  // The loads are unconditional: a branch around them would leave the in-order load counter
  // unknown at the join and force a full drain there.
  constexpr int ESZ = KV8 ? 1 : 2;
  const int last_blk = last_key >> 4;
  const unsigned voff_row = (unsigned)((sv_kr * HD + sv_c * 8) * ESZ);
  const unsigned voff_end = (unsigned)(((sv_kr < (last_key & 15) ? sv_kr : (last_key & 15)) * HD + sv_c * 8) * ESZ);
  const long long blk_bytes = (long long)nkv * BS * HD * ESZ, head_bytes = (long long)g * BS * HD * ESZ;
#define BT_LOAD(KB, B0, B1)                                        \
  do {                                                             \
    const int bi = (KB) >> 4;                                      \
    B0 = bt[bi < last_blk ? bi : last_blk];                        \
    B1 = bt[bi + 1 < last_blk ? bi + 1 : last_blk];                \
  } while (0)
#define KV_LOAD_STEP(KB, B0, B1)                                                              \
  do {                                                                                        \
    const int bi = (KB) >> 4;                                                                 \
    const long long base0 = (long long)(B0) * blk_bytes + head_bytes;                         \
    const long long base1 = (long long)(B1) * blk_bytes + head_bytes;                         \
    const unsigned off0 = bi < last_blk ? voff_row : voff_end;                                \
    const unsigned off1 = bi + 1 < last_blk ? voff_row : voff_end;                            \
    k0 = kv_load8_at<KV8>(reinterpret_cast<const char*>(k_cache) + base0, off0);              \
    v0 = kv_load8_at<KV8>(reinterpret_cast<const char*>(v_cache) + base0, off0);              \
    k1 = kv_load8_at<KV8>(reinterpret_cast<const char*>(k_cache) + base1, off1);              \
    v1 = kv_load8_at<KV8>(reinterpret_cast<const char*>(v_cache) + base1, off1);              \
  } while (0)
  // Three LDS slots per image, software-pipelined by one step.  Iteration i
  //   writes the rows of step i+2 (in registers since the last iteration) into slot (i+2) % 3 and
  //   starts the cache reads of step i+3,
  //   reads the V fragments of step i and the K fragments of step i+1 BEFORE the softmax, whose
  //   VALU work then covers their LDS latency,
  //   issues the S^T MFMAs of step i+1 next to the softmax of step i (independent work in one
  //   basic block: the matrix pipe runs while the softmax occupies the vector ALU),
  //   and ends with the P V MFMAs of step i and ONE workgroup barrier.
  // Slot (i+2) % 3 was last read in iteration i-1, which that iteration's barrier closed.
#define STAGE_TILES(SLOT)                                                                        \
  do {                                                                                           \
    unsigned short* kd = k_lds + (SLOT) * TILE + k_st;                                            \
    *reinterpret_cast<bf16x8*>(kd) = k0;                                                         \
    *reinterpret_cast<bf16x8*>(kd + 16 * HD) = k1;                                               \
    unsigned short* vd = v_lds + (SLOT) * TILE + v_st;                                            \
    *reinterpret_cast<bf16x8*>(vd) = v0;                                                         \
    *reinterpret_cast<bf16x8*>(vd + 16 * HD) = v1;                                               \
  } while (0)
  // fragment read offsets (elements) inside an image
  const int kf_row = lr * HD;                                       // + 16*b rows; chunk (4*ks + lq) ^ lr
  const int vf_base = (4 * lq + (lr >> 2)) * HD + ((lr & 3) << 2);  // + 16*b rows; piece nt ^ vf_sw
  const int vf_sw = 4 * (lq & 1) + (lr >> 2);                       // (row & 7) of this lane's row
  // K fragments (lane = K[key 16*b + lr][dims 32*ks + 8*lq .. +8]) of a slot and S^T = K Q^T on
  // them: ST[b][r] = score(key 16*b + 4*lq + r of the step, query lr)
#define S_MFMAS(SLOT, ST)                                                                                                        \
  do {                                                                                                                           \
    mfma_ab kf[2][4];                                                                                                            \
    _Pragma("unroll") for (int b = 0; b < 2; ++b) _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                               \
        kf[b][ks] = *reinterpret_cast<const mfma_ab*>(k_lds + (SLOT) * TILE + 16 * b * HD + kf_row + (((4 * ks + lq) ^ lr) << 3)); \
    _Pragma("unroll") for (int b = 0; b < 2; ++b) {                                                                              \
      ST[b] = (mfma_cd){0.f, 0.f, 0.f, 0.f};                                                                                     \
      _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) ST[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[b][ks], qa[ks], ST[b], 0, 0, 0); \
    }                                                                                                                            \
  } while (0)
  int blk0, blk1;
  mfma_cd st_cur[2];
  BT_LOAD(0, blk0, blk1);
  KV_LOAD_STEP(0, blk0, blk1);
  BT_LOAD(KSTEP, blk0, blk1);
  STAGE_TILES(0);
  KV_LOAD_STEP(KSTEP, blk0, blk1);
  BT_LOAD(2 * KSTEP, blk0, blk1);
  STAGE_TILES(1);
  KV_LOAD_STEP(2 * KSTEP, blk0, blk1);
  BT_LOAD(3 * KSTEP, blk0, blk1);
  __syncthreads();
  if (head_on) S_MFMAS(0, st_cur);

  // One step.  MASKED: the step reaches the diagonal of some query of the tile (the last one or
  // two steps of a pass); the others run without any mask code.
  auto step = [&](auto masked_c, int kb, int s_v, int s_k, int s_w) __attribute__((always_inline)) {
    constexpr bool MASKED = decltype(masked_c)::value;
    STAGE_TILES(s_w);
    KV_LOAD_STEP(kb + 3 * KSTEP, blk0, blk1);
    BT_LOAD(kb + 4 * KSTEP, blk0, blk1);
    if (head_on) {
      // V^T fragments of this step (A lane = V^T[dim 16*nt + lr][k-slots 8*lq .. +8]): one pair of
      // transposed reads (keys 16*b + 4*lq .. +3, b = 0, 1) per 16 dims
      const unsigned short* vt = v_lds + s_v * TILE + vf_base;
      mfma_ab va[8];
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        struct { tr_b64 lo, hi; } vf;
        vf.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_b64*)(vt + ((nt ^ vf_sw) << 4)));
        vf.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_b64*)(vt + 16 * HD + ((nt ^ vf_sw) << 4)));
        va[nt] = __builtin_bit_cast(mfma_ab, vf);
      }
      mfma_cd st_nxt[2];
      S_MFMAS(s_k, st_nxt);
      // ---- online softmax in units of scale * log2(e): the multiply comes first on purpose (a
      // product is known to be a quiet number, so the maxima below need no canonicalising
      // v_max x, x, x per MFMA result)
      float sc[8];
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) sc[4 * b + r] = st_cur[b][r] * scale_log2e;
      if constexpr (MASKED) {
        const int lim = pos0 + lr - kb - 4 * lq;  // key offset 16*b + r is visible iff <= lim
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) sc[4 * b + r] = (16 * b + r <= lim) ? sc[4 * b + r] : -1e30f;
      }
      float mx = fmaxf(fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3])), fmaxf(fmaxf(sc[4], sc[5]), fmaxf(sc[6], sc[7])));
      mx = quad_rows_max(mx);  // the 4 lanes of a query
      const float m_old = m_run;
      const float m_new = fmaxf(m_old, mx);
      // v_exp_f32 as is (arguments <= 0; a result below the normal range flushes to 0, which
      // is what a softmax weight of 2^-126 is worth)
      float pr[8], psum = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        pr[j] = __builtin_amdgcn_exp2f(sc[j] - m_new);
        psum += pr[j];
      }
      const float corr = __builtin_amdgcn_exp2f(m_old - m_new);
      l_run = l_run * corr + psum;  // this lane's keys only; folded after the loop
      m_run = m_new;
      // ---- P^T as the B operand, straight from the accumulator layout (hardware bf16 conversion)
      mfma_ab pb;
#pragma unroll
      for (int j = 0; j < 8; ++j) pb[j] = (__bf16)pr[j];
      if (__builtin_amdgcn_ballot_w64(m_new != m_old) != 0) {  // rescale O only when some query's maximum moved
#pragma unroll
        for (int nt = 0; nt < 8; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) o[nt][r] *= corr;
      }
      // ---- O^T += V^T P^T
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) o[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va[nt], pb, o[nt], 0, 0, 0);
      st_cur[0] = st_nxt[0];
      st_cur[1] = st_nxt[1];
    }
    __syncthreads();  // the tiles of step i+2 are staged, and every wave is done with slot i % 3
  };

  int kb = 0, s_v = 0, s_k = 1, s_w = 2;
#define ROTATE_SLOTS()       \
  do {                       \
    const int s_t = s_v;     \
    s_v = s_k;               \
    s_k = s_w;               \
    s_w = s_t;               \
  } while (0)
  for (; kb + KSTEP - 1 <= pos0; kb += KSTEP) {
    step(std::false_type{}, kb, s_v, s_k, s_w);
    ROTATE_SLOTS();
  }
  for (; kb <= last_key; kb += KSTEP) {
    step(std::true_type{}, kb, s_v, s_k, s_w);
    ROTATE_SLOTS();
  }
#undef ROTATE_SLOTS
#undef S_MFMAS
#undef STAGE_TILES
#undef KV_LOAD_STEP
#undef BT_LOAD
  // ---- normalise and store: lane holds O[query lr][dims 16*nt + 4*lq + r]
  const float l = quad_rows_sum(l_run);  // per-lane partial sums: fold the 4 lanes of the query (before any lane leaves)
  if (!head_on || lr >= n_q) return;
  const float inv = l > 0.f ? v_scale / l : 0.f;
  unsigned short* op = out + (long long)(row0 + lr) * nh * HD + (long long)(g * REP + wave) * HD + 4 * lq;
#pragma unroll
  for (int nt = 0; nt < 8; ++nt) {
    typedef __attribute__((ext_vector_type(4))) unsigned short us4;
    us4 w;
#pragma unroll
    for (int r = 0; r < 4; ++r) w[r] = f2bf(o[nt][r] * inv);
    *reinterpret_cast<us4*>(op + nt * 16) = w;
  }
}

// The same pass on mfma_f32_32x32x16_bf16, a wave's 32 columns being either 32 consecutive queries
// of one sequence for ONE head (HPW = 1: head groups of at most 4; two whole tiles, the second
// possibly partial) or one 16-query tile for the wave's TWO heads (HPW = 2: head groups of 5-8,
// column = 16 * (which head) + query) — the K / V fragment traffic is the same.
// Why a second shape: a 16x16x32 MFMA holds its SIMD's vector issue for 8 of its 16 cycles, a
// 32x32x16 for 8 of its 32, and the two waves of a SIMD share that issue bandwidth.  The softmax
// needs about 24 issue cycles per score and lane (multiply, maximum, subtract, v_exp_f32 at 8,
// add, half a conversion) against 32 cycles of matrix time: with the small shape's 16 cycles of
// hold per score on top the loop is bound by vector issue whatever the schedule, with the large
// shape's 4 it is not.  The cross-lane part of the row maximum also shrinks from 4 lanes to 2.
//   S^T = K Q^T   A = K rows (lane l: key l & 31, dims 16*ks + 8*(l >> 5) ..), B = Q (lane: query
//                 l & 31, same dims) -> accumulator lane (query l & 31, half hi = l >> 5), register
//                 r = score(key (r & 3) + 8*(r >> 2) + 4*hi): a lane owns ONE query and 16 of the
//                 step's 32 keys, the other 16 live in lane l ^ 32;
//   O^T += V^T P^T  registers 8*s .. 8*s+7, packed to bf16, are the B operand of k-step s (keys
//                 16*s ..+15) with element j <-> key 16*s + 8*(j >> 2) + 4*hi + (j & 3); the A operand
//                 V^T[dim 32*nt + (l & 31)][those keys] is two transposed reads (j >> 2 = 0, 1) of
//                 4 keys x 16 dims per 16-lane group.
// LDS images of THIS path are padded, not swizzled: every lane group of its reads has one value of
// hi, so 272-byte K rows (16 rows of a ds_read_b128 group land on 16 different 16-byte slots) and
// 320-byte V rows (the 4 rows x 64 bytes of a transposed read's 32-lane half cover the 256-byte
// bank window once) are conflict-free, and every fragment address is lane base + slot base +
// immediate: one vector add per slot and step instead of one per read.
constexpr int PA32_K_RS = 136, PA32_V_RS = 160;
typedef __attribute__((ext_vector_type(16))) float mfma_cd16;
__device__ inline float half_rows_max(float x) {  // over lanes l and l ^ 32
  const unsigned u = __float_as_uint(x);
  auto a = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
}
__device__ inline float half_rows_sum(float x) {
  const unsigned u = __float_as_uint(x);
  auto a = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(a[0]) + __uint_as_float(a[1]);
}

// PART (a chunk deep inside a long context, its key range split over several workgroups): the pass covers the keys
// [key_lo, key_hi) only (key_lo a multiple of 32; the causal bound still applies) and leaves, per query and head, its
// output normalised by ITS OWN sum as fp32 in `part_o` ([query][nh][128], query 0 = the tile's first) and
// (reference, sum) in `part_ml` ([query][nh][2], reference in units of log2) for prefill_split_combine_kernel.
template <int REP, int HPW, bool KV8, bool PART = false>
__device__ __forceinline__ void prefill_attention_tiles32(const unsigned short* __restrict__ q, long long q_row_stride, const void* __restrict__ k_cache,
                                                          const void* __restrict__ v_cache, const int* __restrict__ bt, int row0, int pos0, int n_q,
                                                          unsigned short* __restrict__ out, int nh, int nkv, float scale_log2e, float v_scale, unsigned short* v_lds,
                                                          unsigned short* k_lds, int key_lo = 0, int key_hi = 0x7fffffff, float* __restrict__ part_o = nullptr,
                                                          float* __restrict__ part_ml = nullptr) {
  static_assert((HPW == 1 || HPW == 2) && REP <= 4 * HPW && REP > 4 * (HPW - 1), "32 columns = HPW heads x 32 / HPW queries");
  constexpr int HD = 128, BS = 16, KSTEP = 32;
  constexpr int K_RS = PA32_K_RS, V_RS = PA32_V_RS;         // row strides (elements) of the padded images
  constexpr int K_TILE = KSTEP * K_RS, V_TILE = KSTEP * V_RS;  // elements of one K / V image
  const int g = blockIdx.x;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int lr = lane & 31, hi = lane >> 5;
  // column lr of the 32-column tile: query qi of head wave + 4*hh
  const int qi = HPW == 2 ? (lr & 15) : lr, hh = HPW == 2 ? (lr >> 4) : 0;
  const int h = wave + 4 * hh;
  const int last_key = pos0 + n_q - 1;                // highest key index any query of the tiles may see
  const int kb0 = PART ? key_lo : 0;                  // first key of the pass
  const int lim_key = PART ? (last_key < key_hi - 1 ? last_key : key_hi - 1) : last_key;  // last key of the pass (cache reads run ahead up to last_key)
  const bool head_on = REP % 4 == 0 || wave < REP;   // wave-uniform: a wave without any head only stages
  const bool col_on = REP % 4 == 0 || h < REP;       // head groups of 5-7: some waves own one head, their other 16 columns idle

  // Q fragments: lane = Q[query lr][dims 16*ks + 8*hi .. +8]
  mfma_ab qa[8];
  if (head_on) {
    const int qrow = row0 + (qi < n_q ? qi : n_q - 1);
    const unsigned short* qp = q + (long long)qrow * q_row_stride + (long long)(g * REP + (col_on ? h : wave)) * HD + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qa[ks] = *reinterpret_cast<const mfma_ab*>(qp + ks * 16);
  }
  mfma_cd16 o[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[nt][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;
  // staging role and cache reads: as in prefill_attention_tiles16 (16 lanes per 256-byte row, keys
  // sv_kr and sv_kr + 16 of a step, one step ahead of the LDS write, block-table entries one
  // step earlier still, keys past the end clamped to the last key)
  const int sv_c = tid & 15, sv_kr = tid >> 4;
  const int k_st = sv_kr * K_RS + sv_c * 8;
  const int v_st = sv_kr * V_RS + sv_c * 8;
  bf16x8 k0, k1, v0, v1;
  constexpr int ESZ = KV8 ? 1 : 2;
  const int last_blk = last_key >> 4;
  const unsigned voff_row = (unsigned)((sv_kr * HD + sv_c * 8) * ESZ);
  const unsigned voff_end = (unsigned)(((sv_kr < (last_key & 15) ? sv_kr : (last_key & 15)) * HD + sv_c * 8) * ESZ);
  const long long blk_bytes = (long long)nkv * BS * HD * ESZ, head_bytes = (long long)g * BS * HD * ESZ;
#define BT_LOAD(KB, B0, B1)                                        \
  do {                                                             \
    const int bi = (KB) >> 4;                                      \
    B0 = bt[bi < last_blk ? bi : last_blk];                        \
    B1 = bt[bi + 1 < last_blk ? bi + 1 : last_blk];                \
  } while (0)
#define KV_LOAD_STEP(KB, B0, B1)                                                              \
  do {                                                                                        \
    const int bi = (KB) >> 4;                                                                 \
    const long long base0 = (long long)(B0) * blk_bytes + head_bytes;                         \
    const long long base1 = (long long)(B1) * blk_bytes + head_bytes;                         \
    const unsigned off0 = bi < last_blk ? voff_row : voff_end;                                \
    const unsigned off1 = bi + 1 < last_blk ? voff_row : voff_end;                            \
    k0 = kv_load8_at<KV8>(reinterpret_cast<const char*>(k_cache) + base0, off0);              \
    v0 = kv_load8_at<KV8>(reinterpret_cast<const char*>(v_cache) + base0, off0);              \
    k1 = kv_load8_at<KV8>(reinterpret_cast<const char*>(k_cache) + base1, off1);              \
    v1 = kv_load8_at<KV8>(reinterpret_cast<const char*>(v_cache) + base1, off1);              \
  } while (0)
#define STAGE_TILES(KSLOT, VSLOT)                                                                \
  do {                                                                                           \
    unsigned short* kd = k_lds + (KSLOT) * K_TILE + k_st;                                         \
    *reinterpret_cast<bf16x8*>(kd) = k0;                                                         \
    *reinterpret_cast<bf16x8*>(kd + 16 * K_RS) = k1;                                             \
    unsigned short* vd = v_lds + (VSLOT) * V_TILE + v_st;                                         \
    *reinterpret_cast<bf16x8*>(vd) = v0;                                                         \
    *reinterpret_cast<bf16x8*>(vd + 16 * V_RS) = v1;                                             \
  } while (0)
  // fragment read offsets (elements) inside an image
  const int kf_base = lr * K_RS + hi * 8;  // + 16*ks: the rest of every K fragment address is an immediate
  const int vq = (lane >> 2) & 3, vp = lane & 3, vodd = (lane >> 4) & 1;
  const int vf_base = (4 * hi + vq) * V_RS + vodd * 16 + vp * 4;  // + (16*s + 8*jj) rows + 32*nt: immediates
  // K fragments of a slot and the S^T MFMAs on them
#define S_MFMAS(SLOT, ST)                                                                                             \
  do {                                                                                                                \
    mfma_ab kf[8];                                                                                                    \
    _Pragma("unroll") for (int ks = 0; ks < 8; ++ks) kf[ks] =                                                         \
        *reinterpret_cast<const mfma_ab*>(k_lds + (SLOT) * K_TILE + kf_base + 16 * ks);                               \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) ST[r] = 0.f;                                                       \
    _Pragma("unroll") for (int ks = 0; ks < 8; ++ks) ST = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qa[ks], ST, 0, 0, 0); \
  } while (0)
  int blk0, blk1;
  mfma_cd16 st_cur;
  // P of the previous step (bf16, as the B operand of its P V MFMAs, which run one iteration late);
  // before the first step: zero, against a zeroed V slot
  mfma_ab p_prev[2];
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
    for (int j = 0; j < 8; ++j) p_prev[s2][j] = (__bf16)0.f;
  k0 = k1 = v0 = v1 = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
  {
    unsigned short* vd = v_lds + 3 * V_TILE + v_st;
    *reinterpret_cast<bf16x8*>(vd) = v0;
    *reinterpret_cast<bf16x8*>(vd + 16 * V_RS) = v1;
  }
  BT_LOAD(kb0, blk0, blk1);
  KV_LOAD_STEP(kb0, blk0, blk1);
  BT_LOAD(kb0 + KSTEP, blk0, blk1);
  STAGE_TILES(0, 0);
  KV_LOAD_STEP(kb0 + KSTEP, blk0, blk1);
  BT_LOAD(kb0 + 2 * KSTEP, blk0, blk1);
  STAGE_TILES(1, 1);
  KV_LOAD_STEP(kb0 + 2 * KSTEP, blk0, blk1);
  BT_LOAD(kb0 + 3 * KSTEP, blk0, blk1);
  __syncthreads();
  if (head_on) S_MFMAS(0, st_cur);

  // One step, software-pipelined over THREE steps so that both halves of it carry matrix and vector
  // work (a wave whose P V MFMAs follow its own exponentials leaves the matrix pipe idle meanwhile,
  // and two such waves on a SIMD fall into the same phase rather than filling each other's gaps).
  // Iteration i
  //   writes the rows of step i+2 into K slot (i+2) % 3 and V slot (i+2) % 4, starts the cache reads
  //   of step i+3, reads the K fragments of step i+1 and the V fragments of step i-1;
  //   first half:  S^T MFMAs of step i+1   beside   s*c - reference and the row maximum of step i;
  //   (rare re-reference branch)
  //   second half: P V MFMAs of step i-1   beside   the exponentials, row sum and bf16 packing of step i;
  //   one barrier.  V of step i-1 is still read while step i+2 is staged: four V slots.
  // The P V of the last step follows the loops.
  // EXACT (the first step and the steps that reach the diagonal): causal mask, and the textbook
  // online softmax whose reference m_run is the running maximum.  The other steps let the
  // reference LAG: exponents are taken against the reference as it is, s * c - m_run in ONE fma
  // (no separate scaling, subtraction or correction factor), and only when some query's scores
  // exceed it by more than 2^LAG_LOG2 does the wave re-reference (m_run, l_run and O move
  // together, BEFORE this step's exponentials are taken, so nothing is ever at two scales).
  // P <= 2^LAG_LOG2 then instead of <= 1: the same relative precision in bf16, and the largest
  // term of every row sum is still >= 1.
  constexpr float LAG_LOG2 = 6.f;
  auto step = [&](auto exact_c, int kb, int it, int s_k, int s_w) __attribute__((always_inline)) {
    constexpr bool EXACT = decltype(exact_c)::value;
    STAGE_TILES(s_w, (it + 2) & 3);
    KV_LOAD_STEP(kb + 3 * KSTEP, blk0, blk1);
    BT_LOAD(kb + 4 * KSTEP, blk0, blk1);
    if (head_on) {
      const unsigned short* vt = v_lds + ((it + 3) & 3) * V_TILE + vf_base;  // step i-1
      mfma_ab va[4][2];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          struct { tr_b64 lo, hi; } vf;
          vf.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_b64*)(vt + (16 * s2) * V_RS + 32 * nt));
          vf.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_b64*)(vt + (16 * s2 + 8) * V_RS + 32 * nt));
          va[nt][s2] = __builtin_bit_cast(mfma_ab, vf);
        }
      mfma_cd16 st_nxt;
      S_MFMAS(s_k, st_nxt);
      // sc = s * c - reference, in units of log2.  (A product or fma result is known to be a
      // quiet number, so the maxima below need no canonicalising v_max x, x, x per MFMA result.)
      float sc[16];
      if constexpr (EXACT) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = st_cur[r] * scale_log2e;
        const int lim = pos0 + qi - kb - 4 * hi;  // key offset (r & 3) + 8*(r >> 2) [+ 4*hi] is visible iff <= lim
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = ((r & 3) + 8 * (r >> 2) <= lim) ? sc[r] : -1e30f;
      } else {
        const float neg_ref = -m_run;
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = __builtin_fmaf(st_cur[r], scale_log2e, neg_ref);
      }
      float mx = fmaxf(fmaxf(fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3])), fmaxf(fmaxf(sc[4], sc[5]), fmaxf(sc[6], sc[7]))),
                       fmaxf(fmaxf(fmaxf(sc[8], sc[9]), fmaxf(sc[10], sc[11])), fmaxf(fmaxf(sc[12], sc[13]), fmaxf(sc[14], sc[15]))));
      mx = half_rows_max(mx);  // the 2 lanes of a query
      // how far the reference moves up (0: not at all), and by what factor l_run and O shrink
      float up = 0.f;
      bool move;
      if constexpr (EXACT) {
        const float m_new = fmaxf(m_run, mx);
        up = m_new - m_run;  // m_run = -1e30 before the first step: up is huge, the factor 0, and O and l_run are 0 anyway
        move = __builtin_amdgcn_ballot_w64(up != 0.f) != 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] -= m_new;
        m_run = m_new;
      } else {
        move = __builtin_amdgcn_ballot_w64(mx > LAG_LOG2) != 0;
      }
      if (move) {
        if constexpr (!EXACT) {
          up = fmaxf(mx, 0.f);
          m_run += up;
#pragma unroll
          for (int r = 0; r < 16; ++r) sc[r] -= up;
        }
        const float corr = __builtin_amdgcn_exp2f(-up);
        l_run *= corr;
        // the previous step's P is still on its way into O: it moves with O
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int j = 0; j < 8; ++j) p_prev[s2][j] = (__bf16)((float)p_prev[s2][j] * corr);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[nt][r] *= corr;
      }
      // v_exp_f32 as is (a result below the normal range flushes to 0, which is what a softmax
      // weight of 2^-126 is worth)
      float pr[16], psum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        pr[r] = __builtin_amdgcn_exp2f(sc[r]);
        psum += pr[r];
      }
      l_run += psum;  // this lane's keys only; folded after the loop
      // ---- O^T += V^T P^T of the previous step
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) o[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va[nt][s2], p_prev[s2], o[nt], 0, 0, 0);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int j = 0; j < 8; ++j) p_prev[s2][j] = (__bf16)pr[8 * s2 + j];
      st_cur = st_nxt;
    }
    __syncthreads();  // the tiles of step i+2 are staged, and every wave is done with K slot (i+1) % 3 and V slot (i-1) % 4
  };

  int kb = kb0, it = 0, s_i = 0, s_k = 1, s_w = 2;  // K slots: idle, read (step i+1), written (step i+2)
#define ROTATE_SLOTS()       \
  do {                       \
    const int s_t = s_i;     \
    s_i = s_k;               \
    s_k = s_w;               \
    s_w = s_t;               \
    ++it;                    \
  } while (0)
  if (kb + KSTEP - 1 <= pos0 && kb <= lim_key) {  // the first step sets the reference: exact (its mask hides nothing)
    step(std::true_type{}, kb, it, s_k, s_w);
    ROTATE_SLOTS();
    kb += KSTEP;
  }
  for (; kb + KSTEP - 1 <= pos0 && kb <= lim_key; kb += KSTEP) {
    step(std::false_type{}, kb, it, s_k, s_w);
    ROTATE_SLOTS();
  }
  for (; kb <= lim_key; kb += KSTEP) {
    step(std::true_type{}, kb, it, s_k, s_w);
    ROTATE_SLOTS();
  }
  if (head_on) {  // P V of the last step (its V slot was staged three iterations ago; nothing overwrote it)
    const unsigned short* vt = v_lds + ((it + 3) & 3) * V_TILE + vf_base;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        struct { tr_b64 lo, hi; } vf;
        vf.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_b64*)(vt + (16 * s2) * V_RS + 32 * nt));
        vf.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_b64*)(vt + (16 * s2 + 8) * V_RS + 32 * nt));
        o[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_ab, vf), p_prev[s2], o[nt], 0, 0, 0);
      }
  }
#undef ROTATE_SLOTS
#undef S_MFMAS
#undef STAGE_TILES
#undef KV_LOAD_STEP
#undef BT_LOAD
  // ---- normalise and store: lane holds O[its column's query and head][dims 32*nt + 8*rq + 4*hi + (0..3)]
  const float l = half_rows_sum(l_run);  // before any lane leaves: the exchange wants both lanes of a query
  if (!head_on || !col_on || qi >= n_q) return;
  const float inv = l > 0.f ? v_scale / l : 0.f;
  if constexpr (PART) {
    const long long e = (long long)qi * nh + (g * REP + h);
    float* po = part_o + e * HD + 4 * hi;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        typedef __attribute__((ext_vector_type(4))) float f4;
        f4 w;
#pragma unroll
        for (int r = 0; r < 4; ++r) w[r] = o[nt][4 * rq + r] * inv;
        *reinterpret_cast<f4*>(po + 32 * nt + 8 * rq) = w;
      }
    if (hi == 0) {
      typedef __attribute__((ext_vector_type(2))) float f2;
      *reinterpret_cast<f2*>(part_ml + 2 * e) = (f2){m_run, l};
    }
    return;
  }
  unsigned short* op = out + (long long)(row0 + qi) * nh * HD + (long long)(g * REP + h) * HD + 4 * hi;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      typedef __attribute__((ext_vector_type(4))) unsigned short us4;
      us4 w;
#pragma unroll
      for (int r = 0; r < 4; ++r) w[r] = f2bf(o[nt][4 * rq + r] * inv);
      *reinterpret_cast<us4*>(op + 32 * nt + 8 * rq) = w;
    }
}

// ---------------------------------------------------------------------------------
// The 32-column pass in an 8-wave workgroup: two halves (waves 0-3 and 4-7; a SIMD hosts one wave
// of each) serve two NEIGHBOURING 32-query blocks of one chunk on the SAME K / V images and
// alternate by construction — while one half is in its matrix phase (S^T of step i+1 and P V of
// step i-1: 16 MFMAs, the fragment reads, the staging) the other is in its vector phase (the
// softmax of step i), two workgroup barriers per step, the second half one phase behind the
// first.  Two independent 4-wave workgroups per CU drift into the SAME phase instead (measured:
// matrix pipe 39.5 % busy, vector ALU 34.5 %, hardly overlapping); here the phases are
// complementary whatever the timing, and 512 threads share the staging of a step (one row each).
// The host hands out groups of four neighbouring tiles of one chunk (tile lists padded per chunk
// to a multiple of four: `quads`), so a group is up to 64 consecutive queries of one sequence.
template <int REP, bool KV8>
__global__ __launch_bounds__(512) void prefill_attention_quads_kernel(const unsigned short* __restrict__ q, long long q_row_stride, const void* __restrict__ k_cache,
                                                                      const void* __restrict__ v_cache, const int* __restrict__ block_tables, int max_blocks_per_seq,
                                                                      const int* __restrict__ tile_row0, const int* __restrict__ tile_n,
                                                                      const int* __restrict__ tile_pos0, const int* __restrict__ tile_seq,
                                                                      unsigned short* __restrict__ out, int nh, int nkv, float scale_log2e,
                                                                      const float* __restrict__ kv_scale) {
  static_assert(REP <= 4, "one query head per wave");
  constexpr int HD = 128, BS = 16, KSTEP = 32;
  constexpr int K_RS = PA32_K_RS, V_RS = PA32_V_RS;
  constexpr int K_TILE = KSTEP * K_RS, V_TILE = KSTEP * V_RS;
  __shared__ __attribute__((aligned(256))) unsigned short v_lds[4 * V_TILE];
  __shared__ __attribute__((aligned(256))) unsigned short k_lds[3 * K_TILE];
  const int t0 = 4 * (gridDim.y - 1 - blockIdx.y);  // later groups of a chunk see more keys: longest first
  const int n_total = tile_n[t0] + tile_n[t0 + 1] + tile_n[t0 + 2] + tile_n[t0 + 3];
  if (n_total == 0) return;
  const int row0_wg = tile_row0[t0], pos0_wg = tile_pos0[t0];
  const int* __restrict__ bt = block_tables + (long long)tile_seq[t0] * max_blocks_per_seq;
  float v_scale = 1.f;
  if constexpr (KV8) {
    scale_log2e *= kv_scale[blockIdx.x];
    v_scale = kv_scale[nkv + blockIdx.x];
  }
  const int g = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = wave >> 2, hw = wave & 3;  // which 32-query block, which head of the group
  const int lr = lane & 31, hi = lane >> 5;
  const int row0 = row0_wg + 32 * half, pos0 = pos0_wg + 32 * half;
  const int n_q = n_total - 32 * half < 32 ? n_total - 32 * half : 32;  // <= 0: this half has no queries
  const int last_key = pos0_wg + n_total - 1;  // of the WORKGROUP: both halves walk the same steps
  const bool head_on = (REP == 4 || hw < REP) && n_q > 0;

  mfma_ab qa[8];
  if (head_on) {
    const int qrow = row0 + (lr < n_q ? lr : n_q - 1);
    const unsigned short* qp = q + (long long)qrow * q_row_stride + (long long)(g * REP + hw) * HD + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qa[ks] = *reinterpret_cast<const mfma_ab*>(qp + ks * 16);
  }
  mfma_cd16 o[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[nt][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;
  // staging: thread = 16-byte chunk sv_c of row sv_kr (0..31) of the step; rows 0-15 (first half's
  // waves) are cache block kb/16, rows 16-31 (second half's) block kb/16 + 1: wave-uniform lookups
  const int sv_c = tid & 15, sv_kr = tid >> 4;
  const int k_st = sv_kr * K_RS + sv_c * 8, v_st = sv_kr * V_RS + sv_c * 8;
  bf16x8 k0, v0;
  constexpr int ESZ = KV8 ? 1 : 2;
  const int last_blk = last_key >> 4;
  const unsigned voff_row = (unsigned)(((sv_kr & 15) * HD + sv_c * 8) * ESZ);
  const unsigned voff_end = (unsigned)((((sv_kr & 15) < (last_key & 15) ? (sv_kr & 15) : (last_key & 15)) * HD + sv_c * 8) * ESZ);
  const long long blk_bytes = (long long)nkv * BS * HD * ESZ, head_bytes = (long long)g * BS * HD * ESZ;
#define BT_LOAD(KB, B)                                   \
  do {                                                   \
    const int bi = ((KB) >> 4) + half;                   \
    B = bt[bi < last_blk ? bi : last_blk];               \
  } while (0)
#define KV_LOAD_STEP(KB, B)                                                                   \
  do {                                                                                        \
    const int bi = ((KB) >> 4) + half;                                                        \
    const long long base = (long long)(B) * blk_bytes + head_bytes;                           \
    const unsigned off = bi < last_blk ? voff_row : voff_end;                                 \
    k0 = kv_load8_at<KV8>(reinterpret_cast<const char*>(k_cache) + base, off);                \
    v0 = kv_load8_at<KV8>(reinterpret_cast<const char*>(v_cache) + base, off);                \
  } while (0)
#define STAGE_TILES(KSLOT, VSLOT)                                                  \
  do {                                                                             \
    *reinterpret_cast<bf16x8*>(k_lds + (KSLOT) * K_TILE + k_st) = k0;               \
    *reinterpret_cast<bf16x8*>(v_lds + (VSLOT) * V_TILE + v_st) = v0;               \
  } while (0)
  const int kf_base = lr * K_RS + hi * 8;
  const int vq = (lane >> 2) & 3, vp = lane & 3, vodd = (lane >> 4) & 1;
  const int vf_base = (4 * hi + vq) * V_RS + vodd * 16 + vp * 4;
  // K fragments are read at the end of a vector phase (the LDS is idle then) for the matrix phase
  // that follows; V fragments at the start of the matrix phase, landing behind its S^T MFMAs
#define READ_K_FRAGS(KSLOT)                                                                                           \
  _Pragma("unroll") for (int ks = 0; ks < 8; ++ks) kf[ks] =                                                           \
      *reinterpret_cast<const mfma_ab*>(k_lds + (KSLOT) * K_TILE + kf_base + 16 * ks)
#define READ_V_FRAGS(VSLOT)                                                                                           \
  do {                                                                                                                \
    const unsigned short* vt = v_lds + (VSLOT) * V_TILE + vf_base;                                                    \
    _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2) {               \
      struct { tr_b64 lo, hi; } vf;                                                                                   \
      vf.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_b64*)(vt + (16 * s2) * V_RS + 32 * nt));                \
      vf.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_b64*)(vt + (16 * s2 + 8) * V_RS + 32 * nt));            \
      va[nt][s2] = __builtin_bit_cast(mfma_ab, vf);                                                                   \
    }                                                                                                                 \
  } while (0)
#define S_MFMAS(ST)                                                                                                   \
  do {                                                                                                                \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) ST[r] = 0.f;                                                       \
    _Pragma("unroll") for (int ks = 0; ks < 8; ++ks) ST = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qa[ks], ST, 0, 0, 0); \
  } while (0)
#define PV_MFMAS()                                                                                                    \
  _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2)                   \
      o[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va[nt][s2], pb[s2], o[nt], 0, 0, 0)
  int blk;
  mfma_cd16 st;        // S^T of the step whose softmax comes next
  mfma_ab kf[8], pb[2];
  BT_LOAD(0, blk);
  KV_LOAD_STEP(0, blk);
  BT_LOAD(KSTEP, blk);
  STAGE_TILES(0, 0);
  KV_LOAD_STEP(KSTEP, blk);
  BT_LOAD(2 * KSTEP, blk);
  STAGE_TILES(1, 1);
  KV_LOAD_STEP(2 * KSTEP, blk);
  BT_LOAD(3 * KSTEP, blk);
  STAGE_TILES(2, 2);
  KV_LOAD_STEP(3 * KSTEP, blk);
  BT_LOAD(4 * KSTEP, blk);
  __syncthreads();
  if (head_on) {
    READ_K_FRAGS(0);
    S_MFMAS(st);  // S^T of step 0
  }

  // vector phase of step kb: the softmax of prefill_attention_tiles32 (EXACT: mask + textbook
  // form; otherwise the lagging reference) on `st`, leaving P (bf16) in `pb`.  The P V of the
  // previous step is complete by now, so a re-reference rescales O and l only.
  constexpr float LAG_LOG2 = 6.f;
  auto softmax = [&](auto exact_c, int kb) __attribute__((always_inline)) {
    constexpr bool EXACT = decltype(exact_c)::value;
    float sc[16];
    if constexpr (EXACT) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[r] = st[r] * scale_log2e;
      const int lim = pos0 + lr - kb - 4 * hi;
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[r] = ((r & 3) + 8 * (r >> 2) <= lim) ? sc[r] : -1e30f;
    } else {
      const float neg_ref = -m_run;
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[r] = __builtin_fmaf(st[r], scale_log2e, neg_ref);
    }
    float mx = fmaxf(fmaxf(fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3])), fmaxf(fmaxf(sc[4], sc[5]), fmaxf(sc[6], sc[7]))),
                     fmaxf(fmaxf(fmaxf(sc[8], sc[9]), fmaxf(sc[10], sc[11])), fmaxf(fmaxf(sc[12], sc[13]), fmaxf(sc[14], sc[15]))));
    mx = half_rows_max(mx);
    float up = 0.f;
    bool move;
    if constexpr (EXACT) {
      const float m_new = fmaxf(m_run, mx);
      up = m_new - m_run;
      move = __builtin_amdgcn_ballot_w64(up != 0.f) != 0;
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[r] -= m_new;
      m_run = m_new;
    } else {
      move = __builtin_amdgcn_ballot_w64(mx > LAG_LOG2) != 0;
    }
    if (move) {
      if constexpr (!EXACT) {
        up = fmaxf(mx, 0.f);
        m_run += up;
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] -= up;
      }
      const float corr = __builtin_amdgcn_exp2f(-up);
      l_run *= corr;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[nt][r] *= corr;
    }
    float pr[16], psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      pr[r] = __builtin_amdgcn_exp2f(sc[r]);
      psum += pr[r];
    }
    l_run += psum;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int j = 0; j < 8; ++j) pb[s2][j] = (__bf16)pr[8 * s2 + j];
  };

  // A wave alternates: vector phase of step i (softmax on S^T of step i, then the K fragment reads of
  // step i+1 for its matrix phase — the LDS is idle in this phase, so their latency costs nothing),
  // barrier, matrix phase of step i (writes the rows of step i+3 into K slot i % 3 and V slot
  // (i+3) % 4, starts the cache reads of step i+4; reads the V fragments of step i behind S^T of
  // step i+1, then P V of step i), barrier.  The second half runs one phase behind the first, so
  // one half's matrix phase always runs beside the other's vector phase.  A K image lives from the
  // matrix phase that writes it (step - 3) to the vector phases that read it (step - 1): three
  // slots; a V image until the second half's vector phase of its own step, which runs beside the
  // first half's matrix phase of that step: four slots.
  if (half == 1) __syncthreads();
  int kb = 0, it = 0, ks3 = 0;  // ks3 = it % 3
  auto step = [&](auto exact_c) __attribute__((always_inline)) {
    // ---- vector phase
    if (head_on) {
      softmax(exact_c, kb);
      READ_K_FRAGS(ks3 + 1 < 3 ? ks3 + 1 : 0);
    }
    __syncthreads();
    // ---- matrix phase
    STAGE_TILES(ks3, (it + 3) & 3);
    KV_LOAD_STEP(kb + 4 * KSTEP, blk);
    BT_LOAD(kb + 5 * KSTEP, blk);
    if (head_on) {
      mfma_ab va[4][2];
      READ_V_FRAGS(it & 3);
      S_MFMAS(st);
      PV_MFMAS();
    }
    __syncthreads();
    ks3 = ks3 + 1 < 3 ? ks3 + 1 : 0;
    kb += KSTEP;
    ++it;
  };
  // the two forms of the softmax run in loops of their own (one loop with both would pay register
  // copies where the forms join); every half takes the same number of steps in all
  if (KSTEP - 1 <= pos0) step(std::true_type{});  // the first step sets the reference: exact
  while (kb + KSTEP - 1 <= pos0 && kb <= last_key) step(std::false_type{});
  while (kb <= last_key) step(std::true_type{});
  if (half == 0) __syncthreads();
#undef READ_V_FRAGS
#undef READ_K_FRAGS
#undef PV_MFMAS
#undef S_MFMAS
#undef STAGE_TILES
#undef KV_LOAD_STEP
#undef BT_LOAD
  const float l = half_rows_sum(l_run);
  if (!head_on || lr >= n_q) return;
  const float inv = l > 0.f ? v_scale / l : 0.f;
  unsigned short* op = out + (long long)(row0 + lr) * nh * HD + (long long)(g * REP + hw) * HD + 4 * hi;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      typedef __attribute__((ext_vector_type(4))) unsigned short us4;
      us4 w;
#pragma unroll
      for (int r = 0; r < 4; ++r) w[r] = f2bf(o[nt][4 * rq + r] * inv);
      *reinterpret_cast<us4*>(op + 32 * nt + 8 * rq) = w;
    }
}

// grid = (kv heads, tile groups): consecutive workgroup ids differ in the kv head, so with the
// round-robin placement of workgroups on the 8 XCDs every tile of kv head g (when nkv is a
// multiple of 8) runs on XCD g % 8 and that head's K/V stay in ONE XCD's L2.  A workgroup owns
// QT consecutive entries of the tile arrays: when they are consecutive tiles of one chunk
// (the usual case: the host lists a chunk's tiles in order) they share one pass over the keys,
// otherwise each gets a pass of its own.
template <int REP, int QT, bool KV8>
__global__ __launch_bounds__(256, 2) void prefill_attention_kernel(const unsigned short* __restrict__ q, long long q_row_stride,
                                                                const void* __restrict__ k_cache, const void* __restrict__ v_cache,
                                                                const int* __restrict__ block_tables, int max_blocks_per_seq,
                                                                const int* __restrict__ tile_row0, const int* __restrict__ tile_n,
                                                                const int* __restrict__ tile_pos0, const int* __restrict__ tile_seq, int n_tiles,
                                                                unsigned short* __restrict__ out, int nh, int nkv, float scale_log2e,
                                                                const float* __restrict__ kv_scale, int order) {
  float v_scale = 1.f;
  if constexpr (KV8) {
    scale_log2e *= kv_scale[blockIdx.x];
    v_scale = kv_scale[nkv + blockIdx.x];
  }
  // [slot][key][dim] images (swizzled 256-byte rows on the 16-column path, padded rows on the
  // 32-column path); the bank analysis of both assumes images that start on a multiple of 256 bytes
  __shared__ __attribute__((aligned(256))) unsigned short v_lds[4 * 32 * PA32_V_RS];  // sized for the 32-column path: padded images, four V slots
  __shared__ __attribute__((aligned(256))) unsigned short k_lds[3 * 32 * PA32_K_RS];
  // Later tiles of a chunk see more keys.  order 0: longest first, so the longest passes do not
  // start last.  order 1 (grids of at most two workgroups per CU, all resident at once): the
  // first half of the ids takes the tiles from the longest down, the second half from the
  // shortest up, so the two workgroups that share a CU add up to about the same number of keys
  // (2048 queries at 2048..4095: 158 -> 150 us; pairing neighbouring ids instead: 171 us).
  const int ny = gridDim.y, yi = blockIdx.y;
  const int half = (ny + 1) >> 1;
  const int t0 = (order == 1 && yi >= half ? yi - half : ny - 1 - yi) * QT;
  if constexpr (QT == 2) {
    const int t1 = t0 + 1;
    const int n0 = tile_n[t0], n1 = t1 < n_tiles ? tile_n[t1] : 0;
    if (n0 == 16 && n1 > 0 && tile_seq[t1] == tile_seq[t0] && tile_pos0[t1] == tile_pos0[t0] + 16 && tile_row0[t1] == tile_row0[t0] + 16) {
      prefill_attention_tiles32<REP, 1, KV8>(q, q_row_stride, k_cache, v_cache, block_tables + (long long)tile_seq[t0] * max_blocks_per_seq, tile_row0[t0], tile_pos0[t0],
                                          16 + n1, out, nh, nkv, scale_log2e, v_scale, v_lds, k_lds);
      return;
    }
    if (n0 > 0)
      prefill_attention_tiles16<REP, KV8>(q, q_row_stride, k_cache, v_cache, block_tables + (long long)tile_seq[t0] * max_blocks_per_seq, tile_row0[t0], tile_pos0[t0], n0,
                                      out, nh, nkv, scale_log2e, v_scale, v_lds, k_lds);
    if (n1 > 0) {
      __syncthreads();  // the first pass is done with the LDS tiles
      prefill_attention_tiles16<REP, KV8>(q, q_row_stride, k_cache, v_cache, block_tables + (long long)tile_seq[t1] * max_blocks_per_seq, tile_row0[t1], tile_pos0[t1], n1,
                                      out, nh, nkv, scale_log2e, v_scale, v_lds, k_lds);
    }
  } else {
    const int n0 = tile_n[t0];
    if (n0 > 0) {
      if constexpr (REP > 4)
        prefill_attention_tiles32<REP, 2, KV8>(q, q_row_stride, k_cache, v_cache, block_tables + (long long)tile_seq[t0] * max_blocks_per_seq, tile_row0[t0], tile_pos0[t0],
                                               n0, out, nh, nkv, scale_log2e, v_scale, v_lds, k_lds);
      else
        prefill_attention_tiles16<REP, KV8>(q, q_row_stride, k_cache, v_cache, block_tables + (long long)tile_seq[t0] * max_blocks_per_seq, tile_row0[t0], tile_pos0[t0],
                                             n0, out, nh, nkv, scale_log2e, v_scale, v_lds, k_lds);
    }
  }
}

// ---------------------------------------------------------------------------------
// The split form, for a few tiles deep inside a long context (a short new turn after a cached prefix of 100k tokens):
// grid = (kv heads, tile groups, splits) — workgroup z of a group walks the z-th share of the keys its queries may see
// (shares of whole 32-key steps, the same for every group of one chunk) on the 32-column pass and leaves partial
// results (see PART above); prefill_split_combine_kernel folds them.  Head groups of at most 4: two tiles per group
// when they are consecutive tiles of one chunk (else one pass each); head groups of 5-8: one tile per group.
// Workspace rows go by tile slot: [split][16 * tile + query][nh][..].
template <int REP, bool KV8>
__global__ __launch_bounds__(256, 2) void prefill_attention_split_kernel(const unsigned short* __restrict__ q, long long q_row_stride, const void* __restrict__ k_cache,
                                                                      const void* __restrict__ v_cache, const int* __restrict__ block_tables, int max_blocks_per_seq,
                                                                      const int* __restrict__ tile_row0, const int* __restrict__ tile_n,
                                                                      const int* __restrict__ tile_pos0, const int* __restrict__ tile_seq, int n_tiles, int nh, int nkv,
                                                                      float scale_log2e, const float* __restrict__ kv_scale, float* __restrict__ part_o,
                                                                      float* __restrict__ part_ml) {
  float v_scale = 1.f;
  if constexpr (KV8) {
    scale_log2e *= kv_scale[blockIdx.x];
    v_scale = kv_scale[nkv + blockIdx.x];
  }
  __shared__ __attribute__((aligned(256))) unsigned short v_lds[4 * 32 * PA32_V_RS];
  __shared__ __attribute__((aligned(256))) unsigned short k_lds[3 * 32 * PA32_K_RS];
  constexpr int QT = REP <= 4 ? 2 : 1, HPW = REP <= 4 ? 1 : 2;
  const int z = blockIdx.z, ns = gridDim.z;
  const long long split_rows = (long long)z * n_tiles * 16;
  auto pass = [&](int t, int nq) __attribute__((always_inline)) {
    const int pos0 = tile_pos0[t];
    const int steps = (pos0 + nq + 31) >> 5;           // 32-key steps up to the last visible key
    const int per = ((steps + ns - 1) / ns) << 5;      // keys per split
    float* po = part_o + (split_rows + 16 * t) * nh * 128;
    float* pml = part_ml + (split_rows + 16 * t) * nh * 2;
    if (z * per >= pos0 + nq) {  // nothing left for this split: an empty partial (sum 0), the outputs are not read
      const int e = threadIdx.x;
      if (e < nq * REP) {
        typedef __attribute__((ext_vector_type(2))) float f2;
        *reinterpret_cast<f2*>(pml + 2 * ((long long)(e / REP) * nh + blockIdx.x * REP + e % REP)) = (f2){-1e30f, 0.f};
      }
      return;
    }
    prefill_attention_tiles32<REP, HPW, KV8, true>(q, q_row_stride, k_cache, v_cache, block_tables + (long long)tile_seq[t] * max_blocks_per_seq, tile_row0[t], pos0, nq,
                                                   nullptr, nh, nkv, scale_log2e, v_scale, v_lds, k_lds, z * per, (z + 1) * per, po, pml);
  };
  const int t0 = blockIdx.y * QT;
  const int n0 = tile_n[t0];
  if constexpr (QT == 2) {
    const int t1 = t0 + 1;
    const int n1 = t1 < n_tiles ? tile_n[t1] : 0;
    if (n0 == 16 && n1 > 0 && tile_seq[t1] == tile_seq[t0] && tile_pos0[t1] == tile_pos0[t0] + 16 && tile_row0[t1] == tile_row0[t0] + 16) {
      pass(t0, 16 + n1);
      return;
    }
    if (n0 > 0) pass(t0, n0);
    if (n1 > 0) {
      __syncthreads();  // the first pass is done with the LDS tiles
      pass(t1, n1);
    }
  } else {
    if (n0 > 0) pass(t0, n0);
  }
}

// out[tile_row0[t] + r] = sum over splits of w_s * o_s / sum of w_s,  w_s = l_s * 2^(m_s - max m): grid (tile, query row,
// 8 heads), a thread per 4 dims of a head.
__global__ __launch_bounds__(256) void prefill_split_combine_kernel(const float* __restrict__ part_o, const float* __restrict__ part_ml, int n_split, int n_tiles,
                                                                    const int* __restrict__ tile_row0, const int* __restrict__ tile_n, unsigned short* __restrict__ out,
                                                                    int nh) {
  const int t = blockIdx.x, r = blockIdx.y;
  if (r >= tile_n[t]) return;
  const long long ws_row = 16 * t + r, split_rows = (long long)n_tiles * 16;
  unsigned short* orow = out + (long long)(tile_row0[t] + r) * nh * 128;
  typedef __attribute__((ext_vector_type(4))) float f4;
  typedef __attribute__((ext_vector_type(2))) float f2;
  typedef __attribute__((ext_vector_type(4))) unsigned short us4;
  const int c = blockIdx.z * 256 + threadIdx.x;  // 4 dims of one head
  if (c >= nh * 32) return;
  const int head = c >> 5, d = (c & 31) << 2;
  const f2* ml = reinterpret_cast<const f2*>(part_ml) + ws_row * nh + head;
  const float* po = part_o + (ws_row * nh + head) * 128 + d;
  const long long ml_stride = split_rows * nh, o_stride = split_rows * nh * 128;
  // every load of a loop is independent of the others (an empty share's outputs were never written: they are
  // read all the same and dropped, so that no load waits for a comparison)
  float mx = -1e30f;
#pragma unroll 8
  for (int s = 0; s < n_split; ++s) {
    const f2 v = ml[s * ml_stride];
    mx = v[1] > 0.f ? fmaxf(mx, v[0]) : mx;
  }
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  float wsum = 0.f;
#pragma unroll 8
  for (int s = 0; s < n_split; ++s) {
    const f2 v = ml[s * ml_stride];
    const f4 o = *reinterpret_cast<const f4*>(po + s * o_stride);
    const bool on = v[1] > 0.f;
    const float w = on ? v[1] * __builtin_amdgcn_exp2f(v[0] - mx) : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] += on ? o[j] * w : 0.f;
    wsum += w;
  }
  const float inv = wsum > 0.f ? 1.f / wsum : 0.f;
  us4 w4;
#pragma unroll
  for (int j = 0; j < 4; ++j) w4[j] = f2bf(acc[j] * inv);
  *reinterpret_cast<us4*>(orow + head * 128 + d) = w4;
}

// ---------------------------------------------------------------------------------
// Decode-step fusion: RoPE + KV write + paged attention in ONE launch, for steps in which
// every token is the only new token of its sequence (plain decoding: no prompt chunk needs
// keys that another workgroup of the same launch is still producing).  One workgroup per
// (token, kv head):
//   phase A  sum the split-K slabs of this head group's q rows (REP heads), k row and v row
//            (all slab loads of a thread in flight at once), apply the folded RMSNorm row
//            scale and RoPE, round to bf16 like the two-kernel path; q (pre-scaled) and the
//            token's own k / v stay in LDS, k / v also go to the paged cache for later steps;
//   phase B  the 4 waves walk the sequence's EARLIER positions in the cache (online softmax,
//            same lane roles as paged_attention_kernel) and wave 0 folds in the token's own
//            key / value from LDS — nothing written in phase A is read back from memory.
// Saves one kernel boundary and one slab -> q round trip per layer.  head_dim 128, block 16.
// KV8: the token's own key / value take part in the attention as they will be read back from
// the cache by later steps (rounded to e4m3, in stored units), so this kernel and the two-kernel
// path agree.
//
// SPLIT (long contexts on small grids): gridDim.z workgroups share one (token, kv head); each walks
// a contiguous range of the sequence's KV blocks (at least SPLIT_MIN_BLOCKS each: short contexts
// use fewer workgroups, the rest exit at once), leaves its partial (max, sum, unnormalised output)
// in a workspace, and the last one to arrive merges them — a 128k-token sequence decoded alone is
// then streamed by up to 32 x nkv workgroups instead of nkv.  Every workgroup repeats phase A
// (a few KB of slabs); only split 0 writes the cache and folds in the token's own key / value.
constexpr int SPLIT_MIN_BLOCKS = 16;

// all-reduce over the four lane quarters (lanes l, l ^ 16, l ^ 32, l ^ 48) with the two gfx950 row / half swaps
__device__ inline float quarters_max(float v) {
  typedef __attribute__((ext_vector_type(2))) unsigned int u2;
  const u2 a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  const u2 b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ inline float quarters_sum(float v) {
  typedef __attribute__((ext_vector_type(2))) unsigned int u2;
  const u2 a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  const u2 b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

//
// ATTN (with SPLIT; decode tokens of a step that also carries prompt chunks): phase A is another
// kernel's work — q arrives rotated (`q_in`, bf16 rows of `q_row_stride` elements) and the token's
// own key / value are in the cache already, so the walk covers all `ctx` positions and nothing is
// folded in from LDS.  `row_list` names the token rows to serve (grid.z entries; -1 = none).
//
// MF (long contexts): the scores of a block come from the matrix cores.  S^T[token][head] = K[16 tokens x 128] . Q^T is
// four mfma_f32_16x16x32_bf16 per block with K as the A operand — lane (token = lane & 15, chunk = lane >> 4) holds
// 32 dims of a key exactly as it loads them, and since a dot product may run over the dims in any order, k-step v of
// the instruction is "piece v of every chunk" on both operands — and Q^T as the B operand (lane (head = lane & 15,
// chunk): 16 registers instead of the 16 per head of the dot-product form).  A lane (head, quarter c) then owns the
// scores of tokens 4 c .. 4 c + 3 of its head: block maximum and sum by two row / half swaps, per-head values to
// every lane by v_readlane.  P . V stays on the vector ALU in packed fp32 pairs (V rows are head-dim-major: the
// matrix form would need the block transposed).  With an e4m3 cache this halves the vector work per block, which
// is what bounds that kernel (the bf16 cache is bound by the KV stream already).
template <int REP, int NWAVE, bool KV8, bool SPLIT, bool ATTN = false, bool MF = false>
__device__ __forceinline__ void decode_attention_body(const int bx, const int by, const int bz, const int gdx, const float* __restrict__ slabs, int n_slabs, long long slab_stride,
                                                                     const int* __restrict__ positions, const float* __restrict__ cos_t,
                                                                     const float* __restrict__ sin_t, void* __restrict__ k_cache,
                                                                     void* __restrict__ v_cache, const int* __restrict__ slot_mapping,
                                                                     const int* __restrict__ block_tables, int max_blocks_per_seq,
                                                                     const int* __restrict__ tok_seq, const int* __restrict__ ctx_lens,
                                                                     unsigned short* __restrict__ out, int nh, int nkv,
                                                                     const float* __restrict__ row_ss, int ss_parts, float inv_h, float eps,
                                                                     float scale_log2e, const float* __restrict__ kv_scale,
                                                                     float* __restrict__ part_o, float* __restrict__ part_ml, int* __restrict__ counters,
                                                                     const unsigned short* __restrict__ q_in,
                                                                     long long q_row_stride, const int* __restrict__ row_list) {
  constexpr int HD = 128, BS = 16, HALF = 64, NT = 64 * NWAVE;
  static_assert(!ATTN || SPLIT, "the attention-only form exists in the split layout");
  float inv_ks = 1.f, inv_vs = 1.f, v_scale = 1.f;
  if constexpr (KV8) {
    const float ks = kv_scale[by];
    v_scale = kv_scale[nkv + by];
    inv_ks = 1.f / ks;
    inv_vs = 1.f / v_scale;
    scale_log2e *= ks;
  }
  // rows 0..REP-1: q; row REP: this token's key; row REP+1: its value — the projection output,
  // bf16-rounded like the cache contents.  A.2 leaves q and the key, RoPE'd, as bf16 in xb_lds.
  __shared__ float x_lds[REP + 2][HD];
  __shared__ __attribute__((aligned(16))) unsigned short xb_lds[REP + 1][HD];
  __shared__ float p_lds[NWAVE][REP][BS];
  __shared__ float m_lds[NWAVE][REP], l_lds[NWAVE][REP];
  __shared__ float o_lds[NWAVE][4][REP][HD];  // [wave][token group]: partial outputs, summed in the merge
  // SPLIT: grid = (splits, kv heads, tokens) — consecutive workgroup ids (which the dispatcher
  // deals round-robin to the 8 XCDs) are the splits of one pair, so that a lone long sequence
  // is streamed by every XCD; with the token index fastest, a 16-row step whose only live row
  // is row 0 would put all its workgroups on one XCD.
  const int t = ATTN ? row_list[bz] : SPLIT ? bz : bx, g = by;
  const int n_z = SPLIT ? gdx : 1;
  if constexpr (ATTN) {
    if (t < 0) return;
  }
  // wave index in a scalar register: block-table reads become scalar loads (see paged_attention_kernel)
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int ctx = ctx_lens[t];
  unsigned short* orow = out + (long long)t * nh * HD + (long long)g * REP * HD;
  // the split index is rotated by the pair index: when short contexts leave only split 0 at work,
  // the working workgroups of consecutive pairs still sit on different XCDs
  const int z = SPLIT ? (int)((bx + (unsigned)(t * nkv + g)) % (unsigned)n_z) : 0;
  if (ctx <= 0) {  // padding token: no KV write (slot -1), zero output
    if (z == 0)
      for (int i = tid; i < REP * HD; i += NT) orow[i] = 0;
    return;
  }
  // The keys / values of earlier positions do not depend on phase A: each wave loads its
  // first KV block now, so that round trip overlaps the slab loads and the RoPE phase.
  const int prev = ATTN ? ctx : ctx - 1;
  const int* bt = block_tables + (long long)tok_seq[t] * max_blocks_per_seq;
  const int all_blocks = (prev + BS - 1) / BS;
  int eff = 1, b0 = 0, n_blocks = all_blocks;  // this workgroup walks blocks [b0, n_blocks)
  if constexpr (SPLIT) {
    eff = (all_blocks + SPLIT_MIN_BLOCKS - 1) / SPLIT_MIN_BLOCKS;
    eff = eff < 1 ? 1 : (eff > n_z ? n_z : eff);
    if (z >= eff) return;
    const int per = (all_blocks + eff - 1) / eff;
    b0 = z * per;
    n_blocks = b0 + per < all_blocks ? b0 + per : all_blocks;
  }
  static_assert(!MF || REP <= 16, "one head per lane column");
  const int qk_tok = MF ? (lane & 15) : (lane >> 2), qk_c = MF ? (lane >> 4) : (lane & 3);
  const int pv_tg = lane >> 4, pv_dv = lane & 15;
  KVStage<KV8> st_cur, st_nxt;  // this wave's first block and the one after: two blocks (16 KB of bf16) per wave in flight at any time
  if (b0 + wave < n_blocks) st_cur.load(k_cache, v_cache, ((long long)bt[b0 + wave] * nkv + g) * BS * HD, qk_tok, qk_c, pv_tg, pv_dv);
  if (b0 + wave + NWAVE < n_blocks) st_nxt.load(k_cache, v_cache, ((long long)bt[b0 + wave + NWAVE] * nkv + g) * BS * HD, qk_tok, qk_c, pv_tg, pv_dv);
  if constexpr (ATTN) {
    for (int pc = tid; pc < REP * (HD / 8); pc += NT) {
      const int h = pc >> 4, d0 = (pc & 15) << 3;
      *reinterpret_cast<bf16x8*>(&xb_lds[h][d0]) = *reinterpret_cast<const bf16x8*>(q_in + (long long)t * q_row_stride + (long long)(g * REP + h) * HD + d0);
    }
  } else {
  // likewise the RoPE table row of this token (needed in A.2)
  constexpr int ROT = (REP + 1) * 8;  // (q heads + k) x 8 pieces of 8 dims and their partners
  const int pos = positions[t];
  const int slot = slot_mapping[t];
  f32x4 csr[2], snr[2];
  if (tid < ROT) {
    const int d0 = (tid & 7) << 3;
    const f32x4* cp = reinterpret_cast<const f32x4*>(cos_t + (long long)pos * HALF + d0);
    const f32x4* sp = reinterpret_cast<const f32x4*>(sin_t + (long long)pos * HALF + d0);
    csr[0] = cp[0]; csr[1] = cp[1]; snr[0] = sp[0]; snr[1] = sp[1];
  }
  // ---- phase A.1: every thread sums ONE 4-column piece of this head group's q / k / v rows
  // over all slabs.  The slab loads are independent and issued together (one memory round
  // trip instead of n_slabs), and summed in slab order — the order of the two-kernel path.
  {
    float rs = 1.f;
    if (row_ss != nullptr) rs = rsqrtf(row_ss_total(row_ss + (long long)t * ss_parts, ss_parts, lane) * inv_h + eps);
    const int width = (nh + 2 * nkv) * HD;
    const float* row0 = slabs + (long long)t * width;
    constexpr int PIECES = (REP + 2) * (HD / 4);  // q heads, then k, then v; 32 pieces per row
    constexpr int MAX_S = 16;
    for (int pc = tid; pc < PIECES; pc += NT) {
      const int r = pc >> 5, c4 = (pc & 31) << 2;
      const int col = (r < REP ? (g * REP + r) * HD : r == REP ? (nh + g) * HD : (nh + nkv + g) * HD) + c4;
      f32x4 v[MAX_S];
#pragma unroll
      for (int sidx = 0; sidx < MAX_S; ++sidx)
        if (sidx < n_slabs) v[sidx] = *reinterpret_cast<const f32x4*>(row0 + (long long)sidx * slab_stride + col);
      f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int sidx = 0; sidx < MAX_S; ++sidx)
        if (sidx < n_slabs) a += v[sidx];
#pragma unroll
      for (int j = 0; j < 4; ++j) x_lds[r][c4 + j] = bf2f(f2bf(a[j] * rs));  // the GEMM output as bf16
    }
  }
  __syncthreads();
  // ---- phase A.2: RoPE on q and k (pairs d, d + 64), cache write of k and v
  {
    const long long cache_row = (slot >= 0 && z == 0) ? (((long long)(slot / BS) * nkv + g) * BS + slot % BS) * HD : -1;
    static_assert(ROT + 16 <= 256, "one task per thread");
    {
      const int task = tid;
      if (task < ROT) {
        const int hsel = task >> 3, d0 = (task & 7) << 3;
        bf16x8 oa, ob;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float c = csr[j >> 2][j & 3], sv = snr[j >> 2][j & 3];
          const float x1 = x_lds[hsel][d0 + j], x2 = x_lds[hsel][HALF + d0 + j];
          oa[j] = f2bf(x1 * c - x2 * sv);
          ob[j] = f2bf(x2 * c + x1 * sv);
        }
        if (hsel == REP) {  // the key: to the cache, and (as the cache will hold it) into LDS
          oa = kv_store8<KV8>(k_cache, cache_row + d0, oa, inv_ks, cache_row >= 0);
          ob = kv_store8<KV8>(k_cache, cache_row + HALF + d0, ob, inv_ks, cache_row >= 0);
        }
        *reinterpret_cast<bf16x8*>(&xb_lds[hsel][d0]) = oa;
        *reinterpret_cast<bf16x8*>(&xb_lds[hsel][HALF + d0]) = ob;
      } else if (task < ROT + 16 && (cache_row >= 0 || KV8)) {
        const int d0 = (task - ROT) << 3;
        bf16x8 ov;
#pragma unroll
        for (int j = 0; j < 8; ++j) ov[j] = f2bf(x_lds[REP + 1][d0 + j]);
        ov = kv_store8<KV8>(v_cache, cache_row + d0, ov, inv_vs, cache_row >= 0);
        if constexpr (KV8) {  // this thread's own 8 values: what the cache holds, in stored units
#pragma unroll
          for (int j = 0; j < 8; ++j) x_lds[REP + 1][d0 + j] = bf2f(ov[j]);
        }
      }
    }
  }
  }  // !ATTN
  __syncthreads();
  // ---- phase B: earlier positions from the cache
  // q stays packed as bf16 pairs (16 registers per head instead of 32) and meets the keys in
  // v_dot2_f32_bf16: half the multiply instructions, no bf16 -> fp32 unpacking of K, and a
  // register footprint that lets 8 waves per workgroup (2 per SIMD) run without spilling
  const int tail_c = lane & 3;  // dim chunk of the lane in the quad layout (the token's own key below)
  bf16x2 qp[REP][16];
  auto load_qp = [&]() {
#pragma unroll
    for (int h = 0; h < REP; ++h)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const bf16x2x4 qw = __builtin_bit_cast(bf16x2x4, *reinterpret_cast<const bf16x8*>(&xb_lds[h][tail_c * 32 + v * 8]));
#pragma unroll
        for (int j = 0; j < 4; ++j) qp[h][v * 4 + j] = qw.p[j];
      }
  };
  if constexpr (!MF) load_qp();  // (MF: only the token's own key below needs q in this form)
  float acc[REP][8];
  float m_run[REP], l_run[REP];
#pragma unroll
  for (int h = 0; h < REP; ++h) {
    m_run[h] = -1e30f; l_run[h] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[h][j] = 0.f;
  }
  if constexpr (MF) {
    typedef __attribute__((ext_vector_type(8))) __bf16 mf_ab;
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    const int col = lane & 15;  // the head whose scores this lane ends up with
    mf_ab qb[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      bf16x8 raw = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
      if (col < REP) raw = *reinterpret_cast<const bf16x8*>(&xb_lds[col][qk_c * 32 + v * 8]);
      qb[v] = __builtin_bit_cast(mf_ab, raw);
    }
    f32x2 a2[REP][4];
#pragma unroll
    for (int h = 0; h < REP; ++h)
#pragma unroll
      for (int j = 0; j < 4; ++j) a2[h][j] = (f32x2){0.f, 0.f};
    float m_own = -1e30f;  // running maximum of this lane's own head (equals m_run[col])
    // Three K/V stages in a ring of NAMED registers: a stage is refilled (with the block three rounds ahead) right
    // after its contents were taken, and nothing is ever copied from stage to stage — a copy of registers that a
    // load is still filling waits for that load, which leaves one block in flight per wave instead of three.
    KVStage<KV8> st_third;
    if (b0 + wave + 2 * NWAVE < n_blocks) st_third.load(k_cache, v_cache, ((long long)bt[b0 + wave + 2 * NWAVE] * nkv + g) * BS * HD, qk_tok, qk_c, pv_tg, pv_dv);
    auto one_block = [&](KVStage<KV8>& st, const int b) __attribute__((always_inline)) {
      bf16x8 kv[4], vv[4];
      st.unpack(kv, vv, pv_dv);
      if (b + 3 * NWAVE < n_blocks) st.load(k_cache, v_cache, ((long long)bt[b + 3 * NWAVE] * nkv + g) * BS * HD, qk_tok, qk_c, pv_tg, pv_dv);
      f32x4 sc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int v = 0; v < 4; ++v) sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mf_ab, kv[v]), qb[v], sc, 0, 0, 0);
      // lane (head col, quarter qk_c): sc[e] = q_col . k of token 4 qk_c + e
      float mx = -1e30f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        sc[e] = (col < REP && b * BS + 4 * qk_c + e < prev) ? sc[e] * scale_log2e : -1e30f;
        mx = fmaxf(mx, sc[e]);
      }
      mx = quarters_max(mx);
      const float m_own_new = fmaxf(m_own, mx);
      f32x4 pe;
      float ps = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        pe[e] = __builtin_amdgcn_exp2f(sc[e] - m_own_new);
        ps += pe[e];
      }
      m_own = m_own_new;
      ps = quarters_sum(ps);
      if (col < REP) *reinterpret_cast<f32x4*>(&p_lds[wave][col][4 * qk_c]) = pe;
#pragma unroll
      for (int h = 0; h < REP; ++h) {
        const float m_new = fmaxf(m_run[h], lane_value(mx, h));
        const float corr = __builtin_amdgcn_exp2f(m_run[h] - m_new);
        l_run[h] = l_run[h] * corr + lane_value(ps, h);
        m_run[h] = m_new;
#pragma unroll
        for (int j = 0; j < 4; ++j) a2[h][j] *= (f32x2){corr, corr};
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int tok = pv_tg + 4 * i;
        f32x2 vf[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) vf[j] = (f32x2){bf2f(vv[i][2 * j]), bf2f(vv[i][2 * j + 1])};
#pragma unroll
        for (int h = 0; h < REP; ++h) {
          const float pw = p_lds[wave][h][tok];
#pragma unroll
          for (int j = 0; j < 4; ++j) a2[h][j] += (f32x2){pw, pw} * vf[j];
        }
      }
      __builtin_amdgcn_wave_barrier();
    };
    for (int b = b0 + wave; b < n_blocks; b += 3 * NWAVE) {
      one_block(st_cur, b);
      if (b + NWAVE < n_blocks) one_block(st_nxt, b + NWAVE);
      if (b + 2 * NWAVE < n_blocks) one_block(st_third, b + 2 * NWAVE);
    }
#pragma unroll
    for (int h = 0; h < REP; ++h)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[h][j] = a2[h][j >> 1][j & 1];
  } else
  for (int b = b0 + wave; b < n_blocks; b += NWAVE) {
    // this block's K / V and the next one's are in registers or on their way (loaded before
    // phase A, or by earlier iterations); start the loads of the block after those
    KVStage<KV8> st_far;
    if (b + 2 * NWAVE < n_blocks) st_far.load(k_cache, v_cache, ((long long)bt[b + 2 * NWAVE] * nkv + g) * BS * HD, qk_tok, qk_c, pv_tg, pv_dv);
    bf16x8 kv[4], vv[4];
    st_cur.unpack(kv, vv, pv_dv);
    float s[REP];
#pragma unroll
    for (int h = 0; h < REP; ++h) {
      float d = 0.f;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const bf16x2x4 kw = __builtin_bit_cast(bf16x2x4, kv[v]);
#pragma unroll
        for (int j = 0; j < 4; ++j) d = __builtin_amdgcn_fdot2_f32_bf16(qp[h][v * 4 + j], kw.p[j], d, false);
      }
      d += dpp_f32<0xB1>(d);  // the 4 dim chunks of a key sit in one quad
      d += dpp_f32<0x4E>(d);
      s[h] = (b * BS + qk_tok < prev) ? d * scale_log2e : -1e30f;
    }
#pragma unroll
    for (int h = 0; h < REP; ++h) {
      // all-reduce over the 16 keys: quads hold equal values, so the two mirrors fold the 4
      // quads of a row in registers; only the two cross-row steps go through the LDS crossbar
      float mx = s[h];
      mx = fmaxf(mx, dpp_f32<0x141>(mx));
      mx = fmaxf(mx, dpp_f32<0x140>(mx));
      mx = quad_rows_max(mx);
      const float m_new = fmaxf(m_run[h], mx);
      const float p = __builtin_amdgcn_exp2f(s[h] - m_new);
      float ps = p;
      ps += dpp_f32<0x141>(ps);
      ps += dpp_f32<0x140>(ps);
      ps = quad_rows_sum(ps);
      const float corr = __builtin_amdgcn_exp2f(m_run[h] - m_new);
      l_run[h] = l_run[h] * corr + ps;
      m_run[h] = m_new;
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[h][j] *= corr;
      if (qk_c == 0) p_lds[wave][h][qk_tok] = p;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int tok = pv_tg + 4 * i;
#pragma unroll
      for (int h = 0; h < REP; ++h) {
        const float p = p_lds[wave][h][tok];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[h][j] += p * bf2f(vv[i][j]);
      }
    }
    __builtin_amdgcn_wave_barrier();
    st_cur = st_nxt;
    st_nxt = st_far;
  }
  // ---- the token's own key / value (wave 0 of split 0; one token group adds the value)
  if (!ATTN && wave == 0 && z == 0) {
    if constexpr (MF) load_qp();
    bf16x2 kp2[16];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const bf16x2x4 kw = __builtin_bit_cast(bf16x2x4, *reinterpret_cast<const bf16x8*>(&xb_lds[REP][tail_c * 32 + v * 8]));
#pragma unroll
      for (int j = 0; j < 4; ++j) kp2[v * 4 + j] = kw.p[j];
    }
#pragma unroll
    for (int h = 0; h < REP; ++h) {
      float d = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) d = __builtin_amdgcn_fdot2_f32_bf16(qp[h][j], kp2[j], d, false);
      d += dpp_f32<0xB1>(d);
      d += dpp_f32<0x4E>(d);
      d *= scale_log2e;
      const float m_new = fmaxf(m_run[h], d);
      const float p = __builtin_amdgcn_exp2f(d - m_new);
      const float corr = __builtin_amdgcn_exp2f(m_run[h] - m_new);
      l_run[h] = l_run[h] * corr + p;
      m_run[h] = m_new;
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[h][j] = acc[h][j] * corr + (pv_tg == 0 ? p * x_lds[REP + 1][pv_dv * 8 + j] : 0.f);
    }
  }
  // every token group leaves its partial output in LDS (the merge adds them up): 16-byte
  // stores instead of 64 cross-lane exchanges per lane
#pragma unroll
  for (int h = 0; h < REP; ++h) {
    *reinterpret_cast<f32x4*>(&o_lds[wave][pv_tg][h][pv_dv * 8]) = (f32x4){acc[h][0], acc[h][1], acc[h][2], acc[h][3]};
    *reinterpret_cast<f32x4*>(&o_lds[wave][pv_tg][h][pv_dv * 8 + 4]) = (f32x4){acc[h][4], acc[h][5], acc[h][6], acc[h][7]};
  }
  if (lane == 0) {
#pragma unroll
    for (int h = 0; h < REP; ++h) { m_lds[wave][h] = m_run[h]; l_lds[wave][h] = l_run[h]; }
  }
  __syncthreads();
  const bool direct = !SPLIT || eff == 1;
  const int wt = ATTN ? bz : t;  // workspace entries go by grid row (listed rows may be any token rows of a wide step)
  const long long pbase = SPLIT ? ((long long)(wt * nkv + g) * n_z) : 0;  // this (token, kv head)'s first partial
  for (int i = tid; i < REP * HD; i += NT) {
    const int h = i / HD, d = i % HD;
    float m = m_lds[0][h];
#pragma unroll
    for (int w = 1; w < NWAVE; ++w) m = fmaxf(m, m_lds[w][h]);
    float num = 0.f, den = 0.f;
#pragma unroll
    for (int w = 0; w < NWAVE; ++w) {
      const float c = __builtin_amdgcn_exp2f(m_lds[w][h] - m);
      num += (o_lds[w][0][h][d] + o_lds[w][1][h][d] + o_lds[w][2][h][d] + o_lds[w][3][h][d]) * c;
      den += l_lds[w][h] * c;
    }
    if (direct) {
      orow[i] = f2bf(den > 0.f ? num / den * v_scale : 0.f);
    } else {
      // partials cross XCDs (each has its own L2): device-scope stores write through and the
      // merging workgroup reads with device-scope loads, so no cache write-back / invalidate
      // (a walk over the whole L2 per workgroup) is needed around the arrival counter
      __hip_atomic_store(part_o + (pbase + z) * (REP * HD) + i, num, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (d == 0) {
        __hip_atomic_store(part_ml + ((pbase + z) * REP + h) * 2, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(part_ml + ((pbase + z) * REP + h) * 2 + 1, den, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  if constexpr (SPLIT) {
    if (direct) return;
    __shared__ int is_last;
    // The partial stores are write-through (device scope) but asynchronous: a workgroup-scope release
    // does not wait for them on this target (it only drains the LDS / scalar counter).  Every wave
    // therefore waits until its vector-memory counter is zero — the stores have reached memory —
    // before the barrier, and only then is the arrival counted; the merging workgroup, possibly
    // on another XCD, can no longer see the full count while a partial is still in flight.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // also a compiler barrier: no store may sink below it
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (tid == 0) {
      const int arrived = __hip_atomic_fetch_add(counters + wt * nkv + g, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      is_last = arrived == eff - 1;
      if (is_last) __hip_atomic_store(counters + wt * nkv + g, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
    }
    __syncthreads();
    if (!is_last) return;
    __shared__ float w_lds[REP][64];  // per head: weight of every split, 1 / total
    for (int k = tid; k < REP * 64; k += NT) {  // one wave per head (NT is a multiple of 64: h is uniform in a wave)
      const int h = k >> 6, s2 = k & 63;
      float m = -1e30f, den = 0.f;
      if (s2 < eff) {
        m = __hip_atomic_load(part_ml + ((pbase + s2) * REP + h) * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        den = __hip_atomic_load(part_ml + ((pbase + s2) * REP + h) * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      float mx = m;
      for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
      const float c = s2 < eff ? __builtin_amdgcn_exp2f(m - mx) : 0.f;
      float tot = den * c;
      for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o, 64);
      w_lds[h][s2] = tot > 0.f ? c / tot : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < REP * HD; i += NT) {
      const int h = i / HD;
      const float* po = part_o + pbase * (REP * HD) + i;
      float num = 0.f;
      int s2 = 0;
      for (; s2 + 4 <= eff; s2 += 4) {
        const float a0 = __hip_atomic_load(po + (long long)(s2 + 0) * (REP * HD), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float a1 = __hip_atomic_load(po + (long long)(s2 + 1) * (REP * HD), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float a2 = __hip_atomic_load(po + (long long)(s2 + 2) * (REP * HD), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float a3 = __hip_atomic_load(po + (long long)(s2 + 3) * (REP * HD), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        num += a0 * w_lds[h][s2] + a1 * w_lds[h][s2 + 1] + a2 * w_lds[h][s2 + 2] + a3 * w_lds[h][s2 + 3];
      }
      for (; s2 < eff; ++s2) num += __hip_atomic_load(po + (long long)s2 * (REP * HD), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * w_lds[h][s2];
      orow[i] = f2bf(num * v_scale);
    }
  }
}

// The kernel: one workgroup per (token, kv head[, split]) of the launch grid; grid rows past the kv
// heads of an unsplit launch are weight-prefetch workgroups (see prefetch_role).  The same body runs
// as the attention phase of the persistent decode kernel below, one (token, kv head) at a time.
template <int REP, int NWAVE, bool KV8, bool SPLIT, bool ATTN = false, bool MF = false>
__global__ __launch_bounds__(64 * NWAVE) void decode_attention_fused_kernel(const float* __restrict__ slabs, int n_slabs, long long slab_stride,
                                                                     const int* __restrict__ positions, const float* __restrict__ cos_t,
                                                                     const float* __restrict__ sin_t, void* __restrict__ k_cache,
                                                                     void* __restrict__ v_cache, const int* __restrict__ slot_mapping,
                                                                     const int* __restrict__ block_tables, int max_blocks_per_seq,
                                                                     const int* __restrict__ tok_seq, const int* __restrict__ ctx_lens,
                                                                     unsigned short* __restrict__ out, int nh, int nkv,
                                                                     const float* __restrict__ row_ss, int ss_parts, float inv_h, float eps,
                                                                     float scale_log2e, const float* __restrict__ kv_scale,
                                                                     float* __restrict__ part_o, float* __restrict__ part_ml, int* __restrict__ counters,
                                                                     PrefetchPlan pf, const unsigned short* __restrict__ q_in = nullptr,
                                                                     long long q_row_stride = 0, const int* __restrict__ row_list = nullptr) {
  if constexpr (!SPLIT) {
    // grid rows past the kv heads: weight prefetch for the output projection (see prefetch_role).
    // A workgroup of this kernel fills a CU's registers, so these run where the step's (token, kv
    // head) pairs leave CUs free and queue behind them otherwise.
    if ((int)blockIdx.y >= nkv) {
      prefetch_role(pf, ((int)blockIdx.y - nkv) * (int)gridDim.x + (int)blockIdx.x, (int)(blockIdx.y * gridDim.x + blockIdx.x),
                    __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), NWAVE, (int)threadIdx.x & 63);
      return;
    }
  }
  decode_attention_body<REP, NWAVE, KV8, SPLIT, ATTN, MF>((int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, (int)gridDim.x, slabs, n_slabs, slab_stride, positions, cos_t, sin_t,
                                                      k_cache, v_cache, slot_mapping, block_tables, max_blocks_per_seq, tok_seq, ctx_lens, out, nh, nkv, row_ss,
                                                      ss_parts, inv_h, eps, scale_log2e, kv_scale, part_o, part_ml, counters, q_in, q_row_stride, row_list);
}

// ---------------------------------------------------------------------------------
// Row-wise argmax of bf16 logits [S, V] -> int64 ids.  One 1024-thread workgroup per row.
__global__ __launch_bounds__(1024) void argmax_kernel(const unsigned short* __restrict__ logits, long long row_stride, int V, long long* __restrict__ out) {
  __shared__ float bv[16];
  __shared__ int bi[16];
  const int r = blockIdx.x, tid = threadIdx.x;
  const unsigned short* row = logits + (long long)r * row_stride;
  float best = -INFINITY;
  int besti = 0;
  const int nv = V >> 3;
  const bool aligned = ((((uintptr_t)row) & 15) == 0);
  if (aligned) {
    const bf16x8* rv = reinterpret_cast<const bf16x8*>(row);
    for (int i = tid; i < nv; i += 1024) {
      bf16x8 a = rv[i];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float f = bf2f(a[j]);
        if (f > best) { best = f; besti = i * 8 + j; }
      }
    }
  }
  for (int i = (aligned ? nv * 8 : 0) + tid; i < V; i += 1024) {
    float f = bf2f(row[i]);
    if (f > best) { best = f; besti = i; }
  }
  // first-index tie-break, like torch.argmax
  for (int o = 32; o > 0; o >>= 1) {
    float ov = __shfl_xor(best, o, 64);
    int oi = __shfl_xor(besti, o, 64);
    if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
  }
  if ((tid & 63) == 0) { bv[tid >> 6] = best; bi[tid >> 6] = besti; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 16; ++w)
      if (bv[w] > best || (bv[w] == best && bi[w] < besti)) { best = bv[w]; besti = bi[w]; }
    out[r] = besti;
  }
}

// Embedding lookup whose token id may come from the previous step's sampled ids instead of
// the staged token array: feed[i] = r + 1 selects prev_ids[r] (the host launched this step
// before it had read that id back), feed[i] = 0 selects tokens[i].  One workgroup per row,
// 16 bytes per thread per trip; ids outside [0, V) read row 0.
__global__ __launch_bounds__(256) void embed_fed_kernel(const unsigned short* __restrict__ table, const int* __restrict__ tokens, const int* __restrict__ feed,
                                                        const long long* __restrict__ prev_ids, int n_prev, unsigned short* __restrict__ out, int H, int V) {
  const int i = blockIdx.x;
  const int f = feed[i];
  long long tok = (f > 0 && f <= n_prev) ? prev_ids[f - 1] : (long long)tokens[i];
  if (tok < 0 || tok >= V) tok = 0;
  const bf16x8* src = reinterpret_cast<const bf16x8*>(table + tok * H);
  bf16x8* dst = reinterpret_cast<bf16x8*>(out + (long long)i * H);
  for (int c = threadIdx.x; c < (H >> 3); c += 256) dst[c] = src[c];
}

// y[t] = bf16(x[t] / rms(x[t])) from the per-row partial sums of squares of the residual
// stream (wide token batches: the library GEMM has no row-scale epilogue, so the folded
// RMSNorm is applied to its input).  One workgroup per row, 16 bytes per thread per trip.
__global__ __launch_bounds__(256) void scale_rows_kernel(const unsigned short* __restrict__ x, const float* __restrict__ row_ss, int ss_parts, float inv_h, float eps,
                                                         unsigned short* __restrict__ y, int H) {
  const int t = blockIdx.x;
  const float rs = rsqrtf(row_ss_total(row_ss + (long long)t * ss_parts, ss_parts, threadIdx.x & 63) * inv_h + eps);
  const bf16x8* src = reinterpret_cast<const bf16x8*>(x + (long long)t * H);
  bf16x8* dst = reinterpret_cast<bf16x8*>(y + (long long)t * H);
  for (int c = threadIdx.x; c < (H >> 3); c += 256) {
    bf16x8 v = src[c], o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = f2bf(bf2f(v[j]) * rs);
    dst[c] = o;
  }
}

// silu(gate) * up for the group-of-8 interleaved layout: gu [T, 2I] -> act [T, I]
// slab[t][n] += bias[n] * sqrt(mean square of row t + eps): a projection's bias carried by its first split-K slab,
// divided by the row scale (1 / rms of the folded RMSNorm) that the slabs' consumer still applies to their sum.
// row_ss == nullptr: no row scale is pending, the bias is added as it is.  One workgroup per token row.
__global__ __launch_bounds__(256) void slab_add_bias_kernel(float* __restrict__ slab, long long row_stride, const float* __restrict__ bias, const float* __restrict__ row_ss,
                                                            int ss_parts, float inv_h, float eps, int N) {
  __shared__ float red[4];
  const int t = blockIdx.x, tid = threadIdx.x;
  float f = 1.f;
  if (row_ss != nullptr) {
    float ssum = 0.f;
    for (int i = tid; i < ss_parts; i += 256) ssum += row_ss[(long long)t * ss_parts + i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ssum += __shfl_xor(ssum, o, 64);
    if ((tid & 63) == 0) red[tid >> 6] = ssum;
    __syncthreads();
    f = sqrtf((red[0] + red[1] + red[2] + red[3]) * inv_h + eps);
  }
  float* row = slab + (long long)t * row_stride;
  for (int n = tid; n < N; n += 256) row[n] += bias[n] * f;
}

__global__ __launch_bounds__(256) void swiglu_interleaved_kernel(const unsigned short* __restrict__ gu, unsigned short* __restrict__ act, int I) {
  const int t = blockIdx.y;
  const int grp = blockIdx.x * 256 + threadIdx.x;  // one thread = one group: 8 gate values, then their 8 up values
  const int n_tasks = I >> 3;
  if (grp >= n_tasks) return;
  const unsigned short* src = gu + (long long)t * 2 * I + (long long)grp * 16;
  bf16x8 a = *reinterpret_cast<const bf16x8*>(src);
  bf16x8 b = *reinterpret_cast<const bf16x8*>(src + 8);
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float gte = bf2f(a[j]), up = bf2f(b[j]);
    o[j] = f2bf(gte / (1.f + __expf(-gte)) * up);
  }
  *reinterpret_cast<bf16x8*>(act + (long long)t * I + (long long)grp * 8) = o;
}

// Host-side check of a prefetch plan against the weight allocation it points into: every range
// the prefetch workgroups will read must lie inside [base, base + bytes).
int make_prefetch_plan(const void* base, const long long* v, unsigned int* sink, PrefetchPlan* out) {
  const long long gx = v[0], gy = v[1], nw = v[2], kib = v[3], tile_bytes = v[4], split_bytes = v[5], wgs = v[6], bytes = v[7];
  if (gx < 1 || gy < 1 || nw < 1 || kib < 1 || wgs < 8 || (wgs & 7) || wgs > 65536 || gx * gy > (1 << 20)) return -1;
  if (tile_bytes < 1024 || (tile_bytes & 1023) || split_bytes < 1024 || (split_bytes & 1023) || ((uintptr_t)base & 15)) return -1;
  if (kib * 1024 > split_bytes) return -1;
  // the last range starts at ((gx * nw - 1) * tile_bytes + (gy - 1) * split_bytes) and is kib KiB long
  if ((gx * nw - 1) * tile_bytes + (gy - 1) * split_bytes + kib * 1024 > bytes) return -1;
  *out = PrefetchPlan{(const unsigned char*)base, (int)gx, (int)gy, (int)nw, (int)kib, tile_bytes, split_bytes, (int)wgs, sink};
  return 0;
}


// =================================================================================================
// Persistent decode-layers kernel: all decoder layers of a plain decoding step in ONE launch.
//
// A decode step of a dense model is one long weight stream (Llama-3-8B: 436 MB per layer) cut into
// seven dependent kernels per layer; at every cut HBM drains, the next kernel ramps up again, and
// the latency-bound kernels in between (residual update, decode attention at short contexts) leave
// it idle altogether — 98 us per layer against 69 us of weight bytes at 6.3 TB/s.  Weights do not
// depend on activations, so here the cuts become device-wide barriers INSIDE one kernel and every
// GEMM phase issues the first chunks of its weight ring BEFORE it waits at the barrier in front of
// it: the bytes are in flight while the flags travel and while the residual phase runs.
//
//   * grid = one workgroup of 8 waves per CU (co-resident by LDS size), phases per layer:
//       QKV GEMM | RoPE + KV write + attention | output GEMM | residual | gate/up GEMM (SwiGLU)
//       | down GEMM | residual        — 7 barriers per layer.
//   * GEMM phases are the weight-streaming kernel's loop (csrc/stream_gemm.hip): packed weights
//     global -> registers in a ring of NB chunks per wave, the x chunk shared through LDS, the
//     same (waves, K splits) plans, the same fp32 slabs — so every number equals what the
//     kernel-per-phase path computes, bit for bit.  Work units (feature group, K split) are dealt
//     round-robin to the workgroups.
//   * The attention phase is decode_attention_body, one (token, kv head) per workgroup at a time.
//   * Barrier: workgroup g stores the epoch into flags[g]; thread i of every workgroup polls
//     flags[i].  No read-modify-write, 2.6 us for 256 workgroups (csrc/tests/grid_barrier_bench.hip).
//     A poll is bounded by the real-time counter: a workgroup that never arrives makes the launch
//     finish with an error word set, not hang.
//   * Coherence: the XCDs' L2 caches are not coherent with each other and a kernel boundary's
//     write-back / invalidate is gone, so everything one workgroup writes and another reads inside
//     the launch (slabs, residual stream, attention output, activations, flags) lives in an
//     UNCACHED allocation: plain loads and stores, complete when vmcnt says so.  The residual stream
//     enters and leaves through ordinary memory (read by layer 0 only, written by the last layer
//     only); the KV cache is ordinary memory too (a step's own key / value are used by the
//     workgroup that wrote them).
typedef __attribute__((ext_vector_type(8))) __bf16 dl_frag_ab;
typedef __attribute__((ext_vector_type(4))) float dl_frag_cd;
typedef __attribute__((ext_vector_type(4))) unsigned int dl_raw16;

struct DlLayer { const void* wqkv; const void* wo; const void* wgu; const void* wdown; void* k_cache; void* v_cache; };
struct DlGemm { int N, K, nw, splits, cps; };  // cps: chunks of 128 columns per K split

struct DlParams {
  const DlLayer* layers;
  int n_layers, T, H, nh, nkv, I;
  DlGemm qkv, o, gu, down;
  const int* positions; const float* cos_t; const float* sin_t; const int* slot_mapping; const int* block_tables; int max_blocks_per_seq;
  const int* tok_seq; const int* ctx_lens;
  float eps, scale_log2e;
  const float* kv_scale; int kv_scale_stride;  // FP8 cache: [layers][2][nkv]
  unsigned short* residual; float* ss;         // ordinary memory: [T][H] bf16, [T][ceil(H/512)] fp32 (in: layer 0 reads; out: last layer writes)
  // uncached workspace
  unsigned short* xres; float* ssb; float* qkv_slabs; unsigned short* attn_out; float* proj_slabs; unsigned short* act;
  unsigned int* flags; unsigned int* state;    // state[0]: epoch reached by the previous launch, state[1]: error word
  unsigned long long* timeline;                // [layers][16] real-time counter stamps of workgroup 0 (100 MHz), or nullptr
};

// phase timeline of workgroup 0 (profiling aid: where a layer's time goes, barrier by barrier)
__device__ __forceinline__ void dl_stamp(const DlParams& p, int li, int idx, int wg, int tid) {
  if (p.timeline != nullptr && wg == 0 && tid == 0) p.timeline[li * 16 + idx] = __builtin_amdgcn_s_memrealtime();
}

constexpr unsigned long long DL_SPIN_TICKS = 200000000ull;  // 2 s of the 100 MHz real-time counter

struct DlBarrier {
  unsigned int* flags; unsigned int* state; unsigned int epoch; int G, wg;
};

// arrive: this workgroup's stores are complete (every wave waits for its vector-memory counter,
// then the workgroup barrier), then its flag carries the new epoch
__device__ __forceinline__ void dl_arrive(DlBarrier& b, int tid) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // also a compiler barrier: no store may sink below it
  __syncthreads();
  b.epoch += 1;
  if (tid == 0) __hip_atomic_store(b.flags + b.wg, b.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void dl_wait(const DlBarrier& b, int tid, int* dead) {
  if (tid < b.G && !*dead) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((int)(__hip_atomic_load(b.flags + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - b.epoch) < 0) {
      if (__builtin_amdgcn_s_memrealtime() - t0 > DL_SPIN_TICKS) {
        *dead = 1;
        atomicOr(b.state + 1, 1u);
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  __syncthreads();
}

// residual' = bf16(res_in + bf16(sum of slabs)), sums of squares per 512-column segment: the math of
// residual_reduce_kernel<true>, one (row, segment) per wave, dealt over the workgroups' waves
__device__ __forceinline__ void dl_reduce(const float* __restrict__ slabs, int n_slabs, long long slab_stride, const unsigned short* __restrict__ res_in,
                                          unsigned short* __restrict__ res_out, float* __restrict__ ss_out, int H, int T, int wg, int G, int wave, int lane) {
  const int parts = (H + 511) / 512;
  const int n_items = T * parts;
  for (int item = wg + G * wave; item < n_items; item += G * 8) {
    const int t = item / parts, p = item - t * parts;
    const int col = (p * 64 + lane) * 8;
    float ss = 0.f;
    if (col < H) {
      const long long off = (long long)t * H + col;
      float a[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = 0.f;
      // slabs in groups of 8 (all loads of a group in flight together, summed in slab order): the
      // weight ring of the next GEMM phase is live in registers while this runs
      constexpr int GRP = 8;
      const bf16x8 b = *reinterpret_cast<const bf16x8*>(res_in + off);
      for (int s0 = 0; s0 < n_slabs; s0 += GRP) {
        f32x4 lo[GRP], hi[GRP];
#pragma unroll
        for (int k = 0; k < GRP; ++k)
          if (s0 + k < n_slabs) {
            const f32x4* q = reinterpret_cast<const f32x4*>(slabs + (long long)(s0 + k) * slab_stride + off);
            lo[k] = q[0];
            hi[k] = q[1];
          }
#pragma unroll
        for (int k = 0; k < GRP; ++k)
          if (s0 + k < n_slabs) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { a[j] += lo[k][j]; a[4 + j] += hi[k][j]; }
          }
      }
      bf16x8 c;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        c[j] = f2bf(bf2f(f2bf(a[j])) + bf2f(b[j]));
        const float f = bf2f(c[j]);
        ss += f * f;
      }
      *reinterpret_cast<bf16x8*>(res_out + off) = c;
    }
    ss = wave_sum(ss);
    if (lane == 0) ss_out[(long long)t * parts + p] = ss;
  }
}

// One GEMM phase.  `mid()` runs between the issue of the first unit's first weight chunks and the
// first use of x: it holds the wait at the barrier in front of the phase (and, where a residual
// phase sits in between, that phase and its barrier), so the weight bytes travel meanwhile.
// EPI 1: fp32 slabs out[split][T][N]; EPI 2: SwiGLU, bf16 out[T][N/2] scaled by rs_lds[row].
template <int MT, int NB, int EPI, typename Mid>
__device__ __forceinline__ void dl_gemm(const void* __restrict__ Wp, const DlGemm g, const unsigned short* __restrict__ x, long long ldx, void* __restrict__ out,
                                        const float* rs_lds, dl_frag_ab* xs, unsigned int& seq, int wg, int G, int wave, int lane, Mid&& mid) {
  constexpr int KS = 4, PD = NB - 1, XF = KS * MT, XL = (XF + 7) / 8;
  const int r = lane & 15, q = lane >> 4;
  const int ksteps = g.K >> 5, nchunks = ksteps / KS;
  const int groups = g.N / (16 * g.nw);
  const int n_units = groups * g.splits;
  const bool active = wave < g.nw;
  const unsigned short* xsrc[XL];
#pragma unroll
  for (int i = 0; i < XL; ++i) {
    const int f = wave + 8 * i;
    const int ks = (f < XF ? f : 0) / MT, j = (f < XF ? f : 0) % MT;
    xsrc[i] = x + (long long)(j * 16 + r) * ldx + ks * 32 + q * 8;
  }
  dl_raw16 a[NB][KS];
  dl_frag_ab xr[NB][XL];
  dl_frag_cd acc[MT];

#define DL_ISSUE_W(slot, chunk)                                                                                                     \
  do {                                                                                                                              \
    if (active) {                                                                                                                   \
      _Pragma("unroll") for (int u = 0; u < KS; ++u) a[slot][u] = __builtin_nontemporal_load(wp + ((long long)(chunk) * KS + u) * 64); \
    }                                                                                                                               \
  } while (0)
#define DL_ISSUE_X(slot, chunk)                                                                                                     \
  do {                                                                                                                              \
    _Pragma("unroll") for (int i = 0; i < XL; ++i) if (wave + 8 * i < XF)                                                           \
        xr[slot][i] = *reinterpret_cast<const dl_frag_ab*>(xsrc[i] + (long long)(chunk) * KS * 32);                                 \
  } while (0)
#define DL_CONSUME(slot)                                                                                                            \
  do {                                                                                                                              \
    const int buf = (int)(seq & 1u);                                                                                                \
    seq += 1;                                                                                                                       \
    _Pragma("unroll") for (int i = 0; i < XL; ++i) if (wave + 8 * i < XF) xs[(buf * XF + wave + 8 * i) * 64 + lane] = xr[slot][i];  \
    __syncthreads();                                                                                                                \
    if (active) {                                                                                                                   \
      _Pragma("unroll") for (int u = 0; u < KS; ++u) {                                                                              \
        const dl_frag_ab af = __builtin_bit_cast(dl_frag_ab, a[slot][u]);                                                           \
        _Pragma("unroll") for (int j = 0; j < MT; ++j)                                                                              \
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, xs[(buf * XF + u * MT + j) * 64 + lane], acc[j], 0, 0, 0);         \
      }                                                                                                                             \
    }                                                                                                                               \
  } while (0)

  int unit = wg;
  bool has = unit < n_units;
  int fg = has ? unit % groups : 0, sp = has ? unit / groups : 0;
  long long tile = (long long)fg * g.nw + wave;
  const dl_raw16* wp = reinterpret_cast<const dl_raw16*>(Wp) + tile * (long long)ksteps * 64 + lane;
  int c0 = sp * g.cps, c1 = c0 + g.cps < nchunks ? c0 + g.cps : nchunks;
  if (has) {
#pragma unroll
    for (int s = 0; s < PD; ++s)
      if (c0 + s < c1) DL_ISSUE_W(s, c0 + s);
  }
  mid();
  while (has) {
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[j] = (dl_frag_cd){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < PD; ++s)
      if (c0 + s < c1) DL_ISSUE_X(s, c0 + s);
    int c = c0;
    for (; c + NB + PD <= c1; c += NB) {
#pragma unroll
      for (int s = 0; s < NB; ++s) {
        DL_ISSUE_X((s + PD) % NB, c + s + PD);
        DL_ISSUE_W((s + PD) % NB, c + s + PD);
        DL_CONSUME(s);
      }
    }
    // drain (the macros' own loop variables are i, u, j: the step counter must not be one of them)
#pragma unroll
    for (int dstep = 0; dstep < NB + PD - 1; ++dstep) {
      if (c + dstep < c1) {
        if (c + dstep + PD < c1) {
          DL_ISSUE_X((dstep + PD) % NB, c + dstep + PD);
          DL_ISSUE_W((dstep + PD) % NB, c + dstep + PD);
        }
        DL_CONSUME(dstep % NB);
      }
    }
    if (active) {
      const long long n_base = tile * 16 + q * 4;
      if constexpr (EPI == 1) {
        float* o = reinterpret_cast<float*>(out) + (long long)sp * ((long long)(16 * MT) * g.N);
#pragma unroll
        for (int j = 0; j < MT; ++j) *reinterpret_cast<dl_frag_cd*>(o + (long long)(j * 16 + r) * g.N + n_base) = acc[j];
      } else {
        unsigned short* o = reinterpret_cast<unsigned short*>(out);
        const long long i_base = tile * 8 + q * 4;
#pragma unroll
        for (int j = 0; j < MT; ++j) {
          const float rs = rs_lds[j * 16 + r];
          bf16x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float gt = acc[j][e] * rs, up = __shfl_xor(acc[j][e], 32, 64) * rs;
            v[e] = f2bf(gt / (1.f + __expf(-gt)) * up);
          }
          if (q < 2) *reinterpret_cast<bf16x4*>(o + (long long)(j * 16 + r) * (g.N >> 1) + i_base) = v;
        }
      }
    }
    unit += G;
    has = unit < n_units;
    if (has) {  // further units of this workgroup (more units than workgroups): no barrier to hide behind
      fg = unit % groups; sp = unit / groups;
      tile = (long long)fg * g.nw + wave;
      wp = reinterpret_cast<const dl_raw16*>(Wp) + tile * (long long)ksteps * 64 + lane;
      c0 = sp * g.cps; c1 = c0 + g.cps < nchunks ? c0 + g.cps : nchunks;
#pragma unroll
      for (int s = 0; s < PD; ++s)
        if (c0 + s < c1) DL_ISSUE_W(s, c0 + s);
    }
  }
#undef DL_ISSUE_W
#undef DL_ISSUE_X
#undef DL_CONSUME
}

template <int MT, int REP, bool KV8>
__global__ __launch_bounds__(512) void decode_layers_kernel(const DlParams p) {
  constexpr int XF = 4 * MT;
  __shared__ __attribute__((aligned(16))) dl_frag_ab xs[2 * XF * 64];
  __shared__ float rs_lds[16 * MT];
  __shared__ int dead;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int wg = blockIdx.x, G = gridDim.x;
  if (tid == 0) dead = 0;
  DlBarrier bar{p.flags, p.state, __hip_atomic_load(p.state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), G, wg};
  unsigned int seq = 0;
  __syncthreads();
  const int T = 16 * MT, H = p.H;
  const int parts = (H + 511) / 512;
  const int qkv_w = (p.nh + 2 * p.nkv) * 128;
  const float inv_h = 1.f / (float)H;
  const unsigned short* xin = p.residual;  // the residual stream a layer starts from
  const float* ssin = p.ss;
  for (int li = 0; li < p.n_layers; ++li) {
    const DlLayer L = p.layers[li];
    const bool last = li + 1 == p.n_layers;
    dl_stamp(p, li, 0, wg, tid);
    // ---- QKV projection: x = residual stream (the row scale 1/rms is applied by the attention phase)
    dl_gemm<MT, 4, 1>(L.wqkv, p.qkv, xin, H, p.qkv_slabs, nullptr, xs, seq, wg, G, wave, lane, [&]() {
      if (li > 0) {
        // the previous layer's down projection has arrived at its barrier just before this phase
        dl_wait(bar, tid, &dead);
        dl_stamp(p, li, 1, wg, tid);  // down projection of the previous layer done everywhere
        dl_reduce(p.proj_slabs, p.down.splits, (long long)T * H, p.xres, p.xres, p.ssb, H, T, wg, G, wave, lane);
        dl_arrive(bar, tid);
        dl_wait(bar, tid, &dead);
        dl_stamp(p, li, 2, wg, tid);  // residual update done everywhere
      }
    });
    dl_stamp(p, li, 3, wg, tid);      // own share of the QKV projection done
    dl_arrive(bar, tid);
    dl_wait(bar, tid, &dead);
    dl_stamp(p, li, 4, wg, tid);      // QKV done everywhere
    // ---- RoPE + KV write + attention: one (token, kv head) per workgroup at a time
    {
      const float* kvs = KV8 ? p.kv_scale + (long long)li * p.kv_scale_stride : nullptr;
      const int n_items = T * p.nkv;
      for (int item = wg; item < n_items; item += G) {
        decode_attention_body<REP, 8, KV8, false, false>(item / p.nkv, item % p.nkv, 0, 1, p.qkv_slabs, p.qkv.splits, (long long)T * qkv_w, p.positions, p.cos_t, p.sin_t,
                                                         L.k_cache, L.v_cache, p.slot_mapping, p.block_tables, p.max_blocks_per_seq, p.tok_seq, p.ctx_lens,
                                                         p.attn_out, p.nh, p.nkv, ssin, parts, inv_h, p.eps, p.scale_log2e, kvs, nullptr, nullptr, nullptr, nullptr,
                                                         0, nullptr);
        __syncthreads();
      }
    }
    dl_stamp(p, li, 5, wg, tid);      // own attention items done
    dl_arrive(bar, tid);
    // ---- output projection
    dl_gemm<MT, 4, 1>(L.wo, p.o, p.attn_out, (long long)p.nh * 128, p.proj_slabs, nullptr, xs, seq, wg, G, wave, lane, [&]() {
      dl_wait(bar, tid, &dead);
      dl_stamp(p, li, 6, wg, tid);    // attention done everywhere
    });
    dl_stamp(p, li, 7, wg, tid);      // own share of the output projection done
    dl_arrive(bar, tid);
    // ---- gate/up projection + SwiGLU; the residual update of the output projection runs while its first weights travel
    dl_gemm<MT, 6, 2>(L.wgu, p.gu, p.xres, H, p.act, rs_lds, xs, seq, wg, G, wave, lane, [&]() {
      dl_wait(bar, tid, &dead);
      dl_stamp(p, li, 8, wg, tid);    // output projection done everywhere
      dl_reduce(p.proj_slabs, p.o.splits, (long long)T * H, xin, p.xres, p.ssb, H, T, wg, G, wave, lane);
      dl_arrive(bar, tid);
      dl_wait(bar, tid, &dead);
      dl_stamp(p, li, 9, wg, tid);    // residual update done everywhere
      for (int row = wave; row < T; row += 8) {
        const float tot = row_ss_total(p.ssb + (long long)row * parts, parts, lane);
        if (lane == 0) rs_lds[row] = rsqrtf(tot * inv_h + p.eps);
      }
    });
    dl_stamp(p, li, 10, wg, tid);     // own share of the gate/up projection done
    dl_arrive(bar, tid);
    // ---- down projection
    dl_gemm<MT, 4, 1>(L.wdown, p.down, p.act, p.I, p.proj_slabs, nullptr, xs, seq, wg, G, wave, lane, [&]() {
      dl_wait(bar, tid, &dead);
      dl_stamp(p, li, 11, wg, tid);   // gate/up done everywhere
    });
    dl_stamp(p, li, 12, wg, tid);     // own share of the down projection done
    dl_arrive(bar, tid);
    if (last) {
      // the step's last residual update leaves through ordinary memory
      dl_wait(bar, tid, &dead);
      dl_reduce(p.proj_slabs, p.down.splits, (long long)T * H, p.xres, p.residual, p.ss, H, T, wg, G, wave, lane);
      dl_arrive(bar, tid);
      dl_wait(bar, tid, &dead);
    }
    xin = p.xres;
    ssin = p.ssb;
  }
  if (wg == 0 && tid == 0) __hip_atomic_store(p.state, bar.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace

// Pull a byte range through the memory hierarchy without using it: reads allocate in the
// memory-side Infinity Cache (256 MiB), so a weight matrix touched here while HBM would
// otherwise idle (attention / norm phases) is served from that cache when the next GEMM
// streams it.  A few workgroups suffice: every lane keeps several 16-byte loads in flight.
__global__ __launch_bounds__(256) void prefetch_kernel(const uint4* __restrict__ p, long long n16, unsigned int* __restrict__ sink) {
  const long long stride = (long long)gridDim.x * 256;
  unsigned int acc = 0;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n16; i += 4 * stride) {
    const uint4 a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
    acc ^= a.x ^ b.y ^ c.z ^ d.w;
  }
  for (; i < n16; i += stride) acc ^= p[i].x;
  if (acc == 0x9e3779b9u && sink != nullptr) sink[0] = acc;  // keeps the loads alive; practically never taken
}

extern "C" {

int kv_prefetch(const void* p, long long bytes, int workgroups, unsigned int* sink, hipStream_t stream) {
  if (bytes < 16 || workgroups < 1) return -1;
  hipLaunchKernelGGL(prefetch_kernel, dim3(workgroups), dim3(256), 0, stream, (const uint4*)p, bytes / 16, sink);
  return (int)hipGetLastError();
}

int kv_add_rmsnorm(const void* x, void* residual, const void* w, void* y, int T, int H, float eps, hipStream_t stream) {
  if (T <= 0) return 0;
  if (H % 8) return -1;
  size_t smem = 16 + (size_t)H * 2;
  hipLaunchKernelGGL(add_rmsnorm_kernel, dim3(T), dim3(256), smem, stream, (const unsigned short*)x, (unsigned short*)residual, (const unsigned short*)w,
                     (unsigned short*)y, H, eps);
  return (int)hipGetLastError();
}

int kv_add_rmsnorm_slabs(const float* slabs, int n_slabs, long long slab_stride, void* residual, const void* w, void* y, int T, int H, float eps,
                         hipStream_t stream) {
  if (T <= 0) return 0;
  if (H % 8 || n_slabs < 1) return -1;
  size_t smem = 16 + (size_t)H * 2;
  hipLaunchKernelGGL(add_rmsnorm_slab_kernel, dim3(T), dim3(256), smem, stream, slabs, n_slabs, slab_stride, (unsigned short*)residual,
                     (const unsigned short*)w, (unsigned short*)y, H, eps);
  return (int)hipGetLastError();
}

static inline int rope_grid_y(int nh, int nkv, int hd) {
  const int tasks = (nh + nkv) * (hd >> 4) + nkv * (hd >> 3);
  return (tasks + 63) / 64;
}

// row_ss (nullable): [T][ss_parts] partial sums of squares of the GEMM's input rows; when
// given, the summed projection is scaled by rsqrt(sum/H_in + eps) (folded RMSNorm).
int kv_rope_kv_write_slabs(const float* slabs, int n_slabs, long long slab_stride, void* q_out, const int* positions, const float* cos_t, const float* sin_t,
                           void* k_cache, void* v_cache, const int* slot_mapping, int T, int nh, int nkv, int hd, int block_size, const float* row_ss,
                           int ss_parts, int h_in, float eps, const float* kv_scale, hipStream_t stream) {
  if (T <= 0) return 0;
  if (hd % 16 || n_slabs < 1) return -1;
#define KV_ROPE(K8)                                                                                                                                             \
  hipLaunchKernelGGL((rope_kv_write_kernel<true, K8>), dim3(T, rope_grid_y(nh, nkv, hd)), dim3(64), 0, stream, (unsigned short*)nullptr, slabs, n_slabs, slab_stride, \
                     (unsigned short*)q_out, positions, cos_t, sin_t, k_cache, v_cache, slot_mapping, nh, nkv, hd, block_size, row_ss, ss_parts,                  \
                     h_in > 0 ? 1.f / (float)h_in : 0.f, eps, kv_scale)
  if (kv_scale) KV_ROPE(true); else KV_ROPE(false);
#undef KV_ROPE
  return (int)hipGetLastError();
}

// x: bf16 [T, H] (n_slabs == 0) or fp32 slabs [n_slabs][T][H]; residual: bf16 [T, H] in/out
// (has_residual == 0: treated as zero and only written); ss_part: float [T][ceil(H/512)].
// pf_base / pf (optional): the packed weights of the GEMM that runs next and its plan
// {gx, gy, nw, kib, tile_bytes, split_bytes, prefetch workgroups, weight bytes}; see PrefetchPlan.
int kv_residual_reduce(const float* slabs, int n_slabs, long long slab_stride, const void* x, void* residual, int has_residual, float* ss_part, int T, int H,
                       const void* pf_base, const long long* pf, unsigned int* pf_sink, hipStream_t stream) {
  if (T <= 0) return 0;
  if (H % 8) return -1;
  dim3 grid((H + 511) / 512, T);
  PrefetchPlan plan{};
  if (pf_base != nullptr && pf != nullptr) {
    if (make_prefetch_plan(pf_base, pf, pf_sink, &plan) != 0) return -1;
    grid.y += (plan.wgs + grid.x - 1) / grid.x;
  }
  if (n_slabs > 0)
    hipLaunchKernelGGL(residual_reduce_kernel<true>, grid, dim3(64), 0, stream, slabs, n_slabs, slab_stride, (const unsigned short*)nullptr,
                       (unsigned short*)residual, has_residual, ss_part, H, T, plan);
  else
    hipLaunchKernelGGL(residual_reduce_kernel<false>, grid, dim3(64), 0, stream, (const float*)nullptr, 0, 0LL, (const unsigned short*)x,
                       (unsigned short*)residual, has_residual, ss_part, H, T, plan);
  return (int)hipGetLastError();
}

int kv_rope_kv_write(void* qkv, const int* positions, const float* cos_t, const float* sin_t, void* k_cache, void* v_cache, const int* slot_mapping, int T,
                     int nh, int nkv, int hd, int block_size, const float* kv_scale, hipStream_t stream) {
  if (T <= 0) return 0;
  if (hd % 16) return -1;
#define KV_ROPE(K8)                                                                                                                                               \
  hipLaunchKernelGGL((rope_kv_write_kernel<false, K8>), dim3(T, rope_grid_y(nh, nkv, hd)), dim3(64), 0, stream, (unsigned short*)qkv, (const float*)nullptr, 0, 0LL, \
                     (unsigned short*)nullptr, positions, cos_t, sin_t, k_cache, v_cache, slot_mapping, nh, nkv, hd, block_size, (const float*)nullptr, 0, 0.f,      \
                     0.f, kv_scale)
  if (kv_scale) KV_ROPE(true); else KV_ROPE(false);
#undef KV_ROPE
  return (int)hipGetLastError();
}

int kv_paged_attention(const void* q, long long q_row_stride, const void* k_cache, const void* v_cache, const int* block_tables, int max_blocks_per_seq,
                       const int* tok_seq, const int* ctx_lens, void* out, int T, int nh, int nkv, int hd, int block_size, float scale,
                       const float* kv_scale, hipStream_t stream) {
  if (T <= 0) return 0;
  if ((hd != 128 && hd != 64) || block_size != 16 || nh % nkv) return -1;
  const int rep = nh / nkv;
  const float sl = scale * 1.4426950408889634f;
  // 8 waves per workgroup (twice the K/V blocks in flight) where the head group's registers fit
  // 2 waves per SIMD and the grid does not fill the 256 CUs on its own
  const bool wide = rep <= 4 && (long long)T * nkv <= 256;
  dim3 grid(T, nkv), block(wide ? 512 : 256);
#define KV_LAUNCH_K(R, D, W, K8)                                                                                                          \
  hipLaunchKernelGGL((paged_attention_kernel<R, D, W, K8>), grid, block, 0, stream, (const unsigned short*)q, q_row_stride, k_cache, v_cache, \
                     block_tables, max_blocks_per_seq, tok_seq, ctx_lens, (unsigned short*)out, nh, nkv, sl, kv_scale)
#define KV_LAUNCH_W(R, D, W)       \
  if (kv_scale) {                  \
    KV_LAUNCH_K(R, D, W, true);    \
  } else {                         \
    KV_LAUNCH_K(R, D, W, false);   \
  }
#define KV_LAUNCH(R, D)                       \
  if (wide) {                                 \
    KV_LAUNCH_W(R, D, ((R) <= 4 ? 8 : 4));    \
  } else {                                    \
    KV_LAUNCH_W(R, D, 4);                     \
  }
#define KV_BY_REP(D)              \
  switch (rep) {                  \
    case 1: KV_LAUNCH(1, D); break; \
    case 2: KV_LAUNCH(2, D); break; \
    case 3: KV_LAUNCH(3, D); break; \
    case 4: KV_LAUNCH(4, D); break; \
    case 6: KV_LAUNCH(6, D); break; \
    case 7: KV_LAUNCH(7, D); break; \
    case 8: KV_LAUNCH(8, D); break; \
    default: return -1;           \
  }
  if (hd == 128) { KV_BY_REP(128) } else { KV_BY_REP(64) }
#undef KV_BY_REP
#undef KV_LAUNCH
#undef KV_LAUNCH_W
#undef KV_LAUNCH_K
  return (int)hipGetLastError();
}

// Tiled causal attention for prompt chunks (see prefill_attention_kernel).  Tile arrays have
// n_tiles entries; rows of `out` not covered by a tile are left untouched.
int kv_prefill_attention(const void* q, long long q_row_stride, const void* k_cache, const void* v_cache, const int* block_tables, int max_blocks_per_seq,
                         const int* tile_row0, const int* tile_n, const int* tile_pos0, const int* tile_seq, void* out, int n_tiles, int nh, int nkv, int hd,
                         int block_size, float scale, const float* kv_scale, int quads, hipStream_t stream) {
  if (n_tiles <= 0) return 0;
  if (hd != 128 || block_size != 16 || nh % nkv) return -1;
  const int rep = nh / nkv;
  const float sl = scale * 1.4426950408889634f;
  if (quads) {
    // every aligned group of four tiles is consecutive tiles of ONE chunk (the caller pads each
    // chunk's tiles to a multiple of four with unused ones): the 8-wave kernel, head groups of at most 4
    if (rep > 4 || (n_tiles & 3)) return -1;
    dim3 grid(nkv, n_tiles / 4), block(512);
#define KV_LAUNCH_QUADS(R)                                                                                                                           \
  if (kv_scale)                                                                                                                                      \
    hipLaunchKernelGGL((prefill_attention_quads_kernel<R, true>), grid, block, 0, stream, (const unsigned short*)q, q_row_stride, k_cache, v_cache,   \
                       block_tables, max_blocks_per_seq, tile_row0, tile_n, tile_pos0, tile_seq, (unsigned short*)out, nh, nkv, sl, kv_scale);        \
  else                                                                                                                                               \
    hipLaunchKernelGGL((prefill_attention_quads_kernel<R, false>), grid, block, 0, stream, (const unsigned short*)q, q_row_stride, k_cache, v_cache,  \
                       block_tables, max_blocks_per_seq, tile_row0, tile_n, tile_pos0, tile_seq, (unsigned short*)out, nh, nkv, sl, kv_scale)
    switch (rep) {
      case 1: KV_LAUNCH_QUADS(1); break;
      case 2: KV_LAUNCH_QUADS(2); break;
      case 3: KV_LAUNCH_QUADS(3); break;
      case 4: KV_LAUNCH_QUADS(4); break;
      default: return -1;
    }
#undef KV_LAUNCH_QUADS
    return (int)hipGetLastError();
  }
  // A wave's MFMA tile has 32 columns.  Head groups of at most 4 (one head per wave) fill them with
  // two 16-query tiles per workgroup where the halved grid still fills the chip several times
  // over, and otherwise run one tile per workgroup on the 16-column path; head groups of 5-8 fill
  // them with the two heads of a wave x one tile.
  // (KVMINI_PREFILL_TILES=1|2 pins the choice, for measurements)
  static const int pinned = [] { const char* e = getenv("KVMINI_PREFILL_TILES"); return e ? atoi(e) : 0; }();
  const int qt = rep > 4 ? 1 : (pinned == 1 || pinned == 2 ? pinned : ((long long)nkv * n_tiles >= 512 ? 2 : 1));
  dim3 grid(nkv, (n_tiles + qt - 1) / qt), block(256);
  const int order = (long long)grid.x * grid.y <= 512 ? 1 : 0;
#define KV_LAUNCH_K(R, Q, K8)                                                                                                              \
  hipLaunchKernelGGL((prefill_attention_kernel<R, Q, K8>), grid, block, 0, stream, (const unsigned short*)q, q_row_stride, k_cache, v_cache,   \
                     block_tables, max_blocks_per_seq, tile_row0, tile_n, tile_pos0, tile_seq, n_tiles, (unsigned short*)out, nh, nkv, sl, kv_scale, order)
#define KV_LAUNCH_Q(R, Q)       \
  if (kv_scale) {               \
    KV_LAUNCH_K(R, Q, true);    \
  } else {                      \
    KV_LAUNCH_K(R, Q, false);   \
  }
#define KV_LAUNCH(R)                          \
  if (qt == 2) {                              \
    KV_LAUNCH_Q(R, ((R) <= 4 ? 2 : 1));       \
  } else {                                    \
    KV_LAUNCH_Q(R, 1);                        \
  }
  switch (rep) {
    case 1: KV_LAUNCH(1); break;
    case 2: KV_LAUNCH(2); break;
    case 3: KV_LAUNCH(3); break;
    case 4: KV_LAUNCH(4); break;
    case 6: KV_LAUNCH(6); break;
    case 7: KV_LAUNCH(7); break;
    case 8: KV_LAUNCH(8); break;
    default: return -1;
  }
#undef KV_LAUNCH
#undef KV_LAUNCH_Q
#undef KV_LAUNCH_K
  return (int)hipGetLastError();
}

// The split form of the tiled prompt attention (prefill_attention_split_kernel + combine): the first n_tiles entries of
// the tile arrays, every tile's key range in n_split shares.  part_o: fp32 [n_split][16 * n_tiles][nh][128], part_ml:
// fp32 [n_split][16 * n_tiles][nh][2] (the caller's workspace).
int kv_prefill_attention_split(const void* q, long long q_row_stride, const void* k_cache, const void* v_cache, const int* block_tables, int max_blocks_per_seq,
                               const int* tile_row0, const int* tile_n, const int* tile_pos0, const int* tile_seq, void* out, int n_tiles, int nh, int nkv, int hd,
                               int block_size, float scale, const float* kv_scale, int n_split, float* part_o, float* part_ml, hipStream_t stream) {
  if (n_tiles <= 0) return 0;
  if (hd != 128 || block_size != 16 || nh % nkv || n_split < 1 || n_split > 64 || !part_o || !part_ml) return -1;
  const int rep = nh / nkv;
  const float sl = scale * 1.4426950408889634f;
  const int qt = rep <= 4 ? 2 : 1;
  dim3 grid(nkv, (n_tiles + qt - 1) / qt, n_split), block(256);
#define KV_LAUNCH_S(R)                                                                                                                              \
  if (kv_scale)                                                                                                                                     \
    hipLaunchKernelGGL((prefill_attention_split_kernel<R, true>), grid, block, 0, stream, (const unsigned short*)q, q_row_stride, k_cache, v_cache,  \
                       block_tables, max_blocks_per_seq, tile_row0, tile_n, tile_pos0, tile_seq, n_tiles, nh, nkv, sl, kv_scale, part_o, part_ml);   \
  else                                                                                                                                              \
    hipLaunchKernelGGL((prefill_attention_split_kernel<R, false>), grid, block, 0, stream, (const unsigned short*)q, q_row_stride, k_cache, v_cache, \
                       block_tables, max_blocks_per_seq, tile_row0, tile_n, tile_pos0, tile_seq, n_tiles, nh, nkv, sl, kv_scale, part_o, part_ml)
  switch (rep) {
    case 1: KV_LAUNCH_S(1); break;
    case 2: KV_LAUNCH_S(2); break;
    case 3: KV_LAUNCH_S(3); break;
    case 4: KV_LAUNCH_S(4); break;
    case 6: KV_LAUNCH_S(6); break;
    case 7: KV_LAUNCH_S(7); break;
    case 8: KV_LAUNCH_S(8); break;
    default: return -1;
  }
#undef KV_LAUNCH_S
  hipLaunchKernelGGL(prefill_split_combine_kernel, dim3(n_tiles, 16, (nh * 32 + 255) / 256), dim3(256), 0, stream, (const float*)part_o, (const float*)part_ml, n_split, n_tiles, tile_row0,
                     tile_n, (unsigned short*)out, nh);
  return (int)hipGetLastError();
}

// Split-KV launches (long contexts) of head groups of 4 and 8 over an e4m3 cache take their scores from the matrix cores
// when this is set (decode_attention_body, MF); the dot-product form stays as the A/B, for the other group sizes and
// for the bf16 cache (bound by the KV stream already; measured slower in the matrix form: 8.0 against 6.5 ms per step
// at 128k tokens, the three bf16 stages of the ring cost too many registers).
static int g_attention_mfma = 1;
// The unsplit fused decode kernel takes its scores from the matrix cores too on an e4m3 cache (KVMINI_FUSED_MFMA_UNSPLIT=0:
// the dot-product form).  Measured, Llama-3-8B decode step at batch 20 / 32, dot products -> matrix cores: 98 tokens of context
// 3.41 -> 3.42 / 3.48 -> 3.50 ms, 1060 tokens 3.85 -> 3.80 / 3.92 -> 3.90, 4150 tokens 5.14 -> 4.87 / 5.27 -> 5.07.
static int g_attention_mfma_unsplit = [] { const char* e = getenv("KVMINI_FUSED_MFMA_UNSPLIT"); return e ? (e[0] == '1' ? 1 : 0) : 1; }();
void kv_set_attention_mfma_unsplit(int on) { g_attention_mfma_unsplit = on ? 1 : 0; }
void kv_set_attention_mfma(int on) { g_attention_mfma = on ? 1 : 0; }
int kv_get_attention_mfma() { return g_attention_mfma; }

// slabs: fp32 [n_slabs][T][(nh + 2 nkv) * 128] split-K partial sums of the fused QKV projection.
// Valid only when every token of the step is the sole new token of its sequence.
int kv_decode_attention_fused(const float* slabs, int n_slabs, long long slab_stride, const int* positions, const float* cos_t, const float* sin_t, void* k_cache,
                              void* v_cache, const int* slot_mapping, const int* block_tables, int max_blocks_per_seq, const int* tok_seq, const int* ctx_lens,
                              void* out, int T, int nh, int nkv, int hd, int block_size, const float* row_ss, int ss_parts, int h_in, float eps, float scale,
                              int waves, const float* kv_scale, int n_split, float* part_o, float* part_ml, int* counters, const void* pf_base,
                              const long long* pf, unsigned int* pf_sink, hipStream_t stream) {
  if (T <= 0) return 0;
  if (hd != 128 || block_size != 16 || nh % nkv || n_slabs < 1 || n_slabs > 16 || (waves != 4 && waves != 8)) return -1;
  if (n_split < 1 || n_split > 64 || (n_split > 1 && (!part_o || !part_ml || !counters))) return -1;
  const int rep = nh / nkv;
  const float sl = scale * 1.4426950408889634f;
  const float inv_h = h_in > 0 ? 1.f / (float)h_in : 0.f;
  if (rep > 4) waves = 4;  // 8 waves of the wider head groups do not fit the 256 registers per lane that 2 waves per SIMD leave
  dim3 grid(T, nkv), block(64 * waves);
  PrefetchPlan plan{};
  if (n_split > 1) {
    grid = dim3(n_split, nkv, T);  // splits fastest: see the kernel
  } else if (pf_base != nullptr && pf != nullptr) {  // (long-context steps are bound by the KV stream themselves: no prefetch beside the split variant)
    if (make_prefetch_plan(pf_base, pf, pf_sink, &plan) != 0) return -1;
    grid.y += (plan.wgs + T - 1) / T;
  }
#define KV_LAUNCH_S(R, W, K8, SP)                                                                                                                        \
  hipLaunchKernelGGL((decode_attention_fused_kernel<R, W, K8, SP>), grid, block, 0, stream, slabs, n_slabs, slab_stride, positions, cos_t, sin_t, k_cache, \
                     v_cache, slot_mapping, block_tables, max_blocks_per_seq, tok_seq, ctx_lens, (unsigned short*)out, nh, nkv, row_ss, ss_parts,        \
                     inv_h, eps, sl, kv_scale, part_o, part_ml, counters, plan)
#define KV_LAUNCH_MF(R, W, K8)                                                                                                                            \
  hipLaunchKernelGGL((decode_attention_fused_kernel<R, W, K8, true, false, true>), grid, block, 0, stream, slabs, n_slabs, slab_stride, positions, cos_t, sin_t, \
                     k_cache, v_cache, slot_mapping, block_tables, max_blocks_per_seq, tok_seq, ctx_lens, (unsigned short*)out, nh, nkv, row_ss, ss_parts,      \
                     inv_h, eps, sl, kv_scale, part_o, part_ml, counters, plan)
#define KV_LAUNCH_K(R, W, K8)       \
  if (n_split > 1) {                \
    KV_LAUNCH_S(R, W, K8, true);    \
  } else {                          \
    KV_LAUNCH_S(R, W, K8, false);   \
  }
#define KV_LAUNCH_MF1(R, W, K8)                                                                                                                           \
  hipLaunchKernelGGL((decode_attention_fused_kernel<R, W, K8, false, false, true>), grid, block, 0, stream, slabs, n_slabs, slab_stride, positions, cos_t, sin_t, \
                     k_cache, v_cache, slot_mapping, block_tables, max_blocks_per_seq, tok_seq, ctx_lens, (unsigned short*)out, nh, nkv, row_ss, ss_parts,      \
                     inv_h, eps, sl, kv_scale, part_o, part_ml, counters, plan)
#define KV_LAUNCH_KM(R, W, K8)                        \
  if (n_split > 1 && g_attention_mfma) {              \
    KV_LAUNCH_MF(R, W, K8);                           \
  } else if (g_attention_mfma && g_attention_mfma_unsplit) { \
    KV_LAUNCH_MF1(R, W, K8);                          \
  } else {                                            \
    KV_LAUNCH_K(R, W, K8);                            \
  }
#define KV_LAUNCH_WM(R, W)      \
  if (kv_scale) {               \
    KV_LAUNCH_KM(R, W, true);   \
  } else {                      \
    KV_LAUNCH_K(R, W, false);   \
  }
#define KV_LAUNCHM(R)     \
  if (waves == 8) {       \
    KV_LAUNCH_WM(R, ((R) <= 4 ? 8 : 4)); \
  } else {                \
    KV_LAUNCH_WM(R, 4);   \
  }
#define KV_LAUNCH_W(R, W)       \
  if (kv_scale) {               \
    KV_LAUNCH_K(R, W, true);    \
  } else {                      \
    KV_LAUNCH_K(R, W, false);   \
  }
#define KV_LAUNCH(R)      \
  if (waves == 8) {       \
    KV_LAUNCH_W(R, ((R) <= 4 ? 8 : 4)); /* wider head groups exist with 4 waves only */ \
  } else {                \
    KV_LAUNCH_W(R, 4);    \
  }
  switch (rep) {
    case 1: KV_LAUNCH(1); break;
    case 2: KV_LAUNCH(2); break;
    case 3: KV_LAUNCH(3); break;
    case 4: KV_LAUNCHM(4); break;
    case 6: KV_LAUNCH(6); break;
    case 7: KV_LAUNCH(7); break;
    case 8: KV_LAUNCHM(8); break;
    default: return -1;
  }
#undef KV_LAUNCH
#undef KV_LAUNCH_W
#undef KV_LAUNCH_K
#undef KV_LAUNCH_MF
#undef KV_LAUNCH_MF1
#undef KV_LAUNCH_KM
#undef KV_LAUNCH_WM
#undef KV_LAUNCHM
#undef KV_LAUNCH_S
  return (int)hipGetLastError();
}

// Split-KV attention for the decode tokens of a step that also carries prompt chunks: q is rotated
// and the step's keys / values are in the cache (rope_kv_write ran).  `row_list` (n_rows entries,
// -1 = unused) names the rows; rows outside the list are not written.
int kv_split_decode_attention(const void* q, long long q_row_stride, const void* k_cache, const void* v_cache, const int* block_tables, int max_blocks_per_seq,
                              const int* tok_seq, const int* ctx_lens, const int* row_list, int n_rows, void* out, int nh, int nkv, int hd, int block_size, float scale,
                              int waves, const float* kv_scale, int n_split, float* part_o, float* part_ml, int* counters, hipStream_t stream) {
  if (n_rows <= 0) return 0;
  if (hd != 128 || block_size != 16 || nh % nkv || (waves != 4 && waves != 8) || !row_list) return -1;
  if (n_split < 1 || n_split > 64 || !part_o || !part_ml || !counters) return -1;
  const int rep = nh / nkv;
  const float sl = scale * 1.4426950408889634f;
  if (rep > 4) waves = 4;
  const dim3 grid(n_split, nkv, n_rows), block(64 * waves);
  const PrefetchPlan plan{};
#define KV_LAUNCH_S(R, W, K8)                                                                                                                                  \
  hipLaunchKernelGGL((decode_attention_fused_kernel<R, W, K8, true, true>), grid, block, 0, stream, (const float*)nullptr, 0, 0LL, (const int*)nullptr,        \
                     (const float*)nullptr, (const float*)nullptr, const_cast<void*>(k_cache), const_cast<void*>(v_cache), (const int*)nullptr, block_tables, \
                     max_blocks_per_seq, tok_seq, ctx_lens, (unsigned short*)out, nh, nkv, (const float*)nullptr, 0, 0.f, 0.f, sl, kv_scale, part_o, part_ml,   \
                     counters, plan, (const unsigned short*)q, q_row_stride, row_list)
#define KV_LAUNCH_MF(R, W, K8)                                                                                                                                 \
  hipLaunchKernelGGL((decode_attention_fused_kernel<R, W, K8, true, true, true>), grid, block, 0, stream, (const float*)nullptr, 0, 0LL, (const int*)nullptr,  \
                     (const float*)nullptr, (const float*)nullptr, const_cast<void*>(k_cache), const_cast<void*>(v_cache), (const int*)nullptr, block_tables, \
                     max_blocks_per_seq, tok_seq, ctx_lens, (unsigned short*)out, nh, nkv, (const float*)nullptr, 0, 0.f, 0.f, sl, kv_scale, part_o, part_ml,   \
                     counters, plan, (const unsigned short*)q, q_row_stride, row_list)
#define KV_LAUNCH_X(R, W, K8)                             \
  if (g_attention_mfma) { KV_LAUNCH_MF(R, W, K8); }       \
  else { KV_LAUNCH_S(R, W, K8); }
#define KV_LAUNCH_W(R, W)     \
  if (kv_scale) {             \
    KV_LAUNCH_S(R, W, true);  \
  } else {                    \
    KV_LAUNCH_S(R, W, false); \
  }
#define KV_LAUNCH_WM(R, W)    \
  if (kv_scale) {             \
    KV_LAUNCH_X(R, W, true);  \
  } else {                    \
    KV_LAUNCH_S(R, W, false); \
  }
#define KV_LAUNCHM(R)                       \
  if (waves == 8) {                         \
    KV_LAUNCH_WM(R, ((R) <= 4 ? 8 : 4));    \
  } else {                                  \
    KV_LAUNCH_WM(R, 4);                     \
  }
#define KV_LAUNCH(R)                        \
  if (waves == 8) {                         \
    KV_LAUNCH_W(R, ((R) <= 4 ? 8 : 4));     \
  } else {                                  \
    KV_LAUNCH_W(R, 4);                      \
  }
  switch (rep) {
    case 1: KV_LAUNCH(1); break;
    case 2: KV_LAUNCH(2); break;
    case 3: KV_LAUNCH(3); break;
    case 4: KV_LAUNCHM(4); break;
    case 6: KV_LAUNCH(6); break;
    case 7: KV_LAUNCH(7); break;
    case 8: KV_LAUNCHM(8); break;
    default: return -1;
  }
#undef KV_LAUNCH
#undef KV_LAUNCH_W
#undef KV_LAUNCH_WM
#undef KV_LAUNCHM
#undef KV_LAUNCH_X
#undef KV_LAUNCH_MF
#undef KV_LAUNCH_S
  return (int)hipGetLastError();
}

int kv_argmax(const void* logits, long long row_stride, int S, int V, long long* out, hipStream_t stream) {
  if (S <= 0) return 0;
  hipLaunchKernelGGL(argmax_kernel, dim3(S), dim3(1024), 0, stream, (const unsigned short*)logits, row_stride, V, out);
  return (int)hipGetLastError();
}

int kv_embed_fed(const void* table, const int* tokens, const int* feed, const long long* prev_ids, int n_prev, void* out, int T, int H, int V, hipStream_t stream) {
  if (T <= 0) return 0;
  if (H % 8) return -1;
  hipLaunchKernelGGL(embed_fed_kernel, dim3(T), dim3(256), 0, stream, (const unsigned short*)table, tokens, feed, prev_ids, n_prev, (unsigned short*)out, H, V);
  return (int)hipGetLastError();
}

int kv_scale_rows(const void* x, const float* row_ss, int ss_parts, float eps, void* y, int T, int H, hipStream_t stream) {
  if (T <= 0) return 0;
  if (H % 8 || ss_parts < 1) return -1;
  hipLaunchKernelGGL(scale_rows_kernel, dim3(T), dim3(256), 0, stream, (const unsigned short*)x, row_ss, ss_parts, 1.f / (float)H, eps, (unsigned short*)y, H);
  return (int)hipGetLastError();
}

int kv_slab_add_bias(float* slab, long long row_stride, const float* bias, const float* row_ss, int ss_parts, float inv_h, float eps, int T, int N, hipStream_t stream) {
  if (T <= 0 || N <= 0) return 0;
  if (!slab || !bias || (row_ss != nullptr && ss_parts < 1)) return -1;
  hipLaunchKernelGGL(slab_add_bias_kernel, dim3(T), dim3(256), 0, stream, slab, row_stride, bias, row_ss, ss_parts, inv_h, eps, N);
  return (int)hipGetLastError();
}

int kv_swiglu_interleaved(const void* gu, void* act, int T, int I, hipStream_t stream) {
  if (T <= 0) return 0;
  if (I % 8) return -1;
  dim3 grid(((I >> 3) + 255) / 256, T);
  hipLaunchKernelGGL(swiglu_interleaved_kernel, grid, dim3(256), 0, stream, (const unsigned short*)gu, (unsigned short*)act, I);
  return (int)hipGetLastError();
}


// Bytes of the uncached workspace of kv_decode_layers for a model (rows up to 64), and its layout.
struct DlLayout { long long flags, state, xres, ssb, qkv_slabs, attn_out, proj_slabs, act, timeline, total; };
static DlLayout dl_layout(int H, int nh, int nkv, int I, int s_qkv, int s_proj) {
  auto up = [](long long v) { return (v + 255) / 256 * 256; };
  DlLayout l{};
  const long long R = 64, parts = (H + 511) / 512;
  long long off = 0;
  l.flags = off; off = up(off + 512 * 4);
  l.state = off; off = up(off + 64);
  l.xres = off; off = up(off + R * H * 2);
  l.ssb = off; off = up(off + R * parts * 4);
  l.qkv_slabs = off; off = up(off + (long long)s_qkv * R * (nh + 2 * nkv) * 128 * 4);
  l.attn_out = off; off = up(off + R * nh * 128 * 2);
  l.proj_slabs = off; off = up(off + (long long)s_proj * R * H * 4);
  l.act = off; off = up(off + R * I * 2);
  l.timeline = off; off = up(off + 128 * 16 * 8);  // up to 128 layers
  l.total = off;
  return l;
}

long long kv_decode_layers_workspace_bytes(int H, int nh, int nkv, int I, int s_qkv, int s_proj) { return dl_layout(H, nh, nkv, I, s_qkv, s_proj).total; }

// All decoder layers of a plain decoding step (every token the only new one of its sequence) in one
// persistent launch: see decode_layers_kernel.  layers: device array of n_layers x 6 pointers (packed
// bf16 wqkv, wo, wgu, wdown, K cache, V cache of the layer); plans: 4 x {N, K, waves, splits, chunks
// per split} for the QKV, output, gate/up and down projections; residual / ss: the residual stream
// and its per-row sums of squares (in and out); ws: kv_decode_layers_workspace_bytes of UNCACHED,
// initially zeroed device memory; grid: workgroups (at most one per CU, at most 512).
// Returns 0, -1 for shapes outside the kernel's set.  ws word [state + 4] is non-zero after a launch
// whose barrier timed out.
int kv_decode_layers(const void* layers, int n_layers, int T, int H, int nh, int nkv, int I, const int* plans, const int* positions, const float* cos_t,
                     const float* sin_t, const int* slot_mapping, const int* block_tables, int max_blocks_per_seq, const int* tok_seq, const int* ctx_lens, float eps,
                     float scale, const float* kv_scale, void* residual, float* ss, void* ws, int grid, int timeline, hipStream_t stream) {
  if (!layers || n_layers < 1 || !plans || !ws || !residual || !ss || grid < 1 || grid > 512) return -1;
  if (T != 16 && T != 32 && T != 48 && T != 64) return -1;
  if (nkv < 1 || nh % nkv || nh / nkv != 4 || (H & 127)) return -1;  // head groups of 4 (Llama-3-8B, Mistral-7B); others keep the kernel-per-phase path
  DlParams p{};
  DlGemm* gp[4] = {&p.qkv, &p.o, &p.gu, &p.down};
  for (int i = 0; i < 4; ++i) {
    DlGemm& g = *gp[i];
    g.N = plans[5 * i]; g.K = plans[5 * i + 1]; g.nw = plans[5 * i + 2]; g.splits = plans[5 * i + 3]; g.cps = plans[5 * i + 4];
    if (g.N <= 0 || g.K <= 0 || (g.K & 127) || g.nw < 1 || g.nw > 8 || g.N % (16 * g.nw) || g.splits < 1 || g.splits > 16 || g.cps < 1) return -1;
    const int nchunks = g.K / 128;
    if ((long long)g.splits * g.cps < nchunks || (long long)(g.splits - 1) * g.cps >= nchunks) return -1;
  }
  if (p.qkv.N != (nh + 2 * nkv) * 128 || p.qkv.K != H || p.o.N != H || p.o.K != nh * 128 || p.gu.N != 2 * I || p.gu.K != H || p.gu.splits != 1 || p.down.N != H ||
      p.down.K != I)
    return -1;
  const DlLayout l = dl_layout(H, nh, nkv, I, p.qkv.splits, p.o.splits > p.down.splits ? p.o.splits : p.down.splits);
  char* w = reinterpret_cast<char*>(ws);
  p.layers = reinterpret_cast<const DlLayer*>(layers);
  p.n_layers = n_layers; p.T = T; p.H = H; p.nh = nh; p.nkv = nkv; p.I = I;
  p.positions = positions; p.cos_t = cos_t; p.sin_t = sin_t; p.slot_mapping = slot_mapping; p.block_tables = block_tables; p.max_blocks_per_seq = max_blocks_per_seq;
  p.tok_seq = tok_seq; p.ctx_lens = ctx_lens;
  p.eps = eps; p.scale_log2e = scale * 1.4426950408889634f;
  p.kv_scale = kv_scale; p.kv_scale_stride = 2 * nkv;
  p.residual = (unsigned short*)residual; p.ss = ss;
  p.xres = (unsigned short*)(w + l.xres); p.ssb = (float*)(w + l.ssb); p.qkv_slabs = (float*)(w + l.qkv_slabs); p.attn_out = (unsigned short*)(w + l.attn_out);
  p.proj_slabs = (float*)(w + l.proj_slabs); p.act = (unsigned short*)(w + l.act);
  p.flags = (unsigned int*)(w + l.flags); p.state = (unsigned int*)(w + l.state);
  p.timeline = (timeline && n_layers <= 128) ? (unsigned long long*)(w + l.timeline) : nullptr;
#define KV_DL(MTv)                                                                                                    \
  if (kv_scale) hipLaunchKernelGGL((decode_layers_kernel<MTv, 4, true>), dim3(grid), dim3(512), 0, stream, p);        \
  else hipLaunchKernelGGL((decode_layers_kernel<MTv, 4, false>), dim3(grid), dim3(512), 0, stream, p)
  switch (T >> 4) {
    case 1: KV_DL(1); break;
    case 2: KV_DL(2); break;
    case 3: KV_DL(3); break;
    case 4: KV_DL(4); break;
  }
#undef KV_DL
  return (int)hipGetLastError();
}

// byte offsets of the workspace regions: flags, state, xres, ssb, qkv_slabs, attn_out, proj_slabs, act, total (tests and debugging)
void kv_decode_layers_layout(int H, int nh, int nkv, int I, int s_qkv, int s_proj, long long* out9) {
  const DlLayout l = dl_layout(H, nh, nkv, I, s_qkv, s_proj);
  const long long v[10] = {l.flags, l.state, l.xres, l.ssb, l.qkv_slabs, l.attn_out, l.proj_slabs, l.act, l.total, l.timeline};
  for (int i = 0; i < 10; ++i) out9[i] = v[i];
}

// byte offset of the two state words (epoch, error) inside the workspace
long long kv_decode_layers_state_offset(int H, int nh, int nkv, int I, int s_qkv, int s_proj) { return dl_layout(H, nh, nkv, I, s_qkv, s_proj).state; }

}  // extern "C"

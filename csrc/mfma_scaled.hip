// The gfx950 block-scaled matrix instruction (v_mfma_scale_f32_16x16x128_f8f6f4) and what feeds it:
//   * a probe that runs ONE instruction on caller-supplied operand registers, so that a test can
//     pin the operand lane map with exact integer data (the tiled GEMM's FP8 x FP8 and MXFP4 x FP8
//     paths in csrc/wide_gemm.hip rely on the map stated below);
//   * per-token dynamic FP8 (e4m3) quantisation of activation rows, written in the k order the
//     packed FP8 weights use.
//
// Operand map of the 16x16x128 form (one-hot probes, scripts/mfma_scale_map.py; pinned by a GPU test).
// Lane l = (r = l & 15, q = l >> 4) supplies row r of A / column r of B:
//   8-bit formats: bytes 0..15 of its eight operand dwords are k = 16 q + j, bytes 16..31 are k = 64 + 16 q + j;
//   FP4          : nibble j (low nibble first) of its first four dwords is k = 32 q + j;
//   scales       : the E8M0 byte (byte OPSEL of the scale dword) of lane (r, b) scales the 32 elements
//                  k = 32 b .. 32 b + 31 of row / column r;
//   C/D          : lane (c, q), register e = D[row 4 q + e][column c].
// With unit scales only the PAIRING matters to a GEMM: A and B must hold the same k at the same (lane
// quarter, byte) — which is how the FP8 x FP8 kernel uses it, with its own k order inside a stage.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

typedef __attribute__((ext_vector_type(8))) int v8i;
typedef __attribute__((ext_vector_type(4))) float v4f;
typedef __attribute__((ext_vector_type(4))) unsigned int raw16;
typedef __attribute__((ext_vector_type(2))) unsigned int raw8;

template <int FA, int FB>
__global__ __launch_bounds__(64) void mfma_scale_probe_kernel(const int* __restrict__ a, const int* __restrict__ b, const int* __restrict__ sa, const int* __restrict__ sb,
                                                             float* __restrict__ out) {
  const int l = threadIdx.x;
  v8i va, vb;
#pragma unroll
  for (int i = 0; i < 8; ++i) { va[i] = a[l * 8 + i]; vb[i] = b[l * 8 + i]; }
  v4f c = (v4f){0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(va, vb, c, FA, FB, 0, sa[l], 0, sb[l]);
#pragma unroll
  for (int e = 0; e < 4; ++e) out[l * 4 + e] = c[e];
}

__device__ inline float bf2f(unsigned short b) { return __uint_as_float(((unsigned int)b) << 16); }

// One workgroup (256 threads) per token row.  x: bf16 [M][ldx]; x8: e4m3 [M][K]; scale[t] = absmax / 448
// (1 for an all-zero row), so x ~= x8 * scale[t].  Inside every group of 64 K-elements the bytes
// are stored in the order of the packed FP8 weight fragments (kv_pack_weight_fp8): byte 16 q + j
// holds k = 8 q + j for j < 8 and k = 32 + 8 q + (j - 8) for j >= 8, so that a 16-byte chunk of
// the row is exactly what lane quarter q of a weight fragment multiplies.
template <bool PERMUTE>
__global__ __launch_bounds__(256) void quantize_rows_fp8_kernel(const unsigned short* __restrict__ x, long long ldx, int K, unsigned char* __restrict__ x8,
                                                                float* __restrict__ scale) {
  __shared__ float red[4];
  const int t = blockIdx.x, tid = threadIdx.x;
  const unsigned short* row = x + (long long)t * ldx;
  float amax = 0.f;
  for (int i = tid * 8; i < K; i += 256 * 8) {
    const raw16 v = *reinterpret_cast<const raw16*>(row + i);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      amax = fmaxf(amax, fabsf(bf2f((unsigned short)(v[j] & 0xffffu))));
      amax = fmaxf(amax, fabsf(bf2f((unsigned short)(v[j] >> 16))));
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
  if ((tid & 63) == 0) red[tid >> 6] = amax;
  __syncthreads();
  amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float sc = amax > 0.f ? amax * (1.f / 448.f) : 1.f;
  const float inv = 1.f / sc;
  if (tid == 0) scale[t] = sc;
  unsigned char* orow = x8 + (long long)t * K;
  // thread handles 8 consecutive k (one 16-byte load) -> 8 bytes at the permuted position
  for (int i = tid * 8; i < K; i += 256 * 8) {
    const raw16 v = *reinterpret_cast<const raw16*>(row + i);
    float f[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f[2 * j] = bf2f((unsigned short)(v[j] & 0xffffu)) * inv;
      f[2 * j + 1] = bf2f((unsigned short)(v[j] >> 16)) * inv;
    }
    unsigned int lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
    if constexpr (PERMUTE) {
      const int g = i >> 6, w = i & 63;           // 64-element group, offset inside it (a multiple of 8)
      const int half = w >> 5, qq = (w & 31) >> 3;  // k = 32 half + 8 qq + j
      *reinterpret_cast<raw8*>(orow + g * 64 + 16 * qq + 8 * half) = (raw8){lo, hi};
    } else {
      *reinterpret_cast<raw8*>(orow + i) = (raw8){lo, hi};  // natural k order (MXFP4 weights: the FP4 operand is in k order)
    }
  }
}

}  // namespace

extern "C" {

// fmt: 0 = e4m3, 1 = e5m2, 4 = e2m1 (FP4); a / b: int32 [64][8] operand registers per lane; sa / sb: int32 [64] scale dwords
int kv_mfma_scale_probe(const int* a, const int* b, const int* sa, const int* sb, float* out, int fmt_a, int fmt_b, hipStream_t stream) {
#define KV_PROBE(FA, FB) hipLaunchKernelGGL((mfma_scale_probe_kernel<FA, FB>), dim3(1), dim3(64), 0, stream, a, b, sa, sb, out)
  if (fmt_a == 0 && fmt_b == 0) KV_PROBE(0, 0);
  else if (fmt_a == 4 && fmt_b == 0) KV_PROBE(4, 0);
  else if (fmt_a == 0 && fmt_b == 4) KV_PROBE(0, 4);
  else if (fmt_a == 4 && fmt_b == 4) KV_PROBE(4, 4);
  else if (fmt_a == 1 && fmt_b == 1) KV_PROBE(1, 1);
  else return -1;
#undef KV_PROBE
  return (int)hipGetLastError();
}

// x bf16 [M][ldx] -> x8 e4m3 [M][K] (permute != 0: permuted inside 64-element groups into the FP8 weight fragments' order,
// see the kernel; 0: natural order), scale fp32 [M].  K % 64 == 0, ldx % 8 == 0.
int kv_quantize_rows_fp8(const void* x, long long ldx, int M, int K, void* x8, float* scale, int permute, hipStream_t stream) {
  if (M <= 0) return 0;
  if ((K & 63) || (ldx & 7) || ldx < K) return -1;
  if (permute) hipLaunchKernelGGL(quantize_rows_fp8_kernel<true>, dim3(M), dim3(256), 0, stream, (const unsigned short*)x, ldx, K, (unsigned char*)x8, scale);
  else hipLaunchKernelGGL(quantize_rows_fp8_kernel<false>, dim3(M), dim3(256), 0, stream, (const unsigned short*)x, ldx, K, (unsigned char*)x8, scale);
  return (int)hipGetLastError();
}

}  // extern "C"

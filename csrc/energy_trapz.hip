// Batched trapezoid energy integration for gfx950: one launch integrates the power
// series of all GPUs of a node over the benchmark window.
//
// Semantics (reference energy/collector.py:133-149): keep samples with a value and
// t0 <= ts <= t1, in time order; Wh = sum over consecutive kept samples of
// (p_i + p_j)/2 * (t_j - t_i)/3600, counted only when t_j > t_i; fewer than two kept
// samples -> 0.  Series are rows of a [n_series, max_samples] matrix, time-sorted (the
// sampler appends in time order); NaN watts = missing sample; n[s] = valid length.
// grid.x = series; each thread handles a strided set of "left" samples and looks ahead
// for the next kept sample (normally the very next one).  Also returns the window mean
// power and the number of kept samples.

#include <hip/hip_runtime.h>

namespace {

__device__ inline double wave_sum_d(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__global__ __launch_bounds__(256) void trapz_kernel(const double* __restrict__ ts, const double* __restrict__ watts, const int* __restrict__ n,
                                                    int max_samples, double t0, double t1, double* __restrict__ out /* [n_series][3]: Wh, mean W, kept */) {
  __shared__ double red[3][4];
  const int s = blockIdx.x, tid = threadIdx.x;
  const double* t = ts + (long long)s * max_samples;
  const double* p = watts + (long long)s * max_samples;
  const int ns = n[s];
  double wh = 0.0, psum = 0.0, kept = 0.0;
  if (t1 > t0) {
    for (int i = tid; i < ns; i += 256) {
      double ti = t[i], pi = p[i];
      if (!(pi == pi) || ti < t0 || ti > t1) continue;
      kept += 1.0;
      psum += pi;
      int j = i + 1;
      while (j < ns && !((p[j] == p[j]) && t[j] >= t0 && t[j] <= t1)) ++j;
      if (j < ns) {
        double dt_h = (t[j] - ti) / 3600.0;
        if (dt_h > 0) wh += (pi + p[j]) * 0.5 * dt_h;
      }
    }
  }
  wh = wave_sum_d(wh); psum = wave_sum_d(psum); kept = wave_sum_d(kept);
  if ((tid & 63) == 0) { red[0][tid >> 6] = wh; red[1][tid >> 6] = psum; red[2][tid >> 6] = kept; }
  __syncthreads();
  if (tid == 0) {
    double a = 0, b = 0, c = 0;
    for (int w = 0; w < 4; ++w) { a += red[0][w]; b += red[1][w]; c += red[2][w]; }
    out[s * 3 + 0] = c >= 2.0 ? a : 0.0;
    out[s * 3 + 1] = c > 0 ? b / c : 0.0;
    out[s * 3 + 2] = c;
  }
}

}  // namespace

extern "C" int kv_energy_trapz(const double* ts, const double* watts, const int* n, int n_series, int max_samples, double t0, double t1, double* out,
                               hipStream_t stream) {
  if (n_series <= 0) return 0;
  hipLaunchKernelGGL(trapz_kernel, dim3(n_series), dim3(256), 0, stream, ts, watts, n, max_samples, t0, t1, out);
  return (int)hipGetLastError();
}

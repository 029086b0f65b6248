// Host-side analysis routines of the benchmark pipeline, in C++ (no GPU runtime, no Python).
//
// The analyzer, the quality/quantisation sweeps and the fairness guard each have one inner loop
// that grows with the run: the non-dominated filter over sweep rows (quadratic), the cold/warm
// classification of every request against the cold-start list, exact quantiles of the latency
// column, and a rolling quantile that is read after every response.  The Python versions
// (kvmini/harness/quality.py ParetoAnalyzer.frontier_mask, kvmini/harness/quantization_sweep.py
// pareto_mask, kvmini/analysis/coldwarm.py, kvmini/analysis/quantiles.py) define the semantics;
// these are the same predicates on flat double arrays, reached through ctypes
// (kvmini/analysis/native_host.py) and compiled into libkvmini_rt.so.
//
// Reference behaviour they reproduce: analyze.py:59-81 (linear-interpolated percentile),
// analyze.py:402-460 (30 s cold window), scripts/fairness_dual_tenant.py:46-65 (rolling p95,
// floor index), quality/evaluator.py:446-511 (Pareto buckets).

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

enum Method { LINEAR = 0, FLOOR = 1, NEAREST_RANK = 2 };

double from_sorted(const double* v, int64_t n, double p, int method) {
  if (n == 0) return NAN;
  if (n == 1) return v[0];
  p = std::max(0.0, std::min(p, 1.0));
  if (method == LINEAR) {
    const double pos = p * double(n - 1);
    const int64_t lo = (int64_t)std::floor(pos), hi = (int64_t)std::ceil(pos);
    if (lo == hi) return v[lo];
    return v[lo] + (pos - double(lo)) * (v[hi] - v[lo]);
  }
  if (method == FLOOR) return v[(int64_t)(p * double(n - 1))];
  return v[std::min<int64_t>(n - 1, (int64_t)(p * double(n)))];
}

// Rolling window kept twice: arrival order (ring) and sorted (contiguous, shifted on insert).
struct Rolling {
  int64_t window;
  double p;
  int method;
  std::vector<double> ring;
  int64_t head = 0, count = 0;
  std::vector<double> sorted;
};

}  // namespace

extern "C" {

// out[i] = 1 when no row j dominates row i (all axes lower-is-better; j dominates i when it is
// <= on every axis and < on at least one).  nan_neutral != 0: an axis where either value is NaN
// neither helps nor hurts (missing measurements); otherwise NaN compares false like IEEE.
int kvh_pareto_mask(const double* mat, int64_t n, int64_t d, int nan_neutral, uint8_t* out) {
  if (n < 0 || d < 0 || (!mat && n * d > 0) || (!out && n > 0)) return -1;
  for (int64_t i = 0; i < n; ++i) {
    const double* a = mat + i * d;
    bool dominated = false;
    for (int64_t j = 0; j < n && !dominated; ++j) {
      if (j == i) continue;
      const double* b = mat + j * d;
      bool le = true, lt = false;
      for (int64_t k = 0; k < d; ++k) {
        const double x = a[k], y = b[k];
        if (nan_neutral && (std::isnan(x) || std::isnan(y))) continue;
        if (!(y <= x)) { le = false; break; }
        if (y < x) lt = true;
      }
      dominated = le && lt;
    }
    out[i] = dominated ? 0 : 1;
  }
  return 0;
}

// Cold/warm flags.  start_s / latency_s: request start time and latency in seconds;
// cold_sorted: ascending cold-start times.  A request is cold when it starts inside
// [t_cold, t_cold + window] of the latest cold start before it, or (spanning) when the next
// cold start happens while it is in flight.
int kvh_classify_cold(const double* start_s, const double* latency_s, int64_t n, const double* cold_sorted, int64_t k, double window, int spanning,
                      uint8_t* out) {
  if (n < 0 || k < 0 || (n > 0 && (!start_s || !out)) || (k > 0 && !cold_sorted)) return -1;
  for (int64_t r = 0; r < n; ++r) {
    const double t = start_s[r];
    bool cold = false;
    if (k > 0) {
      const int64_t j = std::upper_bound(cold_sorted, cold_sorted + k, t) - cold_sorted;  // first cold start > t
      if (j >= 1 && t <= cold_sorted[j - 1] + window) {
        cold = true;
      } else if (spanning && latency_s) {
        const double lat = latency_s[r];
        if (lat > 0 && j < k && cold_sorted[j] < t + lat) cold = true;
      }
    }
    out[r] = cold ? 1 : 0;
  }
  return 0;
}

// Exact quantiles of vals (NaNs dropped): one sort, every p read from it.  Returns the number of
// samples that took part, or -1.
int64_t kvh_quantiles(const double* vals, int64_t n, const double* ps, int64_t np, int method, double* out) {
  if (n < 0 || np < 0 || (n > 0 && !vals) || (np > 0 && (!ps || !out)) || method < 0 || method > 2) return -1;
  std::vector<double> v;
  v.reserve((size_t)n);
  for (int64_t i = 0; i < n; ++i)
    if (!std::isnan(vals[i])) v.push_back(vals[i]);
  std::sort(v.begin(), v.end());
  for (int64_t i = 0; i < np; ++i) out[i] = from_sorted(v.data(), (int64_t)v.size(), ps[i], method);
  return (int64_t)v.size();
}

void* kvh_rq_new(int64_t window, double p, int method) {
  if (window < 1 || method < 0 || method > 2) return nullptr;
  Rolling* r = new Rolling();
  r->window = window;
  r->p = p;
  r->method = method;
  r->ring.resize((size_t)window);
  r->sorted.reserve((size_t)window + 1);
  return r;
}

void kvh_rq_free(void* h) { delete static_cast<Rolling*>(h); }

// Add one observation; returns the window's quantile after it (the read that follows every add
// in the fairness guard), so a response costs one call.
double kvh_rq_add(void* h, double x) {
  Rolling* r = static_cast<Rolling*>(h);
  if (r->count == r->window) {
    const double old = r->ring[(size_t)r->head];
    auto it = std::lower_bound(r->sorted.begin(), r->sorted.end(), old);
    r->sorted.erase(it);
  } else {
    r->count += 1;
  }
  r->ring[(size_t)r->head] = x;
  r->head = (r->head + 1) % r->window;
  r->sorted.insert(std::upper_bound(r->sorted.begin(), r->sorted.end(), x), x);
  return from_sorted(r->sorted.data(), (int64_t)r->sorted.size(), r->p, r->method);
}

// Many observations at once; out (optional) receives the quantile after each.
int kvh_rq_add_many(void* h, const double* xs, int64_t n, double* out) {
  if (!h || n < 0 || (n > 0 && !xs)) return -1;
  for (int64_t i = 0; i < n; ++i) {
    const double q = kvh_rq_add(h, xs[i]);
    if (out) out[i] = q;
  }
  return 0;
}

double kvh_rq_value(void* h) {
  Rolling* r = static_cast<Rolling*>(h);
  return from_sorted(r->sorted.data(), (int64_t)r->sorted.size(), r->p, r->method);
}

int64_t kvh_rq_len(void* h) { return static_cast<Rolling*>(h)->count; }

}  // extern "C"

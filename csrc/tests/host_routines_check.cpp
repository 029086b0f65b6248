// Self-check of csrc/host_routines.cpp against the obvious slow versions, on random inputs:
// the rolling quantile against re-sorting the window, the cold/warm flags against a scan over
// every (request, cold start) pair, quantiles against a full sort.  Built by `make sanitize`
// with -fsanitize=address,undefined so that an off-by-one in the ring or the sorted window shows
// up as a report, not as a wrong number.  Exits 0 on success.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

extern "C" {
int kvh_pareto_mask(const double* mat, int64_t n, int64_t d, int nan_neutral, uint8_t* out);
int kvh_classify_cold(const double* start_s, const double* latency_s, int64_t n, const double* cold_sorted, int64_t k, double window, int spanning, uint8_t* out);
int64_t kvh_quantiles(const double* vals, int64_t n, const double* ps, int64_t np, int method, double* out);
void* kvh_rq_new(int64_t window, double p, int method);
void kvh_rq_free(void* h);
double kvh_rq_add(void* h, double x);
double kvh_rq_value(void* h);
int64_t kvh_rq_len(void* h);
}

static int fail(const char* what) {
  std::fprintf(stderr, "host_routines_check: %s\n", what);
  return 1;
}

int main() {
  std::mt19937_64 rng(7);
  std::uniform_real_distribution<double> U(0.0, 1.0);

  // rolling quantile, every window size from 1 to 9 and a large one, floor and linear
  for (int method = 0; method < 3; ++method) {
    for (int64_t w : {1, 2, 3, 5, 9, 50}) {
      void* h = kvh_rq_new(w, 0.95, method);
      if (!h) return fail("rq_new");
      std::vector<double> all;
      for (int i = 0; i < 400; ++i) {
        const double x = std::floor(U(rng) * 20.0);  // many ties
        all.push_back(x);
        const double got = kvh_rq_add(h, x);
        const size_t lo = all.size() > (size_t)w ? all.size() - (size_t)w : 0;
        std::vector<double> win(all.begin() + (long)lo, all.end());
        std::sort(win.begin(), win.end());
        double want;
        const int64_t n = (int64_t)win.size();
        if (n == 1) want = win[0];
        else if (method == 0) {
          const double pos = 0.95 * double(n - 1);
          const int64_t a = (int64_t)std::floor(pos), b = (int64_t)std::ceil(pos);
          want = a == b ? win[a] : win[a] + (pos - double(a)) * (win[b] - win[a]);
        } else if (method == 1) want = win[(int64_t)(0.95 * double(n - 1))];
        else want = win[std::min<int64_t>(n - 1, (int64_t)(0.95 * double(n)))];
        if (got != want || kvh_rq_value(h) != want || kvh_rq_len(h) != n) return fail("rolling quantile differs from the re-sorted window");
      }
      kvh_rq_free(h);
    }
  }

  // cold/warm flags
  for (int trial = 0; trial < 50; ++trial) {
    const int64_t n = 1 + (int64_t)(U(rng) * 300), k = (int64_t)(U(rng) * 6);
    std::vector<double> s((size_t)n), lat((size_t)n), cold((size_t)k);
    for (auto& x : s) x = U(rng) * 600.0;
    for (auto& x : lat) x = U(rng) < 0.1 ? 0.0 : U(rng) * 40.0;
    for (auto& x : cold) x = U(rng) * 600.0;
    std::sort(cold.begin(), cold.end());
    for (int spanning = 0; spanning < 2; ++spanning) {
      std::vector<uint8_t> out((size_t)n);
      if (kvh_classify_cold(s.data(), lat.data(), n, cold.data(), k, 30.0, spanning, out.data()) != 0) return fail("classify_cold rc");
      for (int64_t r = 0; r < n; ++r) {
        bool want = false;
        for (int64_t j = 0; j < k; ++j) {
          if (cold[(size_t)j] <= s[(size_t)r] && s[(size_t)r] <= cold[(size_t)j] + 30.0) want = true;
        }
        if (!want && spanning && lat[(size_t)r] > 0) {
          for (int64_t j = 0; j < k; ++j)
            if (cold[(size_t)j] > s[(size_t)r]) { want = cold[(size_t)j] < s[(size_t)r] + lat[(size_t)r]; break; }
        }
        if ((out[(size_t)r] != 0) != want) return fail("cold flag differs from the pairwise scan");
      }
    }
  }

  // quantiles with NaNs, empty and single inputs
  {
    std::vector<double> v;
    for (int i = 0; i < 1001; ++i) v.push_back(i % 17 == 0 ? NAN : U(rng));
    const double ps[3] = {0.5, 0.95, 0.99};
    double out[3];
    const int64_t used = kvh_quantiles(v.data(), (int64_t)v.size(), ps, 3, 0, out);
    std::vector<double> c;
    for (double x : v) if (!std::isnan(x)) c.push_back(x);
    std::sort(c.begin(), c.end());
    if (used != (int64_t)c.size()) return fail("quantiles: sample count");
    const double pos = 0.95 * double(c.size() - 1);
    const size_t a = (size_t)std::floor(pos), b = (size_t)std::ceil(pos);
    const double want = a == b ? c[a] : c[a] + (pos - double(a)) * (c[b] - c[a]);
    if (out[1] != want) return fail("quantiles: p95");
    if (kvh_quantiles(nullptr, 0, ps, 3, 0, out) != 0 || !std::isnan(out[0])) return fail("quantiles: empty input");
    const double one = 4.0;
    if (kvh_quantiles(&one, 1, ps, 3, 2, out) != 1 || out[2] != 4.0) return fail("quantiles: single input");
  }

  // non-dominated filter: a strict chain plus duplicates plus a NaN row
  {
    const double m[5 * 2] = {1, 1, 2, 2, 1, 1, 0.5, 3, NAN, 0.1};
    uint8_t out[5];
    if (kvh_pareto_mask(m, 5, 2, 0, out) != 0) return fail("pareto rc");
    const uint8_t want[5] = {1, 0, 1, 1, 1};
    for (int i = 0; i < 5; ++i) if (out[i] != want[i]) return fail("pareto mask");
    if (kvh_pareto_mask(m, 5, 2, 1, out) != 0) return fail("pareto rc");
    const uint8_t want_neutral[5] = {0, 0, 0, 0, 1};  // (NaN, 0.1) beats every row on the one axis it has
    for (int i = 0; i < 5; ++i) if (out[i] != want_neutral[i]) return fail("pareto mask (NaN neutral)");
    if (kvh_pareto_mask(nullptr, 0, 2, 0, nullptr) != 0) return fail("pareto: empty input");
  }
  std::printf("host_routines_check: ok\n");
  return 0;
}

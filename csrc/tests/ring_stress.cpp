// Two-thread stress test of the telemetry sampler's ring buffer (csrc/amdsmi_sampler.cpp):
// the sampler thread writes synthetic records as fast as it can into a SMALL ring, so that it
// laps the reader all the time, while this thread keeps copying the ring out.  Every record that
// kvs_read returns must be whole (all eight fields belong to the same record number) and the
// record numbers of one read must be consecutive; restarts with other ring sizes exercise the
// replacement of the ring under a reader.  Built by `make sanitize` with -fsanitize=thread and
// with -fsanitize=address,undefined; exits 0 on success.
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

extern "C" {
int kvs_start_synthetic(double period_s, int n_gpus, long long capacity);
long long kvs_count();
long long kvs_read(double* out, long long max_records);
int kvs_stop();
double kvs_trapz_wh(const double* t, const double* w, long long n);
}

static long long check(const double* rec, long long n, int n_gpus) {
  long long bad = 0;
  for (long long i = 0; i < n; ++i) {
    const double* r = rec + i * 8;
    const double x = r[0];
    const unsigned long long k = (unsigned long long)x;
    const bool whole = r[2] == 100.0 + x && r[3] == x * 0.5 && r[4] == x * 1024.0 && r[5] == 40.0 + (double)(k % 50) && r[6] == 2400.0 - (double)(k % 7) && r[7] == x * 3.0 &&
                       r[1] >= 0 && r[1] < n_gpus;
    if (!whole) ++bad;
    if (i > 0 && r[0] != rec[(i - 1) * 8] + 1.0) ++bad;  // consecutive record numbers within one read
  }
  return bad;
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 1.0;
  long long reads = 0, records = 0, bad = 0, empty = 0;
  const long long caps[] = {64, 100, 4096};
  for (long long cap : caps) {
    const int n_gpus = cap == 100 ? 3 : 1;
    if (kvs_start_synthetic(0.0, n_gpus, cap) != 0) { std::fprintf(stderr, "start failed\n"); return 2; }
    std::vector<double> buf((size_t)cap * 8);
    std::atomic<bool> done{false};
    // a second reader: kvs_read must be callable from several threads at once
    std::thread other([&] {
      std::vector<double> b2((size_t)cap * 8);
      while (!done.load()) { const long long n = kvs_read(b2.data(), cap); if (n > 0 && check(b2.data(), n, n_gpus)) { std::fprintf(stderr, "torn record (second reader)\n"); std::abort(); } }
    });
    const auto t_end = std::chrono::steady_clock::now() + std::chrono::duration<double>(seconds / 3.0);
    while (std::chrono::steady_clock::now() < t_end) {
      const long long want = 1 + (reads % cap);
      const long long n = kvs_read(buf.data(), want);
      ++reads;
      if (n == 0) ++empty;
      if (n < 0 || n > want) { std::fprintf(stderr, "kvs_read returned %lld of %lld\n", n, want); return 3; }
      records += n;
      bad += check(buf.data(), n, n_gpus);
    }
    done.store(true);
    other.join();
    if (kvs_count() <= cap) { std::fprintf(stderr, "the writer never lapped the ring (%lld records)\n", kvs_count()); return 4; }
    if (kvs_stop() != 0) return 5;
  }
  const double t[4] = {0.0, 1.0, 2.0, 3.0}, w[4] = {100.0, NAN, 300.0, 300.0};
  if (std::fabs(kvs_trapz_wh(t, w, 4) - (400.0 + 300.0) / 3600.0) > 1e-12) { std::fprintf(stderr, "trapz\n"); return 6; }
  std::printf("ring_stress: %lld reads, %lld records checked, %lld reads overtaken, %lld bad\n", reads, records, empty, bad);
  return bad ? 1 : 0;
}

// Native GPU telemetry sampler: a C++ thread that polls the AMD SMI library at a fixed
// period and stores fixed-size records in a ring buffer; Python (ctypes) starts / stops it
// and copies the records out afterwards, so the sampling cadence does not depend on the
// Python interpreter (the load generator and the engine thread both hold the GIL in bursts
// and a Python sampler at 10 ms period shows 30-50 ms holes under load).
//
// The ring is written by the sampler thread and read by whoever calls kvs_read, with no lock
// between them: every field is a relaxed atomic word (free on x86-64), the record count `head`
// is published with release / read with acquire, and a reader that was overtaken while copying
// (the writer wrapped around onto the oldest records) drops exactly those records.  `make
// sanitize` runs a two-thread stress test of this protocol under ThreadSanitizer and
// AddressSanitizer + UBSan (csrc/tests/ring_stress.cpp).
//
// libamd_smi.so is opened with dlopen at start-up: a machine without the library (CPU-only
// CI) can still load this runtime; kvs_start() then returns -2 and the caller falls back.
// The structs below copy the leading fields of the amdsmi.h definitions this file was
// written against (ROCm 7.x); each is padded generously past the library's reserved tail.

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <dlfcn.h>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace {

struct PowerInfo { uint64_t socket_power; uint32_t current_socket_power; uint32_t average_socket_power; uint64_t gfx_voltage, soc_voltage, mem_voltage; uint32_t power_limit; uint64_t reserved[32]; };
struct EngineUsage { uint32_t gfx_activity, umc_activity, mm_activity; uint32_t reserved[29]; };
struct VramUsage { uint32_t vram_total, vram_used; uint32_t reserved[14]; };
struct ClkInfo { uint32_t clk, min_clk, max_clk; uint8_t clk_locked, clk_deep_sleep; uint32_t reserved[16]; };

typedef void* handle_t;
typedef int (*fn_init)(uint64_t);
typedef int (*fn_shutdown)();
typedef int (*fn_sockets)(uint32_t*, handle_t*);
typedef int (*fn_procs)(handle_t, uint32_t*, handle_t*);
typedef int (*fn_power)(handle_t, PowerInfo*);
typedef int (*fn_activity)(handle_t, EngineUsage*);
typedef int (*fn_vram)(handle_t, VramUsage*);
typedef int (*fn_temp)(handle_t, int, int, int64_t*);
typedef int (*fn_clk)(handle_t, int, ClkInfo*);
typedef int (*fn_energy)(handle_t, uint64_t*, float*, uint64_t*);

constexpr uint64_t INIT_AMD_GPUS = 1u << 1;
constexpr int TEMP_TYPE_HOTSPOT = 1, TEMP_CURRENT = 0, CLK_TYPE_GFX = 0;

struct Api {
  void* lib = nullptr;
  fn_init init = nullptr; fn_shutdown shutdown = nullptr; fn_sockets sockets = nullptr; fn_procs procs = nullptr;
  fn_power power = nullptr; fn_activity activity = nullptr; fn_vram vram = nullptr; fn_temp temp = nullptr; fn_clk clk = nullptr; fn_energy energy = nullptr;
  std::vector<handle_t> gpus;
  bool open() {
    if (lib) return true;
    const char* names[] = {"libamd_smi.so", "libamd_smi.so.26", "/opt/rocm/lib/libamd_smi.so"};
    for (const char* n : names) { lib = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (lib) break; }
    if (!lib) return false;
#define SYM(field, type, name) field = reinterpret_cast<type>(dlsym(lib, name))
    SYM(init, fn_init, "amdsmi_init"); SYM(shutdown, fn_shutdown, "amdsmi_shut_down"); SYM(sockets, fn_sockets, "amdsmi_get_socket_handles");
    SYM(procs, fn_procs, "amdsmi_get_processor_handles"); SYM(power, fn_power, "amdsmi_get_power_info"); SYM(activity, fn_activity, "amdsmi_get_gpu_activity");
    SYM(vram, fn_vram, "amdsmi_get_gpu_vram_usage"); SYM(temp, fn_temp, "amdsmi_get_temp_metric"); SYM(clk, fn_clk, "amdsmi_get_clock_info");
    SYM(energy, fn_energy, "amdsmi_get_energy_count");
#undef SYM
    if (!init || !sockets || !procs || !power) { dlclose(lib); lib = nullptr; return false; }
    if (init(INIT_AMD_GPUS) != 0) { dlclose(lib); lib = nullptr; return false; }
    uint32_t ns = 0;
    if (sockets(&ns, nullptr) != 0 || ns == 0) return true;  // library present, no GPUs visible
    std::vector<handle_t> socks(ns);
    sockets(&ns, socks.data());
    for (uint32_t s = 0; s < ns; ++s) {
      uint32_t np = 0;
      if (procs(socks[s], &np, nullptr) != 0 || np == 0) continue;
      std::vector<handle_t> ps(np);
      procs(socks[s], &np, ps.data());
      gpus.insert(gpus.end(), ps.begin(), ps.begin() + np);
    }
    return true;
  }
};

// One record per (tick, gpu): 8 doubles.
struct Record { double ts_s, gpu, watts, gfx_util, vram_used_bytes, temp_c, sclk_mhz, energy_uj; };
constexpr int RECORD_WORDS = 8;
static_assert(sizeof(Record) == RECORD_WORDS * sizeof(double), "a record is 8 doubles");

// Ring slot: the same 8 doubles as bit patterns in atomic words.
struct Slot {
  std::atomic<uint64_t> w[RECORD_WORDS];
  void put(const Record& r) {
    uint64_t bits[RECORD_WORDS];
    std::memcpy(bits, &r, sizeof r);
    for (int i = 0; i < RECORD_WORDS; ++i) w[i].store(bits[i], std::memory_order_relaxed);
  }
  void get(double* out) const {
    uint64_t bits[RECORD_WORDS];
    for (int i = 0; i < RECORD_WORDS; ++i) bits[i] = w[i].load(std::memory_order_relaxed);
    std::memcpy(out, bits, sizeof bits);
  }
};

struct Sampler {
  Api api;
  std::thread th;
  std::atomic<bool> stop{false};
  std::atomic<uint64_t> head{0};  // records written
  std::unique_ptr<Slot[]> ring;
  std::atomic<uint64_t> cap{0};  // slots in `ring` (0: none yet)
  std::vector<int> gpu_idx;
  double period_s = 0.05;
  std::mutex mu;
  bool running = false;
  bool synthetic = false;  // records come from a counter instead of the SMI library (tests, dry runs)

  ~Sampler() {  // a process that exits with the thread still sampling (no kvs_stop) must not abort in std::thread's destructor
    stop.store(true);
    if (th.joinable()) th.join();
  }

  static double now_s() {
    using namespace std::chrono;
    return duration<double>(system_clock::now().time_since_epoch()).count();
  }

  void read_one(int gi, Record* r) {
    const double nan = std::nan("");
    *r = Record{now_s(), (double)gi, nan, nan, nan, nan, nan, nan};
    handle_t h = api.gpus[gi];
    PowerInfo p; std::memset(&p, 0, sizeof p);
    if (api.power(h, &p) == 0) {
      uint32_t w = p.current_socket_power;
      if (w == 0 || w == UINT32_MAX) w = p.average_socket_power;
      if (w != UINT32_MAX && w != 0) r->watts = (double)w;
      else if (p.socket_power && p.socket_power != UINT64_MAX) r->watts = (double)p.socket_power;
    }
    EngineUsage u; std::memset(&u, 0, sizeof u);
    if (api.activity && api.activity(h, &u) == 0 && u.gfx_activity != UINT32_MAX) r->gfx_util = (double)u.gfx_activity;
    VramUsage v; std::memset(&v, 0, sizeof v);
    if (api.vram && api.vram(h, &v) == 0) r->vram_used_bytes = (double)v.vram_used * 1048576.0;
    int64_t t = 0;
    if (api.temp && api.temp(h, TEMP_TYPE_HOTSPOT, TEMP_CURRENT, &t) == 0) r->temp_c = (double)t;
    ClkInfo c; std::memset(&c, 0, sizeof c);
    if (api.clk && api.clk(h, CLK_TYPE_GFX, &c) == 0) r->sclk_mhz = (double)c.clk;
    uint64_t acc = 0, ts = 0; float res = 0.f;
    if (api.energy && api.energy(h, &acc, &res, &ts) == 0) r->energy_uj = (double)acc * (double)res;
  }

  // Synthetic record number n of gpu gi: every field is a function of n, so a reader can tell a
  // torn record (fields of two different n) from a whole one.
  static void synth_one(int gi, uint64_t n, Record* r) {
    const double x = (double)n;
    *r = Record{x, (double)gi, 100.0 + x, x * 0.5, x * 1024.0, 40.0 + (double)(n % 50), 2400.0 - (double)(n % 7), x * 3.0};
  }

  void loop() {
    using clock = std::chrono::steady_clock;
    auto next = clock::now();
    const auto period = std::chrono::duration_cast<clock::duration>(std::chrono::duration<double>(period_s));
    const uint64_t n_slots = cap.load(std::memory_order_relaxed);
    while (!stop.load(std::memory_order_relaxed)) {
      for (int gi : gpu_idx) {
        Record r;
        const uint64_t h = head.load(std::memory_order_relaxed);
        if (synthetic) synth_one(gi, h, &r); else read_one(gi, &r);
        ring[h % n_slots].put(r);
        head.store(h + 1, std::memory_order_release);
      }
      if (period_s <= 0.0) continue;  // synthetic stress mode: as fast as the thread can go
      next += period;
      const auto now = clock::now();
      if (next > now) std::this_thread::sleep_until(next); else next = now;
    }
  }

  void begin(double period, long long capacity) {
    period_s = period;
    const uint64_t n = (uint64_t)(capacity < 64 ? 64 : capacity);
    cap.store(0, std::memory_order_release);  // readers see an empty ring while it is replaced
    ring.reset(new Slot[n]);
    for (uint64_t i = 0; i < n; ++i)
      for (int j = 0; j < RECORD_WORDS; ++j) ring[i].w[j].store(0, std::memory_order_relaxed);
    head.store(0, std::memory_order_relaxed);
    cap.store(n, std::memory_order_release);
    stop.store(false);
    th = std::thread([this] { loop(); });
    running = true;
  }
};

Sampler g;

}  // namespace

extern "C" {

// Number of GPUs the SMI library exposes (-2: library not loadable).
int kvs_gpu_count() {
  std::lock_guard<std::mutex> lk(g.mu);
  if (!g.api.open()) return -2;
  return (int)g.api.gpus.size();
}

// Start sampling `n_gpus` devices (indices in `gpus`, or all when n_gpus <= 0) every
// `period_s` seconds into a ring of `capacity` records.  0 on success, -1 already running,
// -2 no library, -3 no GPUs.
int kvs_start(double period_s, const int* gpus, int n_gpus, long long capacity) {
  std::lock_guard<std::mutex> lk(g.mu);
  if (g.running) return -1;
  if (!g.api.open()) return -2;
  if (g.api.gpus.empty()) return -3;
  g.gpu_idx.clear();
  if (n_gpus <= 0) for (size_t i = 0; i < g.api.gpus.size(); ++i) g.gpu_idx.push_back((int)i);
  else for (int i = 0; i < n_gpus; ++i) if (gpus[i] >= 0 && gpus[i] < (int)g.api.gpus.size()) g.gpu_idx.push_back(gpus[i]);
  if (g.gpu_idx.empty()) return -3;
  g.synthetic = false;
  g.begin(period_s < 0.001 ? 0.001 : period_s, capacity);
  return 0;
}

// The same sampler thread and ring fed from a counter instead of the SMI library: `n_gpus` fake
// devices, record n of a device carries f(n) in every field.  period_s <= 0: no sleep between
// ticks (stress test).  0 on success, -1 already running.
int kvs_start_synthetic(double period_s, int n_gpus, long long capacity) {
  std::lock_guard<std::mutex> lk(g.mu);
  if (g.running) return -1;
  g.gpu_idx.clear();
  for (int i = 0; i < (n_gpus < 1 ? 1 : n_gpus); ++i) g.gpu_idx.push_back(i);
  g.synthetic = true;
  g.begin(period_s, capacity);
  return 0;
}

// Records written so far.
long long kvs_count() { return (long long)g.head.load(std::memory_order_acquire); }

// Copy up to `max_records` of the most recent records (oldest first) into `out`
// (8 doubles each: ts_s, gpu, watts, gfx_util, vram_used_bytes, temp_c, sclk_mhz, energy_uj).
long long kvs_read(double* out, long long max_records) {
  const uint64_t cap = g.cap.load(std::memory_order_acquire);
  if (cap == 0 || max_records <= 0) return 0;
  const uint64_t h = g.head.load(std::memory_order_acquire);
  uint64_t n = h < cap ? h : cap;
  if ((long long)n > max_records) n = (uint64_t)max_records;
  const uint64_t first = h - n;
  for (uint64_t i = 0; i < n; ++i) g.ring[(first + i) % cap].get(out + i * RECORD_WORDS);
  // The writer kept going: record i's slot is reused by record i + cap, whose write starts as soon
  // as head reaches i + cap.  Everything at or below (head now) - cap may be torn: drop it.
  const uint64_t h2 = g.head.load(std::memory_order_acquire);
  uint64_t lost = 0;
  if (h2 >= cap && first <= h2 - cap) lost = h2 - cap - first + 1;
  if (lost >= n) return 0;
  if (lost) std::memmove(out, out + lost * RECORD_WORDS, (size_t)(n - lost) * RECORD_WORDS * sizeof(double));
  return (long long)(n - lost);
}

int kvs_stop() {
  std::lock_guard<std::mutex> lk(g.mu);
  if (!g.running) return -1;
  g.stop.store(true);
  if (g.th.joinable()) g.th.join();
  g.running = false;
  return 0;
}

// Trapezoidal energy (Wh) of a (t seconds, watts) series on the host; NaN samples are skipped.
double kvs_trapz_wh(const double* t, const double* w, long long n) {
  double e = 0.0; long long prev = -1;
  for (long long i = 0; i < n; ++i) {
    if (std::isnan(w[i]) || std::isnan(t[i])) continue;
    if (prev >= 0 && t[i] > t[prev]) e += 0.5 * (w[i] + w[prev]) * (t[i] - t[prev]);
    prev = i;
  }
  return e / 3600.0;
}

}  // extern "C"

// Latency of a device-wide barrier among co-resident workgroups on gfx950 (one workgroup per CU),
// in three forms.  A persistent decode kernel separates its phases with such barriers instead of
// kernel boundaries, so this number decides what that design can hide.
//   flat : every workgroup adds 1 to ONE counter (device scope), spins until it reaches G * epoch
//   tree : workgroups of an XCD add to that XCD's counter; the last of each XCD adds to the global one
//   flags: no read-modify-write at all — workgroup g stores the epoch into flags[g] (write-through),
//          then every workgroup polls the whole array (thread i reads flags[i]) until all carry it
// Every spin is bounded (a missing workgroup makes the kernel exit with an error count, not hang).
// Build: hipcc --offload-arch=gfx950 -O3 -o grid_barrier_bench grid_barrier_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr long long SPIN_LIMIT = 4000000;  // polls; ~ seconds

__device__ inline unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}

template <int MODE>
__global__ __launch_bounds__(512) void barrier_kernel(unsigned* __restrict__ counters, unsigned* __restrict__ flags, int iters, unsigned* __restrict__ errors,
                                                      unsigned* __restrict__ xcc_pop, float* __restrict__ payload) {
  const int G = gridDim.x, g = blockIdx.x, tid = threadIdx.x;
  __shared__ int bad;
  if (tid == 0) bad = 0;
  const unsigned xcc = xcc_id();
  unsigned my_pop = 0;
  if (MODE == 1) {
    // census: how many workgroups sit on each XCD (needed by the tree form)
    if (tid == 0) atomicAdd(&xcc_pop[xcc], 1u);
  }
  __syncthreads();
  float acc = 0.f;
  for (int it = 1; it <= iters + (MODE == 1 ? 1 : 0); ++it) {
    // a little work between barriers, so that arrivals are not artificially aligned
    acc += payload[(g * 512 + tid) & 4095] * (float)it;
    __syncthreads();
    if (MODE == 0) {
      if (tid == 0) {
        __hip_atomic_fetch_add(&counters[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        long long n = 0;
        while (__hip_atomic_load(&counters[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)G * (unsigned)it) {
          if (++n > SPIN_LIMIT) { bad = 1; break; }
        }
      }
    } else if (MODE == 1) {
      if (tid == 0) {
        if (it == 1) {
          // first round: a flat barrier so that the census is complete before it is read
          __hip_atomic_fetch_add(&counters[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          long long n = 0;
          while (__hip_atomic_load(&counters[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)G) {
            if (++n > SPIN_LIMIT) { bad = 1; break; }
          }
          my_pop = __hip_atomic_load(&xcc_pop[xcc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          const unsigned e = (unsigned)(it - 1);
          const unsigned prev = __hip_atomic_fetch_add(&counters[16 + 16 * xcc], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (prev + 1 == my_pop * e) __hip_atomic_fetch_add(&counters[1], my_pop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          long long n = 0;
          while (__hip_atomic_load(&counters[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)G * e) {
            if (++n > SPIN_LIMIT) { bad = 1; break; }
          }
        }
      }
    } else {
      if (tid == 0) __hip_atomic_store(&flags[g], (unsigned)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // threads 0..G-1 each watch one flag; the workgroup leaves when all of them saw the epoch
      long long n = 0;
      if (tid < G) {
        while (__hip_atomic_load(&flags[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)it) {
          if (++n > SPIN_LIMIT) { bad = 1; break; }
        }
      }
    }
    __syncthreads();
    if (bad) break;
  }
  if (tid == 0 && bad) atomicAdd(errors, 1u);
  if (acc == 123.456f) payload[0] = acc;
}

int main(int argc, char** argv) {
  int iters = argc > 1 ? atoi(argv[1]) : 2000;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("{\"device\": \"%s\", \"cus\": %d}\n", prop.name, cus);
  unsigned *counters, *flags, *errors, *pop;
  float* payload;
  CHECK(hipMalloc(&counters, 4096));
  CHECK(hipMalloc(&flags, 4096));
  CHECK(hipMalloc(&errors, 4));
  CHECK(hipMalloc(&pop, 64));
  CHECK(hipMalloc(&payload, 4096 * 4));
  CHECK(hipMemset(payload, 0, 4096 * 4));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const char* names[3] = {"flat", "tree", "flags"};
  for (int grid : {cus, cus / 2}) {
    for (int mode = 0; mode < 3; ++mode) {
      for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipMemset(counters, 0, 4096));
        CHECK(hipMemset(flags, 0, 4096));
        CHECK(hipMemset(errors, 0, 4));
        CHECK(hipMemset(pop, 0, 64));
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        if (mode == 0) hipLaunchKernelGGL(barrier_kernel<0>, dim3(grid), dim3(512), 0, 0, counters, flags, iters, errors, pop, payload);
        if (mode == 1) hipLaunchKernelGGL(barrier_kernel<1>, dim3(grid), dim3(512), 0, 0, counters, flags, iters, errors, pop, payload);
        if (mode == 2) hipLaunchKernelGGL(barrier_kernel<2>, dim3(grid), dim3(512), 0, 0, counters, flags, iters, errors, pop, payload);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        unsigned err = 0, hpop[16];
        CHECK(hipMemcpy(&err, errors, 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(hpop, pop, 64, hipMemcpyDeviceToHost));
        if (rep == 1) {
          printf("{\"bench\": \"grid_barrier\", \"form\": \"%s\", \"workgroups\": %d, \"iters\": %d, \"us_per_barrier\": %.3f, \"timeouts\": %u", names[mode], grid, iters,
                 ms * 1000.0 / iters, err);
          if (mode == 1) {
            printf(", \"xcd_population\": [");
            for (int i = 0; i < 8; ++i) printf("%u%s", hpop[i], i < 7 ? ", " : "]");
          }
          printf("}\n");
          fflush(stdout);
        }
        if (err) { printf("barrier timed out: stopping\n"); return 2; }
      }
    }
  }
  return 0;
}

// Micro-experiment: throughput of fp32 atomic adds on device memory by memory scope (MI355X, 8 XCDs).
// Random indices into a 4 MB table (1 M floats), 64 M lane-ops per launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <int MODE>
__global__ __launch_bounds__(256) void k_atom(float* table, unsigned mask, int per_thread) {
  const unsigned tid = blockIdx.x * 256 + threadIdx.x;
  for (int i = 0; i < per_thread; ++i) {
    const unsigned idx = hash(tid * 977u + i * 131071u) & mask;
    float* p = table + idx;
    if (MODE == 0) atomicAdd(p, 1.0f);
    if (MODE == 1) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (MODE == 2) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (MODE == 3) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    if (MODE == 4) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SINGLETHREAD);
    if (MODE == 5) (void)__builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float*)p, 1.0f);
  }
}

template <int MODE>
void run(const char* name, float* table, unsigned n) {
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int blocks = 4096, per = 64;
  CHECK(hipMemset(table, 0, n * 4));
  k_atom<MODE><<<blocks, 256>>>(table, n - 1, per);
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemset(table, 0, n * 4));
  CHECK(hipEventRecord(e0));
  k_atom<MODE><<<blocks, 256>>>(table, n - 1, per);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  // correctness: total must equal the number of adds (sum over table)
  float* h = (float*)malloc(n * 4);
  CHECK(hipMemcpy(h, table, n * 4, hipMemcpyDeviceToHost));
  double s = 0; for (unsigned i = 0; i < n; ++i) s += h[i];
  free(h);
  const double ops = (double)blocks * 256 * per;
  printf("%-28s table %5.1f MB: %7.3f ms  %6.1f G lane-ops/s   sum/ops = %.6f\n", name, n * 4 / 1e6, ms, ops / ms / 1e6, s / ops);
}

int main() {
  for (unsigned n : {1u << 20, 1u << 24}) {
    float* table; CHECK(hipMalloc(&table, (size_t)n * 4));
    run<0>("atomicAdd", table, n);
    run<1>("scope agent", table, n);
    run<2>("scope workgroup", table, n);
    run<3>("scope wavefront", table, n);
    run<4>("scope singlethread", table, n);
    run<5>("builtin global_atomic_fadd", table, n);
    CHECK(hipFree(table));
  }
  return 0;
}

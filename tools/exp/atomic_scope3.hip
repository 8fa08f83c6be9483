// Micro-experiment: which lanes of one fp32 atomic instruction does MI355X merge into one memory-side request?
// G adjacent lanes hit G adjacent floats starting at a random G-aligned position (+ MIS floats of misalignment).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
template <int G, int MIS>
__global__ __launch_bounds__(256) void k_atom(float* table, unsigned n, int per_thread) {
  const unsigned tid = blockIdx.x * 256 + threadIdx.x;
  for (int i = 0; i < per_thread; ++i) {
    const unsigned key = ((hash((tid / G) * 977u + i * 131071u) & (n / G - 1)) * G + (tid % G) + MIS) & (n - 1);
    __hip_atomic_fetch_add(table + key, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
template <int G, int MIS>
void run(float* table, unsigned n) {
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int blocks = 4096, per = 64;
  for (int rep = 0; rep < 2; ++rep) {
    CHECK(hipMemset(table, 0, (size_t)n * 4));
    CHECK(hipEventRecord(e0));
    k_atom<G, MIS><<<blocks, 256>>>(table, n, per);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
  }
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double ops = (double)blocks * 256 * per;
  printf("group %2d lanes (%3d B) misaligned by %d floats: %7.3f ms  %7.1f G lane-ops/s  %6.1f G groups/s\n", G, G * 4, MIS, ms, ops / ms / 1e6, ops / G / ms / 1e6);
}
int main() {
  const unsigned n = 1u << 21;
  float* table; CHECK(hipMalloc(&table, (size_t)n * 4));
  run<1, 0>(table, n); run<2, 0>(table, n); run<2, 1>(table, n); run<4, 0>(table, n); run<4, 2>(table, n); run<4, 1>(table, n);
  run<8, 0>(table, n); run<8, 4>(table, n); run<16, 0>(table, n); run<16, 8>(table, n); run<32, 0>(table, n); run<64, 0>(table, n);
  return 0;
}

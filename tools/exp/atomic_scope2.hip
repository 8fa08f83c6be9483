// Micro-experiment (round 2): fp32 atomic adds that stay inside one XCD's L2.  Device-scope atomics on MI355X go to the
// memory side (the 8 L2s are not coherent); a workgroup-scope atomic is performed by the issuing XCD's L2.  That is only
// correct when every workgroup that touches a given table lives on the same XCD -- so each XCD gets a PRIVATE copy of the
// table (selected by HW_REG_XCC_ID) and a second kernel sums the copies.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 7; }   // HW_REG_XCC_ID[3:0]

template <int MODE, int QUAD>
__global__ __launch_bounds__(256) void k_atom(float* table, unsigned n, int per_thread) {
  const unsigned tid = blockIdx.x * 256 + threadIdx.x;
  float* base = table;
  if (MODE == 2) base = table + (size_t)xcc_id() * n;
  for (int i = 0; i < per_thread; ++i) {
    // QUAD: 4 adjacent lanes hit 4 adjacent floats (the lane-quad layout of k_ngp_scatter)
    const unsigned key = QUAD ? ((hash((tid >> 2) * 977u + i * 131071u) & (n / 4 - 1)) * 4 + (tid & 3)) : (hash(tid * 977u + i * 131071u) & (n - 1));
    float* p = base + key;
    if (MODE == 0) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (MODE == 1) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // WRONG across XCDs: sum check shows it
    if (MODE == 2) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // private copy per XCD
  }
}
__global__ void k_sum8(const float* t, float* out, unsigned n) {
  for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    float s = 0;
    for (int x = 0; x < 8; ++x) s += t[(size_t)x * n + i];
    out[i] = s;
  }
}
__global__ void k_census(unsigned* hist) { if (threadIdx.x == 0) atomicAdd(&hist[xcc_id()], 1u); }

template <int MODE, int QUAD>
void run(const char* name, float* table, float* out, unsigned n) {
  hipEvent_t e0, e1, e2; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&e2));
  const int blocks = 4096, per = 64;
  const size_t bytes = (size_t)n * 4 * (MODE == 2 ? 8 : 1);
  for (int rep = 0; rep < 2; ++rep) {
    CHECK(hipMemset(table, 0, bytes));
    CHECK(hipEventRecord(e0));
    k_atom<MODE, QUAD><<<blocks, 256>>>(table, n, per);
    CHECK(hipEventRecord(e1));
    if (MODE == 2) k_sum8<<<1024, 256>>>(table, out, n);
    CHECK(hipEventRecord(e2));
    CHECK(hipDeviceSynchronize());
  }
  float ms, ms2; CHECK(hipEventElapsedTime(&ms, e0, e1)); CHECK(hipEventElapsedTime(&ms2, e1, e2));
  float* h = (float*)malloc((size_t)n * 4);
  CHECK(hipMemcpy(h, MODE == 2 ? out : table, (size_t)n * 4, hipMemcpyDeviceToHost));
  double s = 0; for (unsigned i = 0; i < n; ++i) s += h[i];
  free(h);
  const double ops = (double)blocks * 256 * per;
  printf("%-34s table %5.1f MB: %7.3f ms (+%.3f ms sum) %7.1f G lane-ops/s   sum/ops = %.6f\n", name, n * 4 / 1e6, ms, ms2, ops / ms / 1e6, s / ops);
}

int main() {
  unsigned* hist; CHECK(hipMalloc(&hist, 64)); CHECK(hipMemset(hist, 0, 64));
  k_census<<<2048, 64>>>(hist);
  unsigned hh[8]; CHECK(hipMemcpy(hh, hist, 32, hipMemcpyDeviceToHost));
  printf("workgroups per XCC id:"); for (int i = 0; i < 8; ++i) printf(" %u", hh[i]); printf("\n");
  for (unsigned n : {1u << 17, 1u << 21}) {       // 0.5 MB (one level of the tiled grid) and 8 MB (the whole table)
    float *table, *out; CHECK(hipMalloc(&table, (size_t)n * 4 * 8)); CHECK(hipMalloc(&out, (size_t)n * 4));
    run<0, 0>("agent scope, scattered", table, out, n);
    run<1, 0>("workgroup scope, shared (WRONG?)", table, out, n);
    run<2, 0>("workgroup scope, per-XCD copies", table, out, n);
    run<0, 1>("agent scope, lane quads", table, out, n);
    run<2, 1>("workgroup, per-XCD, lane quads", table, out, n);
    CHECK(hipFree(table)); CHECK(hipFree(out));
  }
  return 0;
}

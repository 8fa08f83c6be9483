// Micro-experiment: cost of a device-wide barrier inside a persistent kernel vs a kernel boundary (MI355X).
// Each "stage" every block writes 1 KB of a buffer that ANOTHER block (different XCD) reads in the next stage,
// so the barrier has to make data visible across XCD L2s (agent-scope release/acquire).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE);                  // agent scope: writes back this XCD's dirty lines
    while (__atomic_load_n(counter, __ATOMIC_ACQUIRE) < target) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                     // every wave invalidates its stale lines
}

__global__ __launch_bounds__(256) void k_persistent(float* buf, unsigned* counter, int stages, int* err) {
  const int nb = gridDim.x;
  for (int s = 0; s < stages; ++s) {
    // read what block (b + 37) % nb wrote in stage s-1, write own
    const int src = (blockIdx.x + 37) % nb;
    float v = buf[((s + 1) & 1) * nb * 256 + src * 256 + threadIdx.x];
    if (s > 0 && v != (float)(s - 1 + src)) atomicAdd(err, 1);
    buf[(s & 1) * nb * 256 + blockIdx.x * 256 + threadIdx.x] = (float)(s + blockIdx.x);
    grid_barrier(counter, (unsigned)(s + 1) * nb);
  }
}

__global__ __launch_bounds__(256) void k_stage(float* buf, int s, int* err) {
  const int nb = gridDim.x;
  const int src = (blockIdx.x + 37) % nb;
  float v = buf[((s + 1) & 1) * nb * 256 + src * 256 + threadIdx.x];
  if (s > 0 && v != (float)(s - 1 + src)) atomicAdd(err, 1);
  buf[(s & 1) * nb * 256 + blockIdx.x * 256 + threadIdx.x] = (float)(s + blockIdx.x);
}

int main() {
  const int stages = 400;
  for (int nb : {256, 512, 1024}) {
    float* buf; unsigned* counter; int* err;
    CHECK(hipMalloc(&buf, 2 * nb * 256 * 4)); CHECK(hipMalloc(&counter, 4)); CHECK(hipMalloc(&err, 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipStream_t st; CHECK(hipStreamCreate(&st));
    for (int rep = 0; rep < 2; ++rep) {
      CHECK(hipMemsetAsync(counter, 0, 4, st)); CHECK(hipMemsetAsync(err, 0, 4, st)); CHECK(hipMemsetAsync(buf, 0, 2 * nb * 256 * 4, st));
      void* args[] = {&buf, &counter, (void*)&stages, &err};
      CHECK(hipEventRecord(e0, st));
      CHECK(hipLaunchCooperativeKernel((void*)k_persistent, dim3(nb), dim3(256), args, 0, st));
      CHECK(hipEventRecord(e1, st));
      CHECK(hipStreamSynchronize(st));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      int h; CHECK(hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost));
      if (rep) printf("persistent  blocks=%4d: %.2f us per stage (errors %d)\n", nb, ms * 1e3 / stages, h);
    }
    // same through a captured graph of `stages` dependent launches
    CHECK(hipMemsetAsync(err, 0, 4, st)); CHECK(hipMemsetAsync(buf, 0, 2 * nb * 256 * 4, st));
    hipGraph_t g; hipGraphExec_t ge;
    CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int s = 0; s < stages; ++s) k_stage<<<nb, 256, 0, st>>>(buf, s, err);
    CHECK(hipStreamEndCapture(st, &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 2; ++rep) {
      CHECK(hipEventRecord(e0, st));
      CHECK(hipGraphLaunch(ge, st));
      CHECK(hipEventRecord(e1, st));
      CHECK(hipStreamSynchronize(st));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      int h; CHECK(hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost));
      if (rep) printf("graph nodes blocks=%4d: %.2f us per stage (errors %d)\n", nb, ms * 1e3 / stages, h);
    }
  }
  return 0;
}

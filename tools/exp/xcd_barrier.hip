// Micro-experiment for the persistent 4x4-level kernel: cost of an XCD-HIERARCHICAL grid barrier
// with agent-scope atomics, against the chain of kernel boundaries it would replace.  The round-1 experiment
// (grid_barrier.hip: one counter, 256 pollers, system scope) measured 20 us and was rightly criticised; MI355X_MICROARCH.md
// prices the hierarchical form at 4.1 us idle / 4.8-7.2 us behind real phases.  Each stage every workgroup writes a 1 KB
// record that a workgroup of ANOTHER XCD reads in the next stage, so the barrier has to publish data across the XCD L2s.
//   build: hipcc --offload-arch=gfx950 -O3 tools/exp/xcd_barrier.hip -o /tmp/xcd_barrier && /tmp/xcd_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct Bar {                       // every word on its own 128-byte line
  unsigned xcd_cnt[8][32];         // arrivals per XCD
  unsigned top_cnt[32];            // XCD leaders that arrived
  unsigned gen[8][32];             // release word per XCD (monotonic generation)
};

// Workgroups are dealt round-robin over the 8 XCDs (blockIdx & 7); gridDim must be a multiple of 8 and fully resident.
__device__ __forceinline__ void xcd_barrier(Bar* b, unsigned generation) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned x = blockIdx.x & 7, per_xcd = gridDim.x >> 3;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");                                   // publish this workgroup's writes
    const unsigned n = __hip_atomic_fetch_add(&b->xcd_cnt[x][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (n == per_xcd - 1) {                                                               // last of its XCD: go up one level
      __hip_atomic_store(&b->xcd_cnt[x][0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned t = __hip_atomic_fetch_add(&b->top_cnt[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (t == 7) {                                                                       // last XCD: release everybody
        __hip_atomic_store(&b->top_cnt[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int k = 0; k < 8; ++k) __hip_atomic_store(&b->gen[k][0], generation, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    while (__hip_atomic_load(&b->gen[x][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < generation) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                                      // every wave drops its stale lines
}

__device__ __forceinline__ void stage_body(float* buf, int s, int* err, int work) {
  const int nb = gridDim.x;
  const int src = (blockIdx.x + 37) % nb;                                                 // 37 is odd: a different XCD
  float v = buf[((s + 1) & 1) * nb * 256 + src * 256 + threadIdx.x];
  if (s > 0 && v != (float)(s - 1 + src)) atomicAdd(err, 1);
  float acc = v;
  for (int k = 0; k < work; ++k) acc = fmaf(acc, 1.0000001f, 1e-9f);                     // optional phase work (latency only)
  buf[(s & 1) * nb * 256 + blockIdx.x * 256 + threadIdx.x] = (float)(s + blockIdx.x) + (acc != acc ? 1.0f : 0.0f);
}

__global__ __launch_bounds__(256) void k_persistent(float* buf, Bar* bar, int stages, int* err, int work) {
  for (int s = 0; s < stages; ++s) {
    stage_body(buf, s, err, work);
    xcd_barrier(bar, (unsigned)(s + 1));
  }
}

__global__ __launch_bounds__(256) void k_stage(float* buf, int s, int* err, int work) { stage_body(buf, s, err, work); }

int main() {
  const int stages = 400;
  for (int work : {0, 2000}) {
    for (int nb : {256, 512}) {
      float* buf; Bar* bar; int* err;
      CHECK(hipMalloc(&buf, 2 * nb * 256 * 4)); CHECK(hipMalloc(&bar, sizeof(Bar))); CHECK(hipMalloc(&err, 4));
      hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
      hipStream_t st; CHECK(hipStreamCreate(&st));
      float ms_p = 0.f, ms_g = 0.f;
      int h_p = 0, h_g = 0;
      for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipMemsetAsync(bar, 0, sizeof(Bar), st)); CHECK(hipMemsetAsync(err, 0, 4, st)); CHECK(hipMemsetAsync(buf, 0, 2 * nb * 256 * 4, st));
        void* args[] = {&buf, &bar, (void*)&stages, &err, (void*)&work};
        CHECK(hipEventRecord(e0, st));
        CHECK(hipLaunchCooperativeKernel((void*)k_persistent, dim3(nb), dim3(256), args, 0, st));
        CHECK(hipEventRecord(e1, st));
        CHECK(hipStreamSynchronize(st));
        CHECK(hipEventElapsedTime(&ms_p, e0, e1));
        CHECK(hipMemcpy(&h_p, err, 4, hipMemcpyDeviceToHost));
      }
      // the same stages as a captured graph of dependent launches (what the plan executor replays today)
      CHECK(hipMemsetAsync(err, 0, 4, st)); CHECK(hipMemsetAsync(buf, 0, 2 * nb * 256 * 4, st));
      hipGraph_t g; hipGraphExec_t ge;
      CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      for (int s = 0; s < stages; ++s) k_stage<<<nb, 256, 0, st>>>(buf, s, err, work);
      CHECK(hipStreamEndCapture(st, &g));
      CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipMemsetAsync(err, 0, 4, st));
        CHECK(hipEventRecord(e0, st));
        CHECK(hipGraphLaunch(ge, st));
        CHECK(hipEventRecord(e1, st));
        CHECK(hipStreamSynchronize(st));
        CHECK(hipEventElapsedTime(&ms_g, e0, e1));
        CHECK(hipMemcpy(&h_g, err, 4, hipMemcpyDeviceToHost));
      }
      printf("work %4d  workgroups %4d: xcd barrier %.2f us / stage (errors %d)   kernel boundary %.2f us / stage (errors %d)\n", work, nb,
             ms_p * 1e3 / stages, h_p, ms_g * 1e3 / stages, h_g);
      CHECK(hipGraphExecDestroy(ge)); CHECK(hipGraphDestroy(g));
      CHECK(hipFree(buf)); CHECK(hipFree(bar)); CHECK(hipFree(err));
    }
  }
  return 0;
}

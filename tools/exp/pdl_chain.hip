// Micro-experiment for DESIGN.md section 8, item 1 ("software dependent launch"): a chain of weight-streaming "layers" where
// layer L + 1 is launched with a graph edge to layer L - 1 only (two alternating capture streams), pulls its weight slice into
// registers at once, and waits for layer L on a device-side flag (8 per-XCD arrival counters + one top word) before it touches L's output --
// against the same chain as plainly dependent launches.  Each layer streams its own 18.9 MB of weights (74 KB per workgroup,
// as a 1024 -> 1024 3x3 conv of the 4x4 level does), reads a 4 KB record written by workgroups of OTHER XCDs in the previous
// layer, and writes its own.  Prints us per layer for both forms and checks that both produce the same final records.
//   build: hipcc --offload-arch=gfx950 -O3 tools/exp/pdl_chain.hip -o /tmp/pdl_chain && /tmp/pdl_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int NWG = 256, NT = 512, WPT = 9;                   // 256 workgroups x 512 threads x 9 x 16 B = 18.9 MB per layer
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

struct Flags { unsigned cnt[8][32]; };                        // one 128-byte line per XCD shard

// the protocol of csrc/sf_dev.h (sf_pdl_wait / sf_pdl_arrive): per-XCD arrival shards, the last arriver of a shard bumps a top
// word (kept in shard 0's line, word 16), a waiting workgroup polls that one word with one thread
__device__ __forceinline__ void pdl_wait(const Flags* f, unsigned grid_prev) {
  if (threadIdx.x == 0) {
    const unsigned want = grid_prev < 8u ? grid_prev : 8u;
    while (__hip_atomic_load(&f->cnt[0][16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(2);
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

__device__ __forceinline__ void pdl_arrive(Flags* f) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const unsigned x = blockIdx.x & 7, in_shard = (gridDim.x + 7 - x) >> 3;
    const unsigned n = __hip_atomic_fetch_add(&f->cnt[x][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (n + 1 == in_shard) __hip_atomic_fetch_add(&f->cnt[0][16], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// in / out: [NWG][128] floats (one record per workgroup)
__global__ __launch_bounds__(NT) void k_layer(const u32x4* __restrict__ w, const float* __restrict__ in, float* __restrict__ out,
                                              const Flags* wait_on, Flags* arrive_on, int layer) {
  u32x4 r[WPT];
#pragma unroll
  for (int i = 0; i < WPT; ++i) r[i] = __builtin_nontemporal_load(&w[((long)blockIdx.x * WPT + i) * NT + threadIdx.x]);
  if (wait_on) pdl_wait(wait_on, NWG);
  const int src = (blockIdx.x + 37) % NWG;                    // a workgroup of another XCD
  float v = threadIdx.x < 128 ? in[src * 128 + threadIdx.x] : 0.0f;
  unsigned acc = 0;
#pragma unroll
  for (int i = 0; i < WPT; ++i) acc ^= r[i][0] ^ r[i][1] ^ r[i][2] ^ r[i][3];
  __shared__ unsigned red[NT];
  red[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x < 128) {
    unsigned a = 0;
    for (int k = threadIdx.x; k < NT; k += 128) a ^= red[k];
    out[blockIdx.x * 128 + threadIdx.x] = v * 0.5f + (float)(a & 1023u) + (float)layer;
  }
  if (arrive_on) pdl_arrive(arrive_on);
}

int main() {
  const int layers = 48;
  const size_t wbytes = (size_t)NWG * NT * WPT * 16;
  std::vector<u32x4*> w(layers);
  std::vector<unsigned> hw(wbytes / 4);
  for (int l = 0; l < layers; ++l) {
    CHECK(hipMalloc(&w[l], wbytes));
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = (unsigned)(i * 2654435761u + l * 40503u);
    CHECK(hipMemcpy(w[l], hw.data(), wbytes, hipMemcpyHostToDevice));
  }
  float *a0, *a1; Flags* flags;
  CHECK(hipMalloc(&a0, NWG * 128 * 4)); CHECK(hipMalloc(&a1, NWG * 128 * 4)); CHECK(hipMalloc(&flags, sizeof(Flags) * layers));
  hipStream_t s0, s1; CHECK(hipStreamCreate(&s0)); CHECK(hipStreamCreate(&s1));
  hipEvent_t e0, e1, ef, ej; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); CHECK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
  std::vector<float> res[2];
  for (int mode = 0; mode < 2; ++mode) {                      // 0: dependent launches, 1: alternating streams + flags
    hipGraph_t g; hipGraphExec_t ge;
    CHECK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
    CHECK(hipMemsetAsync(flags, 0, sizeof(Flags) * layers, s0));
    CHECK(hipMemsetAsync(a0, 0, NWG * 128 * 4, s0));
    if (mode) { CHECK(hipEventRecord(ef, s0)); CHECK(hipStreamWaitEvent(s1, ef, 0)); }
    for (int l = 0; l < layers; ++l) {
      hipStream_t st = (mode && (l & 1)) ? s1 : s0;
      const float* in = (l & 1) ? a1 : a0;
      float* out = (l & 1) ? a0 : a1;
      k_layer<<<NWG, NT, 0, st>>>(w[l], in, out, (mode && l > 0) ? flags + (l - 1) : nullptr, mode ? flags + l : nullptr, l);
    }
    if (mode) { CHECK(hipEventRecord(ej, s1)); CHECK(hipStreamWaitEvent(s0, ej, 0)); }
    CHECK(hipStreamEndCapture(s0, &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    float ms = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
      CHECK(hipEventRecord(e0, s0));
      CHECK(hipGraphLaunch(ge, s0));
      CHECK(hipEventRecord(e1, s0));
      CHECK(hipStreamSynchronize(s0));
      CHECK(hipEventElapsedTime(&ms, e0, e1));
    }
    res[mode].resize(NWG * 128);
    CHECK(hipMemcpy(res[mode].data(), (layers & 1) ? a1 : a0, NWG * 128 * 4, hipMemcpyDeviceToHost));
    printf("%s: %.2f us per layer (%d layers x %.1f MB of weights, %.0f GB/s)\n", mode ? "alternating streams + flags" : "dependent launches        ",
           ms * 1e3 / layers, layers, wbytes / 1e6, wbytes * layers / (ms * 1e-3) / 1e9);
    CHECK(hipGraphExecDestroy(ge)); CHECK(hipGraphDestroy(g));
  }
  int bad = 0;
  for (int i = 0; i < NWG * 128; ++i) bad += res[0][i] != res[1][i];
  printf("records differing between the two forms: %d\n", bad);
  return bad != 0;
}

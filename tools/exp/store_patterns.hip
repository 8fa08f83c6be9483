// r04 experiment: what would staging the NGP bin kernel's entries in LDS (coalesced runs) buy?  16-byte stores, 0.6 GB per launch,
// by the length of the contiguous run consecutive lanes write: 1 (every lane its own 16-byte slot: k_ngp_bin today), 4, 16, 64
// (one 1 KB run per wave instruction), and the fully streaming order.  build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// entry index for (global thread t): runs of RUN consecutive entries, run bases permuted pseudo-randomly over the buffer
template <int RUN>
__global__ __launch_bounds__(1024) void k_store(float4* out, uint32_t n_runs_log, int iters) {
  const uint32_t tpg = gridDim.x * 1024u;
  for (int it = 0; it < iters; ++it) {
    const uint32_t t = (uint32_t)it * tpg + blockIdx.x * 1024u + threadIdx.x;
    const uint32_t run = t / RUN, in = t % RUN;
    // bijective scramble of the run index inside 2^n_runs_log (odd multiplier + xor-shift)
    uint32_t r = run * 2654435761u;
    r ^= r >> 15;
    r *= 0x2c1b3c6du;
    r &= (1u << n_runs_log) - 1u;
    out[(size_t)r * RUN + in] = make_float4(1.f, 2.f, 3.f, (float)t);
  }
}
__global__ __launch_bounds__(1024) void k_stream(float4* out, int iters) {
  const uint32_t tpg = gridDim.x * 1024u;
  for (int it = 0; it < iters; ++it) out[(size_t)it * tpg + blockIdx.x * 1024u + threadIdx.x] = make_float4(1.f, 2.f, 3.f, 4.f);
}

int main() {
  const uint32_t LOGN = 25;                       // 2^25 entries x 16 B = 512 MiB
  const size_t N = (size_t)1 << LOGN;
  float4* buf;
  CK(hipMalloc(&buf, N * 16));
  CK(hipMemset(buf, 0, N * 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = 512, iters = (int)(N / (512u * 1024u));      // every entry written once per launch
  auto timeit = [&](const char* name, auto launch) {
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int i = 0; i < 5; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-52s %8.1f us   %6.2f TB/s\n", name, ms * 1e3, (double)N * 16 / (ms * 1e-3) / 1e12);
  };
  timeit("streaming (lane-contiguous, in order)", [&] { k_stream<<<grid, 1024>>>(buf, iters); });
  timeit("runs of 64 entries (1 KB), random run bases", [&] { k_store<64><<<grid, 1024>>>(buf, LOGN - 6, iters); });
  timeit("runs of 16 entries (256 B)", [&] { k_store<16><<<grid, 1024>>>(buf, LOGN - 4, iters); });
  timeit("runs of 4 entries (64 B)", [&] { k_store<4><<<grid, 1024>>>(buf, LOGN - 2, iters); });
  timeit("every lane its own 16-byte slot", [&] { k_store<1><<<grid, 1024>>>(buf, LOGN, iters); });
  return 0;
}

// r04 experiment: what does k_ngp_bin_reduce wait for?  LDS atomic rates by type (random addresses in a 32 KB slice) and the
// kernel's own phases (entries streamed with / without the LDS atomics).  build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(1024) void k_rate(const uint32_t* rows, int iters, float* out) {
  __shared__ double accd[4096];
  float* accf = reinterpret_cast<float*>(accd);
  uint32_t* accu = reinterpret_cast<uint32_t*>(accd);
  unsigned long long* accl = reinterpret_cast<unsigned long long*>(accd);
  for (int i = threadIdx.x; i < 4096; i += 1024) accd[i] = 0.0;
  __syncthreads();
  uint32_t r = rows[blockIdx.x * 1024 + threadIdx.x];
  uint32_t s = 0;
  for (int it = 0; it < iters; ++it) {
    const uint32_t q = r & 4095u;
    if (MODE == 0) atomicAdd(&accf[2 * q], 1.0f);
    if (MODE == 1) atomicAdd(&accu[2 * q], 1u);
    if (MODE == 2) s += atomicAdd(&accu[2 * q], 1u);
    if (MODE == 3) atomicAdd(&accl[q], 1ull);
    if (MODE == 4) atomicAdd(&accd[q], 1.0);
    if (MODE == 5) accf[2 * q] += 1.0f;                      // plain (racy) read-modify-write: the LDS pipe without the atomic unit
    r = r * 1664525u + 1013904223u;
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = accf[0] + (float)s;
}

struct Red { const uint32_t* cursor; const uint32_t* rows; const float2* vals; float* tab; uint32_t cap; };
template <int MODE, int NT>
__global__ __launch_bounds__(NT) void k_reduce(Red a) {
  __shared__ float acc[8192];
  const uint32_t gb = blockIdx.x, tid = threadIdx.x;
  const uint32_t n = a.cursor[gb];
  const uint32_t* rows = a.rows + (size_t)gb * a.cap;
  const float2* vals = a.vals + (size_t)gb * a.cap;
  for (uint32_t k = tid; k < 8192; k += NT) acc[k] = 0.f;
  __syncthreads();
  float s0 = 0.f, s1 = 0.f;
  for (uint32_t i0 = 0; i0 < n; i0 += 4 * NT) {
    uint32_t r[4]; float2 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { uint32_t i = i0 + u * NT + tid; if (i > n - 1) i = n - 1; r[u] = rows[i]; v[u] = vals[i]; }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (i0 + u * NT + tid < n) {
        const uint32_t q = r[u] & 4095u;
        if (MODE == 0) { atomicAdd(&acc[2 * q], v[u].x); atomicAdd(&acc[2 * q + 1], v[u].y); }
        else { s0 += v[u].x + (float)q; s1 += v[u].y; }
      }
    }
  }
  if (MODE != 0) { acc[2 * (tid & 4095)] = s0; acc[2 * (tid & 4095) + 1] = s1; }
  __syncthreads();
  float2* tab = reinterpret_cast<float2*>(a.tab) + (size_t)gb * 4096;
  for (uint32_t q = tid; q < 4096; q += NT) { float2 t = tab[q]; t.x += acc[2 * q]; t.y += acc[2 * q + 1]; tab[q] = t; }
}

int main() {
  const int NB = 1152; const uint32_t cap = 59000, n = 32768;
  std::vector<uint32_t> hrows((size_t)NB * cap), hcur(NB, n);
  std::vector<float2> hvals((size_t)NB * cap);
  uint32_t x = 12345;
  for (size_t i = 0; i < hrows.size(); ++i) { x = x * 1664525u + 1013904223u; hrows[i] = x >> 8; hvals[i] = float2{1.f, 2.f}; }
  uint32_t *drows, *dcur; float2* dvals; float *dtab, *dout;
  CK(hipMalloc(&drows, hrows.size() * 4)); CK(hipMalloc(&dvals, hvals.size() * 8)); CK(hipMalloc(&dcur, NB * 4));
  CK(hipMalloc(&dtab, (size_t)NB * 4096 * 8)); CK(hipMalloc(&dout, 4096 * 4));
  CK(hipMemcpy(drows, hrows.data(), hrows.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dvals, hvals.data(), hvals.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dcur, hcur.data(), NB * 4, hipMemcpyHostToDevice)); CK(hipMemset(dtab, 0, (size_t)NB * 4096 * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, auto launch, double lane_ops) {
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int i = 0; i < 5; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-44s %8.1f us", name, ms * 1e3);
    if (lane_ops > 0) printf("   %7.2f lane-ops / clk / CU (2.4 GHz, 256 CUs)", lane_ops / (ms * 1e-3) / 2.4e9 / 256);
    printf("\n");
  };
  const int iters = 256; const double ops = 512.0 * 1024 * iters;
  timeit("ds_add_f32 (no return), random", [&] { k_rate<0><<<512, 1024>>>(drows, iters, dout); }, ops);
  timeit("ds_add_u32 (no return), random", [&] { k_rate<1><<<512, 1024>>>(drows, iters, dout); }, ops);
  timeit("ds_add_rtn_u32, random", [&] { k_rate<2><<<512, 1024>>>(drows, iters, dout); }, ops);
  timeit("ds_add_u64, random", [&] { k_rate<3><<<512, 1024>>>(drows, iters, dout); }, ops);
  timeit("ds_add_f64, random", [&] { k_rate<4><<<512, 1024>>>(drows, iters, dout); }, ops);
  timeit("plain LDS read-add-write, random", [&] { k_rate<5><<<512, 1024>>>(drows, iters, dout); }, ops);
  Red a{dcur, drows, dvals, dtab, cap};
  const double eops = (double)NB * n * 2;
  timeit("reduce, 1024 threads, LDS f32 atomics", [&] { k_reduce<0, 1024><<<NB, 1024>>>(a); }, eops);
  timeit("reduce, 1024 threads, loads only", [&] { k_reduce<1, 1024><<<NB, 1024>>>(a); }, 0);
  timeit("reduce, 256 threads, LDS f32 atomics", [&] { k_reduce<0, 256><<<NB, 256>>>(a); }, eops);
  timeit("reduce, 256 threads, loads only", [&] { k_reduce<1, 256><<<NB, 256>>>(a); }, 0);
  printf("entries: %.0f MB per launch\n", (double)NB * n * 12 / 1e6);
  return 0;
}

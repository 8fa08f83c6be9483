// What does a dependent launch cost when consecutive launches are DIFFERENT kernels with real code footprints?
// (kernarg_lat.hip: 1.7 us per launch for ONE tiny kernel replayed 200 x from a graph; the UNet eval's smallest kernels take
// 4.6-5.8 us each in the rocprofv3 trace although their dependent chains are ~1.5 us.)  Variables: number of distinct kernels
// in the chain (1 / 16), straight-line code executed per wave (0 / 2048 / 8192 FMAs = 0 / 16 / 64 KB of code), workgroup
// geometry (64 x 256 threads / 256 x 512 threads), static LDS (0 / 64 KB).  Every workgroup stamps entry and exit (100 MHz).
//   hipcc --offload-arch=gfx950 -O3 tools/exp/launch_floor.hip -o tools/exp/launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Args { const float* src; float* dst; long long* stamps; int launch, n, wgs; float k; };

template <int ID, int PAD, int LDS_KB, int THREADS>
__global__ void __launch_bounds__(THREADS) k_chain(Args a) {
  const long long t0 = (long long)wall_clock64();
  __shared__ float lds[LDS_KB ? LDS_KB * 256 : 1];
  const int i = blockIdx.x * THREADS + threadIdx.x;
  float v = a.src[(i * 17 + ID) % a.n];                    // produced by other workgroups of the previous launch
  if (LDS_KB) { lds[threadIdx.x] = v; __syncthreads(); v = lds[(threadIdx.x + 1) % THREADS]; }
  float x = v, y = a.k;
#pragma unroll
  for (int p = 0; p < PAD; ++p) x = __builtin_fmaf(x, y, (float)(p + ID));   // straight-line, executed, distinct per ID
  if (i < a.n) a.dst[i] = x * 1e-30f + v * a.k + 1.0f;
  const long long t1 = (long long)wall_clock64();
  if (threadIdx.x == 0) {
    long long* st = a.stamps + ((long)a.launch * a.wgs + blockIdx.x) * 2;
    st[0] = t0; st[1] = t1;
  }
}

typedef void (*kfn)(Args);
template <int PAD, int LDS_KB, int THREADS, int... IDS>
static std::vector<kfn> table(std::integer_sequence<int, IDS...>) { return {k_chain<IDS, PAD, LDS_KB, THREADS>...}; }

template <int PAD, int LDS_KB, int THREADS>
static void run(const char* tag, int wgs, int distinct, float* b0, float* b1, long long* st, int N, hipStream_t s) {
  const int NL = 160;
  auto ks = table<PAD, LDS_KB, THREADS>(std::make_integer_sequence<int, 16>{});
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int l = 0; l < NL; ++l) {
    Args a{(l & 1) ? b1 : b0, (l & 1) ? b0 : b1, st, l, N, wgs, 0.5f};
    hipLaunchKernelGGL(ks[l % distinct], dim3(wgs), dim3(THREADS), 0, s, a);
  }
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0) best = std::min(best, ms);
  }
  std::vector<long long> hs((size_t)NL * wgs * 2);
  CK(hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost));
  double span = 0, gap = 0, first = 0; int cnt = 0;
  long long prev_end = 0;
  for (int l = 16; l < NL; ++l) {
    long long mn = 1LL << 62, mx = 0, mxs = 0;
    for (int w = 0; w < wgs; ++w) { mn = std::min(mn, hs[((size_t)l * wgs + w) * 2]); mxs = std::max(mxs, hs[((size_t)l * wgs + w) * 2]); mx = std::max(mx, hs[((size_t)l * wgs + w) * 2 + 1]); }
    span += (mx - mn) * 0.01; first += (mxs - mn) * 0.01;
    if (l > 16) gap += (mn - prev_end) * 0.01;
    prev_end = mx; ++cnt;
  }
  printf("%-44s wgs %3d x %3d  distinct %2d | %6.2f us/launch | first entry -> last exit %5.2f  (entry spread %5.2f)  last exit -> next first entry %5.2f\n",
         tag, wgs, THREADS, distinct, best * 1000.f / NL, span / cnt, first / cnt, gap / (cnt - 1));
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
}

int main() {
  const int N = 1 << 17;
  float *b0, *b1; long long* st;
  CK(hipMalloc(&b0, N * 4)); CK(hipMalloc(&b1, N * 4)); CK(hipMalloc(&st, (size_t)160 * 256 * 2 * 8));
  CK(hipMemset(b0, 0, N * 4)); CK(hipMemset(b1, 0, N * 4));
  hipStream_t s; CK(hipStreamCreate(&s));
  for (int distinct : {1, 16}) {
    run<0, 0, 256>("tiny code", 64, distinct, b0, b1, st, N, s);
    run<0, 0, 512>("tiny code", 256, distinct, b0, b1, st, N, s);
    run<0, 64, 512>("tiny code, 64 KB LDS", 256, distinct, b0, b1, st, N, s);
    run<2048, 0, 512>("16 KB straight-line code", 256, distinct, b0, b1, st, N, s);
    run<8192, 0, 512>("64 KB straight-line code", 256, distinct, b0, b1, st, N, s);
    run<8192, 64, 512>("64 KB straight-line code, 64 KB LDS", 256, distinct, b0, b1, st, N, s);
  }
  return 0;
}

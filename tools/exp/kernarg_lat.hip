// Where do the ~2 us between wave entry and the first use of a kernel argument go?  (profiles/r02_fconv_phases.log:
// "entry->first loads issued 2.27" on the 4x4 fused conv, whose first loads need the weight pointer.)
// A chain of dependent launches, each reads 4 KB its predecessor wrote on other XCDs and writes 4 KB; three ways of
// getting at the arguments:
//   A  460-byte struct by value (what FConvArgs is today), fields used from the END of the struct
//   B  16-byte kernarg = pointer into a device-resident argument table (written once) + index
//   C  as B, but compiled with -mllvm -amdgpu-kernarg-preload-count=4 (build this file twice: -DPRELOAD_BUILD)
// In-kernel 100 MHz stamps: entry -> argument usable -> first dependent global load back -> store issued.
// Run under HIP_FORCE_DEV_KERNARG=0 and =1, plain stream and captured graph.
//   hipcc --offload-arch=gfx950 -O3 tools/exp/kernarg_lat.hip -o /tmp/kernarg_lat
//   hipcc --offload-arch=gfx950 -O3 -DPRELOAD_BUILD -mllvm -amdgpu-kernarg-preload-count=4 tools/exp/kernarg_lat.hip -o /tmp/kernarg_lat_pre
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Big {
  float pad[104];            // 416 bytes of other arguments
  const float* src;          // used first: sits in the LAST 64-byte line of the segment
  float* dst;
  long long* stamps;         // [launch][4]
  int launch, n;
  float k;
  int pad2[3];
};
static_assert(sizeof(Big) >= 448 && sizeof(Big) <= 480, "about the size of FConvArgs");

struct Small { const float* src; float* dst; long long* stamps; int launch, n; float k; int pad; };

template <class A>
__device__ __forceinline__ void body(const A& a, long long t0) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const float* s = a.src;
  asm volatile("" ::"s"(s));
  // argument usable: force the scalar value to exist before the stamp
  long long t1 = (long long)wall_clock64();
  float v = (i < a.n) ? s[(i * 17) % a.n] : 0.f;                 // permuted: produced by other workgroups / XCDs
  asm volatile("s_waitcnt vmcnt(0)" ::"v"(v) : "memory");
  long long t2 = (long long)wall_clock64();
  if (i < a.n) a.dst[i] = v * a.k + 1.0f;
  long long t3 = (long long)wall_clock64();
  if (blockIdx.x == 3 && threadIdx.x == 0) {
    long long* st = a.stamps + (long)a.launch * 4;
    st[0] = t0; st[1] = t1; st[2] = t2; st[3] = t3;
  }
}

__global__ void __launch_bounds__(256) k_big(Big a) {
  long long t0 = (long long)wall_clock64();
  body(a, t0);
}
__global__ void __launch_bounds__(256) k_tab(const Small* tab, int idx) {
  long long t0 = (long long)wall_clock64();
  const Small a = tab[idx];
  body(a, t0);
}

int main(int argc, char** argv) {
  const int NL = 200, N = 1024, WG = 64;       // 64 workgroups x 16 floats... n = 1024 floats = 4 KB
  float *b0, *b1; long long* st; Small* tab;
  CK(hipMalloc(&b0, N * 4)); CK(hipMalloc(&b1, N * 4)); CK(hipMalloc(&st, NL * 4 * 8)); CK(hipMalloc(&tab, NL * sizeof(Small)));
  CK(hipMemset(b0, 0, N * 4));
  std::vector<Small> ht(NL);
  for (int l = 0; l < NL; ++l) ht[l] = Small{(l & 1) ? b1 : b0, (l & 1) ? b0 : b1, st, l, N, 0.5f, 0};
  CK(hipMemcpy(tab, ht.data(), NL * sizeof(Small), hipMemcpyHostToDevice));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<long long> hs(NL * 4);
  for (int variant = 0; variant < 2; ++variant) {
    for (int mode = 0; mode < 2; ++mode) {      // 0 plain stream, 1 graph
      auto enqueue = [&] {
        for (int l = 0; l < NL; ++l) {
          if (variant == 0) {
            Big a{}; a.src = ht[l].src; a.dst = ht[l].dst; a.stamps = st; a.launch = l; a.n = N; a.k = 0.5f;
            hipLaunchKernelGGL(k_big, dim3(WG), dim3(256), 0, s, a);
          } else {
            hipLaunchKernelGGL(k_tab, dim3(WG), dim3(256), 0, s, (const Small*)tab, l);
          }
        }
      };
      hipGraphExec_t ge = nullptr;
      if (mode == 1) {
        hipGraph_t g;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        enqueue();
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      }
      float best = 1e9f;
      for (int rep = 0; rep < 6; ++rep) {
        CK(hipEventRecord(e0, s));
        if (mode == 1) CK(hipGraphLaunch(ge, s)); else enqueue();
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
      }
      CK(hipMemcpy(hs.data(), st, NL * 4 * 8, hipMemcpyDeviceToHost));
      double d01 = 0, d12 = 0, d23 = 0, gap = 0; int cnt = 0;
      for (int l = 20; l < NL; ++l) {
        d01 += (hs[l * 4 + 1] - hs[l * 4]) * 0.01; d12 += (hs[l * 4 + 2] - hs[l * 4 + 1]) * 0.01; d23 += (hs[l * 4 + 3] - hs[l * 4 + 2]) * 0.01;
        gap += (hs[l * 4] - hs[(l - 1) * 4 + 3]) * 0.01; ++cnt;
      }
#ifdef PRELOAD_BUILD
      const char* tag = variant == 0 ? "A' 460 B by value, preload build" : "C  table pointer + kernarg preload";
#else
      const char* tag = variant == 0 ? "A  460 B by value" : "B  table pointer";
#endif
      printf("%-36s %-6s  %6.2f us/launch | entry->arg %5.2f  arg->load back %5.2f  ->store issued %5.2f  prev store->entry %5.2f\n", tag,
             mode ? "graph" : "stream", best * 1000.f / NL, d01 / cnt, d12 / cnt, d23 / cnt, gap / cnt);
    }
  }
  return 0;
}

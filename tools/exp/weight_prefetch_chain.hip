// Can the weight stream of launch l + 1 run UNDER launch l?  (r05)
//
// The 4x4 / 8x8 GroupNorm-fused convs of the UNet are chains of dependent launches that each pull 19-38 MB of weights nobody has
// touched since the previous eval (800 MB per eval: nothing stays in the 8 x 4 MB L2s or the 256 MB Infinity Cache), as one burst
// at kernel entry.  This chain imitates the 4x4 conv (256 workgroups x 8 waves; a workgroup reads a 64 KB lazy activation slice that
// the previous launch wrote on other XCDs, reduces it, "normalises" it through LDS, then multiplies it with a 72 / 144 KB weight
// slab that it requested at entry with non-temporal loads) and adds a NINTH wave per workgroup that touches one dword per 128-byte
// line of the slab the SAME workgroup index will want in the next launch (same XCD under round-robin placement -> same L2).
//   variants   none      8 waves, no prefetch (the product today)
//              same      9th wave prefetches slab(w) of the next launch            (L2 of the consumer's XCD)
//              shifted   9th wave prefetches slab((w + 1) % 256) of the next launch (another XCD: Infinity Cache only)
//              hot       every launch uses the same weights, no prefetch            (bound: weights already in L2)
//              noweights no weight loads at all                                     (bound: the activation path alone)
//   hipcc --offload-arch=gfx950 -O3 tools/exp/weight_prefetch_chain.hip -o /tmp/weight_prefetch_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <string>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(4))) float f32x4;

struct Args {
  const float* xin;      // [4 slabs][16 px][1024 ch]
  float* xout;
  const f32x4* w;        // this launch's weights: [256 workgroups][KB][64 lanes] float4
  const char* wnext;     // next launch's weights (prefetch target) or null
  int kb;                // KiB of weights per workgroup (72 | 144), multiple of 8
  int pf_shift;          // prefetch slab of workgroup (w + pf_shift) % 256
  int use_w;
};

template <int KB>
__global__ void __launch_bounds__(576) k_layer(Args a) {
  __shared__ float red[16];
  __shared__ float frame[16 * 260];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int w = blockIdx.x, s = w >> 6, nt = w & 63;
  if (wave == 8) {                                   // the prefetcher: KB KiB = KB * 8 lines, 64 lines per load instruction
    if (a.wnext) {
      const char* p = a.wnext + (size_t)((w + a.pf_shift) & 255) * KB * 1024 + lane * 128;
      float t[KB / 8];
#pragma unroll
      for (int i = 0; i < KB / 8; ++i) t[i] = *reinterpret_cast<const float*>(p + (size_t)i * 8192);
#pragma unroll
      for (int i = 0; i < KB / 8; ++i) asm volatile("" :: "v"(t[i]));
    }
    return;
  }
  // (1) the lazy activation slice: 16 px x 256 ch x 4 slabs = 64 KB per workgroup, 8 float4 per thread, all in flight together
  f32x4 v[8];
  {
    const int e = tid;                               // 0..511 -> (px, c4) of a 16 x 64-float4 tile, two tiles per slab pair
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int slab = u & 3, half = u >> 2;
      const int idx = e + half * 512;                // 0..1023 = 16 px x 64 float4
      const int px = idx >> 6, c4 = idx & 63;
      v[u] = *reinterpret_cast<const f32x4*>(a.xin + ((size_t)slab * 16 + px) * 1024 + s * 256 + c4 * 4);
    }
  }
  // (2) weight ring right behind them: KB KiB / 8 waves, non-temporal
  constexpr int NW = KB / 8;                         // float4 loads per thread
  f32x4 wr[NW];
  if (a.use_w) {
#pragma unroll
    for (int i = 0; i < NW; ++i) wr[i] = __builtin_nontemporal_load(&a.w[((size_t)w * KB + wave * NW + i) * 64 + lane]);
  } else {
#pragma unroll
    for (int i = 0; i < NW; ++i) wr[i] = f32x4{1.f, 2.f, 3.f, 4.f};
  }
  // (3) statistics: sum of the slice -> LDS -> every thread
  float sm = 0.f;
#pragma unroll
  for (int u = 0; u < 8; ++u) sm += (v[u][0] + v[u][1]) + (v[u][2] + v[u][3]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 64);
  if (lane == 0) red[wave] = sm;
  __syncthreads();
  float mean = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) mean += red[k];
  mean *= (1.0f / 16384.0f);
  // (4) "normalise + SiLU" into the LDS frame
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    f32x4 x = (v[half * 4] + v[half * 4 + 1]) + (v[half * 4 + 2] + v[half * 4 + 3]);
    const int idx = tid + half * 512, px = idx >> 6, c4 = idx & 63;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float y = (x[j] - mean) * 0.5f; x[j] = y / (1.0f + __expf(-y)); }
    *reinterpret_cast<f32x4*>(&frame[px * 260 + c4 * 4]) = x;
  }
  __syncthreads();
  // (5) main loop: every weight float4 meets 4 frame values
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < NW; ++i) {
    const f32x4 f = *reinterpret_cast<const f32x4*>(&frame[(lane & 15) * 260 + ((wave * NW + i) & 63) * 4]);
    acc = __builtin_fmaf(wr[i][0], f[0], acc); acc = __builtin_fmaf(wr[i][1], f[1], acc);
    acc = __builtin_fmaf(wr[i][2], f[2], acc); acc = __builtin_fmaf(wr[i][3], f[3], acc);
  }
  // (6) epilogue: the 8 waves meet in LDS, 16 x 16 outputs leave as one slab piece
  __syncthreads();
  frame[wave * 64 + lane] = acc;
  __syncthreads();
  if (tid < 256) {
    float r = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) r += frame[k * 64 + (tid & 63)];
    const int px = tid >> 4, c = tid & 15;
    a.xout[((size_t)s * 16 + px) * 1024 + nt * 16 + c] = r * 1e-3f + (float)(tid >> 6) * 0.25f;
  }
}

template <int KB>
static double run(const char* tag, int nbuf, int pf_shift, bool prefetch, bool use_w, const char* wbuf, float* x0, float* x1, hipStream_t st) {
  const int NL = 64;
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  const size_t layer_bytes = (size_t)256 * KB * 1024;
  for (int l = 0; l < NL; ++l) {
    Args a{(l & 1) ? x1 : x0, (l & 1) ? x0 : x1, reinterpret_cast<const f32x4*>(wbuf + (size_t)(l % nbuf) * layer_bytes),
           prefetch ? wbuf + (size_t)((l + 1) % nbuf) * layer_bytes : nullptr, KB, pf_shift, use_w ? 1 : 0};
    hipLaunchKernelGGL(k_layer<KB>, dim3(256), dim3(prefetch ? 576 : 512), 0, st, a);
  }
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int rep = 0; rep < 6; ++rep) {
    CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0) best = std::min(best, ms);
  }
  const double us = best * 1000.0 / NL;
  printf("%-10s slab %3d KiB/wg (%5.1f MB per launch) | %6.2f us/launch | %6.2f TB/s of weights\n", tag, KB, layer_bytes / 1e6, us,
         use_w ? layer_bytes / us * 1e-6 : 0.0);
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  return us;
}

// argv[1] (optional): run only this variant of the 72 KiB chain, once ("none" | "hot" | "noweights" | "same" | "shifted") -- the target of a
// rocprofv3 --pmc FETCH_SIZE pass that calibrates the counter on this access pattern (known bytes: 256 x 72 KiB of nt dwordx4 weight loads).
int main(int argc, char** argv) {
  const int NBUF = 16;                                           // 16 x 37.7 MB = 604 MB of distinct weights: beyond L2 + Infinity Cache
  const size_t wbytes = (size_t)NBUF * 256 * 144 * 1024;
  char* wbuf; float *x0, *x1;
  CK(hipMalloc(&wbuf, wbytes)); CK(hipMalloc(&x0, 4 * 16 * 1024 * 4)); CK(hipMalloc(&x1, 4 * 16 * 1024 * 4));
  CK(hipMemset(wbuf, 0, wbytes)); CK(hipMemset(x0, 0, 4 * 16 * 1024 * 4)); CK(hipMemset(x1, 0, 4 * 16 * 1024 * 4));
  hipStream_t st; CK(hipStreamCreate(&st));
  if (argc > 1) {
    const std::string v = argv[1];
    if (v == "none") run<72>("none", NBUF, 0, false, true, wbuf, x0, x1, st);
    else if (v == "hot") run<72>("hot", 1, 0, false, true, wbuf, x0, x1, st);
    else if (v == "noweights") run<72>("noweights", NBUF, 0, false, false, wbuf, x0, x1, st);
    else if (v == "same") run<72>("same", NBUF, 0, true, true, wbuf, x0, x1, st);
    else if (v == "shifted") run<72>("shifted", NBUF, 1, true, true, wbuf, x0, x1, st);
    return 0;
  }
  for (int pass = 0; pass < 2; ++pass) {
    run<72>("none", NBUF, 0, false, true, wbuf, x0, x1, st);
    run<72>("same", NBUF, 0, true, true, wbuf, x0, x1, st);
    run<72>("shifted", NBUF, 1, true, true, wbuf, x0, x1, st);
    run<72>("hot", 1, 0, false, true, wbuf, x0, x1, st);
    run<72>("noweights", NBUF, 0, false, false, wbuf, x0, x1, st);
    run<144>("none", NBUF, 0, false, true, wbuf, x0, x1, st);
    run<144>("same", NBUF, 0, true, true, wbuf, x0, x1, st);
    run<144>("shifted", NBUF, 1, true, true, wbuf, x0, x1, st);
    run<144>("hot", 1, 0, false, true, wbuf, x0, x1, st);
    run<144>("noweights", NBUF, 0, false, false, wbuf, x0, x1, st);
  }
  return 0;
}

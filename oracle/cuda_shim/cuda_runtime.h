// oracle/cuda_shim -- TEST INFRASTRUCTURE ONLY.  Lets g++ compile the reference's CUDA sources
// (/root/reference/external/gridencoder/src/gridencoder.cu, /root/reference/raymarching/src/raymarching.cu)
// as plain host C++, WHERE THEY LIE, into oracle/_ref/libref_native.so (recipe: oracle/build_ref.py).
// The kernels of those two files use no shared memory, no barriers and no shuffles, so a launch is a serial loop
// over (block, thread): one valid schedule of the CUDA grid (atomics included).  What this shim changes relative
// to nvcc and why it cannot matter for the pinned integer outputs: see oracle/build_ref.py.
#pragma once
#include <cmath>
#include <math.h>   // the C++ wrapper: float overloads of atan2 / ceil / fabs ... in the GLOBAL namespace, as CUDA has them
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <type_traits>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct shim_uint3 { unsigned x, y, z; };
extern thread_local shim_uint3 threadIdx, blockIdx;
extern thread_local dim3 blockDim, gridDim;

namespace shim {
// kernel<<<grid, block>>>(args)  ->  shim::launch(grid, block, [&]{ kernel(args); })   (rewritten by build_ref.py)
template <class F>
inline void launch(dim3 g, dim3 b, F&& body) {
  gridDim = g; blockDim = b;
  for (unsigned bz = 0; bz < g.z; ++bz) for (unsigned by = 0; by < g.y; ++by) for (unsigned bx = 0; bx < g.x; ++bx)
    for (unsigned tz = 0; tz < b.z; ++tz) for (unsigned ty = 0; ty < b.y; ++ty) for (unsigned tx = 0; tx < b.x; ++tx) {
      blockIdx = {bx, by, bz}; threadIdx = {tx, ty, tz};
      body();
    }
}
}  // namespace shim

// atomicAdd returns the old value; launches are serial, so a plain read-modify-write is one valid outcome
template <class T, class U>
inline T atomicAdd(T* a, U v) { T o = *a; *a = o + (T)v; return o; }
// the fp32 atomic is an instruction of its own on the GPU: its operand is a rounded product, never the multiplicand of a
// fused multiply-add -- noinline keeps -ffp-contract=fast from fusing `w * g` into the add
__attribute__((noinline)) inline float atomicAdd(float* a, float v) { float o = *a; *a = o + v; return o; }

// CUDA's global-namespace integer min / max (device overloads)
inline int max(int a, int b) { return a > b ? a : b; }
inline int min(int a, int b) { return a < b ? a : b; }
inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }

// nvcc's fast-math intrinsic: ex2.approx(x * log2 e), max 2 ulp.  The host build uses the correctly rounded expf;
// comparisons of composite outputs against this library therefore carry a few-ulp tolerance (tests say so).
inline float __expf(float x) { return expf(x); }

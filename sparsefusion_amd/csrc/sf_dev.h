// Device-side vocabulary of the fused UNet kernels (gfx950): vector types, the MFMA tile op, wave reductions.
// Built two ways: by hipcc for gfx950 (the product), and by the host clang with -DSF_HOST_EMU for the kernel-logic
// tests of tests/hostemu (every lane a fiber on the CPU, deterministic schedule; test infrastructure, never shipped or timed).
#pragma once
#include <math.h>
#include <stdint.h>
#include "sf_operand.h"

typedef __attribute__((ext_vector_type(8))) sf_opnd bf16x8;
typedef __attribute__((ext_vector_type(4))) sf_opnd bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#ifdef SF_HOST_EMU
#include "hip_emu.h"
#define SF_KERNEL(...)
#define SF_DEV inline
#define SF_DYN_LDS(name) char* name = hipemu::dyn_smem()
#define SF_SHARED static
static inline void sf_sync() { hipemu::syncthreads(); }
template <class T>
static inline T sf_shfl_xor(T v, int m) { return hipemu::shfl_xor(v, m); }
template <class T>
static inline T sf_shfl(T v, int src) { return hipemu::shfl_xor(v, (src ^ hipemu::t_lane) & 63); }
static inline uint32_t sf_readlane(uint32_t v, uint32_t src) { return sf_shfl(v, (int)src); }      // src is wave-uniform
static inline int sf_uniform(int v) { return v; }
static inline float sf_exp(float v) { return expf(v); }
static inline float sf_exp2(float v) { return exp2f(v); }
static inline void sf_lds_add(float* p, float v) {
  uint32_t* u = reinterpret_cast<uint32_t*>(p);
  uint32_t old = __atomic_load_n(u, __ATOMIC_RELAXED), neu;
  do {
    float f;
    memcpy(&f, &old, 4);
    f += v;
    memcpy(&neu, &f, 4);
  } while (!__atomic_compare_exchange_n(u, &old, neu, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
}
static inline void sf_lds_add_f64(double* p, double v) {
  uint64_t* u = reinterpret_cast<uint64_t*>(p);
  uint64_t old = __atomic_load_n(u, __ATOMIC_RELAXED), neu;
  do {
    double f;
    memcpy(&f, &old, 8);
    f += v;
    memcpy(&neu, &f, 8);
  } while (!__atomic_compare_exchange_n(u, &old, neu, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
}
static inline float sf_rsqrt(float v) { return 1.0f / sqrtf(v); }
static inline void sf_wave_sync() { hipemu::t_wave->bar.wait(); }
static inline void sf_global_add(float* p, float v) { sf_lds_add(p, v); }
// v_mfma_f32_16x16x4_f32: A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15], D[i = 4 * (lane >> 4) + r][j = lane & 15]
static inline f32x4 sf_mfma4(float a, float b, f32x4 c) {
  hipemu::WaveState* w = hipemu::t_wave;
  const int lane = hipemu::t_lane;
  static float xa[64][64], xb[64][64];
  const int ws = (int)(threadIdx.x >> 6);
  xa[ws][lane] = a;
  xb[ws][lane] = b;
  w->bar.wait();
  f32x4 d = c;
  const int j = lane & 15;
  for (int r = 0; r < 4; ++r) {
    const int i = 4 * (lane >> 4) + r;
    float acc = d[r];
    for (int k = 0; k < 4; ++k) acc = fmaf(xa[ws][k * 16 + i], xb[ws][k * 16 + j], acc);
    d[r] = acc;
  }
  w->bar.wait();
  return d;
}
static inline float sf_rcp(float v) { return 1.0f / v; }
static inline long long sf_clock() { return 0; }
template <int BYTES>
static inline void sf_touch_kernarg() {}
// D = A (16 x 32, rows = lane & 15) * B (32 x 16, cols = lane & 15) + C; see tests/hostemu/hip_emu.h for the layout
static inline f32x4 sf_mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
  hipemu::WaveState* w = hipemu::t_wave;
  const int lane = hipemu::t_lane;
  struct Pair { bf16x8 a, b; };
  static Pair xa[64][64];                       // [wave slot][lane]; one workgroup alive at a time, <= 64 waves
  const int ws = (int)(threadIdx.x >> 6);
  xa[ws][lane].a = a;
  xa[ws][lane].b = b;
  w->bar.wait();
  f32x4 d = c;
  const int n = lane & 15;
  for (int r = 0; r < 4; ++r) {
    const int m = 4 * (lane >> 4) + r;
    float acc = 0.0f;
    for (int kb = 0; kb < 4; ++kb)
      for (int j = 0; j < 8; ++j) acc += (float)xa[ws][kb * 16 + m].a[j] * (float)xa[ws][kb * 16 + n].b[j];
    d[r] += acc;
  }
  w->bar.wait();
  return d;
}
// LDS-DMA (global_load_lds_dwordx4) under emulation.  The copy is DEFERRED until the sf_vmcnt<N>() that retires it (the latest moment
// the hardware may land it: a read that does not sit behind the right counted wait + barrier sees stale data and the test fails);
// HIPEMU_GLDS_IMMEDIATE=1 lands it at issue instead (the earliest moment: a buffer restaged while another wave still reads it shows).
struct SfGldsPending { char* dst; const char* src; };
struct SfGldsLane { SfGldsPending q[64]; int head = 0, n = 0; };
static SfGldsLane sf_glds_lane[1024];                     // one queue per lane of the workgroup (the lanes are fibers on one OS thread, hip_emu.h)
static inline bool sf_glds_immediate() { static const bool v = getenv("HIPEMU_GLDS_IMMEDIATE") && atoi(getenv("HIPEMU_GLDS_IMMEDIATE")); return v; }
static inline void sf_glds16(char* lds_wave_base, const void* gsrc) {
  char* dst = lds_wave_base + hipemu::t_lane * 16;
  if (sf_glds_immediate()) { memcpy(dst, gsrc, 16); return; }
  SfGldsLane& g = sf_glds_lane[threadIdx.x];
  if (g.n == 64) { fprintf(stderr, "sf_glds16: more than 64 LDS-DMA loads in flight (vmcnt is 6 bits)\n"); abort(); }
  g.q[(g.head + g.n++) & 63] = SfGldsPending{dst, (const char*)gsrc};
}
template <int N>
static inline void sf_vmcnt() {
  SfGldsLane& g = sf_glds_lane[threadIdx.x];
  while (g.n > N) { memcpy(g.q[g.head].dst, g.q[g.head].src, 16); g.head = (g.head + 1) & 63; --g.n; }
}
static inline void sf_lds_barrier() { hipemu::syncthreads(); }
#define SF_SCHED_GROUP(mask, n) do { } while (0)
#define SF_LGKM0() do { } while (0)
#define SF_USE_FROM_HERE(x) do { } while (0)
static inline void sf_glds_done() { if (sf_glds_lane[threadIdx.x].n) { fprintf(stderr, "LDS-DMA loads still in flight at kernel end\n"); abort(); } }
static const uint32_t sf_zero128[32] __attribute__((aligned(128))) = {0};
#else
#include <hip/hip_runtime.h>
#define SF_KERNEL(...) __global__ __launch_bounds__(__VA_ARGS__)
#define SF_DEV __device__ __forceinline__
#define SF_DYN_LDS(name) extern __shared__ __attribute__((aligned(16))) char name[]
#define SF_SHARED __shared__
SF_DEV void sf_sync() { __syncthreads(); }
template <class T>
SF_DEV T sf_shfl_xor(T v, int m) { return __shfl_xor(v, m, 64); }
template <class T>
SF_DEV T sf_shfl(T v, int src) { return __shfl(v, src, 64); }
SF_DEV int sf_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }      // tell the compiler v is wave-uniform (scalar registers)
SF_DEV uint32_t sf_readlane(uint32_t v, uint32_t src) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(src)); }   // src wave-uniform
SF_DEV float sf_exp(float v) { return __expf(v); }
SF_DEV float sf_exp2(float v) { return __builtin_amdgcn_exp2f(v); }      // raw v_exp_f32 (no denormal fix-up: the result feeds 1 + e)
SF_DEV void sf_lds_add(float* p, float v) { atomicAdd(p, v); }
// gfx950: ds_add_f32 retires 0.33 lane-operations per clock and CU, ds_add_f64 2.3, ds_add_u32 5.5 (random addresses in a 32 KB
// slice, tools/exp/lds_atomic_rate.hip, profiles/r04_lds_atomic_rate.log) -- LDS accumulators that take many atomics are doubles
SF_DEV void sf_lds_add_f64(double* p, double v) { atomicAdd(p, v); }
SF_DEV float sf_rsqrt(float v) { return rsqrtf(v); }
// same-wave LDS hand-off: DS operations of one wave execute in order; this only keeps the compiler from reordering them
SF_DEV void sf_wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
typedef __attribute__((address_space(1))) float sf_dev_gfloat;
SF_DEV void sf_global_add(float* p, float v) { (void)__builtin_amdgcn_global_atomic_fadd_f32((sf_dev_gfloat*)p, v); }
SF_DEV f32x4 sf_mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
SF_DEV float sf_rcp(float v) { return __builtin_amdgcn_rcpf(v); }
SF_DEV long long sf_clock() { return (long long)wall_clock64(); }      // 100 MHz constant clock
#if SF_OPERAND_F16
SF_DEV f32x4 sf_mfma16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
#else
SF_DEV f32x4 sf_mfma16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
#endif
// Touch every 64-byte line of the kernel-argument segment with independent scalar loads and wait ONCE.  The compiler
// fetches a large by-value argument struct piecemeal, each piece right before its first use and each behind its own
// s_waitcnt: ~8 serialised cold misses of the scalar cache (~0.4 us each) in front of the first vector load of a 460-byte
// FConvArgs.  After this call the lines sit in the scalar cache and those loads are hits.
template <int BYTES>
SF_DEV void sf_touch_kernarg() {
  typedef __attribute__((address_space(4))) const uint32_t* kptr;
  kptr kp = (kptr)__builtin_amdgcn_kernarg_segment_ptr();
  uint32_t acc = 0;
#pragma unroll
  for (int o = 0; o < BYTES; o += 64) acc |= kp[o / 4];
  asm volatile("" ::"s"(acc));
}
// LDS-DMA: every lane's 16 bytes at gsrc land at lds_wave_base + lane * 16 (the base is wave-uniform and goes through M0).  As an
// asm statement the load is INVISIBLE to hipcc's s_waitcnt bookkeeping -- that is the point: a builtin LDS-DMA makes the compiler
// drain vmcnt(0) at every barrier and before every ds_read, which serialises the ring.  The caller counts: sf_vmcnt<N>() (in issue
// order), then sf_lds_barrier(), then the ds_read.  No compiler-visible vector load may be in flight while these are.
typedef __attribute__((address_space(3))) char sf_lds_char;
SF_DEV void sf_glds16(char* lds_wave_base, const void* gsrc) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(sf_lds_char*)lds_wave_base);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}
template <int N>
SF_DEV void sf_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }
// this wave's LDS reads have returned, then the workgroup meets (a raw s_barrier: __syncthreads() would drain the LDS-DMA queue)
SF_DEV void sf_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
SF_DEV void sf_glds_done() {}
// scheduling hint: the next n instructions of class `mask` (0x008 MFMA, 0x100 LDS read) go here, in program order of the groups
#define SF_SCHED_GROUP(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
#define SF_LGKM0() __builtin_amdgcn_s_waitcnt(0xc07f)      // lgkmcnt(0), visible to the compiler's own counting
// pins the first use of a loaded VGPR value to this point of the program: whatever is computed from x cannot be scheduled (and waited
// for) earlier.  hipcc moved the one-dword load of k_conv4_gn_mb's context-logit weight to the kernel's first instruction and waited for it
// there -- a cold round trip in front of every other load of the launch -- when the select on it stood next to the load.
#define SF_USE_FROM_HERE(x) asm volatile("" : "+v"(x))
__device__ __attribute__((aligned(128))) static const uint32_t sf_zero128[32] = {0};      // one zero line: an out-of-image pixel's 8 lanes read it like any other line
#endif

SF_DEV float sf_silu(float v) { return v / (1.0f + sf_exp(-v)); }
// SiLU with the hardware reciprocal (1 ulp) instead of an IEEE division: the operand is rounded to bf16 right after
SF_DEV float sf_silu_fast(float v) { return v * sf_rcp(1.0f + sf_exp(-v)); }
SF_DEV float sf_sigmoid(float v) { return 1.0f / (1.0f + sf_exp(-v)); }
SF_DEV float sf_gelu(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f)); }

SF_DEV float sf_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += sf_shfl_xor(v, o);
  return v;
}
SF_DEV float sf_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, sf_shfl_xor(v, o));
  return v;
}
// sum over aligned groups of `width` lanes (width = power of two <= 64)
SF_DEV float sf_group_sum(float v, int width) {
  for (int o = width >> 1; o > 0; o >>= 1) v += sf_shfl_xor(v, o);
  return v;
}
SF_DEV bf16x8 sf_zero8() { return bf16x8{0, 0, 0, 0, 0, 0, 0, 0}; }

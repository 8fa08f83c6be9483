// Multi-resolution hash / tiled grid encoding for gfx950 (MI355X).
//
// Replaces the reference's `_gridencoder` extension
// (external/gridencoder/src/gridencoder.cu:75-223 forward, :226-313 backward,
// :316-342 input backward; index rule get_grid_index :54-72, fast_hash :35-51).
// Same results, different construction:
//   * the per-level (scale, resolution, offset, hashmap_size) table is computed
//     ONCE on the host (glibc exp2f) and travels in the kernarg segment, so the
//     kernel and the CPU oracle use bit-identical level geometry (a 1-ulp
//     exp2f difference flips `resolution` at levels whose scale is integral);
//   * feature pairs are fetched/stored as 8-byte vectors (C==2 is one
//     global_load_dwordx2 per corner), outputs are written coalesced;
//   * scatter uses hardware fp32 L2 atomics (global_atomic_add_f32, no CAS loop).
// HBM-bound integer/gather work: no MFMA, no LDS needed (table is L2 resident,
// level-major launch order keeps one level's table hot per XCD L2).

#include "sf_common.h"
#include <math.h>

#define SF_MAX_LEVELS 32

struct GridLevels {
  float scale[SF_MAX_LEVELS];
  uint32_t resolution[SF_MAX_LEVELS];
  uint32_t offset[SF_MAX_LEVELS];   // in rows
  uint32_t hsize[SF_MAX_LEVELS];    // rows in this level
};

template <uint32_t D>
__device__ __forceinline__ uint32_t sf_fast_hash(const uint32_t (&pg)[D]) {
  constexpr uint32_t primes[7] = {1u, 2654435761u, 805459861u, 3674653429u,
                                  2097192037u, 1434869437u, 2165219737u};
  uint32_t r = 0;
#pragma unroll
  for (uint32_t i = 0; i < D; ++i) r ^= pg[i] * primes[i];
  return r;
}

// Row index inside one level (reference get_grid_index without the *C+ch).
template <uint32_t D>
__device__ __forceinline__ uint32_t sf_grid_row(uint32_t gridtype, bool align_corners,
                                                uint32_t hsize, uint32_t resolution,
                                                const uint32_t (&pg)[D]) {
  uint32_t stride = 1, index = 0;
  const uint32_t step = align_corners ? resolution : resolution + 1;
#pragma unroll
  for (uint32_t d = 0; d < D; ++d) {
    if (stride <= hsize) {
      index += pg[d] * stride;
      stride *= step;
    }
  }
  if (gridtype == 0 && stride > hsize) index = sf_fast_hash<D>(pg);
  return index % hsize;
}

template <uint32_t C> struct FeatVec;
template <> struct FeatVec<1> { using T = float;  };
template <> struct FeatVec<2> { using T = float2; };
template <> struct FeatVec<4> { using T = float4; };
template <> struct FeatVec<8> { using T = float4; };  // two of them

template <uint32_t C>
__device__ __forceinline__ void sf_load_feat(const float* __restrict__ p, float (&v)[C]) {
  if constexpr (C == 1) { v[0] = p[0]; }
  else if constexpr (C == 2) { float2 t = *reinterpret_cast<const float2*>(p); v[0] = t.x; v[1] = t.y; }
  else {
#pragma unroll
    for (uint32_t c = 0; c < C; c += 4) {
      float4 t = *reinterpret_cast<const float4*>(p + c);
      v[c] = t.x; v[c + 1] = t.y; v[c + 2] = t.z; v[c + 3] = t.w;
    }
  }
}
template <uint32_t C>
__device__ __forceinline__ void sf_store_feat(float* __restrict__ p, const float (&v)[C]) {
  if constexpr (C == 1) { p[0] = v[0]; }
  else if constexpr (C == 2) { *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]); }
  else {
#pragma unroll
    for (uint32_t c = 0; c < C; c += 4)
      *reinterpret_cast<float4*>(p + c) = make_float4(v[c], v[c + 1], v[c + 2], v[c + 3]);
  }
}

// One thread = one (point, level).  grid = (ceil(B/256), L).
template <uint32_t D, uint32_t C>
__global__ __launch_bounds__(256) void k_grid_forward(
    const float* __restrict__ inputs, const float* __restrict__ grid,
    float* __restrict__ outputs, float* __restrict__ dy_dx,
    uint32_t B, uint32_t L, GridLevels lv, uint32_t gridtype, uint32_t align_corners) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const uint32_t level = blockIdx.y;

  const float* tab = grid + (size_t)lv.offset[level] * C;
  const float* x = inputs + (size_t)b * D;
  float* out = outputs + ((size_t)level * B + b) * C;

  float xin[D];
  bool oob = false;
#pragma unroll
  for (uint32_t d = 0; d < D; ++d) {
    xin[d] = x[d];
    if (xin[d] < 0.0f || xin[d] > 1.0f) oob = true;
  }
  if (oob) {
    float z[C];
#pragma unroll
    for (uint32_t c = 0; c < C; ++c) z[c] = 0.0f;
    sf_store_feat<C>(out, z);
    if (dy_dx) {
      float* g = dy_dx + (size_t)b * D * L * C + (size_t)level * D * C;
#pragma unroll
      for (uint32_t i = 0; i < D * C; ++i) g[i] = 0.0f;
    }
    return;
  }

  const uint32_t hsize = lv.hsize[level];
  const float scale = lv.scale[level];
  const uint32_t resolution = lv.resolution[level];
  const bool ac = align_corners != 0;

  float pos[D];
  uint32_t pg[D];
#pragma unroll
  for (uint32_t d = 0; d < D; ++d) {
    pos[d] = fmaf(xin[d], scale, ac ? 0.0f : 0.5f);  // nvcc contracts this to an FMA
    pg[d] = (uint32_t)floorf(pos[d]);
    pos[d] -= (float)pg[d];
  }

  float res[C];
#pragma unroll
  for (uint32_t c = 0; c < C; ++c) res[c] = 0.0f;

#pragma unroll
  for (uint32_t idx = 0; idx < (1u << D); ++idx) {
    float w = 1.0f;
    uint32_t pl[D];
#pragma unroll
    for (uint32_t d = 0; d < D; ++d) {
      if ((idx & (1u << d)) == 0) { w *= 1.0f - pos[d]; pl[d] = pg[d]; }
      else                        { w *= pos[d];        pl[d] = pg[d] + 1; }
    }
    const uint32_t row = sf_grid_row<D>(gridtype, ac, hsize, resolution, pl);
    float f[C];
    sf_load_feat<C>(tab + (size_t)row * C, f);
#pragma unroll
    for (uint32_t c = 0; c < C; ++c) res[c] = fmaf(w, f[c], res[c]);
  }
  sf_store_feat<C>(out, res);

  if (dy_dx) {
    float* g = dy_dx + (size_t)b * D * L * C + (size_t)level * D * C;
#pragma unroll
    for (uint32_t gd = 0; gd < D; ++gd) {
      float rg[C];
#pragma unroll
      for (uint32_t c = 0; c < C; ++c) rg[c] = 0.0f;
#pragma unroll
      for (uint32_t idx = 0; idx < (1u << (D - 1)); ++idx) {
        float w = scale;
        uint32_t pl[D];
#pragma unroll
        for (uint32_t nd = 0; nd < D - 1; ++nd) {
          const uint32_t d = (nd >= gd) ? (nd + 1) : nd;
          if ((idx & (1u << nd)) == 0) { w *= 1.0f - pos[d]; pl[d] = pg[d]; }
          else                         { w *= pos[d];        pl[d] = pg[d] + 1; }
        }
        pl[gd] = pg[gd];
        const uint32_t rl = sf_grid_row<D>(gridtype, ac, hsize, resolution, pl);
        pl[gd] = pg[gd] + 1;
        const uint32_t rr = sf_grid_row<D>(gridtype, ac, hsize, resolution, pl);
        float fl[C], fr[C];
        sf_load_feat<C>(tab + (size_t)rl * C, fl);
        sf_load_feat<C>(tab + (size_t)rr * C, fr);
#pragma unroll
        for (uint32_t c = 0; c < C; ++c) rg[c] = fmaf(w, fr[c] - fl[c], rg[c]);
      }
#pragma unroll
      for (uint32_t c = 0; c < C; ++c) g[gd * C + c] = rg[c];
    }
  }
}

// One thread = one (point, level, channel pair).  grid = (ceil(B*C/NC/256), L).
template <uint32_t D, uint32_t C, uint32_t NC>
__global__ __launch_bounds__(256) void k_grid_backward(
    const float* __restrict__ grad, const float* __restrict__ inputs,
    float* __restrict__ grad_grid, uint32_t B, uint32_t L, GridLevels lv,
    uint32_t gridtype, uint32_t align_corners) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t b = t * NC / C;
  if (b >= B) return;
  const uint32_t level = blockIdx.y;
  const uint32_t ch = t * NC - b * C;

  float* gtab = grad_grid + (size_t)lv.offset[level] * C;
  const float* x = inputs + (size_t)b * D;
  const float* gin = grad + ((size_t)level * B + b) * C + ch;

  float xin[D];
#pragma unroll
  for (uint32_t d = 0; d < D; ++d) {
    xin[d] = x[d];
    if (xin[d] < 0.0f || xin[d] > 1.0f) return;  // grad_grid is zero-initialised
  }
  const uint32_t hsize = lv.hsize[level];
  const float scale = lv.scale[level];
  const uint32_t resolution = lv.resolution[level];
  const bool ac = align_corners != 0;

  float pos[D];
  uint32_t pg[D];
#pragma unroll
  for (uint32_t d = 0; d < D; ++d) {
    pos[d] = fmaf(xin[d], scale, ac ? 0.0f : 0.5f);  // nvcc contracts this to an FMA
    pg[d] = (uint32_t)floorf(pos[d]);
    pos[d] -= (float)pg[d];
  }
  float gc[NC];
#pragma unroll
  for (uint32_t c = 0; c < NC; ++c) gc[c] = gin[c];

#pragma unroll
  for (uint32_t idx = 0; idx < (1u << D); ++idx) {
    float w = 1.0f;
    uint32_t pl[D];
#pragma unroll
    for (uint32_t d = 0; d < D; ++d) {
      if ((idx & (1u << d)) == 0) { w *= 1.0f - pos[d]; pl[d] = pg[d]; }
      else                        { w *= pos[d];        pl[d] = pg[d] + 1; }
    }
    const uint32_t row = sf_grid_row<D>(gridtype, ac, hsize, resolution, pl);
    float* dst = gtab + (size_t)row * C + ch;
#pragma unroll
    for (uint32_t c = 0; c < NC; ++c)
      (void)__builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float*)(dst + c), w * gc[c]);
  }
}

template <uint32_t D, uint32_t C>
__global__ __launch_bounds__(256) void k_grid_input_backward(
    const float* __restrict__ grad, const float* __restrict__ dy_dx,
    float* __restrict__ grad_inputs, uint32_t B, uint32_t L) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= B * D) return;
  const uint32_t b = t / D, d = t - b * D;
  const float* g = dy_dx + (size_t)b * L * D * C;
  float r = 0.0f;
  for (uint32_t l = 0; l < L; ++l) {
#pragma unroll
    for (uint32_t c = 0; c < C; ++c)
      r = fmaf(grad[((size_t)l * B + b) * C + c], g[l * D * C + d * C + c], r);
  }
  grad_inputs[t] = r;
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------

// Level geometry exactly as the reference evaluates it per thread
// (gridencoder.cu:124-126), but once, on the host.
int sf_fill_levels(GridLevels* lv, const int32_t* offsets_dev, const int32_t* h_offsets,
                   uint32_t L, float S, uint32_t H, hipStream_t st) {
  if (L > SF_MAX_LEVELS) SF_FAIL(SF_ERR_INVALID, "GridEncoding: L must be <= %d", SF_MAX_LEVELS);
  int32_t tmp[SF_MAX_LEVELS + 1];
  if (!h_offsets) {
    if (hipMemcpyAsync(tmp, offsets_dev, sizeof(int32_t) * (L + 1), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
      SF_FAIL(SF_ERR_LAUNCH, "GridEncoding: cannot read offsets");
    h_offsets = tmp;
  }
  for (uint32_t l = 0; l < L; ++l) {
    const float scale = exp2f((float)l * S) * (float)H - 1.0f;
    lv->scale[l] = scale;
    lv->resolution[l] = (uint32_t)ceil(scale) + 1;
    lv->offset[l] = (uint32_t)h_offsets[l];
    lv->hsize[l] = (uint32_t)(h_offsets[l + 1] - h_offsets[l]);
  }
  return SF_OK;
}

template <uint32_t D>
static int launch_forward(const float* inputs, const float* emb, float* outputs, float* dy_dx,
                          uint32_t B, uint32_t C, uint32_t L, const GridLevels& lv,
                          uint32_t gridtype, uint32_t ac, hipStream_t st) {
  const dim3 grid(sf_div_up(B, 256), L), block(256);
  switch (C) {
    case 1: k_grid_forward<D, 1><<<grid, block, 0, st>>>(inputs, emb, outputs, dy_dx, B, L, lv, gridtype, ac); break;
    case 2: k_grid_forward<D, 2><<<grid, block, 0, st>>>(inputs, emb, outputs, dy_dx, B, L, lv, gridtype, ac); break;
    case 4: k_grid_forward<D, 4><<<grid, block, 0, st>>>(inputs, emb, outputs, dy_dx, B, L, lv, gridtype, ac); break;
    case 8: k_grid_forward<D, 8><<<grid, block, 0, st>>>(inputs, emb, outputs, dy_dx, B, L, lv, gridtype, ac); break;
    default: SF_FAIL(SF_ERR_INVALID, "GridEncoding: C must be 1, 2, 4, or 8.");
  }
  SF_CHECK_LAUNCH("grid_encode_forward");
  return SF_OK;
}

template <uint32_t D>
static int launch_backward(const float* grad, const float* inputs, float* gemb,
                           const float* dy_dx, float* grad_inputs, uint32_t B, uint32_t C,
                           uint32_t L, const GridLevels& lv, uint32_t gridtype, uint32_t ac,
                           hipStream_t st) {
  const dim3 block(256);
  const uint32_t NC = C < 2 ? C : 2;
  const dim3 grid(sf_div_up((uint64_t)B * C / NC, 256), L);
  const dim3 gridi(sf_div_up((uint64_t)B * D, 256));
  switch (C) {
    case 1:
      k_grid_backward<D, 1, 1><<<grid, block, 0, st>>>(grad, inputs, gemb, B, L, lv, gridtype, ac);
      if (dy_dx) k_grid_input_backward<D, 1><<<gridi, block, 0, st>>>(grad, dy_dx, grad_inputs, B, L);
      break;
    case 2:
      k_grid_backward<D, 2, 2><<<grid, block, 0, st>>>(grad, inputs, gemb, B, L, lv, gridtype, ac);
      if (dy_dx) k_grid_input_backward<D, 2><<<gridi, block, 0, st>>>(grad, dy_dx, grad_inputs, B, L);
      break;
    case 4:
      k_grid_backward<D, 4, 2><<<grid, block, 0, st>>>(grad, inputs, gemb, B, L, lv, gridtype, ac);
      if (dy_dx) k_grid_input_backward<D, 4><<<gridi, block, 0, st>>>(grad, dy_dx, grad_inputs, B, L);
      break;
    case 8:
      k_grid_backward<D, 8, 2><<<grid, block, 0, st>>>(grad, inputs, gemb, B, L, lv, gridtype, ac);
      if (dy_dx) k_grid_input_backward<D, 8><<<gridi, block, 0, st>>>(grad, dy_dx, grad_inputs, B, L);
      break;
    default: SF_FAIL(SF_ERR_INVALID, "GridEncoding: C must be 1, 2, 4, or 8.");
  }
  SF_CHECK_LAUNCH("grid_encode_backward");
  return SF_OK;
}

extern "C" int sf_grid_encode_forward(const float* inputs, const float* embeddings,
                                      const int32_t* offsets, float* outputs, uint32_t B,
                                      uint32_t D, uint32_t C, uint32_t L, float S, uint32_t H,
                                      float* dy_dx, uint32_t gridtype, int align_corners,
                                      const int32_t* h_offsets, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (B == 0) return SF_OK;                      // empty batch: nothing to do (empty tensors have null data)
  if (!inputs || !embeddings || !outputs) SF_FAIL(SF_ERR_INVALID, "grid_encode_forward: null tensor");
  GridLevels lv;
  if (int rc = sf_fill_levels(&lv, offsets, h_offsets, L, S, H, st)) return rc;
  const uint32_t ac = align_corners ? 1u : 0u;
  switch (D) {
    case 1: return launch_forward<1>(inputs, embeddings, outputs, dy_dx, B, C, L, lv, gridtype, ac, st);
    case 2: return launch_forward<2>(inputs, embeddings, outputs, dy_dx, B, C, L, lv, gridtype, ac, st);
    case 3: return launch_forward<3>(inputs, embeddings, outputs, dy_dx, B, C, L, lv, gridtype, ac, st);
    case 4: return launch_forward<4>(inputs, embeddings, outputs, dy_dx, B, C, L, lv, gridtype, ac, st);
    case 5: return launch_forward<5>(inputs, embeddings, outputs, dy_dx, B, C, L, lv, gridtype, ac, st);
    default: SF_FAIL(SF_ERR_INVALID, "GridEncoding: D must be 1, 2, 3, 4, or 5.");
  }
}

extern "C" int sf_grid_encode_backward(const float* grad, const float* inputs,
                                       const float* embeddings, const int32_t* offsets,
                                       float* grad_embeddings, uint32_t B, uint32_t D, uint32_t C,
                                       uint32_t L, float S, uint32_t H, const float* dy_dx,
                                       float* grad_inputs, uint32_t gridtype, int align_corners,
                                       const int32_t* h_offsets, void* stream) {
  (void)embeddings;
  hipStream_t st = (hipStream_t)stream;
  if (B == 0) return SF_OK;
  if (!grad || !inputs || !grad_embeddings) SF_FAIL(SF_ERR_INVALID, "grid_encode_backward: null tensor");
  if ((dy_dx == nullptr) != (grad_inputs == nullptr))
    SF_FAIL(SF_ERR_INVALID, "grid_encode_backward: dy_dx and grad_inputs must be given together");
  GridLevels lv;
  if (int rc = sf_fill_levels(&lv, offsets, h_offsets, L, S, H, st)) return rc;
  const uint32_t ac = align_corners ? 1u : 0u;
  switch (D) {
    case 1: return launch_backward<1>(grad, inputs, grad_embeddings, dy_dx, grad_inputs, B, C, L, lv, gridtype, ac, st);
    case 2: return launch_backward<2>(grad, inputs, grad_embeddings, dy_dx, grad_inputs, B, C, L, lv, gridtype, ac, st);
    case 3: return launch_backward<3>(grad, inputs, grad_embeddings, dy_dx, grad_inputs, B, C, L, lv, gridtype, ac, st);
    case 4: return launch_backward<4>(grad, inputs, grad_embeddings, dy_dx, grad_inputs, B, C, L, lv, gridtype, ac, st);
    case 5: return launch_backward<5>(grad, inputs, grad_embeddings, dy_dx, grad_inputs, B, C, L, lv, gridtype, ac, st);
    default: SF_FAIL(SF_ERR_INVALID, "GridEncoding: D must be 1, 2, 3, 4, or 5.");
  }
}

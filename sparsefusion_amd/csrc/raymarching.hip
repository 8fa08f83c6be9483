// Ray utilities for gfx950: AABB slab test, Morton codes, density-bitfield packing.
//
// Replaces the `_raymarching` entry points that the default (cuda_ray=False)
// distillation path and its density-grid maintenance use:
//   near_far_from_aabb   raymarching/src/raymarching.cu:91-156
//   morton3D / invert    raymarching/src/raymarching.cu:56-82, :214-254
//   packbits             raymarching/src/raymarching.cu:267-289
// fp32 compare order and the 1/d reciprocal are kept exactly (bit-exact near/far).
// All kernels are one element per lane, grid-stride, coalesced; HBM-bound.

#include "sf_common.h"
#include <float.h>

__global__ __launch_bounds__(256) void k_near_far_from_aabb(
    const float* __restrict__ rays_o, const float* __restrict__ rays_d,
    const float* __restrict__ aabb, uint32_t N, float min_near,
    float* __restrict__ nears, float* __restrict__ fars) {
  const float a0 = aabb[0], a1 = aabb[1], a2 = aabb[2], a3 = aabb[3], a4 = aabb[4], a5 = aabb[5];
  for (uint32_t n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
    const float ox = rays_o[n * 3 + 0], oy = rays_o[n * 3 + 1], oz = rays_o[n * 3 + 2];
    const float dx = rays_d[n * 3 + 0], dy = rays_d[n * 3 + 1], dz = rays_d[n * 3 + 2];
    // IEEE division (not v_rcp_f32): the reference computes 1/d with a correctly rounded divide.
    const float rdx = __fdiv_rn(1.0f, dx), rdy = __fdiv_rn(1.0f, dy), rdz = __fdiv_rn(1.0f, dz);

    float near = (a0 - ox) * rdx, far = (a3 - ox) * rdx;
    if (near > far) { float t = near; near = far; far = t; }
    float ny = (a1 - oy) * rdy, fy = (a4 - oy) * rdy;
    if (ny > fy) { float t = ny; ny = fy; fy = t; }
    if (near > fy || ny > far) { nears[n] = FLT_MAX; fars[n] = FLT_MAX; continue; }
    if (ny > near) near = ny;
    if (fy < far) far = fy;
    float nz = (a2 - oz) * rdz, fz = (a5 - oz) * rdz;
    if (nz > fz) { float t = nz; nz = fz; fz = t; }
    if (near > fz || nz > far) { nears[n] = FLT_MAX; fars[n] = FLT_MAX; continue; }
    if (nz > near) near = nz;
    if (fz < far) far = fz;
    if (near < min_near) near = min_near;
    nears[n] = near;
    fars[n] = far;
  }
}

__device__ __forceinline__ uint32_t sf_expand_bits(uint32_t v) {
  v = (v * 0x00010001u) & 0xFF0000FFu;
  v = (v * 0x00000101u) & 0x0F00F00Fu;
  v = (v * 0x00000011u) & 0xC30C30C3u;
  v = (v * 0x00000005u) & 0x49249249u;
  return v;
}
__device__ __forceinline__ uint32_t sf_compact_bits(uint32_t x) {
  x = x & 0x49249249u;
  x = (x | (x >> 2)) & 0xc30c30c3u;
  x = (x | (x >> 4)) & 0x0f00f00fu;
  x = (x | (x >> 8)) & 0xff0000ffu;
  x = (x | (x >> 16)) & 0x0000ffffu;
  return x;
}

__global__ __launch_bounds__(256) void k_morton3D(const int32_t* __restrict__ coords, uint32_t N,
                                                  int32_t* __restrict__ indices) {
  for (uint32_t n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
    const uint32_t x = (uint32_t)coords[n * 3 + 0], y = (uint32_t)coords[n * 3 + 1], z = (uint32_t)coords[n * 3 + 2];
    indices[n] = (int32_t)(sf_expand_bits(x) | (sf_expand_bits(y) << 1) | (sf_expand_bits(z) << 2));
  }
}

__global__ __launch_bounds__(256) void k_morton3D_invert(const int32_t* __restrict__ indices, uint32_t N,
                                                         int32_t* __restrict__ coords) {
  for (uint32_t n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
    const int32_t i = indices[n];  // signed shifts, as the reference (`ind >> 1` on int)
    coords[n * 3 + 0] = (int32_t)sf_compact_bits((uint32_t)(i >> 0));
    coords[n * 3 + 1] = (int32_t)sf_compact_bits((uint32_t)(i >> 1));
    coords[n * 3 + 2] = (int32_t)sf_compact_bits((uint32_t)(i >> 2));
  }
}

// 8 cells -> 1 byte: each lane reads two float4 (32 B, coalesced) and writes one byte.
__global__ __launch_bounds__(256) void k_packbits(const float* __restrict__ grid, uint32_t N,
                                                  float thresh, uint8_t* __restrict__ bitfield) {
  for (uint32_t n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
    const float4 lo = *reinterpret_cast<const float4*>(grid + (size_t)n * 8);
    const float4 hi = *reinterpret_cast<const float4*>(grid + (size_t)n * 8 + 4);
    uint32_t bits = 0;
    bits |= (lo.x > thresh) ? 1u : 0u;
    bits |= (lo.y > thresh) ? 2u : 0u;
    bits |= (lo.z > thresh) ? 4u : 0u;
    bits |= (lo.w > thresh) ? 8u : 0u;
    bits |= (hi.x > thresh) ? 16u : 0u;
    bits |= (hi.y > thresh) ? 32u : 0u;
    bits |= (hi.z > thresh) ? 64u : 0u;
    bits |= (hi.w > thresh) ? 128u : 0u;
    bitfield[n] = (uint8_t)bits;
  }
}

extern "C" int sf_near_far_from_aabb(const float* rays_o, const float* rays_d, const float* aabb,
                                     uint32_t N, float min_near, float* nears, float* fars,
                                     void* stream) {
  if (N == 0) return SF_OK;
  if (!rays_o || !rays_d || !aabb || !nears || !fars) SF_FAIL(SF_ERR_INVALID, "near_far_from_aabb: null tensor");
  k_near_far_from_aabb<<<sf_grid_cap(sf_div_up(N, 256)), 256, 0, (hipStream_t)stream>>>(
      rays_o, rays_d, aabb, N, min_near, nears, fars);
  SF_CHECK_LAUNCH("near_far_from_aabb");
  return SF_OK;
}

extern "C" int sf_morton3D(const int32_t* coords, uint32_t N, int32_t* indices, void* stream) {
  if (N == 0) return SF_OK;
  if (!coords || !indices) SF_FAIL(SF_ERR_INVALID, "morton3D: null tensor");
  k_morton3D<<<sf_grid_cap(sf_div_up(N, 256)), 256, 0, (hipStream_t)stream>>>(coords, N, indices);
  SF_CHECK_LAUNCH("morton3D");
  return SF_OK;
}

extern "C" int sf_morton3D_invert(const int32_t* indices, uint32_t N, int32_t* coords, void* stream) {
  if (N == 0) return SF_OK;
  if (!coords || !indices) SF_FAIL(SF_ERR_INVALID, "morton3D_invert: null tensor");
  k_morton3D_invert<<<sf_grid_cap(sf_div_up(N, 256)), 256, 0, (hipStream_t)stream>>>(indices, N, coords);
  SF_CHECK_LAUNCH("morton3D_invert");
  return SF_OK;
}

extern "C" int sf_packbits(const float* grid, uint32_t N, float density_thresh, uint8_t* bitfield,
                           void* stream) {
  if (N == 0) return SF_OK;
  if (!grid || !bitfield) SF_FAIL(SF_ERR_INVALID, "packbits: null tensor");
  k_packbits<<<sf_grid_cap(sf_div_up(N, 256)), 256, 0, (hipStream_t)stream>>>(grid, N, density_thresh, bitfield);
  SF_CHECK_LAUNCH("packbits");
  return SF_OK;
}

/*
 * oracle/ngp_ref.c -- CPU restatement (plain C) of the reference's CUDA-only entry points
 * on the NGP side of the hot path.  TEST INFRASTRUCTURE ONLY: imported by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by sparsefusion_amd/.
 *
 * Follows, statement by statement:
 *   grid index          external/gridencoder/src/gridencoder.cu:54-72  (get_grid_index)
 *   hash                external/gridencoder/src/gridencoder.cu:35-51  (fast_hash)
 *   forward             external/gridencoder/src/gridencoder.cu:75-223 (kernel_grid)
 *   backward            external/gridencoder/src/gridencoder.cu:226-313 (kernel_grid_backward)
 *   input backward      external/gridencoder/src/gridencoder.cu:316-342
 *   near/far            raymarching/src/raymarching.cu:91-145
 *   morton / packbits   raymarching/src/raymarching.cu:56-82, :214-289
 *
 * PARITY UNPINNED by the reference: it ships no test, golden vector or CPU path for these
 * kernels; tests/test_oracle_first_principles.py cross-checks
 * this file against an independent dense trilinear interpolation instead.
 *
 * a*b+c patterns that nvcc contracts into one FMA under its default -fmad=true are written
 * as fmaf() (compile with -ffp-contract=off so nothing else is contracted).  Level geometry
 * uses glibc exp2f, the same call the HIP library makes on the host.
 */
#include <math.h>
#include <float.h>
#include <stdint.h>
#include <string.h>

#define MAXD 5
#define MAXC 8

static uint32_t fast_hash(uint32_t D, const uint32_t* pg) {
  static const uint32_t primes[7] = {1u, 2654435761u, 805459861u, 3674653429u,
                                     2097192037u, 1434869437u, 2165219737u};
  uint32_t r = 0;
  for (uint32_t i = 0; i < D; ++i) r ^= pg[i] * primes[i];
  return r;
}

/* returns element index (row*C + ch) inside the level */
static uint32_t get_grid_index(uint32_t D, uint32_t C, uint32_t gridtype, int align_corners, uint32_t ch,
                               uint32_t hashmap_size, uint32_t resolution, const uint32_t* pg) {
  uint32_t stride = 1, index = 0;
  for (uint32_t d = 0; d < D && stride <= hashmap_size; d++) {
    index += pg[d] * stride;
    stride *= align_corners ? resolution : (resolution + 1);
  }
  if (gridtype == 0 && stride > hashmap_size) index = fast_hash(D, pg);
  return (index % hashmap_size) * C + ch;
}

/* outputs [L,B,C]; dy_dx [B, L*D*C] or NULL; level_index (optional) [L,B,2^D] rows, for bit-exact
 * index tests (oracle-only diagnostic). */
void oracle_grid_encode_forward(const float* inputs, const float* embeddings, const int32_t* offsets,
                                float* outputs, uint32_t B, uint32_t D, uint32_t C, uint32_t L, float S,
                                uint32_t H, float* dy_dx, uint32_t gridtype, int align_corners,
                                uint32_t* level_index) {
#pragma omp parallel for schedule(static)
  for (int64_t lb = 0; lb < (int64_t)L * B; ++lb) {
    const uint32_t level = (uint32_t)(lb / B), b = (uint32_t)(lb % B);
    const float* grid = embeddings + (size_t)(uint32_t)offsets[level] * C;
    const float* in = inputs + (size_t)b * D;
    float* out = outputs + ((size_t)level * B + b) * C;
    float* dd = dy_dx ? dy_dx + (size_t)b * D * L * C + (size_t)level * D * C : 0;

    int oob = 0;
    for (uint32_t d = 0; d < D; d++)
      if (in[d] < 0 || in[d] > 1) oob = 1;
    if (oob) {
      for (uint32_t ch = 0; ch < C; ch++) out[ch] = 0;
      if (dd) for (uint32_t i = 0; i < D * C; i++) dd[i] = 0;
      if (level_index) for (uint32_t i = 0; i < (1u << D); i++) level_index[((size_t)level * B + b) * (1u << D) + i] = 0xffffffffu;
      continue;
    }
    const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
    const float scale = exp2f(level * S) * H - 1.0f;
    const uint32_t resolution = (uint32_t)ceil(scale) + 1;

    float pos[MAXD];
    uint32_t pos_grid[MAXD];
    for (uint32_t d = 0; d < D; d++) {
      pos[d] = fmaf(in[d], scale, align_corners ? 0.0f : 0.5f);
      pos_grid[d] = (uint32_t)floorf(pos[d]);
      pos[d] -= (float)pos_grid[d];
    }
    float results[MAXC] = {0};
    for (uint32_t idx = 0; idx < (1u << D); idx++) {
      float w = 1;
      uint32_t pgl[MAXD];
      for (uint32_t d = 0; d < D; d++) {
        if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; pgl[d] = pos_grid[d]; }
        else { w *= pos[d]; pgl[d] = pos_grid[d] + 1; }
      }
      const uint32_t index = get_grid_index(D, C, gridtype, align_corners, 0, hashmap_size, resolution, pgl);
      if (level_index) level_index[((size_t)level * B + b) * (1u << D) + idx] = index / C;
      for (uint32_t ch = 0; ch < C; ch++) results[ch] = fmaf(w, grid[index + ch], results[ch]);
    }
    for (uint32_t ch = 0; ch < C; ch++) out[ch] = results[ch];

    if (dd) {
      for (uint32_t gd = 0; gd < D; gd++) {
        float rg[MAXC] = {0};
        for (uint32_t idx = 0; idx < (1u << (D - 1)); idx++) {
          float w = scale;
          uint32_t pgl[MAXD];
          for (uint32_t nd = 0; nd < D - 1; nd++) {
            const uint32_t d = (nd >= gd) ? (nd + 1) : nd;
            if ((idx & (1u << nd)) == 0) { w *= 1 - pos[d]; pgl[d] = pos_grid[d]; }
            else { w *= pos[d]; pgl[d] = pos_grid[d] + 1; }
          }
          pgl[gd] = pos_grid[gd];
          const uint32_t il = get_grid_index(D, C, gridtype, align_corners, 0, hashmap_size, resolution, pgl);
          pgl[gd] = pos_grid[gd] + 1;
          const uint32_t ir = get_grid_index(D, C, gridtype, align_corners, 0, hashmap_size, resolution, pgl);
          for (uint32_t ch = 0; ch < C; ch++) rg[ch] = fmaf(w, grid[ir + ch] - grid[il + ch], rg[ch]);
        }
        for (uint32_t ch = 0; ch < C; ch++) dd[gd * C + ch] = rg[ch];
      }
    }
  }
}

/* grad [L,B,C]; grad_embeddings zero-initialised by the caller.  Levels own disjoint table rows,
 * so threads split over levels and each level accumulates in point order (deterministic, unlike
 * the reference's atomicAdd whose order is unspecified). */
void oracle_grid_encode_backward(const float* grad, const float* inputs, const int32_t* offsets,
                                 float* grad_embeddings, uint32_t B, uint32_t D, uint32_t C, uint32_t L,
                                 float S, uint32_t H, const float* dy_dx, float* grad_inputs,
                                 uint32_t gridtype, int align_corners) {
#pragma omp parallel for schedule(dynamic, 1)
  for (int32_t level = 0; level < (int32_t)L; ++level) {
    float* gg = grad_embeddings + (size_t)(uint32_t)offsets[level] * C;
    const uint32_t hashmap_size = (uint32_t)(offsets[level + 1] - offsets[level]);
    const float scale = exp2f(level * S) * H - 1.0f;
    const uint32_t resolution = (uint32_t)ceil(scale) + 1;
    for (uint32_t b = 0; b < B; ++b) {
      const float* in = inputs + (size_t)b * D;
      const float* g = grad + ((size_t)level * B + b) * C;
      int oob = 0;
      for (uint32_t d = 0; d < D; d++)
        if (in[d] < 0 || in[d] > 1) oob = 1;
      if (oob) continue;
      float pos[MAXD];
      uint32_t pos_grid[MAXD];
      for (uint32_t d = 0; d < D; d++) {
        pos[d] = fmaf(in[d], scale, align_corners ? 0.0f : 0.5f);
        pos_grid[d] = (uint32_t)floorf(pos[d]);
        pos[d] -= (float)pos_grid[d];
      }
      for (uint32_t idx = 0; idx < (1u << D); idx++) {
        float w = 1;
        uint32_t pgl[MAXD];
        for (uint32_t d = 0; d < D; d++) {
          if ((idx & (1u << d)) == 0) { w *= 1 - pos[d]; pgl[d] = pos_grid[d]; }
          else { w *= pos[d]; pgl[d] = pos_grid[d] + 1; }
        }
        const uint32_t index = get_grid_index(D, C, gridtype, align_corners, 0, hashmap_size, resolution, pgl);
        for (uint32_t c = 0; c < C; c++) gg[index + c] += w * g[c];
      }
    }
  }
  if (dy_dx && grad_inputs) {
#pragma omp parallel for schedule(static)
    for (int64_t t = 0; t < (int64_t)B * D; ++t) {
      const uint32_t b = (uint32_t)(t / D), d = (uint32_t)(t % D);
      const float* dd = dy_dx + (size_t)b * L * D * C;
      float result = 0;
      for (uint32_t l = 0; l < L; l++)
        for (uint32_t ch = 0; ch < C; ch++)
          result = fmaf(grad[((size_t)l * B + b) * C + ch], dd[l * D * C + d * C + ch], result);
      grad_inputs[t] = result;
    }
  }
}

void oracle_near_far_from_aabb(const float* rays_o, const float* rays_d, const float* aabb, uint32_t N,
                               float min_near, float* nears, float* fars) {
#pragma omp parallel for schedule(static)
  for (int64_t n = 0; n < (int64_t)N; ++n) {
    const float* o = rays_o + n * 3;
    const float* dv = rays_d + n * 3;
    const float ox = o[0], oy = o[1], oz = o[2];
    const float dx = dv[0], dy = dv[1], dz = dv[2];
    const float rdx = 1 / dx, rdy = 1 / dy, rdz = 1 / dz;
    float near = (aabb[0] - ox) * rdx, far = (aabb[3] - ox) * rdx, t;
    if (near > far) { t = near; near = far; far = t; }
    float near_y = (aabb[1] - oy) * rdy, far_y = (aabb[4] - oy) * rdy;
    if (near_y > far_y) { t = near_y; near_y = far_y; far_y = t; }
    if (near > far_y || near_y > far) { nears[n] = fars[n] = FLT_MAX; continue; }
    if (near_y > near) near = near_y;
    if (far_y < far) far = far_y;
    float near_z = (aabb[2] - oz) * rdz, far_z = (aabb[5] - oz) * rdz;
    if (near_z > far_z) { t = near_z; near_z = far_z; far_z = t; }
    if (near > far_z || near_z > far) { nears[n] = fars[n] = FLT_MAX; continue; }
    if (near_z > near) near = near_z;
    if (far_z < far) far = far_z;
    if (near < min_near) near = min_near;
    nears[n] = near;
    fars[n] = far;
  }
}

static uint32_t expand_bits(uint32_t v) {
  v = (v * 0x00010001u) & 0xFF0000FFu;
  v = (v * 0x00000101u) & 0x0F00F00Fu;
  v = (v * 0x00000011u) & 0xC30C30C3u;
  v = (v * 0x00000005u) & 0x49249249u;
  return v;
}
static uint32_t morton_invert1(uint32_t x) {
  x = x & 0x49249249;
  x = (x | (x >> 2)) & 0xc30c30c3;
  x = (x | (x >> 4)) & 0x0f00f00f;
  x = (x | (x >> 8)) & 0xff0000ff;
  x = (x | (x >> 16)) & 0x0000ffff;
  return x;
}
void oracle_morton3D(const int32_t* coords, uint32_t N, int32_t* indices) {
  for (uint32_t n = 0; n < N; ++n)
    indices[n] = (int32_t)(expand_bits((uint32_t)coords[n * 3]) | (expand_bits((uint32_t)coords[n * 3 + 1]) << 1) |
                           (expand_bits((uint32_t)coords[n * 3 + 2]) << 2));
}
void oracle_morton3D_invert(const int32_t* indices, uint32_t N, int32_t* coords) {
  for (uint32_t n = 0; n < N; ++n) {
    const int ind = indices[n];
    coords[n * 3 + 0] = (int32_t)morton_invert1((uint32_t)(ind >> 0));
    coords[n * 3 + 1] = (int32_t)morton_invert1((uint32_t)(ind >> 1));
    coords[n * 3 + 2] = (int32_t)morton_invert1((uint32_t)(ind >> 2));
  }
}
void oracle_packbits(const float* grid, uint32_t N, float density_thresh, uint8_t* bitfield) {
  for (uint32_t n = 0; n < N; ++n) {
    uint8_t bits = 0;
    for (uint8_t i = 0; i < 8; i++) bits |= (grid[(size_t)n * 8 + i] > density_thresh) ? ((uint8_t)1 << i) : 0;
    bitfield[n] = bits;
  }
}

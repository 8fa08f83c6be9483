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
 * PINNED (round 3): the reference ships no test, golden vector or CPU path for these kernels
 * (SURVEY.md section 4 / 8(c)), so its own .cu sources are compiled for the host by
 * oracle/build_ref.py (oracle/_ref/libref_native*.so) and this file is compared with them bit
 * for bit: tests/test_oracle_native_pin.py (live, dev container) and the committed vectors
 * tests/golden/ngp_native.pt (made by tests/golden/make_golden_native.py; checked everywhere).
 * First-principles cross-checks (dense trilinear interpolation, adjoint identity, slab test):
 * tests/test_oracle_ngp.py, tests/test_oracle_occ.py.
 *
 * a*b+c patterns that nvcc contracts into one FMA under its default -fmad=true are written
 * as fmaf() (compile with -ffp-contract=off so nothing else is contracted).  Level geometry
 * uses glibc exp2f, the same call the HIP library makes on the host.
 */
#include <math.h>
#include <float.h>
#include <stdint.h>
#include <string.h>

#ifdef ORACLE_NO_FMA          /* second build (liboracle_ngp_nofma.so): every fmaf() unfused, to be compared bit for bit */
#define fmaf(a, b, c) ((a) * (b) + (c)) /* with the reference's own sources compiled -ffp-contract=off (oracle/build_ref.py) */
#endif

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

/* =====================================================================================
 * Occupancy-grid ray marching / compositing (`cuda_ray=True` entry points).
 *   helpers                         raymarching/src/raymarching.cu:20-54 (constants, signf, clamp, mip_from_pos/dt)
 *   sph_from_ray                    raymarching/src/raymarching.cu:159-204
 *   march_rays_train                raymarching/src/raymarching.cu:302-492
 *   composite_rays_train_forward    raymarching/src/raymarching.cu:495-583
 *   composite_rays_train_backward   raymarching/src/raymarching.cu:586-688
 *   march_rays / composite_rays     raymarching/src/raymarching.cu:695-913
 * The kernels are one thread per ray; here the rays run serially in index order, which is ONE of the schedules
 * the reference's two atomicAdd (point slot, ray slot) allow.  PARITY UNPINNED by the reference (no test / golden /
 * CPU path); tests/test_oracle_ngp.py checks the geometry from first principles (samples lie in occupied cells,
 * deltas telescoping, counters).  __expf is restated as expf (the fast intrinsic has no portable definition).
 * ===================================================================================== */
#define O_SQRT3 1.7320508075688772f
#define O_RPI 0.3183098861837907f

static float o_signf(float x) { return copysignf(1.0f, x); }
static float o_clamp(float x, float lo, float hi) { return fminf(hi, fmaxf(lo, x)); }

static int o_mip_from_pos(float x, float y, float z, float max_cascade) {
  const float mx = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
  int exponent;
  frexpf(mx, &exponent);
  return (int)fminf(max_cascade - 1, fmaxf(0, (float)exponent));
}

static int o_mip_from_dt(float dt, float H, float max_cascade) {
  const float mx = (float)(dt * H * 0.5);          /* float product, then a double multiply by the literal 0.5 */
  int exponent;
  frexpf(mx, &exponent);
  return (int)fminf(max_cascade - 1, fmaxf(0, (float)exponent));
}

typedef struct {
  float ox, oy, oz, dx, dy, dz, rdx, rdy, rdz, rH, H3, bound, dt_gamma, dt_min, dt_max;
  uint32_t C, H;
  const uint8_t* grid;
} o_ray;

static void o_ray_init(o_ray* r, const float* o, const float* d, const uint8_t* grid, float bound, float dt_gamma,
                       uint32_t max_steps, uint32_t C, uint32_t H) {
  r->ox = o[0]; r->oy = o[1]; r->oz = o[2]; r->dx = d[0]; r->dy = d[1]; r->dz = d[2];
  r->rdx = 1 / r->dx; r->rdy = 1 / r->dy; r->rdz = 1 / r->dz;
  r->rH = 1 / (float)H;
  r->H3 = (float)(H * H * H);
  r->bound = bound; r->dt_gamma = dt_gamma; r->C = C; r->H = H; r->grid = grid;
  r->dt_min = 2 * O_SQRT3 / max_steps;
  r->dt_max = 2 * O_SQRT3 * (1 << (C - 1)) / H;
}

/* one probe of the marching loop: position, dt, cell, occupancy (raymarching.cu:357-377 == :420-441 == :752-771) */
static int o_probe(const o_ray* r, float t, float* x, float* y, float* z, float* dt, int* nx, int* ny, int* nz,
                   float* mip_bound) {
  *x = o_clamp(fmaf(t, r->dx, r->ox), -r->bound, r->bound);
  *y = o_clamp(fmaf(t, r->dy, r->oy), -r->bound, r->bound);
  *z = o_clamp(fmaf(t, r->dz, r->oz), -r->bound, r->bound);
  *dt = o_clamp(t * r->dt_gamma, r->dt_min, r->dt_max);
  const int a = o_mip_from_pos(*x, *y, *z, (float)r->C), b = o_mip_from_dt(*dt, (float)r->H, (float)r->C);
  const int level = a > b ? a : b;
  *mip_bound = fminf(scalbnf(1.0f, level), r->bound);
  const float mip_rbound = 1 / *mip_bound;
  *nx = (int)o_clamp((float)(0.5 * fmaf(*x, mip_rbound, 1.0f) * r->H), 0.0f, (float)(r->H - 1));
  *ny = (int)o_clamp((float)(0.5 * fmaf(*y, mip_rbound, 1.0f) * r->H), 0.0f, (float)(r->H - 1));
  *nz = (int)o_clamp((float)(0.5 * fmaf(*z, mip_rbound, 1.0f) * r->H), 0.0f, (float)(r->H - 1));
  const uint32_t morton = expand_bits((uint32_t)*nx) | (expand_bits((uint32_t)*ny) << 1) | (expand_bits((uint32_t)*nz) << 2);
  const uint32_t index = (uint32_t)fmaf((float)level, r->H3, (float)morton);      /* `level * H3 + morton` in float */
  return (r->grid[index / 8] & (1 << (index % 8))) != 0;
}

/* skip to the next voxel (raymarching.cu:386-396) */
static float o_skip(const o_ray* r, float t, float x, float y, float z, int nx, int ny, int nz, float mip_bound) {
  const float tx = fmaf((nx + 0.5f + 0.5f * o_signf(r->dx)) * r->rH * 2 - 1, mip_bound, -x) * r->rdx;
  const float ty = fmaf((ny + 0.5f + 0.5f * o_signf(r->dy)) * r->rH * 2 - 1, mip_bound, -y) * r->rdy;
  const float tz = fmaf((nz + 0.5f + 0.5f * o_signf(r->dz)) * r->rH * 2 - 1, mip_bound, -z) * r->rdz;
  const float tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
  do {
    t += o_clamp(t * r->dt_gamma, r->dt_min, r->dt_max);
  } while (t < tt);
  return t;
}

void oracle_sph_from_ray(const float* rays_o, const float* rays_d, float radius, uint32_t N, float* coords) {
  for (uint32_t n = 0; n < N; ++n) {
    const float ox = rays_o[n * 3], oy = rays_o[n * 3 + 1], oz = rays_o[n * 3 + 2];
    const float dx = rays_d[n * 3], dy = rays_d[n * 3 + 1], dz = rays_d[n * 3 + 2];
    const float A = dx * dx + dy * dy + dz * dz;
    const float B = ox * dx + oy * dy + oz * dz;
    const float C = ox * ox + oy * oy + oz * oz - radius * radius;
    const float t = (-B + sqrtf(B * B - A * C)) / A;
    const float x = ox + t * dx, y = oy + t * dy, z = oz + t * dz;
    const float theta = atan2f(sqrtf(x * x + z * z), y);
    const float phi = atan2f(z, x);
    coords[n * 2] = 2 * theta * O_RPI - 1;
    coords[n * 2 + 1] = phi * O_RPI;
  }
}

void oracle_march_rays_train(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound, float dt_gamma,
                             uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float* nears,
                             const float* fars, float* xyzs, float* dirs, float* deltas, int32_t* rays, int32_t* counter,
                             const float* noises) {
  for (uint32_t n = 0; n < N; ++n) {
    o_ray r;
    o_ray_init(&r, rays_o + n * 3, rays_d + n * 3, grid, bound, dt_gamma, max_steps, C, H);
    const float far = fars[n];
    float t0 = nears[n];
    t0 = fmaf(o_clamp(t0 * dt_gamma, r.dt_min, r.dt_max), noises[n], t0);
    float t = t0;
    uint32_t num_steps = 0;
    while (t < far && num_steps < max_steps) {                 /* first pass: count (:349-399) */
      float x, y, z, dt, mb;
      int nx, ny, nz;
      if (o_probe(&r, t, &x, &y, &z, &dt, &nx, &ny, &nz, &mb)) { num_steps++; t += dt; }
      else t = o_skip(&r, t, x, y, z, nx, ny, nz, mb);
    }
    const uint32_t point_index = (uint32_t)counter[0];         /* atomicAdd(counter, num_steps) (:404) */
    counter[0] += (int32_t)num_steps;
    const uint32_t ray_index = (uint32_t)counter[1];           /* atomicAdd(counter + 1, 1) (:405) */
    counter[1] += 1;
    rays[ray_index * 3] = (int32_t)n;
    rays[ray_index * 3 + 1] = (int32_t)point_index;
    rays[ray_index * 3 + 2] = (int32_t)num_steps;
    if (num_steps == 0) continue;
    if (point_index + num_steps > M) continue;
    float* px = xyzs + (size_t)point_index * 3;
    float* pd = dirs + (size_t)point_index * 3;
    float* pl = deltas + (size_t)point_index * 2;
    t = t0;
    uint32_t step = 0;
    float last_t = t;
    while (t < far && step < num_steps) {                      /* second pass: write (:426-480) */
      float x, y, z, dt, mb;
      int nx, ny, nz;
      if (o_probe(&r, t, &x, &y, &z, &dt, &nx, &ny, &nz, &mb)) {
        px[0] = x; px[1] = y; px[2] = z;
        pd[0] = r.dx; pd[1] = r.dy; pd[2] = r.dz;
        t += dt;
        pl[0] = dt;
        pl[1] = t - last_t;
        last_t = t;
        px += 3; pd += 3; pl += 2;
        step++;
      } else {
        t = o_skip(&r, t, x, y, z, nx, ny, nz, mb);
      }
    }
  }
}

void oracle_composite_rays_train_forward(const float* sigmas, const float* rgbs, const float* deltas, const int32_t* rays,
                                         uint32_t M, uint32_t N, float T_thresh, float* weights_sum, float* depth,
                                         float* image) {
  for (uint32_t n = 0; n < N; ++n) {
    const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
    if (num_steps == 0 || offset + num_steps > M) {
      weights_sum[index] = 0; depth[index] = 0;
      image[index * 3] = 0; image[index * 3 + 1] = 0; image[index * 3 + 2] = 0;
      continue;
    }
    const float* sg = sigmas + offset;
    const float* cl = rgbs + (size_t)offset * 3;
    const float* dl = deltas + (size_t)offset * 2;
    uint32_t step = 0;
    float T = 1.0f, r = 0, g = 0, b = 0, ws = 0, t = 0, d = 0;
    while (step < num_steps) {
      const float alpha = 1.0f - expf(-sg[0] * dl[0]);
      const float weight = alpha * T;
      r += weight * cl[0]; g += weight * cl[1]; b += weight * cl[2];
      t += dl[1];
      d += weight * t;
      ws += weight;
      T *= 1.0f - alpha;
      if (T < T_thresh) break;
      sg++; cl += 3; dl += 2;
      step++;
    }
    weights_sum[index] = ws; depth[index] = d;
    image[index * 3] = r; image[index * 3 + 1] = g; image[index * 3 + 2] = b;
  }
}

void oracle_composite_rays_train_backward(const float* grad_weights_sum, const float* grad_image, const float* sigmas,
                                          const float* rgbs, const float* deltas, const int32_t* rays,
                                          const float* weights_sum, const float* image, uint32_t M, uint32_t N,
                                          float T_thresh, float* grad_sigmas, float* grad_rgbs) {
  for (uint32_t n = 0; n < N; ++n) {
    const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
    if (num_steps == 0 || offset + num_steps > M) continue;
    const float* gw = grad_weights_sum + index;
    const float* gi = grad_image + index * 3;
    const float* sg = sigmas + offset;
    const float* cl = rgbs + (size_t)offset * 3;
    const float* dl = deltas + (size_t)offset * 2;
    float* gs = grad_sigmas + offset;
    float* gc = grad_rgbs + (size_t)offset * 3;
    uint32_t step = 0;
    float T = 1.0f;
    const float r_final = image[index * 3], g_final = image[index * 3 + 1], b_final = image[index * 3 + 2];
    const float ws_final = weights_sum[index];
    float r = 0, g = 0, b = 0, ws = 0;
    while (step < num_steps) {
      const float alpha = 1.0f - expf(-sg[0] * dl[0]);
      const float weight = alpha * T;
      r += weight * cl[0]; g += weight * cl[1]; b += weight * cl[2];
      ws += weight;
      T *= 1.0f - alpha;
      gc[0] = gi[0] * weight; gc[1] = gi[1] * weight; gc[2] = gi[2] * weight;
      gs[0] = dl[0] * (gi[0] * (T * cl[0] - (r_final - r)) + gi[1] * (T * cl[1] - (g_final - g)) +
                       gi[2] * (T * cl[2] - (b_final - b)) + gw[0] * (1 - ws_final));
      if (T < T_thresh) break;
      sg++; cl += 3; dl += 2; gs++; gc += 3;
      step++;
    }
  }
}

void oracle_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t, const float* rays_o,
                       const float* rays_d, float bound, float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H,
                       const uint8_t* grid, const float* nears, const float* fars, float* xyzs, float* dirs, float* deltas,
                       const float* noises) {
  (void)nears;
  for (uint32_t n = 0; n < n_alive; ++n) {
    const int index = rays_alive[n];
    o_ray r;
    o_ray_init(&r, rays_o + (size_t)index * 3, rays_d + (size_t)index * 3, grid, bound, dt_gamma, max_steps, C, H);
    float* px = xyzs + (size_t)n * n_step * 3;
    float* pd = dirs + (size_t)n * n_step * 3;
    float* pl = deltas + (size_t)n * n_step * 2;
    float t = rays_t[index];
    const float far = fars[index];
    uint32_t step = 0;
    t = fmaf(o_clamp(t * dt_gamma, r.dt_min, r.dt_max), noises[n], t);
    float last_t = t;
    while (t < far && step < n_step) {
      float x, y, z, dt, mb;
      int nx, ny, nz;
      if (o_probe(&r, t, &x, &y, &z, &dt, &nx, &ny, &nz, &mb)) {
        px[0] = x; px[1] = y; px[2] = z;
        pd[0] = r.dx; pd[1] = r.dy; pd[2] = r.dz;
        t += dt;
        pl[0] = dt;
        pl[1] = t - last_t;
        last_t = t;
        px += 3; pd += 3; pl += 2;
        step++;
      } else {
        t = o_skip(&r, t, x, y, z, nx, ny, nz, mb);
      }
    }
  }
}

void oracle_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t* rays_alive, float* rays_t,
                           const float* sigmas, const float* rgbs, const float* deltas, float* weights_sum, float* depth,
                           float* image) {
  for (uint32_t n = 0; n < n_alive; ++n) {
    const int index = rays_alive[n];
    const float* sg = sigmas + (size_t)n * n_step;
    const float* cl = rgbs + (size_t)n * n_step * 3;
    const float* dl = deltas + (size_t)n * n_step * 2;
    float t = rays_t[index];
    float weight_sum = weights_sum[index], d = depth[index];
    float r = image[index * 3], g = image[index * 3 + 1], b = image[index * 3 + 2];
    uint32_t step = 0;
    while (step < n_step) {
      if (dl[0] == 0) break;
      const float alpha = 1.0f - expf(-sg[0] * dl[0]);
      const float T = 1 - weight_sum;
      const float weight = alpha * T;
      weight_sum += weight;
      t += dl[1];
      d += weight * t;
      r += weight * cl[0]; g += weight * cl[1]; b += weight * cl[2];
      if (T < T_thresh) break;
      sg++; cl += 3; dl += 2;
      step++;
    }
    if (step < n_step) rays_alive[n] = -1;
    else rays_t[index] = t;
    weights_sum[index] = weight_sum; depth[index] = d;
    image[index * 3] = r; image[index * 3 + 1] = g; image[index * 3 + 2] = b;
  }
}

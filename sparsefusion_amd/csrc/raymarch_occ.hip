// Occupancy-grid ray marching and compositing for gfx950: the `cuda_ray=True` entry points of `_raymarching`
// (SURVEY.md 8(f) row 3).
//
// Replaces
//   sph_from_ray                    raymarching/src/raymarching.cu:159-204
//   march_rays_train                raymarching/src/raymarching.cu:302-492
//   composite_rays_train_forward    raymarching/src/raymarching.cu:495-583
//   composite_rays_train_backward   raymarching/src/raymarching.cu:586-688
//   march_rays                      raymarching/src/raymarching.cu:695-818
//   composite_rays                  raymarching/src/raymarching.cu:821-913
//
// MI355X notes.  The reference hands out point / ray slots with two same-address atomicAdd per ray; on MI355X
// those execute at the memory side and serialise (~50-80 ns each: 16 384 rays = 2.6 ms of pure queueing) and make
// the packing order run-dependent.  Here the sample counts go through a block scan instead: slots are assigned in
// ray order, the result is one of the reference's valid outcomes and is reproducible bit for bit.  Everything is
// one ray per lane and HBM/latency bound (byte loads from the 256 KiB..2 MiB bitfield, 32 B of output per sample);
// fp32 arithmetic is written with the contraction nvcc applies (-fmad=true) so that cell indices and sample
// positions are bit-exact against oracle/ngp_ref.c.

#include "sf_common.h"
#include "ngp_field_lds.h"
#include <float.h>

#define SF_SQRT3 1.7320508075688772f
#define SF_RPI 0.3183098861837907f
#define MARCH_BLOCK 256
#define SCAN_RAYS 1024                       // rays per scan workgroup (256 lanes x 4)

__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(hi, fmaxf(lo, x)); }
__device__ __forceinline__ float signf1(float x) { return copysignf(1.0f, x); }

__device__ __forceinline__ uint32_t occ_expand_bits(uint32_t v) {
  v = (v * 0x00010001u) & 0xFF0000FFu;
  v = (v * 0x00000101u) & 0x0F00F00Fu;
  v = (v * 0x00000011u) & 0xC30C30C3u;
  v = (v * 0x00000005u) & 0x49249249u;
  return v;
}

__device__ __forceinline__ int mip_from_pos(float x, float y, float z, float max_cascade) {
  const float mx = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
  int e;
  (void)frexpf(mx, &e);
  return (int)fminf(max_cascade - 1.0f, fmaxf(0.0f, (float)e));
}
__device__ __forceinline__ int mip_from_dt(float dt, float H, float max_cascade) {
  const float mx = (float)((double)(dt * H) * 0.5);
  int e;
  (void)frexpf(mx, &e);
  return (int)fminf(max_cascade - 1.0f, fmaxf(0.0f, (float)e));
}

// State of one marching ray; `probe` classifies the sample at the current t, `skip` jumps an empty cell.
struct Marcher {
  float ox, oy, oz, dx, dy, dz, rdx, rdy, rdz, rH, H3, bound, dt_gamma, dt_min, dt_max;
  uint32_t C, H;
  const uint8_t* grid;

  __device__ __forceinline__ void init(const float* o, const float* d, const uint8_t* g, float bnd, float gamma,
                                       uint32_t max_steps, uint32_t c, uint32_t h) {
    ox = o[0]; oy = o[1]; oz = o[2]; dx = d[0]; dy = d[1]; dz = d[2];
    rdx = __fdiv_rn(1.0f, dx); rdy = __fdiv_rn(1.0f, dy); rdz = __fdiv_rn(1.0f, dz);
    rH = __fdiv_rn(1.0f, (float)h);
    H3 = (float)(h * h * h);
    bound = bnd; dt_gamma = gamma; C = c; H = h; grid = g;
    dt_min = __fdiv_rn(2.0f * SF_SQRT3, (float)max_steps);
    dt_max = __fdiv_rn(2.0f * SF_SQRT3 * (float)(1 << (c - 1)), (float)h);
  }
  __device__ __forceinline__ float step_size(float t) const { return clampf(t * dt_gamma, dt_min, dt_max); }

  // returns occupancy of the cell holding the sample at t; fills the clamped position, dt and the cell geometry
  __device__ __forceinline__ bool probe(float t, float& x, float& y, float& z, float& dt, int& nx, int& ny, int& nz,
                                        float& mip_bound) const {
    x = clampf(fmaf(t, dx, ox), -bound, bound);
    y = clampf(fmaf(t, dy, oy), -bound, bound);
    z = clampf(fmaf(t, dz, oz), -bound, bound);
    dt = step_size(t);
    const int l0 = mip_from_pos(x, y, z, (float)C), l1 = mip_from_dt(dt, (float)H, (float)C);
    const int level = l0 > l1 ? l0 : l1;
    mip_bound = fminf(scalbnf(1.0f, level), bound);
    const float mip_rbound = __fdiv_rn(1.0f, mip_bound);
    const float hi = (float)(H - 1);
    // `0.5 * (...) * H` is evaluated in double (0.5 is a double literal), then narrowed for clamp()
    nx = (int)clampf((float)(0.5 * (double)fmaf(x, mip_rbound, 1.0f) * (double)H), 0.0f, hi);
    ny = (int)clampf((float)(0.5 * (double)fmaf(y, mip_rbound, 1.0f) * (double)H), 0.0f, hi);
    nz = (int)clampf((float)(0.5 * (double)fmaf(z, mip_rbound, 1.0f) * (double)H), 0.0f, hi);
    // level * H3 is a float product converted to uint32 after the integer Morton code is added in float
    const uint32_t morton = occ_expand_bits((uint32_t)nx) | (occ_expand_bits((uint32_t)ny) << 1) | (occ_expand_bits((uint32_t)nz) << 2);
    const uint32_t index = (uint32_t)((float)level * H3 + (float)morton);
    return (grid[index >> 3] >> (index & 7u)) & 1u;
  }
  // advance t past the empty cell (nx, ny, nz)
  __device__ __forceinline__ float skip(float t, float x, float y, float z, int nx, int ny, int nz, float mip_bound) const {
    const float tx = (fmaf(((float)nx + 0.5f + 0.5f * signf1(dx)) * rH * 2.0f - 1.0f, mip_bound, -x)) * rdx;
    const float ty = (fmaf(((float)ny + 0.5f + 0.5f * signf1(dy)) * rH * 2.0f - 1.0f, mip_bound, -y)) * rdy;
    const float tz = (fmaf(((float)nz + 0.5f + 0.5f * signf1(dz)) * rH * 2.0f - 1.0f, mip_bound, -z)) * rdz;
    const float tt = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
    do { t += step_size(t); } while (t < tt);
    return t;
  }
};

// ---------------------------------------------------------------------------------------------------------------
// sph_from_ray
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sph_from_ray(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                      float radius, uint32_t N, float* __restrict__ coords) {
  for (uint32_t n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
    const float ox = rays_o[n * 3], oy = rays_o[n * 3 + 1], oz = rays_o[n * 3 + 2];
    const float dx = rays_d[n * 3], dy = rays_d[n * 3 + 1], dz = rays_d[n * 3 + 2];
    const float A = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
    const float B = fmaf(oz, dz, fmaf(oy, dy, ox * dx));
    const float Cq = fmaf(-radius, radius, fmaf(oz, oz, fmaf(oy, oy, ox * ox)));
    const float t = __fdiv_rn(-B + __fsqrt_rn(fmaf(B, B, -(A * Cq))), A);
    const float x = fmaf(t, dx, ox), y = fmaf(t, dy, oy), z = fmaf(t, dz, oz);
    const float theta = atan2f(__fsqrt_rn(fmaf(z, z, x * x)), y);
    const float phi = atan2f(z, x);
    coords[n * 2] = fmaf(2.0f * theta, SF_RPI, -1.0f);
    coords[n * 2 + 1] = phi * SF_RPI;
  }
}

extern "C" int sf_sph_from_ray(const float* rays_o, const float* rays_d, float radius, uint32_t N, float* coords,
                               void* stream) {
  if (N == 0) return SF_OK;
  if (!rays_o || !rays_d || !coords) SF_FAIL(SF_ERR_INVALID, "sph_from_ray: null tensor");
  k_sph_from_ray<<<sf_grid_cap(sf_div_up(N, 256)), 256, 0, (hipStream_t)stream>>>(rays_o, rays_d, radius, N, coords);
  SF_CHECK_LAUNCH("sph_from_ray");
  return SF_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// march_rays_train: (1) count samples per ray, (2) block scan, (3) assign slots in ray order and write the samples
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(MARCH_BLOCK) void k_march_count(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                             const uint8_t* __restrict__ grid, float bound, float dt_gamma,
                                                             uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H,
                                                             const float* __restrict__ nears, const float* __restrict__ fars,
                                                             const float* __restrict__ noises, int32_t* __restrict__ steps) {
  const uint32_t n = blockIdx.x * MARCH_BLOCK + threadIdx.x;
  if (n >= N) return;
  Marcher m;
  m.init(rays_o + n * 3, rays_d + n * 3, grid, bound, dt_gamma, max_steps, C, H);
  const float far = fars[n];
  float t = nears[n];
  t = fmaf(m.step_size(t), noises[n], t);
  uint32_t num = 0;
  while (t < far && num < max_steps) {
    float x, y, z, dt, mb;
    int nx, ny, nz;
    if (m.probe(t, x, y, z, dt, nx, ny, nz, mb)) { ++num; t += dt; }
    else t = m.skip(t, x, y, z, nx, ny, nz, mb);
  }
  steps[n] = (int32_t)num;
}

// exclusive scan of `steps` inside each group of SCAN_RAYS rays -> local[n]; group totals -> totals[g]
// (also snapshots the caller's (points, rays) counter into cbase: k_march_write updates the counter in place)
__global__ __launch_bounds__(256) void k_march_scan(const int32_t* __restrict__ steps, uint32_t N, int32_t* __restrict__ local,
                                                    int32_t* __restrict__ totals, const int32_t* __restrict__ counter,
                                                    int32_t* __restrict__ cbase) {
  __shared__ int32_t wsum[4];
  if (blockIdx.x == 0 && threadIdx.x < 2) cbase[threadIdx.x] = counter[threadIdx.x];
  const uint32_t base = blockIdx.x * SCAN_RAYS + threadIdx.x * 4;
  int32_t v[4], s = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) { v[k] = (base + k < N) ? steps[base + k] : 0; s += v[k]; }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int32_t incl = s;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int32_t o = __shfl_up(incl, off);
    if (lane >= off) incl += o;
  }
  if (lane == 63) wsum[wv] = incl;
  __syncthreads();
  int32_t pre = incl - s;
  for (int w = 0; w < wv; ++w) pre += wsum[w];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (base + k < N) local[base + k] = pre;
    pre += v[k];
  }
  if (threadIdx.x == 255) totals[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ __launch_bounds__(MARCH_BLOCK) void k_march_write(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                             const uint8_t* __restrict__ grid, float bound, float dt_gamma,
                                                             uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M,
                                                             const float* __restrict__ nears, const float* __restrict__ fars,
                                                             const float* __restrict__ noises, const int32_t* __restrict__ steps,
                                                             const int32_t* __restrict__ local, const int32_t* __restrict__ totals,
                                                             uint32_t n_groups, const int32_t* __restrict__ counter_in,
                                                             float* __restrict__ xyzs, float* __restrict__ dirs,
                                                             float* __restrict__ deltas, int32_t* __restrict__ rays,
                                                             int32_t* __restrict__ counter_out) {
  __shared__ int32_t red[MARCH_BLOCK / 64];
  __shared__ int32_t gbase;
  // slot base of this workgroup's scan group = counter[0] + totals of all earlier groups (<= 4096 values)
  const uint32_t grp = (blockIdx.x * MARCH_BLOCK) / SCAN_RAYS;
  int32_t part = 0;
  for (uint32_t g = threadIdx.x; g < grp; g += MARCH_BLOCK) part += totals[g];
  for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
  __syncthreads();
  if (threadIdx.x == 0) {
    int32_t b = counter_in[0];
    for (int w = 0; w < MARCH_BLOCK / 64; ++w) b += red[w];
    gbase = b;
    if (blockIdx.x == gridDim.x - 1) {                   // the last workgroup owns the group with the grand total
      int32_t total = b;
      for (uint32_t g = grp; g < n_groups; ++g) total += totals[g];
      counter_out[0] = total;
      counter_out[1] = counter_in[1] + (int32_t)N;
    }
  }
  __syncthreads();
  const uint32_t n = blockIdx.x * MARCH_BLOCK + threadIdx.x;
  if (n >= N) return;
  const uint32_t num_steps = (uint32_t)steps[n];
  const uint32_t point_index = (uint32_t)(gbase + local[n]);
  const uint32_t ray_index = (uint32_t)counter_in[1] + n;
  if (ray_index < N) {
    rays[ray_index * 3] = (int32_t)n;
    rays[ray_index * 3 + 1] = (int32_t)point_index;
    rays[ray_index * 3 + 2] = (int32_t)num_steps;
  }
  if (num_steps == 0 || point_index + num_steps > M) return;
  Marcher m;
  m.init(rays_o + n * 3, rays_d + n * 3, grid, bound, dt_gamma, max_steps, C, H);
  const float far = fars[n];
  float t = nears[n];
  t = fmaf(m.step_size(t), noises[n], t);
  float last_t = t;
  float* px = xyzs + (size_t)point_index * 3;
  float* pd = dirs + (size_t)point_index * 3;
  float* pl = deltas + (size_t)point_index * 2;
  uint32_t step = 0;
  while (t < far && step < num_steps) {
    float x, y, z, dt, mb;
    int nx, ny, nz;
    if (m.probe(t, x, y, z, dt, nx, ny, nz, mb)) {
      px[0] = x; px[1] = y; px[2] = z;
      pd[0] = m.dx; pd[1] = m.dy; pd[2] = m.dz;
      t += dt;
      pl[0] = dt; pl[1] = t - last_t;
      last_t = t;
      px += 3; pd += 3; pl += 2;
      ++step;
    } else {
      t = m.skip(t, x, y, z, nx, ny, nz, mb);
    }
  }
}

extern "C" uint64_t sf_march_rays_train_workspace_bytes(uint32_t N) {
  return ((uint64_t)N * 2 + sf_div_up((uint64_t)N, (uint64_t)SCAN_RAYS) + 2) * sizeof(int32_t);
}

extern "C" int sf_march_rays_train(const float* rays_o, const float* rays_d, const uint8_t* grid, float bound, float dt_gamma,
                                   uint32_t max_steps, uint32_t N, uint32_t C, uint32_t H, uint32_t M, const float* nears,
                                   const float* fars, float* xyzs, float* dirs, float* deltas, int32_t* rays, int32_t* counter,
                                   const float* noises, void* workspace, uint64_t workspace_bytes, void* stream) {
  if (N == 0) return SF_OK;
  if (!rays_o || !rays_d || !grid || !nears || !fars || !xyzs || !dirs || !deltas || !rays || !counter || !noises)
    SF_FAIL(SF_ERR_INVALID, "march_rays_train: null tensor");
  if (C < 1 || C > 16 || H < 1 || H > 1024) SF_FAIL(SF_ERR_INVALID, "march_rays_train: cascade / grid size out of range");
  if (!workspace || workspace_bytes < sf_march_rays_train_workspace_bytes(N))
    SF_FAIL(SF_ERR_INVALID, "march_rays_train: workspace too small (need %llu bytes)",
            (unsigned long long)sf_march_rays_train_workspace_bytes(N));
  const uint32_t n_groups = (uint32_t)sf_div_up((uint64_t)N, (uint64_t)SCAN_RAYS);
  if (n_groups > 65536) SF_FAIL(SF_ERR_INVALID, "march_rays_train: at most 64 M rays per call");
  hipStream_t st = (hipStream_t)stream;
  int32_t* steps = (int32_t*)workspace;
  int32_t* local = steps + N;
  int32_t* totals = local + N;
  const uint32_t blocks = (uint32_t)sf_div_up(N, (uint32_t)MARCH_BLOCK);
  k_march_count<<<blocks, MARCH_BLOCK, 0, st>>>(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, nears, fars, noises, steps);
  SF_CHECK_LAUNCH("march_count");
  int32_t* cbase = totals + n_groups;
  k_march_scan<<<n_groups, 256, 0, st>>>(steps, N, local, totals, counter, cbase);
  SF_CHECK_LAUNCH("march_scan");
  k_march_write<<<blocks, MARCH_BLOCK, 0, st>>>(rays_o, rays_d, grid, bound, dt_gamma, max_steps, N, C, H, M, nears, fars, noises,
                                                steps, local, totals, n_groups, cbase, xyzs, dirs, deltas, rays, counter);
  SF_CHECK_LAUNCH("march_write");
  return SF_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// composite_rays_train forward / backward (one ray per lane; a ray's samples are contiguous)
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_composite_train_fwd(const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                                                             const float* __restrict__ deltas, const int32_t* __restrict__ rays,
                                                             uint32_t M, uint32_t N, float T_thresh,
                                                             float* __restrict__ weights_sum, float* __restrict__ depth,
                                                             float* __restrict__ image) {
  const uint32_t n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
  if (num_steps == 0 || offset + num_steps > M) {
    weights_sum[index] = 0.0f; depth[index] = 0.0f;
    image[index * 3] = 0.0f; image[index * 3 + 1] = 0.0f; image[index * 3 + 2] = 0.0f;
    return;
  }
  const float* sg = sigmas + offset;
  const float* cl = rgbs + (size_t)offset * 3;
  const float* dl = deltas + (size_t)offset * 2;
  float T = 1.0f, r = 0.0f, g = 0.0f, b = 0.0f, ws = 0.0f, t = 0.0f, d = 0.0f;
  for (uint32_t step = 0; step < num_steps; ++step) {
    const float alpha = 1.0f - __expf(-sg[step] * dl[step * 2]);
    const float weight = alpha * T;
    r = fmaf(weight, cl[step * 3], r);
    g = fmaf(weight, cl[step * 3 + 1], g);
    b = fmaf(weight, cl[step * 3 + 2], b);
    t += dl[step * 2 + 1];
    d = fmaf(weight, t, d);
    ws += weight;
    T *= 1.0f - alpha;
    if (T < T_thresh) break;
  }
  weights_sum[index] = ws; depth[index] = d;
  image[index * 3] = r; image[index * 3 + 1] = g; image[index * 3 + 2] = b;
}

__global__ __launch_bounds__(256) void k_composite_train_bwd(const float* __restrict__ grad_weights_sum,
                                                             const float* __restrict__ grad_image,
                                                             const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                                                             const float* __restrict__ deltas, const int32_t* __restrict__ rays,
                                                             const float* __restrict__ weights_sum, const float* __restrict__ image,
                                                             uint32_t M, uint32_t N, float T_thresh,
                                                             float* __restrict__ grad_sigmas, float* __restrict__ grad_rgbs) {
  const uint32_t n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const uint32_t index = (uint32_t)rays[n * 3], offset = (uint32_t)rays[n * 3 + 1], num_steps = (uint32_t)rays[n * 3 + 2];
  if (num_steps == 0 || offset + num_steps > M) return;
  const float gws = grad_weights_sum[index];
  const float g0 = grad_image[index * 3], g1 = grad_image[index * 3 + 1], g2 = grad_image[index * 3 + 2];
  const float r_final = image[index * 3], g_final = image[index * 3 + 1], b_final = image[index * 3 + 2];
  const float ws_final = weights_sum[index];
  const float* sg = sigmas + offset;
  const float* cl = rgbs + (size_t)offset * 3;
  const float* dl = deltas + (size_t)offset * 2;
  float* gs = grad_sigmas + offset;
  float* gc = grad_rgbs + (size_t)offset * 3;
  float T = 1.0f, r = 0.0f, g = 0.0f, b = 0.0f;
  for (uint32_t step = 0; step < num_steps; ++step) {
    const float c0 = cl[step * 3], c1 = cl[step * 3 + 1], c2 = cl[step * 3 + 2], dlt = dl[step * 2];
    const float alpha = 1.0f - __expf(-sg[step] * dlt);
    const float weight = alpha * T;
    r = fmaf(weight, c0, r); g = fmaf(weight, c1, g); b = fmaf(weight, c2, b);
    T *= 1.0f - alpha;
    gc[step * 3] = g0 * weight; gc[step * 3 + 1] = g1 * weight; gc[step * 3 + 2] = g2 * weight;
    // d image / d sigma_i = delta_i * (T_{i+1} * c_i - (C_final - C_{<=i})); the weights_sum term likewise
    float acc = g0 * fmaf(T, c0, -(r_final - r));
    acc = fmaf(g1, fmaf(T, c1, -(g_final - g)), acc);
    acc = fmaf(g2, fmaf(T, c2, -(b_final - b)), acc);
    acc = fmaf(gws, 1.0f - ws_final, acc);
    gs[step] = dlt * acc;
    if (T < T_thresh) break;
  }
}

extern "C" int sf_composite_rays_train_forward(const float* sigmas, const float* rgbs, const float* deltas, const int32_t* rays,
                                               uint32_t M, uint32_t N, float T_thresh, float* weights_sum, float* depth,
                                               float* image, void* stream) {
  if (N == 0) return SF_OK;
  if (!rays || !weights_sum || !depth || !image || (M && (!sigmas || !rgbs || !deltas)))
    SF_FAIL(SF_ERR_INVALID, "composite_rays_train_forward: null tensor");
  k_composite_train_fwd<<<sf_div_up(N, 256u), 256, 0, (hipStream_t)stream>>>(sigmas, rgbs, deltas, rays, M, N, T_thresh,
                                                                            weights_sum, depth, image);
  SF_CHECK_LAUNCH("composite_rays_train_forward");
  return SF_OK;
}

extern "C" int sf_composite_rays_train_backward(const float* grad_weights_sum, const float* grad_image, const float* sigmas,
                                                const float* rgbs, const float* deltas, const int32_t* rays,
                                                const float* weights_sum, const float* image, uint32_t M, uint32_t N,
                                                float T_thresh, float* grad_sigmas, float* grad_rgbs, void* stream) {
  if (N == 0 || M == 0) return SF_OK;
  if (!grad_weights_sum || !grad_image || !sigmas || !rgbs || !deltas || !rays || !weights_sum || !image || !grad_sigmas ||
      !grad_rgbs)
    SF_FAIL(SF_ERR_INVALID, "composite_rays_train_backward: null tensor");
  k_composite_train_bwd<<<sf_div_up(N, 256u), 256, 0, (hipStream_t)stream>>>(grad_weights_sum, grad_image, sigmas, rgbs, deltas,
                                                                            rays, weights_sum, image, M, N, T_thresh,
                                                                            grad_sigmas, grad_rgbs);
  SF_CHECK_LAUNCH("composite_rays_train_backward");
  return SF_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// inference: march n_step samples for every alive ray, composite them in place
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(MARCH_BLOCK) void k_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t* __restrict__ rays_alive,
                                                            const float* __restrict__ rays_t, const float* __restrict__ rays_o,
                                                            const float* __restrict__ rays_d, float bound, float dt_gamma,
                                                            uint32_t max_steps, uint32_t C, uint32_t H,
                                                            const uint8_t* __restrict__ grid, const float* __restrict__ nears,
                                                            const float* __restrict__ fars, float* __restrict__ xyzs,
                                                            float* __restrict__ dirs, float* __restrict__ deltas,
                                                            const float* __restrict__ noises) {
  const uint32_t n = blockIdx.x * MARCH_BLOCK + threadIdx.x;
  if (n >= n_alive) return;
  const int32_t index = rays_alive[n];
  Marcher m;
  m.init(rays_o + (size_t)index * 3, rays_d + (size_t)index * 3, grid, bound, dt_gamma, max_steps, C, H);
  float* px = xyzs + (size_t)n * n_step * 3;
  float* pd = dirs + (size_t)n * n_step * 3;
  float* pl = deltas + (size_t)n * n_step * 2;
  float t = rays_t[index];
  const float far = fars[index];
  t = fmaf(m.step_size(t), noises[n], t);
  float last_t = t;
  uint32_t step = 0;
  while (t < far && step < n_step) {
    float x, y, z, dt, mb;
    int nx, ny, nz;
    if (m.probe(t, x, y, z, dt, nx, ny, nz, mb)) {
      px[0] = x; px[1] = y; px[2] = z;
      pd[0] = m.dx; pd[1] = m.dy; pd[2] = m.dz;
      t += dt;
      pl[0] = dt; pl[1] = t - last_t;
      last_t = t;
      px += 3; pd += 3; pl += 2;
      ++step;
    } else {
      t = m.skip(t, x, y, z, nx, ny, nz, mb);
    }
  }
}

__global__ __launch_bounds__(256) void k_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh,
                                                        int32_t* __restrict__ rays_alive, float* __restrict__ rays_t,
                                                        const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                                                        const float* __restrict__ deltas, float* __restrict__ weights_sum,
                                                        float* __restrict__ depth, float* __restrict__ image) {
  const uint32_t n = blockIdx.x * 256 + threadIdx.x;
  if (n >= n_alive) return;
  const int32_t index = rays_alive[n];
  const float* sg = sigmas + (size_t)n * n_step;
  const float* cl = rgbs + (size_t)n * n_step * 3;
  const float* dl = deltas + (size_t)n * n_step * 2;
  float t = rays_t[index];
  float weight_sum = weights_sum[index], d = depth[index];
  float r = image[index * 3], g = image[index * 3 + 1], b = image[index * 3 + 2];
  uint32_t step = 0;
  while (step < n_step) {
    if (dl[step * 2] == 0.0f) break;                       // the marcher ran out of the volume: ray finished
    const float alpha = 1.0f - __expf(-sg[step] * dl[step * 2]);
    const float T = 1.0f - weight_sum;
    const float weight = alpha * T;
    weight_sum += weight;
    t += dl[step * 2 + 1];
    d = fmaf(weight, t, d);
    r = fmaf(weight, cl[step * 3], r);
    g = fmaf(weight, cl[step * 3 + 1], g);
    b = fmaf(weight, cl[step * 3 + 2], b);
    if (T < T_thresh) break;
    ++step;
  }
  if (step < n_step) rays_alive[n] = -1;
  else rays_t[index] = t;
  weights_sum[index] = weight_sum; depth[index] = d;
  image[index * 3] = r; image[index * 3 + 1] = g; image[index * 3 + 2] = b;
}

extern "C" int sf_march_rays(uint32_t n_alive, uint32_t n_step, const int32_t* rays_alive, const float* rays_t,
                             const float* rays_o, const float* rays_d, float bound, float dt_gamma, uint32_t max_steps, uint32_t C,
                             uint32_t H, const uint8_t* grid, const float* nears, const float* fars, float* xyzs, float* dirs,
                             float* deltas, const float* noises, void* stream) {
  if (n_alive == 0 || n_step == 0) return SF_OK;
  if (!rays_alive || !rays_t || !rays_o || !rays_d || !grid || !nears || !fars || !xyzs || !dirs || !deltas || !noises)
    SF_FAIL(SF_ERR_INVALID, "march_rays: null tensor");
  if (C < 1 || C > 16 || H < 1 || H > 1024) SF_FAIL(SF_ERR_INVALID, "march_rays: cascade / grid size out of range");
  k_march_rays<<<sf_div_up(n_alive, (uint32_t)MARCH_BLOCK), MARCH_BLOCK, 0, (hipStream_t)stream>>>(
      n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, C, H, grid, nears, fars, xyzs, dirs, deltas,
      noises);
  SF_CHECK_LAUNCH("march_rays");
  return SF_OK;
}

extern "C" int sf_composite_rays(uint32_t n_alive, uint32_t n_step, float T_thresh, int32_t* rays_alive, float* rays_t,
                                 const float* sigmas, const float* rgbs, const float* deltas, float* weights_sum, float* depth,
                                 float* image, void* stream) {
  if (n_alive == 0 || n_step == 0) return SF_OK;
  if (!rays_alive || !rays_t || !sigmas || !rgbs || !deltas || !weights_sum || !depth || !image)
    SF_FAIL(SF_ERR_INVALID, "composite_rays: null tensor");
  k_composite_rays<<<sf_div_up(n_alive, 256u), 256, 0, (hipStream_t)stream>>>(n_alive, n_step, T_thresh, rays_alive, rays_t,
                                                                             sigmas, rgbs, deltas, weights_sum, depth, image);
  SF_CHECK_LAUNCH("composite_rays");
  return SF_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Fused occupancy-grid EVALUATION render: march + field + composite for one ray per lane in ONE launch.
// The reference's inference path (external/nerf/renderer_df.py:543-584) alternates march_rays / network / composite_rays
// rounds over a shrinking list of alive rays, with a host read of the list length before every round.  A ray's samples,
// their order and the arithmetic applied to them do not depend on that batching: here each lane simply walks its ray
// to the end (transmittance below T_thresh, far plane, or max_steps samples), evaluating the field (hash-grid encode +
// MLP, weights in LDS) at every occupied sample -- no alive lists, no intermediate buffers, no host synchronisation.
// Same per-sample formulas as k_march_rays / k_ngp_field<2> / k_composite_rays above.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_render_occ_eval(FieldPtrs f, NgpLevels lv, const float* __restrict__ rays_o,
                                                         const float* __restrict__ rays_d, const float* __restrict__ nears,
                                                         const float* __restrict__ fars, const uint8_t* __restrict__ grid,
                                                         float dt_gamma, uint32_t max_steps, uint32_t C, uint32_t H,
                                                         const float* __restrict__ noises, float T_thresh, uint32_t N,
                                                         float* __restrict__ weights_sum, float* __restrict__ depth,
                                                         float* __restrict__ image) {
  __shared__ __attribute__((aligned(16))) float W[NGP_WTOTAL];
  load_weights_lds(W, f);
  __syncthreads();
  const uint32_t n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  Marcher m;
  m.init(rays_o + (size_t)n * 3, rays_d + (size_t)n * 3, grid, f.bound, dt_gamma, max_steps, C, H);
  const float far = fars[n];
  float t = nears[n];
  if (noises) t = fmaf(m.step_size(t), noises[n], t);
  float tc = nears[n];                                     // the compositor's running depth (starts at rays_t = near)
  float last_t = t;
  float wsum = 0.0f, d = 0.0f, r = 0.0f, g = 0.0f, b = 0.0f;
  uint32_t step = 0;
  while (t < far && step < max_steps) {
    asm volatile("" ::: "memory");                         // keep the LDS weight reads inside the loop (see k_ngp_field)
    float x[3], dt, mb;
    int nx, ny, nz;
    if (!m.probe(t, x[0], x[1], x[2], dt, nx, ny, nz, mb)) {
      t = m.skip(t, x[0], x[1], x[2], nx, ny, nz, mb);
      continue;
    }
    t += dt;
    const float gap = t - last_t;
    last_t = t;
    ++step;
    float x01[3];
    const bool inside = ngp_unit(x, f.bound, x01);
    float feat[NGP_FEAT], h1[NGP_HID], h2[NGP_HID], out[NGP_OUT];
    ngp_encode(lv, f.table, x01, inside, feat);
    ngp_mlp_forward(W, feat, h1, h2, out);
    const float sigma = expf(out[0] + ngp_blob(x));
    const float alpha = 1.0f - __expf(-sigma * dt);
    const float T = 1.0f - wsum;
    const float weight = alpha * T;
    wsum += weight;
    tc += gap;
    d = fmaf(weight, tc, d);
    r = fmaf(weight, ngp_sigmoid(out[1]), r);
    g = fmaf(weight, ngp_sigmoid(out[2]), g);
    b = fmaf(weight, ngp_sigmoid(out[3]), b);
    if (T < T_thresh) break;
  }
  weights_sum[n] = wsum; depth[n] = d;
  image[n * 3] = r; image[n * 3 + 1] = g; image[n * 3 + 2] = b;
}

extern "C" int sf_ngp_render_occ_eval(const sf_ngp_field* f, const float* rays_o, const float* rays_d, const float* nears,
                                      const float* fars, const uint8_t* grid, float dt_gamma, uint32_t max_steps, uint32_t C,
                                      uint32_t H, const float* noises, float T_thresh, uint32_t N, float* weights_sum,
                                      float* depth, float* image, void* stream) {
  if (N == 0) return SF_OK;
  if (!f || !rays_o || !rays_d || !nears || !fars || !grid || !weights_sum || !depth || !image)
    SF_FAIL(SF_ERR_INVALID, "render_occ_eval: null tensor");
  if (C < 1 || C > 16 || H < 1 || H > 1024 || max_steps < 1) SF_FAIL(SF_ERR_INVALID, "render_occ_eval: cascade / grid size / steps out of range");
  NgpLevels lv;
  if (int rc = sf_ngp_make_levels(f, &lv, (hipStream_t)stream)) return rc;
  k_render_occ_eval<<<sf_div_up(N, 256u), 256, 0, (hipStream_t)stream>>>(sf_ngp_field_ptrs(f), lv, rays_o, rays_d, nears, fars, grid,
                                                                         dt_gamma, max_steps, C, H, noises, T_thresh, N,
                                                                         weights_sum, depth, image);
  SF_CHECK_LAUNCH("render_occ_eval");
  return SF_OK;
}

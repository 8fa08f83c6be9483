// Fused Instant-NGP volume render for gfx950 (forward + backward w.r.t. field parameters).
//
// What it replaces: NeRFRenderer.run (external/nerf/renderer_df.py:310-468) driving
// NeRFNetwork.common_forward (external/nerf/network_grid.py:77-88) three times per render plus
// ~40 unfused torch elementwise/sort/gather/cumprod kernels.  MI355X-first restructuring:
//   * the field is evaluated ONCE per sample (128/ray instead of the reference's 256: its third
//     pass re-evaluates the same points; value-identical, and the gradient of one evaluation
//     with both upstream grads equals the sum of the reference's two paths);
//   * encode + MLP + activations are one kernel, MLP weights live in LDS and are read as
//     broadcasts; nothing but sigma/rgb per sample is written to HBM;
//   * sampling (inverse-CDF), merge-sort and alpha compositing are per-ray kernels whose
//     per-thread scratch columns live in LDS ([k][lane] layout: bank-conflict free);
//   * backward RECOMPUTES the field forward instead of saving activations (VALU is cheap, HBM is
//     not); MLP weight gradients are formed per 256-point tile as LDS-staged outer-product
//     sums, one atomic flush per workgroup; table gradients use fp32 L2 atomics.
// Per-thread math lives in ngp_device.h (also compiled for the host by tests/hostemu).
//
// Layouts (row-major): z_c,sig_c [N,T]; rgb_c [N,T,3]; z_f,sig_f,rgb_f same; z_sorted,sigma_s
// [N,2T]; rgb_s [N,2T,3]; image [N,3]; depth, weights_sum, nears, fars [N].

#include "sf_common.h"
#include "ngp_device.h"
#include "ngp_field_lds.h"
#include "ngp_bwd_mfma.h"
#include "ngp_composite_wave.h"
#include "ngp_scatter_bin.h"
#include <stdlib.h>
#include <mutex>

struct GridLevels;  // gridencoder.hip
int sf_fill_levels(GridLevels* lv, const int32_t* offsets_dev, const int32_t* h_offsets, uint32_t L, float S,
                   uint32_t H, hipStream_t st);

// mode 0: z from stratified coarse rule (writes z_out); mode 1: z read from z_in; mode 2: xyz given.
template <int MODE>
__global__ __launch_bounds__(256) void k_ngp_field(
    FieldPtrs f, NgpLevels lv, const float* __restrict__ rays_o, const float* __restrict__ rays_d,
    const float* __restrict__ aabb, const float* __restrict__ nears, const float* __restrict__ fars,
    const float* __restrict__ lin, const float* __restrict__ u, const float* __restrict__ z_in,
    const float* __restrict__ xyz_in, uint32_t P, uint32_t T, float* __restrict__ z_out,
    float* __restrict__ sigma, float* __restrict__ rgb, float* __restrict__ feat_out = nullptr) {
  __shared__ __attribute__((aligned(16))) float W[NGP_WTOTAL];
  load_weights_lds(W, f);
  __syncthreads();
  float box[6];
  if (MODE != 2) {
#pragma unroll
    for (int i = 0; i < 6; ++i) box[i] = aabb[i];
  }
  for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
    // Memory clobber: stops LLVM from hoisting the 6.5k loop-invariant LDS weight reads out of the
    // grid-stride loop (which would pin them in 512 registers and spill).
    asm volatile("" ::: "memory");
    float x[3];
    if (MODE == 2) {
      x[0] = xyz_in[p * 3 + 0]; x[1] = xyz_in[p * 3 + 1]; x[2] = xyz_in[p * 3 + 2];
    } else {
      const uint32_t n = p / T, k = p - n * T;
      float z;
      if (MODE == 0) {
        z = ngp_coarse_z(nears[n], fars[n], lin[k], u ? u[p] : -1.0f, T);
        z_out[p] = z;
      } else {
        z = z_in[p];
      }
      const float o[3] = {rays_o[n * 3], rays_o[n * 3 + 1], rays_o[n * 3 + 2]};
      const float d[3] = {rays_d[n * 3], rays_d[n * 3 + 1], rays_d[n * 3 + 2]};
      ngp_point(o, d, z, box, x);
    }
    float x01[3];
    const bool inside = ngp_unit(x, f.bound, x01);
    float feat[NGP_FEAT], h1[NGP_HID], h2[NGP_HID], out[NGP_OUT];
    ngp_encode(lv, f.table, x01, inside, feat);
    if (feat_out) {                                   // field cache for the backward (sf_ngp_render_forward): one 128-byte row per point
#pragma unroll
      for (int j = 0; j < NGP_FEAT / 4; ++j)
        *reinterpret_cast<float4*>(feat_out + (size_t)p * NGP_FEAT + 4 * j) = make_float4(feat[4 * j], feat[4 * j + 1], feat[4 * j + 2], feat[4 * j + 3]);
    }
    ngp_mlp_forward(W, feat, h1, h2, out);
    sigma[p] = expf(out[0] + ngp_blob(x));
    rgb[p * 3 + 0] = ngp_sigmoid(out[1]);
    rgb[p * 3 + 1] = ngp_sigmoid(out[2]);
    rgb[p * 3 + 2] = ngp_sigmoid(out[3]);
  }
}

// thread per ray; dynamic LDS = 2 * T * 64 floats (cdf, bins columns)
__global__ __launch_bounds__(64) void k_ngp_sample_fine(
    const float* __restrict__ z_c, const float* __restrict__ sig_c, const float* __restrict__ u,
    uint32_t u_row_stride, const float* __restrict__ nears, const float* __restrict__ fars, uint32_t N,
    uint32_t T, float* __restrict__ z_f) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const uint32_t n = blockIdx.x * 64 + threadIdx.x;
  if (n >= N) return;
  SfCol cdf{smem + threadIdx.x, 64};
  SfCol bins{smem + (size_t)T * 64 + threadIdx.x, 64};
  ngp_sample_fine(z_c + (size_t)n * T, sig_c + (size_t)n * T, u + (size_t)n * u_row_stride, nears[n], fars[n], T,
                  cdf, bins, z_f + (size_t)n * T);
}

// ---------------------------------------------------------------------------
// Table-gradient scatter.  Device-memory fp32 atomics execute at the memory side on MI355X (the 8 XCD
// L2s are not coherent): ~17 G scattered lane-ops/s, so the scatter is bound by the NUMBER of atomics.
// One workgroup owns a tile of 64 rays (an 8x8 image patch when the ray layout is known) = 8192 sorted
// samples, and walks the levels one at a time:
//   * coarse levels (cell larger than the patch footprint): contributions are first summed in an LDS
//     direct-mapped cache keyed by table row (LDS atomics are ~free), then each touched row is flushed
//     with ONE pair of global atomics per workgroup -- >95 % fewer global atomics on levels 0-5;
//   * fine levels: direct global atomics, z-corner pairs merged on z-dropped levels (ngp_scatter rule).
// ---------------------------------------------------------------------------
#define SC_RAYS 64
#define SC_SLOTS 4096
#define SC_EMPTY 0xffffffffu
// NT threads share one 80 KB LDS cache (r04: 4096 slots of doubles; r02: 8192 of floats) (one workgroup per CU): NT = 1024 puts 4 waves on every SIMD -- the loop body is a
// chain of dependent global loads, hash arithmetic and LDS atomics, and with the first version's 256 threads (ONE wave per
// SIMD) every one of those latencies was exposed.  `last_level`: levels [0, last_level) are walked by this kernel.
template <int NT>
__global__ __launch_bounds__(NT) void k_ngp_scatter(
    NgpLevels lv, float bound, float* __restrict__ gtable, const float* __restrict__ rays_o,
    const float* __restrict__ rays_d, const float* __restrict__ aabb, const float* __restrict__ z_s,
    const float* __restrict__ dfeat, uint32_t N, uint32_t T2, uint32_t rays_per_row, uint32_t cached_levels,
    uint32_t last_level, uint32_t sc_run) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  uint32_t* tags = reinterpret_cast<uint32_t*>(smem);            // [SC_SLOTS]
  double* vals = reinterpret_cast<double*>(smem + SC_SLOTS);     // [SC_SLOTS][2], doubles: ds_add_f64 is 7x the rate of ds_add_f32 (sf_dev.h)
  const uint32_t P = N * T2;
  float box[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) box[i] = aabb[i];
  // tile -> ray ids: 8x8 patch of the image if the layout is known, else 64 consecutive rays
  uint32_t tile_x = 0, tile_y = 0, tiles_x = 0;
  const bool patch = rays_per_row >= 8 && (rays_per_row % 8) == 0 && (N % rays_per_row) == 0 && ((N / rays_per_row) % 8) == 0;
  if (patch) { tiles_x = rays_per_row / 8; tile_y = blockIdx.x / tiles_x; tile_x = blockIdx.x % tiles_x; }

  for (uint32_t l = 0; l < last_level; ++l) {
    const bool cached = l < cached_levels;
    if (cached) {
      for (uint32_t s = threadIdx.x; s < SC_SLOTS; s += NT) { tags[s] = SC_EMPTY; vals[2 * s] = 0.0; vals[2 * s + 1] = 0.0; }
      __syncthreads();
    }
    float* tab = gtable + (size_t)lv.offset[l] * 2;
    const uint32_t step = lv.resolution[l] + 1;
    const bool z_dropped = lv.gridtype == 1 && (uint64_t)step * step > lv.hsize[l] && step <= lv.hsize[l];
    // lane quads share a sample: lane&1 = channel, lane&2 = x-corner.  The four adds of an x-corner pair (two
    // adjacent table rows x 2 channels = 16 contiguous bytes) sit in adjacent lanes of ONE atomic instruction: the
    // memory-side atomic unit merges lanes of one granule (measured 13.5 -> 7.6 ms with channel pairs alone).
    // A thread-item = (ray, run of `sc_run` consecutive sorted samples, lane of the quad).  Consecutive samples of a ray share
    // their cell on the coarse levels, so equal rows are first summed in REGISTERS (run-length merge per corner pair) and
    // only the run totals go to the LDS cache: on levels 0-5 nearly every lane of every atomic instruction used to hit
    // the same handful of LDS addresses, and same-address LDS atomics serialise (0.18 ms per level for ~1 us of work).
    const uint32_t runs = (T2 + sc_run - 1) / sc_run;
    auto flush = [&](uint32_t row, float v, uint32_t ch) {
      if (cached) {
        const uint32_t slot = (row * 2654435761u) >> 20;         // 12 bits -> SC_SLOTS
        const uint32_t prev = atomicCAS(&tags[slot], SC_EMPTY, row);
        if (prev == SC_EMPTY || prev == row) {
          sf_lds_add_f64(&vals[2 * slot + ch], (double)v);
          return;
        }
      }
      SF_ATOMIC_ADD(tab + (size_t)row * 2 + ch, v);
    };
    for (uint32_t it = threadIdx.x; it < 4 * SC_RAYS * runs; it += NT) {
      const uint32_t q = it >> 2, ch = it & 1, xb = (it >> 1) & 1;
      const uint32_t r = q / runs, k0 = (q - r * runs) * sc_run;
      uint32_t n = patch ? ((tile_y * 8 + (r >> 3)) * rays_per_row + tile_x * 8 + (r & 7)) : (blockIdx.x * SC_RAYS + r);
      if (n >= N) continue;
      const float o[3] = {rays_o[n * 3], rays_o[n * 3 + 1], rays_o[n * 3 + 2]};
      const float d[3] = {rays_d[n * 3], rays_d[n * 3 + 1], rays_d[n * 3 + 2]};
      uint32_t prow[4] = {SC_EMPTY, SC_EMPTY, SC_EMPTY, SC_EMPTY};
      float pacc[4] = {0.f, 0.f, 0.f, 0.f};
      const uint32_t k1 = k0 + sc_run < T2 ? k0 + sc_run : T2;
      // (r04: the run walked in batches of 4 / 8 / 16 samples with their gradient and depth loads issued together measured SLOWER,
      // 3.76 / 3.76 / 3.86 against 3.55 ms render fwd + bwd, profiles/r04_ngp_binned_tuning.log: 128 VGPRs at 1024 threads, and the
      // kernel runs beside the last chunk's bin + reduce, whose memory traffic it then competes with)
      for (uint32_t k = k0; k < k1; ++k) {
        const uint32_t p = n * T2 + k;
        const float dfc = dfeat[((size_t)l * P + p) * 2 + ch];
        if (dfc == 0.0f) continue;                               // outside points / dead samples contribute nothing
        float x[3], x01[3];
        ngp_point(o, d, z_s[p], box, x);
        if (!ngp_unit(x, bound, x01)) continue;
        NgpCell c;
        ngp_cell(lv, l, x01, c);
#pragma unroll
        for (int yz = 0; yz < 4; ++yz) {
          if (z_dropped && yz >= 2) continue;
          const int i0 = yz << 1, i1 = i0 | 1;        // the two x-corners of this (y, z) corner pair
          const float w0 = z_dropped ? SF_ADD(c.w[i0 & 3], c.w[(i0 & 3) + 4]) : c.w[i0];
          const float w1 = z_dropped ? SF_ADD(c.w[i1 & 3], c.w[(i1 & 3) + 4]) : c.w[i1];
          const float w = xb ? w1 : w0;
          const uint32_t row = xb ? c.row[i1] : c.row[i0];
          const float v = SF_MUL(w, dfc);
          if (row == prow[yz]) {
            pacc[yz] += v;
          } else {
            if (prow[yz] != SC_EMPTY) flush(prow[yz], pacc[yz], ch);
            prow[yz] = row;
            pacc[yz] = v;
          }
        }
      }
#pragma unroll
      for (int yz = 0; yz < 4; ++yz)
        if (prow[yz] != SC_EMPTY) flush(prow[yz], pacc[yz], ch);
    }
    if (cached) {
      __syncthreads();
      for (uint32_t s = threadIdx.x; s < SC_SLOTS; s += NT) {
        const uint32_t row = tags[s];
        if (row != SC_EMPTY) {
          SF_ATOMIC_ADD(tab + (size_t)row * 2, (float)vals[2 * s]);
          SF_ATOMIC_ADD(tab + (size_t)row * 2 + 1, (float)vals[2 * s + 1]);
        }
      }
      __syncthreads();
    }
  }
}

// Fine levels (cell smaller than the patch footprint: nothing to merge in LDS): no LDS frame, so 8 waves per SIMD hide the
// latency of the loads in front of every atomic.  Work item = (level, sample) in level-major order (the dfeat layout),
// lane quads as above: 16 contiguous bytes per atomic instruction and quad.
__global__ __launch_bounds__(256) void k_ngp_scatter_fine(
    NgpLevels lv, float bound, float* __restrict__ gtable, const float* __restrict__ rays_o,
    const float* __restrict__ rays_d, const float* __restrict__ aabb, const float* __restrict__ z_s,
    const float* __restrict__ dfeat, uint32_t N, uint32_t T2, uint32_t first_level, uint32_t P_stride, uint32_t p_off) {
  // this launch walks the N rays behind the (pre-offset) ray / sample pointers = points [p_off, p_off + N * T2) of a set of
  // P_stride points, whose level-major layout dfeat keeps
  const uint32_t P = N * T2;
  float box[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) box[i] = aabb[i];
  const uint64_t total = (uint64_t)4 * P * (lv.L - first_level);
  for (uint64_t it = (uint64_t)blockIdx.x * 256 + threadIdx.x; it < total; it += (uint64_t)gridDim.x * 256) {
    const uint32_t ch = (uint32_t)it & 1, xb = ((uint32_t)it >> 1) & 1;
    const uint64_t q = it >> 2;
    const uint32_t l = first_level + (uint32_t)(q / P), p = (uint32_t)(q % P);
    const float dfc = dfeat[((size_t)l * P_stride + p_off + p) * 2 + ch];
    if (dfc == 0.0f) continue;
    const uint32_t n = p / T2;
    const float o[3] = {rays_o[n * 3], rays_o[n * 3 + 1], rays_o[n * 3 + 2]};
    const float d[3] = {rays_d[n * 3], rays_d[n * 3 + 1], rays_d[n * 3 + 2]};
    float x[3], x01[3];
    ngp_point(o, d, z_s[p], box, x);
    if (!ngp_unit(x, bound, x01)) continue;
    NgpCell c;
    ngp_cell(lv, l, x01, c);
    float* tab = gtable + (size_t)lv.offset[l] * 2;
    const uint32_t step = lv.resolution[l] + 1;
    const bool z_dropped = lv.gridtype == 1 && (uint64_t)step * step > lv.hsize[l] && step <= lv.hsize[l];
#pragma unroll
    for (int yz = 0; yz < 4; ++yz) {
      if (z_dropped && yz >= 2) continue;
      const int i0 = yz << 1, i1 = i0 | 1;
      const float w0 = z_dropped ? SF_ADD(c.w[i0 & 3], c.w[(i0 & 3) + 4]) : c.w[i0];
      const float w1 = z_dropped ? SF_ADD(c.w[i1 & 3], c.w[(i1 & 3) + 4]) : c.w[i1];
      const float w = xb ? w1 : w0;
      const uint32_t row = xb ? c.row[i1] : c.row[i0];
      SF_ATOMIC_ADD(tab + (size_t)row * 2 + ch, SF_MUL(w, dfc));
    }
  }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
int sf_ngp_make_levels(const sf_ngp_field* f, NgpLevels* out, hipStream_t st) {
  if (!f || !f->h_offsets) SF_FAIL(SF_ERR_INVALID, "ngp: field/h_offsets must be given");
  if (f->L > NGP_MAX_LEVELS) SF_FAIL(SF_ERR_INVALID, "ngp: at most %d levels", NGP_MAX_LEVELS);
  struct { float scale[32]; uint32_t resolution[32]; uint32_t offset[32]; uint32_t hsize[32]; } tmp;
  if (int rc = sf_fill_levels(reinterpret_cast<GridLevels*>(&tmp), nullptr, f->h_offsets, f->L, f->S, f->H, st)) return rc;
  for (uint32_t l = 0; l < NGP_MAX_LEVELS; ++l) {
    const bool on = l < f->L;
    out->scale[l] = on ? tmp.scale[l] : 0.f;
    out->resolution[l] = on ? tmp.resolution[l] : 1;
    out->offset[l] = on ? tmp.offset[l] : 0;
    out->hsize[l] = on ? tmp.hsize[l] : 1;
  }
  out->L = f->L;
  out->gridtype = f->gridtype;
  return SF_OK;
}

static FieldPtrs field_ptrs(const sf_ngp_field* f) { return sf_ngp_field_ptrs(f); }

extern "C" int sf_ngp_density(const sf_ngp_field* f, const float* xyz, uint32_t P, float* sigma, float* albedo,
                              void* stream) {
  hipStream_t st = (hipStream_t)stream;
  NgpLevels lv;
  if (int rc = sf_ngp_make_levels(f, &lv, st)) return rc;
  if (P == 0) return SF_OK;
  k_ngp_field<2><<<sf_grid_cap(sf_div_up(P, 256)), 256, 0, st>>>(field_ptrs(f), lv, nullptr, nullptr, nullptr, nullptr,
                                                                 nullptr, nullptr, nullptr, nullptr, xyz, P, 1, nullptr,
                                                                 sigma, albedo);
  SF_CHECK_LAUNCH("ngp_density");
  return SF_OK;
}

// workspace (floats): forward z_c, sig_c [N*T]; rgb_c [3NT]; z_f, sig_f [N*T]; rgb_f [3NT]  -> 10*N*T
// backward: dsig [2NT] + drgb [6NT] + d(features) level-major [16][2NT][2] = 72*N*T, then the binned scatter's cursors and entries.
// the backward cuts the rays into chunks (pipeline below): as many as keep a chunk a multiple of 256 rays and >= 2048 rays
#ifndef NGP_BWD_CHUNKS
#define NGP_BWD_CHUNKS 2u       // r03 (atomic scatters): 1 / 2 / 4 / 8 chunks = 5.96 / 5.65 / 5.64 / 6.00 ms; r04 (binned): 1 / 2 / 4 / 8 = 3.71 / 3.55 / 3.68 / 4.11 ms
#endif
struct NgpChunks { uint32_t n, max_rays, start[66]; };      // chunk c = rays [start[c], start[c + 1])
static NgpChunks ngp_bwd_plan(uint32_t N) {
  NgpChunks c{};
  auto equal = [&](uint32_t k) {
    c.n = k; c.max_rays = N / k;
    for (uint32_t i = 0; i <= k; ++i) c.start[i] = i * (N / k);
  };
  auto ok = [&](uint32_t rays, uint32_t k) { return k >= 1 && rays % k == 0 && (rays / k) % 256 == 0 && rays / k >= 2048; };
  // (a small first chunk, so that the side stream starts early, measured slower: first chunk 1/4, 1/8, 1/16 of the rays = 3.63 / 3.63 /
  // 3.66 ms against 3.54 for two equal chunks, profiles/r04_ngp_binned_tuning.log -- co-running kernels cost their sum, not their maximum)
  // large ray sets: chunks of at most 8192 rays, so that the bins of one chunk (the workspace) stay at 1.6 GB
  for (uint32_t k = (N + 8191) / 8192; k > NGP_BWD_CHUNKS && k <= 64; ++k)
    if (ok(N, k)) { equal(k); return c; }
  uint32_t k = NGP_BWD_CHUNKS;
  while (k > 1 && !ok(N, k)) --k;
  if (N / k > 8192) {                   // (r06, ADVICE r05: also when an equal split exists but is coarser than 8192 rays -- N = 51 712 = 2 x 25 856)
    // r05: a ray count with no equal split (N = 200^2, 65 535, ...) used to fall back to ONE chunk, and the bins -- sized by
    // the largest chunk -- then grew with all rays (7.9 GB at 40 000 rays, 13 GB at 65 535).  Unequal chunks: as many chunks of `per`
    // rays (a multiple of 256, 8192 unless that needs more than 64 chunks) as fit, and the remainder as the last one; the kernels take
    // any (offset, count) -- only the bins' size depends on max_rays.
    uint32_t per = 8192;
    if ((N + per - 1) / per > 64) per = (((N + 63) / 64 + 255) / 256) * 256;
    c.n = (N + per - 1) / per; c.max_rays = per;
    for (uint32_t i = 0; i < c.n; ++i) c.start[i] = i * per;
    c.start[c.n] = N;
    return c;
  }
  equal(k);
  return c;
}
// binned scatter (ngp_scatter_bin.h): cursors + 16-byte entries of ONE chunk: 4 corner pairs x every level + slack per bucket
#define SB_CURSOR_WORDS (NGP_MAX_LEVELS * SB_MAX_BUCKETS)
#ifndef SB_LEVELS_BUDGET
#define SB_LEVELS_BUDGET 24                       // entries for 24 level-slots per sample (12 levels are binned): a bucket holds 2 x its uniform
#endif                                            // share (max / mean of the bucket loads on the reference scene: <= 1.9; 16 / 24 / 32 slots: 3.61 / 3.55 / 3.58 ms)
static uint64_t ngp_bin_entries(uint32_t N, uint32_t T) {
  return (uint64_t)ngp_bwd_plan(N).max_rays * 2 * T * 4 * SB_LEVELS_BUDGET + (uint64_t)SB_CURSOR_WORDS * 64;
}
extern "C" uint64_t sf_ngp_render_workspace_bytes(uint32_t N, uint32_t T) {
  return (uint64_t)(8 + 4 * NGP_MAX_LEVELS) * N * T * sizeof(float) + (uint64_t)SB_CURSOR_WORDS * 4 + ngp_bin_entries(N, T) * 16;
}

extern "C" uint64_t sf_ngp_render_forward_workspace_bytes(uint32_t N, uint32_t T) { return (uint64_t)10 * N * T * sizeof(float); }

extern "C" uint64_t sf_ngp_render_cache_bytes(uint32_t N, uint32_t T) {
  return ((uint64_t)2 * N * T * NGP_FEAT + (uint64_t)N * 2 * T) * sizeof(float);
}

extern "C" int sf_ngp_render_forward(const sf_ngp_field* f, const float* rays_o, const float* rays_d,
                                     const float* aabb, uint32_t N, uint32_t T, float min_near,
                                     const float* lin, const float* u_coarse, const float* u_fine,
                                     uint32_t u_fine_row_stride, float bg_color, float* nears, float* fars,
                                     float* z_sorted, float* sigma_s, float* rgb_s, float* image, float* depth,
                                     float* weights_sum, float* field_cache, float* workspace, uint64_t workspace_bytes,
                                     void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (N == 0) return SF_OK;
  if (T < 4 || T > 64) SF_FAIL(SF_ERR_INVALID, "ngp_render: T must be in [4,64]");
  if (workspace_bytes < sf_ngp_render_forward_workspace_bytes(N, T)) SF_FAIL(SF_ERR_INVALID, "ngp_render: workspace too small");
  if (!lin || !u_fine) SF_FAIL(SF_ERR_INVALID, "ngp_render: lin and u_fine tables are required");
  NgpLevels lv;
  if (int rc = sf_ngp_make_levels(f, &lv, st)) return rc;
  if (int rc = sf_near_far_from_aabb(rays_o, rays_d, aabb, N, min_near, nears, fars, stream)) return rc;
  const uint64_t NT = (uint64_t)N * T;
  float* z_c = workspace;           float* sig_c = z_c + NT;     float* rgb_c = sig_c + NT;
  float* z_f = rgb_c + 3 * NT;      float* sig_f = z_f + NT;     float* rgb_f = sig_f + NT;
  const FieldPtrs fp = field_ptrs(f);
  const uint32_t P = (uint32_t)NT;
  const uint32_t gridp = sf_grid_cap(sf_div_up(P, 256));
  // field cache (or NULL): [N*T][32] coarse features | [N*T][32] fine features | [N][2T] u32 sort permutation
  float* feat_c = field_cache;
  float* feat_f = field_cache ? field_cache + NT * NGP_FEAT : nullptr;
  uint32_t* perm = field_cache ? reinterpret_cast<uint32_t*>(field_cache + 2 * NT * NGP_FEAT) : nullptr;
  k_ngp_field<0><<<gridp, 256, 0, st>>>(fp, lv, rays_o, rays_d, aabb, nears, fars, lin, u_coarse, nullptr, nullptr, P, T,
                                        z_c, sig_c, rgb_c, feat_c);
  SF_CHECK_LAUNCH("ngp_field_coarse");
  const uint32_t gridr = sf_div_up(N, 64);
  k_ngp_sample_fine<<<gridr, 64, 2 * T * 64 * sizeof(float), st>>>(z_c, sig_c, u_fine, u_fine_row_stride, nears, fars,
                                                                  N, T, z_f);
  SF_CHECK_LAUNCH("ngp_sample_fine");
  k_ngp_field<1><<<gridp, 256, 0, st>>>(fp, lv, rays_o, rays_d, aabb, nears, fars, nullptr, nullptr, z_f, nullptr, P, T,
                                        nullptr, sig_f, rgb_f, feat_f);
  SF_CHECK_LAUNCH("ngp_field_fine");
  // one WAVE per ray: rank sort of cat([coarse, fine]) by readlane keys + scans (ngp_composite_wave.h)
  k_ngp_composite_wave<<<sf_div_up(N, 4), 256, 4 * 5 * 2 * T * sizeof(float), st>>>(
      CompositeArgs{z_c, sig_c, rgb_c, z_f, sig_f, rgb_f, nears, fars, N, T, bg_color, z_sorted, sigma_s, rgb_s, image, depth,
                    weights_sum, perm});
  SF_CHECK_LAUNCH("ngp_composite");
  return SF_OK;
}

extern "C" int sf_ngp_render_backward(const sf_ngp_field* f, const sf_ngp_field_grad* g, const float* rays_o,
                                      const float* rays_d, const float* aabb, uint32_t N, uint32_t T,
                                      const float* nears, const float* fars, const float* z_sorted,
                                      const float* sigma_s, const float* rgb_s, float bg_color,
                                      const float* grad_image, const float* grad_weights_sum, uint32_t rays_per_row,
                                      const float* field_cache, float* workspace, uint64_t workspace_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (N == 0) return SF_OK;
  if (T < 4 || T > 64) SF_FAIL(SF_ERR_INVALID, "ngp_render: T must be in [4,64]");
  if (workspace_bytes < sf_ngp_render_workspace_bytes(N, T)) SF_FAIL(SF_ERR_INVALID, "ngp_render: workspace too small");
  if (!grad_image) SF_FAIL(SF_ERR_INVALID, "ngp_render_backward: grad_image required");
  NgpLevels lv;
  if (int rc = sf_ngp_make_levels(f, &lv, st)) return rc;
  const uint64_t M = (uint64_t)N * 2 * T;
  float* dsig = workspace;
  float* drgb = dsig + M;
  // wave-per-ray backward: the forward's scans run in reverse (suffix sums in double)
  k_ngp_composite_bwd_wave<<<sf_div_up(N, 4), 256, 0, st>>>(
      CompositeBwdArgs{z_sorted, sigma_s, rgb_s, nears, fars, N, T, bg_color, grad_image, grad_weights_sum, dsig, drgb});
  SF_CHECK_LAUNCH("ngp_composite_bwd");
  int dev_id = 0;                                  // raised dynamic-LDS limits are per-device function attributes
  if (hipGetDevice(&dev_id) != hipSuccess) SF_FAIL(SF_ERR_LAUNCH, "hipGetDevice failed");
  float* dfeat = g->g_embeddings ? drgb + 3 * M : nullptr;        // NULL table gradient = table frozen
  const size_t lds2 = (size_t)FB_LDS_FLOATS * sizeof(float);
  const size_t lds_sc = (size_t)SC_SLOTS * (sizeof(uint32_t) + 2 * sizeof(double));
  static unsigned attr_mask = 0;
  if (dev_id >= 32 || !(attr_mask & (1u << dev_id))) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ngp_field_bwd_mfma), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ngp_scatter<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sc) != hipSuccess)
      SF_FAIL(SF_ERR_LAUNCH, "ngp_render_backward: cannot raise the dynamic LDS limits");
    if (dev_id < 32) attr_mask |= 1u << dev_id;
  }
  // scatter shape, measured on MI355X in r02 (the A/B environment switches were retired in r04): 1024 threads per cached-level
  // workgroup (256 / 512 / 1024: 8.55 / - / 7.84 ms render fwd+bwd), LDS cache up to scale 640 (160 / 320 / 640 / all levels: 7.82 /
  // 7.63 / 7.51 / 8.27 ms), the finer levels in their own high-occupancy launch, 32 sorted samples merged per thread before the cache
  // (4 / 8 / 16 / 32: 6.95 / 6.85 / 6.81 / 6.76 ms)
  constexpr float sc_cutoff = 640.0f;
#ifndef SC_RUN
#define SC_RUN 32u
#endif
  constexpr uint32_t sc_run = SC_RUN;
  // levels up to scale ~640 profit from the LDS cache (measured r02: cut-off 160 / 320 / 640 / none = 7.82 / 7.63 / 7.51 / 8.27 ms render fwd+bwd)
  uint32_t cached = 0;
  while (cached < lv.L && lv.scale[cached] <= sc_cutoff) ++cached;
  // r04: the levels whose cell is smaller than a ray patch's footprint (no merging in the LDS cache) are BINNED by table slice and
  // reduced in LDS instead of sending one device atomic per corner (ngp_scatter_bin.h); tables beyond 2^22 rows keep the atomics
  // (measured r04, render fwd + bwd: atomics 5.02 ms; binned from scale > 45 / 60 / 80 / 112 / 160: 4.97 / 4.45 / 4.50 / 4.43 / 4.58 ms with
  // one entry per corner, 60 / 112: 4.20 / 4.21 ms with pair entries, profiles/r04_ngp_binned_scatter_ab.log; the A/B switches are
  // retired: the dense levels keep the LDS cache, the hashed ones are binned)
  constexpr float bin_cutoff = 60.0f;
  uint32_t first_bin = lv.L;
  uint32_t bucket0[NGP_MAX_LEVELS + 1] = {};
  if (dfeat) {
    first_bin = 0;
    while (first_bin < lv.L && lv.scale[first_bin] <= bin_cutoff) ++first_bin;
    uint32_t tb = 0;
    for (uint32_t l = 0; l <= NGP_MAX_LEVELS; ++l) {
      bucket0[l] = tb;
      if (l >= first_bin && l < lv.L) {
        const uint32_t nb = (lv.hsize[l] + SB_ROWS - 1) >> SB_ROWS_LOG;
        if (nb > SB_MAX_BUCKETS) { first_bin = lv.L; break; }
        tb += nb;
      }
    }
  }
  const bool binned = first_bin < lv.L;
  if (binned && cached > first_bin) cached = first_bin;
  const uint32_t last = !dfeat ? lv.L : binned ? first_bin : cached;     // levels [0, last): k_ngp_scatter, [last, L): binned / k_ngp_scatter_fine
  const uint32_t total_buckets = bucket0[NGP_MAX_LEVELS];
  uint32_t* bin_cursor = reinterpret_cast<uint32_t*>(workspace + (size_t)(8 + 4 * NGP_MAX_LEVELS) * N * T);
  f32x4* bin_ent = reinterpret_cast<f32x4*>(bin_cursor + SB_CURSOR_WORDS);      // (16-byte aligned: 72 N T floats + 2^13 words)
  const uint64_t bin_entries = ngp_bin_entries(N, T);
  const uint32_t bin_cap = binned ? (uint32_t)(bin_entries / total_buckets) : 0;
  if (binned && hipMemsetAsync(bin_cursor, 0, (size_t)total_buckets * 4, st) != hipSuccess) SF_FAIL(SF_ERR_LAUNCH, "ngp_render_backward: memset failed");

  // ---- the pipeline (r03).  Three kernels with three different bottlenecks: the field backward (fp32 MFMA + 113 KB of LDS,
  // one workgroup per CU), the fine-level scatter (memory-side atomic unit, no LDS, VALU idle) and the cached-level scatter
  // (LDS atomics, 96 KB).  The rays are cut into chunks; the fine-level scatter of chunk c runs on a side stream while the
  // field backward of chunk c + 1 occupies the CUs, and the last one runs beside the cached-level scatter (one launch over
  // all rays, its LDS cache wants every ray of an 8x8 patch).  dfeat keeps the level-major layout of the whole ray set.
  // (r03 A/B: 1 / 2 / 4 / 8 chunks 5.96 / 5.65 / 5.64 / 6.00 ms, no overlap 6.33 ms; the switches were retired in r04.)
  struct Side { hipStream_t s; hipEvent_t fork, join; };
  static Side side[32] = {};
  const bool fork = dfeat && last < lv.L && dev_id < 32;
  NgpChunks plan = ngp_bwd_plan(N);                                            // (the entries buffer is sized for this chunking)
  if (!(dfeat && last < lv.L)) { plan.n = 1; plan.start[0] = 0; plan.start[1] = N; }
  const uint32_t n_chunks = plan.n;
  // The side stream and its fork / join events are ONE set per device: two host threads (or two caller streams) rendering on the
  // same device would re-record each other's events, so the whole fork .. join section runs under the lock (host-side enqueue only:
  // microseconds).  The guard below joins the side stream into the caller's stream on EVERY exit path, error returns included.
  static std::mutex side_mu;
  std::unique_lock<std::mutex> side_lk(side_mu, std::defer_lock);
  struct SideJoin {
    Side* sd = nullptr; hipStream_t st = nullptr; bool forked = false;
    int join() { const bool f = forked; forked = false; return (!f || (hipEventRecord(sd->join, sd->s) == hipSuccess && hipStreamWaitEvent(st, sd->join, 0) == hipSuccess)) ? 0 : 1; }
    ~SideJoin() { (void)join(); }
  } side_join;
  if (fork) {
    side_lk.lock();
    Side& sd = side[dev_id];
    if (!sd.s && (hipStreamCreateWithFlags(&sd.s, hipStreamNonBlocking) != hipSuccess ||
                  hipEventCreateWithFlags(&sd.fork, hipEventDisableTiming) != hipSuccess ||
                  hipEventCreateWithFlags(&sd.join, hipEventDisableTiming) != hipSuccess))
      SF_FAIL(SF_ERR_LAUNCH, "ngp_render_backward: cannot create the side stream");
    side_join.sd = &sd; side_join.st = st;
  }
  const uint32_t T2 = 2 * T;
  for (uint32_t c = 0; c < n_chunks; ++c) {
    const uint32_t n0 = plan.start[c], Nc = plan.start[c + 1] - n0;
    const uint32_t Pc = Nc * T2, P0 = n0 * T2;
    // matrix-core field backward of chunk c: one wave per 32 points and trip, six fp32 GEMMs on v_mfma_f32_16x16x4_f32 (ngp_bwd_mfma.h)
    FBArgs a{};
    a.table = f->embeddings; a.w0 = f->w0; a.b0 = f->b0; a.w1 = f->w1; a.b1 = f->b1; a.w2 = f->w2; a.b2 = f->b2; a.bound = f->bound;
    a.g_w0 = g->g_w0; a.g_b0 = g->g_b0; a.g_w1 = g->g_w1; a.g_b1 = g->g_b1; a.g_w2 = g->g_w2; a.g_b2 = g->g_b2;
    a.lv = lv;
    a.rays_o = rays_o + (size_t)n0 * 3; a.rays_d = rays_d + (size_t)n0 * 3; a.aabb = aabb;
    a.z_s = z_sorted + (size_t)P0; a.dsig = dsig + (size_t)P0; a.drgb = drgb + (size_t)P0 * 3; a.dfeat_out = dfeat;
    a.P = Pc; a.T2 = T2; a.dfeat_P = (uint32_t)M; a.p_off = P0;
    a.feat_c = field_cache;                           // (or all three null: the kernel re-gathers the features)
    a.feat_f = field_cache ? field_cache + (size_t)N * T * NGP_FEAT : nullptr;
    a.perm = field_cache ? reinterpret_cast<const uint32_t*>(field_cache + 2 * (size_t)N * T * NGP_FEAT) : nullptr;
    const uint32_t trips = sf_div_up(Pc, FB_PTS);
    const uint32_t grid = trips < 1024 ? sf_div_up(trips, 4) : 256;   // one resident workgroup per CU (LDS-bound), 4 waves each
    k_ngp_field_bwd_mfma<<<grid, 256, lds2, st>>>(a);
    SF_CHECK_LAUNCH("ngp_field_bwd_mfma");
    if (dfeat && last < lv.L) {
      hipStream_t sf = st;
      if (fork) {
        Side& sd = side[dev_id];
        if (hipEventRecord(sd.fork, st) != hipSuccess || hipStreamWaitEvent(sd.s, sd.fork, 0) != hipSuccess)
          SF_FAIL(SF_ERR_LAUNCH, "ngp_render_backward: fork failed");
        sf = sd.s;
        side_join.forked = true;
      }
      if (binned) {
        SBArgs b{};
        b.lv = lv; b.bound = f->bound; b.rays_o = a.rays_o; b.rays_d = a.rays_d; b.aabb = aabb; b.z_s = a.z_s; b.dfeat = dfeat;
        b.gtable = g->g_embeddings; b.cursor = bin_cursor; b.ent = bin_ent;
        b.P = Pc; b.T2 = T2; b.first_level = first_bin; b.P_stride = (uint32_t)M; b.p_off = P0; b.cap = bin_cap;
        SBRArgs r{};
        r.lv = lv; r.gtable = g->g_embeddings; r.cursor = bin_cursor; r.ent = bin_ent; r.first_level = first_bin; r.cap = bin_cap;
        for (uint32_t l = 0; l <= NGP_MAX_LEVELS; ++l) { b.bucket0[l] = bucket0[l]; r.bucket0[l] = bucket0[l]; }
        const uint32_t tiles = sf_div_up(Pc, SB_THREADS);
        k_ngp_bin<<<tiles < 1024 ? tiles : 1024, SB_THREADS, 0, sf>>>(b);
        SF_CHECK_LAUNCH("ngp_bin");
        k_ngp_bin_reduce<<<total_buckets, SBR_THREADS, 0, sf>>>(r);           // leaves the cursors at zero for the next chunk
        SF_CHECK_LAUNCH("ngp_bin_reduce");
      } else {
        const uint64_t items = (uint64_t)4 * Pc * (lv.L - last);
        const uint32_t grid_f = (uint32_t)(items / 256 < 16384 ? (items + 255) / 256 : 16384);
        k_ngp_scatter_fine<<<grid_f, 256, 0, sf>>>(lv, f->bound, g->g_embeddings, a.rays_o, a.rays_d, aabb, a.z_s, dfeat, Nc, T2, last,
                                                   (uint32_t)M, P0);
        SF_CHECK_LAUNCH("ngp_scatter_fine");
      }
    }
  }
  if (dfeat && last > 0) {
    const uint32_t grid_sc = sf_div_up(N, SC_RAYS);
    k_ngp_scatter<1024><<<grid_sc, 1024, lds_sc, st>>>(lv, f->bound, g->g_embeddings, rays_o, rays_d, aabb, z_sorted, dfeat, N, T2, rays_per_row, cached, last, sc_run);
    SF_CHECK_LAUNCH("ngp_scatter");
  }
  if (side_join.join()) SF_FAIL(SF_ERR_LAUNCH, "ngp_render_backward: join failed");
  return SF_OK;
}

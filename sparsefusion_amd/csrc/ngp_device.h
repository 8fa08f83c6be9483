// Per-thread building blocks of the fused Instant-NGP render for gfx950.
//
// What they compute (reference file:line):
//   field        external/nerf/network_grid.py:69-88 (common_forward), :14-33 (MLP),
//                external/gridencoder/src/gridencoder.cu:54-223 (encode), ngp_activation.py:10-23
//   coarse z     external/nerf/renderer_df.py:356-368
//   fine z       external/nerf/renderer_df.py:381-395 + sample_pdf :15-49
//   composite    external/nerf/renderer_df.py:404-456
// The functions are plain C++ over pointers (no wave intrinsics) so that the SAME source is
// compiled by hipcc for the kernels in ngp_render.hip and by g++ for the host emulation that
// tests/ uses to check the kernel logic on a GPU-less machine (tests/hostemu/).
//
// Arithmetic notes: expressions that PyTorch evaluates as separate elementwise ops are written
// with SF_MUL/SF_ADD/SF_SUB/SF_DIV (no FMA contraction) so that sample positions -- and
// therefore grid cell indices -- are bit-identical to the fp32 oracle; running products/sums
// that torch accumulates in double on the CPU (cumsum/cumprod) use double here as well.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define SF_HD __host__ __device__ __forceinline__
#define SF_HDM __host__ __device__ __forceinline__
#else
#define SF_HD static inline
#define SF_HDM inline
#endif

#if defined(__HIP_DEVICE_COMPILE__)
// r05: ROCm 7.2's __fmul_rn / __fadd_rn / __fsub_rn are PLAIN a * b, a + b, a - b (clang's __clang_hip_math.h without
// OCML_BASIC_ROUNDED_OPERATIONS) and hipcc's default -ffp-contract=fast-honor-pragmas fuses SF_ADD(x, SF_MUL(y, z)) into v_fmac_f32:
// r01-r04 believed these intrinsics were contraction barriers, and the coarse sample depths differed from the oracle's in the last
// place on 19 % of the samples (found by tests/test_gpu_ngp.py::test_render_sample_bookkeeping_vs_oracle).  The pragma is honoured
// per instruction and survives inlining (checked in the ISA: v_mul_f32 + v_add_f32).
__device__ __forceinline__ float sf_mul_rn(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ float sf_add_rn(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
}
__device__ __forceinline__ float sf_sub_rn(float a, float b) {
#pragma clang fp contract(off)
  return a - b;
}
#define SF_MUL(a, b) sf_mul_rn((a), (b))
#define SF_ADD(a, b) sf_add_rn((a), (b))
#define SF_SUB(a, b) sf_sub_rn((a), (b))
#define SF_DIV(a, b) __fdiv_rn((a), (b))
// fp32 add straight to the L2 atomic unit (global_atomic_add_f32).  The pointer is cast to the
// global address space explicitly: on a flat pointer hipcc (ROCm 7.2) emits an is_shared test per
// atomic and miscompiles it ("V_CMP_NE_U32 0, $src_shared_base: incorrect register class").
typedef __attribute__((address_space(1))) float sf_gfloat;
#define SF_ATOMIC_ADD(p, v) ((void)__builtin_amdgcn_global_atomic_fadd_f32((sf_gfloat*)(p), (v)))
#else   // host build: compiled with -ffp-contract=off
#define SF_MUL(a, b) ((a) * (b))
#define SF_ADD(a, b) ((a) + (b))
#define SF_SUB(a, b) ((a) - (b))
#define SF_DIV(a, b) ((a) / (b))
#define SF_ATOMIC_ADD(p, v) (*(p) += (v))
#endif

#define NGP_MAX_LEVELS 16
#define NGP_FEAT 32      // 16 levels x 2
#define NGP_HID 64
#define NGP_OUT 4

struct NgpLevels {       // host-computed level geometry (see gridencoder.hip sf_fill_levels)
  float scale[NGP_MAX_LEVELS];
  uint32_t resolution[NGP_MAX_LEVELS];
  uint32_t offset[NGP_MAX_LEVELS];
  uint32_t hsize[NGP_MAX_LEVELS];
  uint32_t L;
  uint32_t gridtype;
};

// Offsets (in floats) of the MLP parameters inside one packed weight block.
#define NGP_W0 0                                   // [64][32]
#define NGP_B0 (NGP_W0 + NGP_HID * NGP_FEAT)       // [64]
#define NGP_W1 (NGP_B0 + NGP_HID)                  // [64][64]
#define NGP_B1 (NGP_W1 + NGP_HID * NGP_HID)        // [64]
#define NGP_W2 (NGP_B1 + NGP_HID)                  // [4][64]
#define NGP_B2 (NGP_W2 + NGP_OUT * NGP_HID)        // [4]
#define NGP_WTOTAL (NGP_B2 + NGP_OUT)              // 6532 floats

SF_HD uint32_t ngp_row3(uint32_t gridtype, uint32_t hsize, uint32_t resolution, uint32_t px, uint32_t py,
                        uint32_t pz) {
  // get_grid_index for D=3, align_corners=false (gridencoder.cu:54-72), uint32 wrap-around kept.
  uint32_t stride = 1, index = 0;
  const uint32_t step = resolution + 1;
  if (stride <= hsize) { index += px * stride; stride *= step; }
  if (stride <= hsize) { index += py * stride; stride *= step; }
  if (stride <= hsize) { index += pz * stride; stride *= step; }
  if (gridtype == 0 && stride > hsize) index = (px * 1u) ^ (py * 2654435761u) ^ (pz * 805459861u);
  return index % hsize;
}

struct NgpCell {          // per (point, level) interpolation record
  uint32_t row[8];
  float w[8];
};

SF_HD void ngp_cell(const NgpLevels& lv, uint32_t level, const float x01[3], NgpCell& c) {
  const float scale = lv.scale[level];
  float pos[3];
  uint32_t pg[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    pos[d] = fmaf(x01[d], scale, 0.5f);
    pg[d] = (uint32_t)floorf(pos[d]);
    pos[d] -= (float)pg[d];
  }
#pragma unroll
  for (uint32_t idx = 0; idx < 8; ++idx) {
    float w = 1.0f;
    uint32_t p[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      if ((idx & (1u << d)) == 0) { w = SF_MUL(w, SF_SUB(1.0f, pos[d])); p[d] = pg[d]; }
      else                        { w = SF_MUL(w, pos[d]);               p[d] = pg[d] + 1; }
    }
    c.w[idx] = w;
    c.row[idx] = ngp_row3(lv.gridtype, lv.hsize[level], lv.resolution[level], p[0], p[1], p[2]);
  }
}

// World position -> unit cube (grid.py:142).  Returns false if outside [0,1] (features = 0).
SF_HD bool ngp_unit(const float x[3], float bound, float x01[3]) {
  bool ok = true;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    x01[d] = SF_DIV(SF_ADD(x[d], bound), SF_MUL(2.0f, bound));
    if (x01[d] < 0.0f || x01[d] > 1.0f) ok = false;
  }
  return ok;
}

SF_HD void ngp_encode(const NgpLevels& lv, const float* __restrict__ table, const float x01[3], bool inside,
                      float feat[NGP_FEAT]) {
#pragma unroll
  for (uint32_t l = 0; l < NGP_MAX_LEVELS; ++l) {
    float r0 = 0.0f, r1 = 0.0f;
    if (inside && l < lv.L) {
      NgpCell c;
      ngp_cell(lv, l, x01, c);
      const float* tab = table + (size_t)lv.offset[l] * 2;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float2 f = *reinterpret_cast<const float2*>(tab + (size_t)c.row[i] * 2);
        r0 = fmaf(c.w[i], f.x, r0);
        r1 = fmaf(c.w[i], f.y, r1);
      }
    }
    feat[2 * l] = r0;
    feat[2 * l + 1] = r1;
  }
}

// 32 -> 64 -> 64 -> 4 ReLU MLP.  `W` is the packed block (LDS on the GPU: every lane reads the
// same address, i.e. a broadcast read).  Keeps the post-ReLU hidden activations for backward.
SF_HD void ngp_mlp_forward(const float* __restrict__ W, const float feat[NGP_FEAT], float h1[NGP_HID],
                           float h2[NGP_HID], float out[NGP_OUT]) {
#pragma unroll
  for (int j = 0; j < NGP_HID; ++j) {
    float a = W[NGP_B0 + j];
#pragma unroll
    for (int k = 0; k < NGP_FEAT; ++k) a = fmaf(W[NGP_W0 + j * NGP_FEAT + k], feat[k], a);
    h1[j] = fmaxf(a, 0.0f);
  }
#pragma unroll
  for (int j = 0; j < NGP_HID; ++j) {
    float a = W[NGP_B1 + j];
#pragma unroll
    for (int k = 0; k < NGP_HID; ++k) a = fmaf(W[NGP_W1 + j * NGP_HID + k], h1[k], a);
    h2[j] = fmaxf(a, 0.0f);
  }
#pragma unroll
  for (int j = 0; j < NGP_OUT; ++j) {
    float a = W[NGP_B2 + j];
#pragma unroll
    for (int k = 0; k < NGP_HID; ++k) a = fmaf(W[NGP_W2 + j * NGP_HID + k], h2[k], a);
    out[j] = a;
  }
}

// density blob: 5 * exp(-|x|^2 / (2 * 0.2^2))   (network_grid.py:69-75)
SF_HD float ngp_blob(const float x[3]) {
  const float d = SF_ADD(SF_ADD(SF_MUL(x[0], x[0]), SF_MUL(x[1], x[1])), SF_MUL(x[2], x[2]));
  return SF_MUL(5.0f, expf(SF_DIV(-d, 0.08f)));
}

SF_HD float ngp_sigmoid(float v) { return 1.0f / (1.0f + expf(-v)); }

// sample position on a ray, clipped to the aabb (renderer_df.py:367-368)
SF_HD void ngp_point(const float o[3], const float d[3], float z, const float aabb[6], float x[3]) {
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    float v = SF_ADD(o[i], SF_MUL(d[i], z));
    v = fmaxf(v, aabb[i]);          // torch.max / torch.min element-wise
    v = fminf(v, aabb[3 + i]);
    x[i] = v;
  }
}

// stratified coarse depth (renderer_df.py:356-363); lin = linspace(0,1,T)[k], u in [0,1) or <0 for none
SF_HD float ngp_coarse_z(float near, float far, float lin, float u, uint32_t T) {
  const float span = SF_SUB(far, near);
  float z = SF_ADD(near, SF_MUL(span, lin));
  if (u >= 0.0f) {
    const float sample_dist = SF_DIV(span, (float)T);
    z = SF_ADD(z, SF_MUL(SF_SUB(u, 0.5f), sample_dist));
  }
  return z;
}

// Strided per-thread column view: element k of this thread lives at base[k * stride].
struct SfCol {
  float* base;
  uint32_t stride;
  SF_HDM float& operator[](uint32_t k) const { return base[(size_t)k * stride]; }
};
struct SfColC {
  const float* base;
  uint32_t stride;
  SF_HDM float operator[](uint32_t k) const { return base[(size_t)k * stride]; }
};

// Importance sampling of T fine depths from the coarse pass of ONE ray (renderer_df.py:381-393).
// z, sigma: coarse row [T]; u: [T] uniforms (the host passes linspace(.5/T, 1-.5/T, T) for det=True); cdf/bins: scratch
// columns of T entries; writes zf[T].
SF_HD void ngp_sample_fine(const float* z, const float* sigma, const float* u, float near, float far, uint32_t T,
                           const SfCol& cdf, const SfCol& bins, float* zf) {
  const float sample_dist = SF_DIV(SF_SUB(far, near), (float)T);
  // weights of the coarse pass (float alphas, double running product like torch CPU cumprod)
  double trans = 1.0;
  double wsum = 0.0;
  for (uint32_t k = 0; k < T; ++k) {
    const float delta = (k + 1 < T) ? SF_SUB(z[k + 1], z[k]) : sample_dist;
    const float alpha = SF_SUB(1.0f, expf(SF_MUL(-delta, sigma[k])));
    const float w = SF_MUL(alpha, (float)trans);
    trans *= (double)SF_ADD(SF_SUB(1.0f, alpha), 1e-15f);
    if (k + 1 < T) bins[k] = SF_ADD(z[k], SF_MUL(0.5f, delta));      // z_vals_mid, T-1 entries
    if (k >= 1 && k + 1 < T) {                                         // weights[:, 1:-1] + 1e-5
      const float wk = SF_ADD(w, 1e-5f);
      cdf[k] = wk;                                                     // stash pdf numerator at [1..T-2]
      wsum += (double)wk;
    }
  }
  const float total = (float)wsum;
  double run = 0.0;
  cdf[0] = 0.0f;
  for (uint32_t k = 1; k + 1 < T; ++k) {                               // cdf[k] = sum_{i<=k} pdf_i
    run += (double)SF_DIV(cdf[k], total);
    cdf[k] = (float)run;
  }
  const uint32_t nc = T - 1;                                           // cdf / bins entries: 0..T-2
  for (uint32_t j = 0; j < T; ++j) {
    const float uj = u[j];
    // searchsorted(cdf, u, right=True): number of entries <= u
    uint32_t lo = 0, hi = nc;
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      if (cdf[mid] <= uj) lo = mid + 1; else hi = mid;
    }
    const uint32_t below = lo > 0 ? lo - 1 : 0;
    const uint32_t above = lo < nc - 1 ? lo : nc - 1;
    const float c0 = cdf[below], c1 = cdf[above];
    float denom = SF_SUB(c1, c0);
    if (denom < 1e-5f) denom = 1.0f;
    const float t = SF_DIV(SF_SUB(uj, c0), denom);
    const float b0 = bins[below], b1 = bins[above];
    zf[j] = SF_ADD(b0, SF_MUL(t, SF_SUB(b1, b0)));
  }
}

// Merge the sorted coarse samples with the (unsorted) fine samples of ONE ray into depth order
// (torch.cat + torch.sort + gather, renderer_df.py:404-412) and alpha-composite (:414-456).
// key/ord: scratch columns of T entries.  Outputs sorted rows zs/sig_s [2T], rgb_s [2T*3] and
// the ray results.
struct NgpRayOut { float image[3]; float depth; float weights_sum; };

SF_HD void ngp_merge_composite(const float* zc, const float* sc, const float* rc, const float* zf, const float* sf,
                               const float* rf, float near, float far, uint32_t T, float bg,
                               const SfCol& key, const SfCol& ord, float* zs, float* sig_s, float* rgb_s,
                               NgpRayOut& out) {
  // stable insertion sort of the fine depths (key = z, ord = source index)
  for (uint32_t i = 0; i < T; ++i) {
    const float v = zf[i];
    uint32_t j = i;
    while (j > 0 && key[j - 1] > v) { key[j] = key[j - 1]; ord[j] = ord[j - 1]; --j; }
    key[j] = v;
    ord[j] = (float)i;
  }
  // two-way merge, coarse first on ties (stable w.r.t. cat([coarse, fine]))
  uint32_t a = 0, b = 0;
  for (uint32_t m = 0; m < 2 * T; ++m) {
    const bool take_c = (b >= T) || (a < T && zc[a] <= key[b]);
    if (take_c) {
      zs[m] = zc[a]; sig_s[m] = sc[a];
      rgb_s[m * 3 + 0] = rc[a * 3 + 0]; rgb_s[m * 3 + 1] = rc[a * 3 + 1]; rgb_s[m * 3 + 2] = rc[a * 3 + 2];
      ++a;
    } else {
      const uint32_t s = (uint32_t)ord[b];
      zs[m] = key[b]; sig_s[m] = sf[s];
      rgb_s[m * 3 + 0] = rf[s * 3 + 0]; rgb_s[m * 3 + 1] = rf[s * 3 + 1]; rgb_s[m * 3 + 2] = rf[s * 3 + 2];
      ++b;
    }
  }
  const float sample_dist = SF_DIV(SF_SUB(far, near), (float)T);
  const float span = SF_SUB(far, near);
  double trans = 1.0;
  float ws = 0.0f, dep = 0.0f, r = 0.0f, g = 0.0f, bl = 0.0f;
  for (uint32_t m = 0; m < 2 * T; ++m) {
    const float delta = (m + 1 < 2 * T) ? SF_SUB(zs[m + 1], zs[m]) : sample_dist;
    const float alpha = SF_SUB(1.0f, expf(SF_MUL(-delta, sig_s[m])));
    const float w = SF_MUL(alpha, (float)trans);
    trans *= (double)SF_ADD(SF_SUB(1.0f, alpha), 1e-15f);
    const float raw = SF_DIV(SF_SUB(zs[m], near), span);
    const float oz = (raw != raw) ? raw : fminf(fmaxf(raw, 0.0f), 1.0f);   // NaN (miss rays: 0/0) propagates like torch.clamp
    ws = SF_ADD(ws, w);
    dep = SF_ADD(dep, SF_MUL(w, oz));
    r = SF_ADD(r, SF_MUL(w, rgb_s[m * 3 + 0]));
    g = SF_ADD(g, SF_MUL(w, rgb_s[m * 3 + 1]));
    bl = SF_ADD(bl, SF_MUL(w, rgb_s[m * 3 + 2]));
  }
  const float rest = SF_MUL(SF_SUB(1.0f, ws), bg);
  out.image[0] = SF_ADD(r, rest); out.image[1] = SF_ADD(g, rest); out.image[2] = SF_ADD(bl, rest);
  out.depth = dep;
  out.weights_sum = ws;
}

// Backward of the composite for ONE ray: d(loss)/d(sigma_m), d(loss)/d(rgb_m) from
// gI = d(loss)/d(image), gW = d(loss)/d(weights_sum).  tr/wt: scratch columns of 2T entries.
SF_HD void ngp_composite_backward(const float* zs, const float* sig_s, const float* rgb_s, float near, float far,
                                  uint32_t T, float bg, const float gI[3], float gW, const SfCol& tr,
                                  const SfCol& wt, float* dsig, float* drgb) {
  const uint32_t M = 2 * T;
  const float sample_dist = SF_DIV(SF_SUB(far, near), (float)T);
  double trans = 1.0;
  for (uint32_t m = 0; m < M; ++m) {
    const float delta = (m + 1 < M) ? SF_SUB(zs[m + 1], zs[m]) : sample_dist;
    const float alpha = SF_SUB(1.0f, expf(SF_MUL(-delta, sig_s[m])));
    tr[m] = (float)trans;
    wt[m] = SF_MUL(alpha, (float)trans);
    trans *= (double)SF_ADD(SF_SUB(1.0f, alpha), 1e-15f);
  }
  const float gsum = gI[0] + gI[1] + gI[2];
  double suffix = 0.0;                                  // sum_{j>m} a_j w_j
  for (uint32_t mm = M; mm-- > 0;) {
    const float w = wt[mm];
    const float a = gI[0] * rgb_s[mm * 3 + 0] + gI[1] * rgb_s[mm * 3 + 1] + gI[2] * rgb_s[mm * 3 + 2] - bg * gsum + gW;
    const float delta = (mm + 1 < M) ? SF_SUB(zs[mm + 1], zs[mm]) : sample_dist;
    const float e = expf(SF_MUL(-delta, sig_s[mm]));    // 1 - alpha
    const float one_m = SF_ADD(SF_SUB(1.0f, SF_SUB(1.0f, e)), 1e-15f);
    const float dalpha = a * tr[mm] - (float)(suffix / (double)one_m);
    dsig[mm] = dalpha * delta * e;
    drgb[mm * 3 + 0] = w * gI[0];
    drgb[mm * 3 + 1] = w * gI[1];
    drgb[mm * 3 + 2] = w * gI[2];
    suffix += (double)a * (double)w;
  }
}

// MLP backward for ONE point: given d(out)[4] and the saved post-ReLU activations, produce the
// gradients w.r.t. the (post-mask) hidden pre-activations and the encoder features.
// `W` as in ngp_mlp_forward (transposed access = strided broadcast reads).
SF_HD void ngp_mlp_backward(const float* __restrict__ W, const float h1[NGP_HID], const float h2[NGP_HID],
                            const float dout[NGP_OUT], float dh2[NGP_HID], float dh1[NGP_HID],
                            float dfeat[NGP_FEAT]) {
#pragma unroll
  for (int k = 0; k < NGP_HID; ++k) {
    float a = 0.0f;
#pragma unroll
    for (int j = 0; j < NGP_OUT; ++j) a = fmaf(W[NGP_W2 + j * NGP_HID + k], dout[j], a);
    dh2[k] = h2[k] > 0.0f ? a : 0.0f;
  }
#pragma unroll
  for (int k = 0; k < NGP_HID; ++k) {
    float a = 0.0f;
#pragma unroll
    for (int j = 0; j < NGP_HID; ++j) a = fmaf(W[NGP_W1 + j * NGP_HID + k], dh2[j], a);
    dh1[k] = h1[k] > 0.0f ? a : 0.0f;
  }
#pragma unroll
  for (int k = 0; k < NGP_FEAT; ++k) {
    float a = 0.0f;
#pragma unroll
    for (int j = 0; j < NGP_HID; ++j) a = fmaf(W[NGP_W0 + j * NGP_FEAT + k], dh1[j], a);
    dfeat[k] = a;
  }
}

// Scatter d(loss)/d(features) of ONE point into the table gradient (kernel_grid_backward,
// gridencoder.cu:226-313) -- hardware fp32 atomics on the GPU.  On tiled levels whose (res+1)^2 already
// exceeds the level size the index rule drops z (gridencoder.cu:60), so the two z-corners of every (x,y)
// pair hit the SAME row with weights w_xy*(1-pz) and w_xy*pz: they are merged into one add of w_xy*g
// (identical sum, half the atomics on levels >= 7 of the reference geometry).
SF_HD void ngp_scatter(const NgpLevels& lv, float* __restrict__ gtable, const float x01[3], bool inside,
                       const float dfeat[NGP_FEAT]) {
  if (!inside) return;
#pragma unroll
  for (uint32_t l = 0; l < NGP_MAX_LEVELS; ++l) {
    if (l < lv.L) {
      NgpCell c;
      ngp_cell(lv, l, x01, c);
      float* tab = gtable + (size_t)lv.offset[l] * 2;
      const uint32_t step = lv.resolution[l] + 1;
      const bool z_dropped = lv.gridtype == 1 && (uint64_t)step * step > lv.hsize[l] && step <= lv.hsize[l];
      if (z_dropped) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {                  // corners i and i+4 differ only in z
          const float w = SF_ADD(c.w[i], c.w[i + 4]);
          SF_ATOMIC_ADD(tab + (size_t)c.row[i] * 2 + 0, SF_MUL(w, dfeat[2 * l]));
          SF_ATOMIC_ADD(tab + (size_t)c.row[i] * 2 + 1, SF_MUL(w, dfeat[2 * l + 1]));
        }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          SF_ATOMIC_ADD(tab + (size_t)c.row[i] * 2 + 0, SF_MUL(c.w[i], dfeat[2 * l]));
          SF_ATOMIC_ADD(tab + (size_t)c.row[i] * 2 + 1, SF_MUL(c.w[i], dfeat[2 * l + 1]));
        }
      }
    }
  }
}

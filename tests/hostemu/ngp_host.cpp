// Host emulation of the fused NGP render kernels: compiles the SAME per-thread device functions
// (sparsefusion_amd/csrc/ngp_device.h) with g++ and runs them thread by thread, so the kernel logic
// can be checked against the oracle on a machine without a GPU.  TEST INFRASTRUCTURE ONLY -- never
// loaded by the sparsefusion_amd package.
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
struct float2 { float x, y; };
#include "../../sparsefusion_amd/csrc/ngp_device.h"

static void fill_levels(NgpLevels* lv, const int32_t* h_offsets, uint32_t L, float S, uint32_t H, uint32_t gridtype) {
  memset(lv, 0, sizeof(*lv));
  for (uint32_t l = 0; l < NGP_MAX_LEVELS; ++l) {
    if (l < L) {
      const float scale = exp2f((float)l * S) * (float)H - 1.0f;
      lv->scale[l] = scale;
      lv->resolution[l] = (uint32_t)ceil(scale) + 1;
      lv->offset[l] = (uint32_t)h_offsets[l];
      lv->hsize[l] = (uint32_t)(h_offsets[l + 1] - h_offsets[l]);
    } else { lv->resolution[l] = 1; lv->hsize[l] = 1; }
  }
  lv->L = L; lv->gridtype = gridtype;
}

static void pack_weights(std::vector<float>& W, const float* w0, const float* b0, const float* w1, const float* b1,
                         const float* w2, const float* b2) {
  W.resize(NGP_WTOTAL);
  memcpy(&W[NGP_W0], w0, sizeof(float) * NGP_HID * NGP_FEAT); memcpy(&W[NGP_B0], b0, sizeof(float) * NGP_HID);
  memcpy(&W[NGP_W1], w1, sizeof(float) * NGP_HID * NGP_HID);  memcpy(&W[NGP_B1], b1, sizeof(float) * NGP_HID);
  memcpy(&W[NGP_W2], w2, sizeof(float) * NGP_OUT * NGP_HID);  memcpy(&W[NGP_B2], b2, sizeof(float) * NGP_OUT);
}

static void near_far(const float* o, const float* d, const float* aabb, float min_near, float* pn, float* pf) {
  const float rdx = 1 / d[0], rdy = 1 / d[1], rdz = 1 / d[2];
  float near = (aabb[0] - o[0]) * rdx, far = (aabb[3] - o[0]) * rdx, t;
  if (near > far) { t = near; near = far; far = t; }
  float ny = (aabb[1] - o[1]) * rdy, fy = (aabb[4] - o[1]) * rdy;
  if (ny > fy) { t = ny; ny = fy; fy = t; }
  if (near > fy || ny > far) { *pn = *pf = 3.402823466e+38f; return; }
  if (ny > near) near = ny; if (fy < far) far = fy;
  float nz = (aabb[2] - o[2]) * rdz, fz = (aabb[5] - o[2]) * rdz;
  if (nz > fz) { t = nz; nz = fz; fz = t; }
  if (near > fz || nz > far) { *pn = *pf = 3.402823466e+38f; return; }
  if (nz > near) near = nz; if (fz < far) far = fz;
  if (near < min_near) near = min_near;
  *pn = near; *pf = far;
}

static void field_point(const NgpLevels& lv, const float* table, const float* W, float bound, const float x[3],
                        float* sigma, float* rgb) {
  float x01[3], feat[NGP_FEAT], h1[NGP_HID], h2[NGP_HID], out[NGP_OUT];
  const bool inside = ngp_unit(x, bound, x01);
  ngp_encode(lv, table, x01, inside, feat);
  ngp_mlp_forward(W, feat, h1, h2, out);
  *sigma = expf(out[0] + ngp_blob(x));
  for (int c = 0; c < 3; ++c) rgb[c] = ngp_sigmoid(out[1 + c]);
}

extern "C" void emu_render_forward(const float* table, const int32_t* h_offsets, uint32_t L, float S, uint32_t H,
                                   uint32_t gridtype, const float* w0, const float* b0, const float* w1,
                                   const float* b1, const float* w2, const float* b2, float bound,
                                   const float* rays_o, const float* rays_d, const float* aabb, uint32_t N,
                                   uint32_t T, float min_near, const float* lin, const float* u_coarse,
                                   const float* u_fine, uint32_t u_fine_row_stride, float bg, float* nears,
                                   float* fars, float* z_sorted, float* sigma_s, float* rgb_s, float* image,
                                   float* depth, float* weights_sum, float* z_fine_out) {
  NgpLevels lv; fill_levels(&lv, h_offsets, L, S, H, gridtype);
  std::vector<float> W; pack_weights(W, w0, b0, w1, b1, w2, b2);
#pragma omp parallel for schedule(dynamic, 4)
  for (int64_t n = 0; n < (int64_t)N; ++n) {
    std::vector<float> zc(T), sc(T), rc(3 * T), zf(T), sf(T), rf(3 * T), s1(T), s2(T);
    const float* o = rays_o + n * 3; const float* d = rays_d + n * 3;
    near_far(o, d, aabb, min_near, &nears[n], &fars[n]);
    for (uint32_t k = 0; k < T; ++k) {
      zc[k] = ngp_coarse_z(nears[n], fars[n], lin[k], u_coarse ? u_coarse[n * T + k] : -1.0f, T);
      float x[3]; ngp_point(o, d, zc[k], aabb, x);
      field_point(lv, table, W.data(), bound, x, &sc[k], &rc[3 * k]);
    }
    SfCol c1{s1.data(), 1}, c2{s2.data(), 1};
    ngp_sample_fine(zc.data(), sc.data(), u_fine + (size_t)n * u_fine_row_stride, nears[n], fars[n], T, c1, c2, zf.data());
    for (uint32_t k = 0; k < T; ++k) {
      float x[3]; ngp_point(o, d, zf[k], aabb, x);
      field_point(lv, table, W.data(), bound, x, &sf[k], &rf[3 * k]);
      if (z_fine_out) z_fine_out[n * T + k] = zf[k];
    }
    NgpRayOut r;
    ngp_merge_composite(zc.data(), sc.data(), rc.data(), zf.data(), sf.data(), rf.data(), nears[n], fars[n], T, bg, c1, c2,
                        z_sorted + (size_t)n * 2 * T, sigma_s + (size_t)n * 2 * T, rgb_s + (size_t)n * 6 * T, r);
    image[n * 3] = r.image[0]; image[n * 3 + 1] = r.image[1]; image[n * 3 + 2] = r.image[2];
    depth[n] = r.depth; weights_sum[n] = r.weights_sum;
  }
}

extern "C" void emu_render_backward(const float* table, const int32_t* h_offsets, uint32_t L, float S, uint32_t H,
                                    uint32_t gridtype, const float* w0, const float* b0, const float* w1,
                                    const float* b1, const float* w2, const float* b2, float bound,
                                    const float* rays_o, const float* rays_d, const float* aabb, uint32_t N,
                                    uint32_t T, const float* nears, const float* fars, const float* z_sorted,
                                    const float* sigma_s, const float* rgb_s, float bg, const float* g_image,
                                    const float* g_ws, float* g_table, float* g_w0, float* g_b0, float* g_w1,
                                    float* g_b1, float* g_w2, float* g_b2) {
  NgpLevels lv; fill_levels(&lv, h_offsets, L, S, H, gridtype);
  std::vector<float> W; pack_weights(W, w0, b0, w1, b1, w2, b2);
  const uint32_t M = 2 * T;
  std::vector<float> dsig(M), drgb(3 * M), s1(M), s2(M);
  for (uint32_t n = 0; n < N; ++n) {
    const float gI[3] = {g_image[n * 3], g_image[n * 3 + 1], g_image[n * 3 + 2]};
    SfCol c1{s1.data(), 1}, c2{s2.data(), 1};
    ngp_composite_backward(z_sorted + (size_t)n * M, sigma_s + (size_t)n * M, rgb_s + (size_t)n * 3 * M, nears[n], fars[n],
                           T, bg, gI, g_ws ? g_ws[n] : 0.0f, c1, c2, dsig.data(), drgb.data());
    for (uint32_t m = 0; m < M; ++m) {
      float x[3], x01[3], feat[NGP_FEAT], h1[NGP_HID], h2[NGP_HID], out[NGP_OUT], dout[NGP_OUT];
      ngp_point(rays_o + n * 3, rays_d + n * 3, z_sorted[(size_t)n * M + m], aabb, x);
      const bool inside = ngp_unit(x, bound, x01);
      ngp_encode(lv, table, x01, inside, feat);
      ngp_mlp_forward(W.data(), feat, h1, h2, out);
      const float pre = out[0] + ngp_blob(x);
      dout[0] = dsig[m] * expf(fminf(fmaxf(pre, -15.0f), 15.0f));
      for (int c = 0; c < 3; ++c) { const float s = ngp_sigmoid(out[1 + c]); dout[1 + c] = drgb[m * 3 + c] * s * (1.0f - s); }
      float dh2[NGP_HID], dh1[NGP_HID], dfeat[NGP_FEAT];
      ngp_mlp_backward(W.data(), h1, h2, dout, dh2, dh1, dfeat);
      for (int j = 0; j < NGP_OUT; ++j) { g_b2[j] += dout[j]; for (int k = 0; k < NGP_HID; ++k) g_w2[j * NGP_HID + k] += dout[j] * h2[k]; }
      for (int j = 0; j < NGP_HID; ++j) { g_b1[j] += dh2[j]; for (int k = 0; k < NGP_HID; ++k) g_w1[j * NGP_HID + k] += dh2[j] * h1[k]; }
      for (int j = 0; j < NGP_HID; ++j) { g_b0[j] += dh1[j]; for (int k = 0; k < NGP_FEAT; ++k) g_w0[j * NGP_FEAT + k] += dh1[j] * feat[k]; }
      ngp_scatter(lv, g_table, x01, inside, dfeat);
    }
  }
}

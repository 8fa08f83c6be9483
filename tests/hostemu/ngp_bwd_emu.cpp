// CPU harness for the MFMA field-backward kernel (sparsefusion_amd/csrc/ngp_bwd_mfma.h): the kernel source runs on CPU
// threads (hip_emu.h) and is compared by tests/test_hostemu_ngp_bwd.py with the per-point reference math of
// ngp_device.h (ngp_mlp_forward / ngp_mlp_backward, the functions the oracle-checked host emulation uses).
#ifndef SF_HOST_EMU
#define SF_HOST_EMU
#endif
#define HIPEMU_IMPLEMENTATION
#include "hip_emu.h"
#include <vector>
struct float2 { float x, y; };
#include "../../sparsefusion_amd/csrc/ngp_bwd_mfma.h"

static void fill_levels(NgpLevels* lv, const int32_t* h_offsets, uint32_t L, float S, uint32_t H, uint32_t gridtype) {
  for (uint32_t l = 0; l < NGP_MAX_LEVELS; ++l) {
    const bool on = l < L;
    const float scale = on ? exp2f((float)l * S) * (float)H - 1.0f : 0.f;
    lv->scale[l] = scale;
    lv->resolution[l] = on ? (uint32_t)ceilf(scale) + 1 : 1;
    lv->offset[l] = on ? (uint32_t)h_offsets[l] : 0;
    lv->hsize[l] = on ? (uint32_t)(h_offsets[l + 1] - h_offsets[l]) : 1;
  }
  lv->L = L;
  lv->gridtype = gridtype;
}

extern "C" void emu_field_bwd(const float* table, const int32_t* h_offsets, uint32_t L, float S, uint32_t H, uint32_t gridtype,
                              const float* w0, const float* b0, const float* w1, const float* b1, const float* w2, const float* b2,
                              float bound, const float* rays_o, const float* rays_d, const float* aabb, const float* z_s,
                              const float* dsig, const float* drgb, uint32_t P, uint32_t T2, uint32_t grid, int use_ref,
                              float* g_w0, float* g_b0, float* g_w1, float* g_b1, float* g_w2, float* g_b2, float* dfeat) {
  // use_ref == 2: the kernel on a field cache (features + permutation as the forward would leave them) instead of the re-gather
  std::vector<float> fc, ff;
  std::vector<uint32_t> perm;
  if (use_ref == 2) {
    NgpLevels lv0;
    fill_levels(&lv0, h_offsets, L, S, H, gridtype);
    const uint32_t T = T2 / 2, N = (P + T2 - 1) / T2;
    fc.assign((size_t)N * T * NGP_FEAT, 0.f); ff.assign((size_t)N * T * NGP_FEAT, 0.f); perm.assign((size_t)N * T2, 0);
    for (uint32_t p = 0; p < P; ++p) {
      const uint32_t n = p / T2, m = p - n * T2;
      const uint32_t src = (m * 7 + 3) % T2;                 // some permutation of the ray's samples (7 is odd, T2 a power of two... or coprime)
      perm[p] = src;
      float x[3], x01[3], feat[NGP_FEAT];
      ngp_point(rays_o + n * 3, rays_d + n * 3, z_s[p], aabb, x);
      const bool inside = ngp_unit(x, bound, x01);
      ngp_encode(lv0, table, x01, inside, feat);
      float* row = src < T ? &fc[((size_t)n * T + src) * NGP_FEAT] : &ff[((size_t)n * T + (src - T)) * NGP_FEAT];
      for (int i = 0; i < NGP_FEAT; ++i) row[i] = feat[i];
    }
  }
  FBArgs a{};                                      // (feat_c / feat_f / perm stay null: the re-gather path)
  a.table = table; a.w0 = w0; a.b0 = b0; a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2; a.bound = bound;
  a.g_w0 = g_w0; a.g_b0 = g_b0; a.g_w1 = g_w1; a.g_b1 = g_b1; a.g_w2 = g_w2; a.g_b2 = g_b2;
  fill_levels(&a.lv, h_offsets, L, S, H, gridtype);
  a.rays_o = rays_o; a.rays_d = rays_d; a.aabb = aabb; a.z_s = z_s; a.dsig = dsig; a.drgb = drgb; a.dfeat_out = dfeat; a.dfeat_P = P; a.p_off = 0;
  a.P = P; a.T2 = T2;
  if (use_ref == 2) { a.feat_c = fc.data(); a.feat_f = ff.data(); a.perm = perm.data(); }
  if (use_ref != 1) {
    hipemu::launch(grid, 256, FB_LDS_FLOATS * sizeof(float), [&] { k_ngp_field_bwd_mfma(a); });
    return;
  }
  // reference: per-point math (ngp_device.h), serial
  std::vector<float> W(NGP_WTOTAL);
  for (int i = 0; i < NGP_HID * NGP_FEAT; ++i) W[NGP_W0 + i] = w0[i];
  for (int i = 0; i < NGP_HID * NGP_HID; ++i) W[NGP_W1 + i] = w1[i];
  for (int i = 0; i < NGP_OUT * NGP_HID; ++i) W[NGP_W2 + i] = w2[i];
  for (int i = 0; i < NGP_HID; ++i) { W[NGP_B0 + i] = b0[i]; W[NGP_B1 + i] = b1[i]; }
  for (int i = 0; i < NGP_OUT; ++i) W[NGP_B2 + i] = b2[i];
  for (uint32_t p = 0; p < P; ++p) {
    const uint32_t n = p / T2;
    float x[3], x01[3], feat[NGP_FEAT], h1[NGP_HID], h2[NGP_HID], out[NGP_OUT], dout[NGP_OUT];
    ngp_point(rays_o + n * 3, rays_d + n * 3, z_s[p], aabb, x);
    const bool inside = ngp_unit(x, bound, x01);
    ngp_encode(a.lv, table, x01, inside, feat);
    ngp_mlp_forward(W.data(), feat, h1, h2, out);
    const float pre = out[0] + ngp_blob(x);
    dout[0] = dsig[p] * expf(fminf(fmaxf(pre, -15.0f), 15.0f));
    for (int c = 0; c < 3; ++c) { const float s = ngp_sigmoid(out[1 + c]); dout[1 + c] = drgb[p * 3 + c] * s * (1.0f - s); }
    float dh2[NGP_HID], dh1[NGP_HID], df[NGP_FEAT];
    ngp_mlp_backward(W.data(), h1, h2, dout, dh2, dh1, df);
    for (int j = 0; j < NGP_OUT; ++j) { g_b2[j] += dout[j]; for (int k = 0; k < NGP_HID; ++k) g_w2[j * NGP_HID + k] += dout[j] * h2[k]; }
    for (int j = 0; j < NGP_HID; ++j) { g_b1[j] += dh2[j]; for (int k = 0; k < NGP_HID; ++k) g_w1[j * NGP_HID + k] += dh2[j] * h1[k]; }
    for (int j = 0; j < NGP_HID; ++j) { g_b0[j] += dh1[j]; for (int k = 0; k < NGP_FEAT; ++k) g_w0[j * NGP_FEAT + k] += dh1[j] * feat[k]; }
    for (uint32_t l = 0; l < L; ++l) {
      dfeat[((size_t)l * P + p) * 2] = inside ? df[2 * l] : 0.0f;
      dfeat[((size_t)l * P + p) * 2 + 1] = inside ? df[2 * l + 1] : 0.0f;
    }
  }
}

// ---- binned table-gradient scatter (sparsefusion_amd/csrc/ngp_scatter_bin.h) against ngp_scatter (ngp_device.h), levels
// [first_level, L); `chunks` rounds of bin + reduce over consecutive ray ranges share the entries buffer (cursor reset by the reducer)
#include "../../sparsefusion_amd/csrc/ngp_scatter_bin.h"
extern "C" int emu_bin_scatter(const int32_t* h_offsets, uint32_t L, float S, uint32_t H, uint32_t gridtype, float bound,
                               const float* rays_o, const float* rays_d, const float* aabb, const float* z_s, const float* dfeat,
                               uint32_t N, uint32_t T2, uint32_t first_level, uint32_t cap, uint32_t chunks, uint32_t grid, int use_ref,
                               float* gtable) {
  NgpLevels lv;
  fill_levels(&lv, h_offsets, L, S, H, gridtype);
  const uint32_t P = N * T2;
  if (use_ref) {
    for (uint32_t p = 0; p < P; ++p) {
      const uint32_t n = p / T2;
      float x[3], x01[3], df[NGP_FEAT] = {0};
      ngp_point(rays_o + n * 3, rays_d + n * 3, z_s[p], aabb, x);
      const bool inside = ngp_unit(x, bound, x01);
      for (uint32_t l = first_level; l < L; ++l) { df[2 * l] = dfeat[((size_t)l * P + p) * 2]; df[2 * l + 1] = dfeat[((size_t)l * P + p) * 2 + 1]; }
      ngp_scatter(lv, gtable, x01, inside, df);
    }
    return 0;
  }
  SBArgs b{};
  SBRArgs r{};
  uint32_t tb = 0;
  for (uint32_t l = 0; l <= NGP_MAX_LEVELS; ++l) {
    b.bucket0[l] = r.bucket0[l] = tb;
    if (l >= first_level && l < L) {
      const uint32_t nb = (lv.hsize[l] + SB_ROWS - 1) >> SB_ROWS_LOG;
      if (nb > SB_MAX_BUCKETS) return 1;
      tb += nb;
    }
  }
  std::vector<uint32_t> cursor(tb, 0);
  std::vector<f32x4> ent((size_t)tb * cap);
  b.lv = lv; b.bound = bound; b.aabb = aabb; b.dfeat = dfeat; b.gtable = gtable; b.cursor = cursor.data(); b.ent = ent.data();
  b.T2 = T2; b.first_level = first_level; b.P_stride = P; b.cap = cap;
  r.lv = lv; r.gtable = gtable; r.cursor = cursor.data(); r.ent = ent.data(); r.first_level = first_level; r.cap = cap;
  const uint32_t Nc = (N + chunks - 1) / chunks;
  for (uint32_t c = 0; c * Nc < N; ++c) {
    const uint32_t n0 = c * Nc, nn = n0 + Nc <= N ? Nc : N - n0;
    b.rays_o = rays_o + (size_t)n0 * 3; b.rays_d = rays_d + (size_t)n0 * 3; b.z_s = z_s + (size_t)n0 * T2; b.P = nn * T2; b.p_off = n0 * T2;
    hipemu::launch(grid, SB_THREADS, 0, [&] { k_ngp_bin(b); });
    hipemu::launch(tb, SBR_THREADS, 0, [&] { k_ngp_bin_reduce(r); });
    for (uint32_t k = 0; k < tb; ++k) if (cursor[k]) return 2;          // the reducer leaves every cursor at zero
  }
  return 0;
}

// CPU harness for the MFMA field-forward kernel (sparsefusion_amd/csrc/ngp_fwd_mfma.h): the kernel source runs on CPU threads
// (hip_emu.h) and is compared by tests/test_hostemu_ngp_fwd.py with the per-point reference math of ngp_device.h
// (ngp_coarse_z / ngp_encode / ngp_mlp_forward: the functions the oracle-checked host emulation of the render uses).
#ifndef SF_HOST_EMU
#define SF_HOST_EMU
#endif
#define HIPEMU_IMPLEMENTATION
#include "hip_emu.h"
#include <vector>
struct float2 { float x, y; };
#include "../../sparsefusion_amd/csrc/ngp_fwd_mfma.h"

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

extern "C" void emu_field_fwd(const float* table, const int32_t* h_offsets, uint32_t L, float S, uint32_t H, uint32_t gridtype,
                              const float* w0, const float* b0, const float* w1, const float* b1, const float* w2, const float* b2,
                              float bound, const float* rays_o, const float* rays_d, const float* aabb, const float* nears,
                              const float* fars, const float* lin, const float* u, const float* z_in, uint32_t P, uint32_t T, int mode,
                              uint32_t grid, int use_ref, float* z_out, float* sigma, float* rgb) {
  FFArgs a;
  a.table = table; a.w0 = w0; a.b0 = b0; a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2; a.bound = bound;
  fill_levels(&a.lv, h_offsets, L, S, H, gridtype);
  a.rays_o = rays_o; a.rays_d = rays_d; a.aabb = aabb; a.nears = nears; a.fars = fars; a.lin = lin; a.u = u; a.z_in = z_in;
  a.P = P; a.T = T; a.mode = mode; a.z_out = z_out; a.sigma = sigma; a.rgb = rgb;
  if (!use_ref) {
    hipemu::launch(grid, 256, FF_LDS_FLOATS * sizeof(float), [&] { k_ngp_field_fwd_mfma(a); });
    return;
  }
  std::vector<float> W(NGP_WTOTAL);
  for (int i = 0; i < NGP_HID * NGP_FEAT; ++i) W[NGP_W0 + i] = w0[i];
  for (int i = 0; i < NGP_HID * NGP_HID; ++i) W[NGP_W1 + i] = w1[i];
  for (int i = 0; i < NGP_OUT * NGP_HID; ++i) W[NGP_W2 + i] = w2[i];
  for (int i = 0; i < NGP_HID; ++i) { W[NGP_B0 + i] = b0[i]; W[NGP_B1 + i] = b1[i]; }
  for (int i = 0; i < NGP_OUT; ++i) W[NGP_B2 + i] = b2[i];
  for (uint32_t p = 0; p < P; ++p) {                      // the per-point loop of k_ngp_field<0/1>
    const uint32_t n = p / T, k = p - n * T;
    float z;
    if (mode == 0) { z = ngp_coarse_z(nears[n], fars[n], lin[k], u ? u[p] : -1.0f, T); z_out[p] = z; }
    else z = z_in[p];
    float x[3], x01[3], feat[NGP_FEAT], h1[NGP_HID], h2[NGP_HID], out[NGP_OUT];
    ngp_point(rays_o + n * 3, rays_d + n * 3, z, aabb, x);
    const bool inside = ngp_unit(x, bound, x01);
    ngp_encode(a.lv, table, x01, inside, feat);
    ngp_mlp_forward(W.data(), feat, h1, h2, out);
    sigma[p] = expf(out[0] + ngp_blob(x));
    rgb[p * 3 + 0] = ngp_sigmoid(out[1]); rgb[p * 3 + 1] = ngp_sigmoid(out[2]); rgb[p * 3 + 2] = ngp_sigmoid(out[3]);
  }
}

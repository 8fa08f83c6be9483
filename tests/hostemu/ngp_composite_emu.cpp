// CPU harness for the wave-per-ray composite (sparsefusion_amd/csrc/ngp_composite_wave.h): the kernel source runs on CPU
// threads (hip_emu.h) and is compared by tests/test_hostemu_composite.py with the per-ray loop of ngp_device.h
// (ngp_merge_composite: the function the oracle-pinned host emulation of the render uses).
#ifndef SF_HOST_EMU
#define SF_HOST_EMU
#endif
#define HIPEMU_IMPLEMENTATION
#include "hip_emu.h"
#include <vector>
struct float2 { float x, y; };
#include "../../sparsefusion_amd/csrc/ngp_composite_wave.h"

extern "C" void emu_composite(const float* z_c, const float* sig_c, const float* rgb_c, const float* z_f, const float* sig_f,
                              const float* rgb_f, const float* nears, const float* fars, uint32_t N, uint32_t T, float bg,
                              int use_ref, float* z_s, float* sig_s, float* rgb_s, float* image, float* depth, float* ws) {
  if (!use_ref) {
    CompositeArgs a{z_c, sig_c, rgb_c, z_f, sig_f, rgb_f, nears, fars, N, T, bg, z_s, sig_s, rgb_s, image, depth, ws};
    hipemu::launch((N + 3) / 4, 256, 4 * 5 * 2 * T * sizeof(float), [&] { k_ngp_composite_wave(a); });
    return;
  }
  std::vector<float> key(T), ord(T);
  for (uint32_t n = 0; n < N; ++n) {
    NgpRayOut r;
    ngp_merge_composite(z_c + (size_t)n * T, sig_c + (size_t)n * T, rgb_c + (size_t)n * T * 3, z_f + (size_t)n * T,
                        sig_f + (size_t)n * T, rgb_f + (size_t)n * T * 3, nears[n], fars[n], T, bg, SfCol{key.data(), 1},
                        SfCol{ord.data(), 1}, z_s + (size_t)n * 2 * T, sig_s + (size_t)n * 2 * T, rgb_s + (size_t)n * 6 * T, r);
    image[n * 3 + 0] = r.image[0]; image[n * 3 + 1] = r.image[1]; image[n * 3 + 2] = r.image[2];
    depth[n] = r.depth;
    ws[n] = r.weights_sum;
  }
}

extern "C" void emu_composite_bwd(const float* z_s, const float* sig_s, const float* rgb_s, const float* nears, const float* fars,
                                  uint32_t N, uint32_t T, float bg, const float* g_image, const float* g_ws, int use_ref, float* dsig,
                                  float* drgb) {
  if (!use_ref) {
    CompositeBwdArgs a{z_s, sig_s, rgb_s, nears, fars, N, T, bg, g_image, g_ws, dsig, drgb};
    hipemu::launch((N + 3) / 4, 256, 0, [&] { k_ngp_composite_bwd_wave(a); });
    return;
  }
  std::vector<float> tr(2 * T), wt(2 * T);
  for (uint32_t n = 0; n < N; ++n) {
    const float gI[3] = {g_image[n * 3], g_image[n * 3 + 1], g_image[n * 3 + 2]};
    ngp_composite_backward(z_s + (size_t)n * 2 * T, sig_s + (size_t)n * 2 * T, rgb_s + (size_t)n * 6 * T, nears[n], fars[n], T, bg, gI,
                           g_ws ? g_ws[n] : 0.0f, SfCol{tr.data(), 1}, SfCol{wt.data(), 1}, dsig + (size_t)n * 2 * T,
                           drgb + (size_t)n * 6 * T);
  }
}

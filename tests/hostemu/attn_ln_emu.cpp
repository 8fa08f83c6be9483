// CPU harness for k_layernorm / k_attn16 (sparsefusion_amd/csrc/attn_ln.h): the kernel source runs on CPU threads (hip_emu.h).
#ifndef SF_HOST_EMU
#define SF_HOST_EMU
#endif
#define HIPEMU_IMPLEMENTATION
#include "hip_emu.h"
#include "../../sparsefusion_amd/csrc/attn_ln.h"

extern "C" void emu_layernorm(const float* in, const float* gain, const float* bias, void* out, const float* resid, int R, int C, float eps,
                              int pre_gelu, int out_f32) {
  if (C == 256 && R >= 6) {                     // (the launcher takes this kernel from 1024 rows on; the test drives it with fewer)
    hipemu::launch((unsigned)((R + 3) / 4), 256, 0, [&] { k_layernorm_w256(in, gain, bias, out, resid, R, eps, pre_gelu, out_f32, nullptr); });
    return;
  }
  if (R <= 256 && (C == 512 || C == 1024 || C == 2048) && !(pre_gelu & 4)) {     // the launcher's rule (unet_ops.hip::run_ln); bit 2 of pre_gelu = flag 4
    const unsigned grid = (unsigned)((R + 3) / 4);
    if (C == 512) hipemu::launch(grid, 256, 0, [&] { k_layernorm_wave<2>(in, gain, bias, out, resid, R, eps, pre_gelu & 1, out_f32); });
    else if (C == 1024) hipemu::launch(grid, 256, 0, [&] { k_layernorm_wave<4>(in, gain, bias, out, resid, R, eps, pre_gelu & 1, out_f32); });
    else hipemu::launch(grid, 256, 0, [&] { k_layernorm_wave<8>(in, gain, bias, out, resid, R, eps, pre_gelu & 1, out_f32); });
    return;
  }
  hipemu::launch((unsigned)R, 256, 0, [&] { k_layernorm(in, gain, bias, out, resid, R, C, eps, pre_gelu & 1, out_f32); });
}

// segment s: keys ks[s] / values vs[s] with (rows, row_stride, batch_stride, head_stride) in geo[4 s ..]
extern "C" void emu_attn16(const float* q, void* out, const float* const* ks, const float* const* vs, const int* geo, int B, int heads,
                           int ldq, float scale, int out_f32) {
  AttnSeg s[3];
  for (int k = 0; k < 3; ++k) s[k] = AttnSeg{ks[k], vs[k], geo[4 * k], geo[4 * k + 1], geo[4 * k + 2], geo[4 * k + 3]};
  hipemu::launch((unsigned)(B * heads), 256, 0, [&] { k_attn16(q, out, s[0], s[1], s[2], heads, ldq, scale, out_f32); });
}

// Where do the 11.8 us of a 4x4 GroupNorm-self conv launch go INSIDE the replayed graph?  (r05)
// tools/exp/weight_prefetch_chain.hip imitates the launch (64 KB lazy slice -> statistics -> SiLU -> LDS frame -> 72 KB weight slab -> LDS
// reduction) and costs 6.9 us per launch; the product kernel costs 11.8.  This file instantiates ONLY k_conv_fused<1, 1, 12, GN_SELF, LAZY 0 | 1>
// from the product header under -DSF_FCX=<n> (phase knock-outs, fused_kernels.h) so that a variant compiles in seconds:
//   0 product | 1 no weight loads | 2 one load per lazy element instead of six | 3 no statistics | 4 no gamma / beta / scale-shift loads
//   5 no SiLU | 6 no MFMA | 7 no cross-wave sum | 8 empty body | 9 no lazy materialisation | 1000 + mask: several at once
//   -DSF_FCONV_WAVES=4: the same kernel with 4 waves per workgroup (256 threads)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DSF_FCX=<n> -Iinclude tools/exp/fconv4_knockout.hip -o /tmp/fcx_<n>.so
#include "../../sparsefusion_amd/csrc/sf_common.h"
#include "../../sparsefusion_amd/csrc/fused_host.h"

thread_local char sf_err_buf[512];

extern "C" int fcx_run(const sf_op* op, void* stream) {
  FConvArgs a;
  int WM, WN;
  uint32_t grid, lds;
  if (fconv_setup(*op, a, WM, WN, grid, lds, sf_err_buf, sizeof(sf_err_buf))) return 1;
  if (WM != 1 || WN != 1 || a.norm != FNORM_GN_SELF || (op->flags & (16 | 32)) || a.s1.mode > 1) return 2;
  hipStream_t st = (hipStream_t)stream;
  static bool big = false;
  if (!big) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv_fused<1, 1, 12, FNORM_GN_SELF, 0, SF_FCONV_WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, SF_LDS_MAX);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv_fused<1, 1, 12, FNORM_GN_SELF, 1, SF_FCONV_WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, SF_LDS_MAX);
    big = true;
  }
  if (a.s1.mode == 1) k_conv_fused<1, 1, 12, FNORM_GN_SELF, 1, SF_FCONV_WAVES><<<grid, SF_FCONV_WAVES * 64, lds, st>>>(a);
  else k_conv_fused<1, 1, 12, FNORM_GN_SELF, 0, SF_FCONV_WAVES><<<grid, SF_FCONV_WAVES * 64, lds, st>>>(a);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}
extern "C" const char* fcx_error() { return sf_err_buf; }

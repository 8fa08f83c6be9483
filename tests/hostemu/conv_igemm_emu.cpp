// CPU harness for k_conv_igemm (sparsefusion_amd/csrc/conv_igemm.h): the kernel source runs on CPU threads (hip_emu.h).  Argument
// set-up mirrors run_conv of csrc/unet_ops.hip for tile codes < 256.
#ifndef SF_HOST_EMU
#define SF_HOST_EMU
#endif
#define HIPEMU_IMPLEMENTATION
#include "hip_emu.h"
#include <algorithm>
using std::min;
using std::max;
#include "../../sparsefusion_amd/csrc/conv_igemm.h"

template <int WM, int WN>
static void go(const ConvArgs& a, int a_f32, unsigned blocks) {
  if (a_f32) hipemu::launch(blocks, 256, 0, [&] { k_conv_igemm<WM, WN, true>(a); });
  else hipemu::launch(blocks, 256, 0, [&] { k_conv_igemm<WM, WN, false>(a); });
}

extern "C" int emu_conv_igemm(const void* in, const uint16_t* w, const float* bias, float* out, const float* resid, float* ws, int B, int H,
                              int W, int Cin, int Ho, int Wo, int Cout, int ldc, int co_off, int k, int stride, int pad, int groups,
                              int WM, int WN, int a_f32, int accum, int ups, int relu, int pixshuf, float* slots) {
  ConvArgs a{};
  a.slots_out = slots;
  a.in = in; a.w = reinterpret_cast<const bf16x8*>(w); a.bias = bias; a.out = out; a.resid = resid; a.ws = ws;
  a.accum = accum; a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Ho = Ho; a.Wo = Wo; a.Cout = Cout; a.ldc = ldc; a.co_off = co_off;
  a.kh = a.kw = k; a.stride = stride; a.pad = pad; a.groups = groups; a.pixshuf = pixshuf; a.ups = ups; a.relu = relu;
  a.cchunks = Cin / 32;
  a.KS = k * k * a.cchunks;
  a.m_frags = (B * Ho * Wo + 15) / 16;
  a.n_frags = (Cout + 15) / 16;
  a.m_tiles = (a.m_frags + WM - 1) / WM;
  a.n_tiles = (a.n_frags + WN - 1) / WN;
  a.npad = a.n_frags * 16;
  a.steps_per_wave = (a.KS + a.groups * 4 - 1) / (a.groups * 4);
  const unsigned blocks = (unsigned)(a.m_tiles * a.n_tiles * a.groups);
  switch (WM * 16 + WN) {
    case 1 * 16 + 1: go<1, 1>(a, a_f32, blocks); break;
    case 1 * 16 + 2: go<1, 2>(a, a_f32, blocks); break;
    case 2 * 16 + 2: go<2, 2>(a, a_f32, blocks); break;
    case 4 * 16 + 2: go<4, 2>(a, a_f32, blocks); break;
    case 4 * 16 + 4: go<4, 4>(a, a_f32, blocks); break;
    default: return 1;
  }
  return 0;
}

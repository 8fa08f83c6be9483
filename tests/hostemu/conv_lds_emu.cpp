// CPU harness for k_conv_lds (sparsefusion_amd/csrc/conv_lds.h): the kernel source runs on CPU threads (hip_emu.h).  Argument
// set-up mirrors run_conv of csrc/unet_ops.hip for tile codes >= 256.
#ifndef SF_HOST_EMU
#define SF_HOST_EMU
#endif
#define HIPEMU_IMPLEMENTATION
#include "hip_emu.h"
#include <algorithm>
using std::min;
using std::max;
#include "../../sparsefusion_amd/csrc/conv_halo.h"
#include "../../sparsefusion_amd/csrc/conv_halo_small.h"

// gn_part != null: the GroupNorm-partials variant (k_conv_lds_gn) followed by k_gn_finalize into `stats` [B][G][2]
// glds > 0: k_conv_glds (conv_glds.h) with a ring of `glds` stage buffers instead of k_conv_lds; bf16 activations only
template <int BNF, int NST, bool GN>
static void run_glds(const ConvArgs& a, unsigned nblk, double* gn_part, int gn_cg) {
  hipemu::launch(nblk, 512, conv_glds_lds_bytes(BNF, NST), [&] { k_conv_glds<BNF, NST, GN>(a, gn_part, gn_cg); });
}
template <int BNF, int NST, bool GN>
static void run_halo(const ConvArgs& a, unsigned nblk, double* gn_part, int gn_cg) {
  hipemu::launch(nblk, 512, conv_halo_lds_bytes(BNF, NST), [&] { k_conv3_halo<BNF, NST, GN>(a, gn_part, gn_cg); });
}
template <int BNF, int NST>
static int run_halo_sm(const ConvArgs& a, unsigned nblk) {
  if (!conv_halo_sm_ok(a)) return 4;
  if (a.H == 4) hipemu::launch(nblk, 512, conv_halo_sm_lds_bytes(BNF, NST, 2), [&] { k_conv3_halo_sm<BNF, NST, 2>(a); });
  else hipemu::launch(nblk, 512, conv_halo_sm_lds_bytes(BNF, NST, 3), [&] { k_conv3_halo_sm<BNF, NST, 3>(a); });
  return 0;
}
template <int BNF, bool GN>
static int run_glds_nst(const ConvArgs& a, unsigned nblk, double* gn_part, int gn_cg, int nst) {
  if (nst == 8 || nst == 9) {                                // k_conv3_halo_sm (whole 4x4 / 8x8 maps) with a 3 / 4 deep weight ring
    if (GN) return 7;
    return nst == 8 ? run_halo_sm<BNF, 3>(a, nblk) : run_halo_sm<BNF, 4>(a, nblk);
  }
  if (nst == 3) run_glds<BNF, 3, GN>(a, nblk, gn_part, gn_cg);
  else if (nst == 4) run_glds<BNF, 4, GN>(a, nblk, gn_part, gn_cg);
  else if (nst == 6 || nst == 7) {                           // k_conv3_halo with a 3 / 4 deep weight ring
    if (!conv_halo_ok(a) || (a.B * a.Ho * a.Wo) % 128) return 4;
    if (nst == 6) run_halo<BNF, 3, GN>(a, nblk, gn_part, gn_cg); else run_halo<BNF, 4, GN>(a, nblk, gn_part, gn_cg);
  } else return 1;
  return 0;
}

// r06: split-K (workspace [groups][M][npad], no bias) and the SiLU + PixelShuffle(2) epilogue of k_conv_lds / k_conv_glds: set for the NEXT
// emu_conv_lds call (its `out` is then the shuffled tensor [B][2 Ho][2 Wo][ldc]; `twin` must be null), cleared by it.
static int g_groups = 1, g_pixshuf = 0;
static float* g_ws = nullptr;
extern "C" void emu_conv_lds_mode(int groups, int pixshuf, float* ws) { g_groups = groups; g_pixshuf = pixshuf; g_ws = ws; }

extern "C" int emu_conv_lds(const void* in, const uint16_t* w, const float* bias, float* out, const float* resid, int B, int H, int W,
                            int Cin, int Ho, int Wo, int Cout, int ldc, int co_off, int k, int stride, int pad, int bnf, int a_f32,
                            int accum, int ups, int relu, double* gn_part, int gn_cg, double* stats, int glds, uint16_t* twin) {
  ConvArgs a{};
  a.in = in; a.w = reinterpret_cast<const bf16x8*>(w); a.bias = bias; a.out = out; a.resid = resid;
  a.ws = reinterpret_cast<float*>(twin);          // operand-type twin of the output (dense [M][Cout]) or null
  a.accum = accum; a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Ho = Ho; a.Wo = Wo; a.Cout = Cout; a.ldc = ldc; a.co_off = co_off;
  a.kh = a.kw = k; a.stride = stride; a.pad = pad; a.groups = 1; a.pixshuf = 0; a.ups = ups; a.relu = relu;
  a.cchunks = Cin / 32;
  a.KS = k * k * a.cchunks;
  a.m_frags = (B * Ho * Wo + 15) / 16;
  a.n_frags = (Cout + 15) / 16;
  a.npad = a.n_frags * 16;
  a.m_tiles = (a.m_frags + 7) / 8;
  a.n_tiles = (a.n_frags + bnf - 1) / bnf;
  a.steps_per_wave = 0;
  a.groups = g_groups; a.pixshuf = g_pixshuf;
  if (g_groups > 1) a.ws = g_ws;
  g_groups = 1; g_pixshuf = 0; g_ws = nullptr;
  if ((a.groups > 1 || a.pixshuf) && (twin || gn_part || glds == 6 || glds == 7)) return 5;
  if (a.groups > 1 && (a.pixshuf || a.groups > ((glds ? a.KS : a.KS + 1) >> 1))) return 6;
  const unsigned nblk = (unsigned)(a.m_tiles * a.n_tiles * a.groups);
  if (glds) {
    if (a_f32 || Cin % 64 || Cout % 4 || ldc % 4 || co_off % 4) return 3;
    if (gn_part && ((Ho * Wo) % 128 || co_off || ldc != Cout || (gn_cg != 4 && gn_cg != 8 && gn_cg != 16))) return 2;
    int rc;
    if (bnf == 8) rc = gn_part ? run_glds_nst<8, true>(a, nblk, gn_part, gn_cg, glds) : run_glds_nst<8, false>(a, nblk, nullptr, 0, glds);
    else if (bnf == 4) rc = gn_part ? run_glds_nst<4, true>(a, nblk, gn_part, gn_cg, glds) : run_glds_nst<4, false>(a, nblk, nullptr, 0, glds);
    else return 1;
    if (rc) return rc;
    if (gn_part) {
      const int G = Cout / gn_cg, tiles_per_image = Ho * Wo / 128;
      hipemu::launch((unsigned)(B * G), 256, 0, [&] { k_gn_finalize(gn_part, stats, tiles_per_image, G); });
    }
    return 0;
  }
  if (gn_part) {
    if ((Ho * Wo) % 128 || co_off || ldc != Cout || (gn_cg != 4 && gn_cg != 8 && gn_cg != 16)) return 2;
    if (bnf == 8) {
      if (a_f32) hipemu::launch(nblk, 256, 0, [&] { k_conv_lds_gn<8, true>(a, gn_part, gn_cg); });
      else hipemu::launch(nblk, 256, 0, [&] { k_conv_lds_gn<8, false>(a, gn_part, gn_cg); });
    } else if (bnf == 4) {
      if (a_f32) hipemu::launch(nblk, 256, 0, [&] { k_conv_lds_gn<4, true>(a, gn_part, gn_cg); });
      else hipemu::launch(nblk, 256, 0, [&] { k_conv_lds_gn<4, false>(a, gn_part, gn_cg); });
    } else {
      return 1;
    }
    const int G = Cout / gn_cg, tiles_per_image = Ho * Wo / 128;
    hipemu::launch((unsigned)(B * G), 256, 0, [&] { k_gn_finalize(gn_part, stats, tiles_per_image, G); });
    return 0;
  }
  if (bnf == 8) {
    if (a_f32) hipemu::launch(nblk, 256, 0, [&] { k_conv_lds<8, true>(a); }); else hipemu::launch(nblk, 256, 0, [&] { k_conv_lds<8, false>(a); });
  } else if (bnf == 4) {
    if (a_f32) hipemu::launch(nblk, 256, 0, [&] { k_conv_lds<4, true>(a); }); else hipemu::launch(nblk, 256, 0, [&] { k_conv_lds<4, false>(a); });
  } else {
    return 1;
  }
  return 0;
}

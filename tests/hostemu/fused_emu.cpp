// CPU harness for the fused UNet kernels (sparsefusion_amd/csrc/fused_kernels.h): the SAME kernel source, compiled
// by the host clang with one fiber per lane (hip_emu.h), driven by the SAME op decoding as the gfx950 launchers
// (fused_host.h).  Test infrastructure only -- checks kernel logic (indexing, LDS layout, fragment order, lazy
// sources, statistics) on a GPU-less machine; the product path is unet_fused.hip on the GPU.
#ifndef SF_HOST_EMU
#define SF_HOST_EMU
#endif
#define HIPEMU_IMPLEMENTATION
#include "hip_emu.h"
#include "../../sparsefusion_amd/csrc/fused_host.h"

template <int WM, int WN, int D, int NORM, int LAZY>
static void emu_fconv(const FConvArgs& a, uint32_t grid, uint32_t lds) {
  hipemu::launch(grid, SF_FCONV_WAVES * 64, lds, [&] { k_conv_fused<WM, WN, D, NORM, LAZY, SF_FCONV_WAVES>(a); });
}

template <int WM, int WN, int D, int NORM, int LAZY>
static void emu_fconv_pair(const FConvPairArgs& p, uint32_t grid, uint32_t lds) {
  hipemu::launch(grid, SF_FCONV_WAVES * 64, lds, [&] { k_conv_fused_pair<WM, WN, D, NORM, LAZY, SF_FCONV_WAVES>(p); });
}

static int g_conv4_launches = 0, g_conv4_mb_launches = 0;
extern "C" int emu_conv4_mb_launches() { return g_conv4_mb_launches; }  // ops that ran on k_conv4_gn_mb (several images per workgroup)
extern "C" int emu_conv4_launches() { return g_conv4_launches; }  // how many ops ran on k_conv4_gn (tests assert the path was taken)
static int g_conv3s_launches = 0;
extern "C" int emu_conv3s_launches() { return g_conv3s_launches; }      // ops that ran on k_conv3s (tests assert the path was taken)
static int g_rc_launches = 0;
extern "C" int emu_rc_launches() { return g_rc_launches; }      // how many pairs ran as k_conv_fused_pipe_rc (tests assert the path was taken)

static int emu_run_pair(const sf_op* op1, const sf_op* op2, char* err, int errn) {
  if (op2->type == SF_OP_GCA) {                    // res_conv || GlobalContext pooling (k_gca_pool_rc)
    FConvArgs b;
    int WM, WN;
    uint32_t gb, lds, gp;
    GcaPoolArgs pa;
    GcaNetArgs na;
    GcaGateArgs ga;
    if (fconv_setup(*op1, b, WM, WN, gb, lds, err, (size_t)errn) || gca_setup(*op2, pa, na, ga, gp, err, (size_t)errn)) return 1;
    if (op2->flags != 1 || b.norm != FNORM_NONE || b.s1.mode || WM != 1 || WN != 1) { snprintf(err, errn, "pool || res_conv pair: bad operands"); return 1; }
    hipemu::launch(gb + gp, SF_FCONV_WAVES * 64, lds, [&] { k_gca_pool_rc<1, 1, 12, SF_FCONV_WAVES>(pa, b, (int)gb); });
    return 0;
  }
  FConvPairArgs p;
  int WM, WN;
  uint32_t grid, lds;
  if (fconv_pair_setup(*op1, *op2, p, WM, WN, grid, lds, err, (size_t)errn)) return 1;
  if (op1->flags & 32) {
    const int EPT = fconv_pipe_ept(p.a);
    {
      FConvArgs a1;
      int wm1, wn1;
      uint32_t g1, l1;
      if (!fconv_setup(*op1, a1, wm1, wn1, g1, l1, err, (size_t)errn)) {       // the same dispatch as unet_fused.hip (r06: k_conv3s_rc)
        const int twl = conv3s_rc_twl(*op1, a1, p.b, WM, WN);
        if (twl >= 0) {
          a1.rc_w = p.b.w; a1.rc_bias = p.b.bias; a1.rc_out = p.b.out;
#define SF_TRY3R(hl_, c1_, c2_, co_, twl_, wm_, wn_) \
          if (a1.H == (1 << hl_) && a1.s1.C == c1_ && a1.s2.C == c2_ && a1.Cout == co_ && twl == twl_ && WM == wm_ && WN == wn_) { \
            hipemu::launch(g1, 512, Conv3sGeom<hl_, c1_, c2_, co_, twl_, wm_, wn_, false, true>::LDS_BYTES, [&] { k_conv3s_rc<hl_, c1_, c2_, co_, twl_, wm_, wn_>(a1); }); \
            ++g_conv3s_launches; \
            return 0; \
          }
          SF_CONV3S_RC_VARIANTS(SF_TRY3R)
#undef SF_TRY3R
          a1.rc_w = nullptr; a1.rc_bias = nullptr; a1.rc_out = nullptr;
        }
        if (op1->i[19] >> 2) { snprintf(err, errn, "fconv pipe pair: no k_conv3s_rc variant"); return 1; }
      }
      if (!fconv_setup(*op1, a1, wm1, wn1, g1, l1, err, (size_t)errn) && fconv_pipe_rc_merge(a1, p.b, WM, WN, l1)) {
#define SF_TRYR(wm, wn, ept) \
        if (WM == wm && WN == wn && EPT == ept) { \
          hipemu::launch(g1, SF_FCONV_WAVES * 64, l1, [&] { k_conv_fused_pipe_rc<wm, wn, ept, SF_FCONV_WAVES>(a1); }); \
          ++g_rc_launches; \
          return 0; \
        }
        SF_FCONV_PIPE_RC_VARIANTS(SF_TRYR)
#undef SF_TRYR
      }
    }
#define SF_TRYP(wm, wn, ept) \
    if (WM == wm && WN == wn && EPT == ept) { \
      hipemu::launch(grid, SF_FCONV_WAVES * 64, lds, [&] { k_conv_fused_pipe_pair<wm, wn, ept, SF_FCONV_WAVES>(p); }); \
      return 0; \
    }
    SF_FCONV_PIPE_VARIANTS(SF_TRYP)
#undef SF_TRYP
    snprintf(err, errn, "fconv pipe pair: no kernel variant for tile %dx%d, %d staging elements", WM, WN, EPT);
    return 1;
  }
#define SF_TRY(wm, wn, d, nm_, lz_) \
  if (WM == wm && WN == wn && p.a.norm == nm_ && p.a.s1.mode == lz_) { emu_fconv_pair<wm, wn, d, nm_, lz_>(p, grid, lds); return 0; }
  SF_FCONV_PAIR_VARIANTS(SF_TRY)
#undef SF_TRY
  snprintf(err, errn, "fconv pair: no kernel variant for tile %dx%d norm %d lazy %d", WM, WN, p.a.norm, p.a.s1.mode);
  return 1;
}

extern "C" int emu_run_op(const sf_op* op, char* err, int errn) {
  err[0] = 0;
  if (op->type == SF_OP_FCONV) {
    FConvArgs a;
    int WM, WN;
    uint32_t grid, lds;
    if (fconv_setup(*op, a, WM, WN, grid, lds, err, (size_t)errn)) return 1;
    if (op->flags & 32) {
      const int twl = conv3s_twl(*op, a, WM, WN);          // the same dispatch as unet_fused.hip::run_fconv (r06: k_conv3s)
      if (twl >= 0) {
#define SF_TRY3(hl_, c_, twl_, wm_, wn_) \
        if (a.H == (1 << hl_) && a.C == c_ && twl == twl_ && WM == wm_ && WN == wn_) { \
          if (a.weff) hipemu::launch(grid, 512, Conv3sGeom1<hl_, c_, twl_, wm_, wn_, true>::LDS_BYTES, [&] { k_conv3s<hl_, c_, twl_, wm_, wn_, true>(a); }); \
          else hipemu::launch(grid, 512, Conv3sGeom1<hl_, c_, twl_, wm_, wn_, false>::LDS_BYTES, [&] { k_conv3s<hl_, c_, twl_, wm_, wn_, false>(a); }); \
          ++g_conv3s_launches; \
          return 0; \
        }
        SF_CONV3S_VARIANTS(SF_TRY3)
#undef SF_TRY3
      }
      if (op->i[19] >> 2) { snprintf(err, errn, "fconv pipe: no k_conv3s variant"); return 1; }
      const int EPT = fconv_pipe_ept(a);
#define SF_TRYP(wm, wn, ept) \
      if (WM == wm && WN == wn && EPT == ept) { \
        if (a.weff) hipemu::launch(grid, SF_FCONV_WAVES * 64, lds, [&] { k_conv_fused_pipe<wm, wn, ept, SF_FCONV_WAVES, true>(a); }); \
        else hipemu::launch(grid, SF_FCONV_WAVES * 64, lds, [&] { k_conv_fused_pipe<wm, wn, ept, SF_FCONV_WAVES, false>(a); }); \
        return 0; \
      }
      SF_FCONV_PIPE_VARIANTS(SF_TRYP)
#undef SF_TRYP
      snprintf(err, errn, "fconv pipe: no kernel variant for tile %dx%d, %d staging elements", WM, WN, EPT);
      return 1;
    }
    if (const int cs4 = conv4_cs4(*op, a, WM, WN)) {        // the same dispatch as unet_fused.hip::run_fconv (r05: k_conv4_gn)
      if (const int nb = conv4_mb_setup(*op, a, cs4, grid, lds)) {
#define SF_TRY4M(c4_, lz_, nb_) \
        if (cs4 == c4_ && a.s1.mode == lz_ && nb == nb_) { hipemu::launch(grid, 512, lds, [&] { k_conv4_gn_mb<c4_, lz_, nb_>(a); }); ++g_conv4_mb_launches; return 0; }
        SF_CONV4_MB_VARIANTS(SF_TRY4M)
#undef SF_TRY4M
        snprintf(err, errn, "fconv: no k_conv4_gn_mb variant");
        return 1;
      }
      if (a.norm == FNORM_NONE) { hipemu::launch(grid, 512, lds, [&] { k_conv4_gn<64, 0, false>(a); }); ++g_conv4_launches; return 0; }
#define SF_TRY4(c4_, lz_) \
      if (cs4 == c4_ && a.s1.mode == lz_) { hipemu::launch(grid, 512, lds, [&] { k_conv4_gn<c4_, lz_>(a); }); ++g_conv4_launches; return 0; }
      SF_TRY4(64, 0) SF_TRY4(64, 1) SF_TRY4(64, 2) SF_TRY4(128, 0) SF_TRY4(128, 1) SF_TRY4(128, 2)
#undef SF_TRY4
    }
    if (const int wn = lin4_attn_wn(*op, a, WM, WN)) {       // r06: k_lin4_attn
      if (wn == 1) hipemu::launch(grid, 512, lds, [&] { k_lin4_attn<1>(a); }); else hipemu::launch(grid, 512, lds, [&] { k_lin4_attn<2>(a); });
      ++g_conv4_launches;
      return 0;
    }
    if (const int c4t = lin4_c4t(*op, a, WM, WN)) {
#define SF_TRYL(c_, wn_) \
      if (c4t == c_ && WN == wn_) { hipemu::launch(grid, 512, lds, [&] { k_lin4_ln<c_, wn_>(a); }); ++g_conv4_launches; return 0; }
      SF_TRYL(8, 1) SF_TRYL(8, 2) SF_TRYL(16, 1) SF_TRYL(16, 2)
#undef SF_TRYL
    }
#define SF_TRY(wm, wn, d, nm_, lz_) \
    if (WM == wm && WN == wn && a.norm == nm_ && a.s1.mode == lz_) { emu_fconv<wm, wn, d, nm_, lz_>(a, grid, lds); return 0; }
    SF_FCONV_VARIANTS(SF_TRY)
#undef SF_TRY
    snprintf(err, errn, "fconv: no kernel variant for tile %dx%d norm %d lazy %d", WM, WN, a.norm, a.s1.mode);
    return 1;
  }
  if (op->type == SF_OP_SLOTS) {
    const int M = op->i[0], C = op->i[1], HW = op->i[2];
    if (M % 16 || C % 16 || (!op->p[0] && !op->p[5]) || !op->p[4]) { snprintf(err, errn, "slots: bad operands"); return 1; }
    const uint32_t waves = (uint32_t)(M / 16) * (C / 16);
    hipemu::launch((waves + 3) / 4, 256, 0, [&] {
      k_slots((const float*)op->p[0], (const float*)op->p[1], (const float*)op->p[2], (float*)op->p[3], (float*)op->p[4], M, C, HW,
              (const float*)op->p[5], (const float*)op->p[6], op->i[3], op->i[4]);
    });
    return 0;
  }
  if (op->type == SF_OP_GCA) {
    GcaPoolArgs pa;
    GcaNetArgs na;
    GcaGateArgs ga;
    uint32_t grid;
    if (gca_setup(*op, pa, na, ga, grid, err, (size_t)errn)) return 1;
    if (op->flags == 1) hipemu::launch(grid, 256, 0, [&] { k_gca_pool(pa); });
    else if (op->flags == 2) {
#define SF_TRYN(c_, n_) if (!(op->i[5] & 1) && na.C == c_ && na.Kp == c_ && na.chunks <= n_ && (n_ == 8 || na.chunks > n_ / 2)) { hipemu::launch(grid, 256, 0, [&] { k_gca_net0_t<c_, n_>(na); }); return 0; }
      SF_TRYN(256, 64) SF_TRYN(256, 8) SF_TRYN(512, 16) SF_TRYN(512, 8) SF_TRYN(1024, 8)
#undef SF_TRYN
      if (na.chunks <= 8) hipemu::launch(grid, 256, 0, [&] { k_gca_net0<8>(na); });
      else if (na.chunks <= 16) hipemu::launch(grid, 256, 0, [&] { k_gca_net0<16>(na); });
      else if (na.chunks <= 32) hipemu::launch(grid, 256, 0, [&] { k_gca_net0<32>(na); });
      else hipemu::launch(grid, 256, 0, [&] { k_gca_net0<64>(na); });
    }
    else if (ga.HID == 128 && !(op->i[5] & 1)) hipemu::launch(grid, 256, 0, [&] { k_gca_gate_t<128>(ga); });
    else if (ga.HID == 256 && !(op->i[5] & 1)) hipemu::launch(grid, 256, 0, [&] { k_gca_gate_t<256>(ga); });
    else if (ga.HID == 512 && !(op->i[5] & 1)) hipemu::launch(grid, 256, 0, [&] { k_gca_gate_t<512>(ga); });
    else hipemu::launch(grid, 256, 0, [&] { k_gca_gate(ga); });
    return 0;
  }
  snprintf(err, errn, "emu: op type %d not supported", op->type);
  return 1;
}

extern "C" int emu_plan_run(const sf_op* ops, uint32_t n, char* err, int errn) {
  for (uint32_t k = 0; k < n; ++k) {
    if (ops[k].type == SF_OP_FCONV && (ops[k].flags & 16)) {
      if (k + 1 >= n) { snprintf(err, errn, "emu: a paired fconv needs a successor"); return 1; }
      if (int rc = emu_run_pair(&ops[k], &ops[k + 1], err, errn)) return rc;
      ++k;
      continue;
    }
    if (int rc = emu_run_op(&ops[k], err, errn)) return rc;
  }
  return 0;
}

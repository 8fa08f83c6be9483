// gfx950 launchers of the fused UNet ops (kernels: fused_kernels.h / fused_gca.h; operand decoding: fused_host.h).
#include "sf_common.h"
#include "plan_ops.h"
#include "fused_host.h"


// Dynamic LDS above 64 KiB must be enabled per kernel AND per device (a process may drive several GPUs).
template <class K>
static int allow_big_lds(K kernel, uint32_t bytes, unsigned& device_mask, int limit = SF_LDS_MAX) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) SF_FAIL(SF_ERR_LAUNCH, "hipGetDevice failed");
  if (dev < 32 && (device_mask & (1u << dev))) return SF_OK;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, limit) != hipSuccess)
    SF_FAIL(SF_ERR_LAUNCH, "hipFuncSetAttribute(max dynamic LDS) failed");
  if (dev < 32) device_mask |= 1u << dev;
  return SF_OK;
}

template <int WM, int WN, int D, int NORM, int LAZY>
static int launch_fconv(const FConvArgs& a, uint32_t grid, uint32_t lds, hipStream_t st) {
  static unsigned mask = 0;
  if (int rc = allow_big_lds(k_conv_fused<WM, WN, D, NORM, LAZY, SF_FCONV_WAVES>, lds, mask)) return rc;
  k_conv_fused<WM, WN, D, NORM, LAZY, SF_FCONV_WAVES><<<grid, SF_FCONV_WAVES * 64, lds, st>>>(a);
  SF_CHECK_LAUNCH("conv_fused");
  return SF_OK;
}

template <int CS4, int LAZY, bool NORM = true>
static int launch_conv4(const FConvArgs& a, uint32_t grid, uint32_t lds, hipStream_t st) {
  static unsigned mask = 0;
  if (int rc = allow_big_lds(k_conv4_gn<CS4, LAZY, NORM>, lds, mask)) return rc;
  k_conv4_gn<CS4, LAZY, NORM><<<grid, 512, lds, st>>>(a);
  SF_CHECK_LAUNCH("conv4_gn");
  return SF_OK;
}

template <int CS4, int LAZY, int NB>
static int launch_conv4_mb(const FConvArgs& a, uint32_t grid, uint32_t lds, hipStream_t st) {
  static unsigned mask = 0;
  if (int rc = allow_big_lds(k_conv4_gn_mb<CS4, LAZY, NB>, lds, mask)) return rc;
  k_conv4_gn_mb<CS4, LAZY, NB><<<grid, 512, lds, st>>>(a);
  SF_CHECK_LAUNCH("conv4_gn_mb");
  return SF_OK;
}

template <int C4T, int WN>
static int launch_lin4(const FConvArgs& a, uint32_t grid, uint32_t lds, hipStream_t st) {
  static unsigned mask = 0;
  if (int rc = allow_big_lds(k_lin4_ln<C4T, WN>, lds, mask)) return rc;
  k_lin4_ln<C4T, WN><<<grid, 512, lds, st>>>(a);
  SF_CHECK_LAUNCH("lin4_ln");
  return SF_OK;
}

template <int WN>
static int launch_lin4_attn(const FConvArgs& a, uint32_t grid, uint32_t lds, hipStream_t st) {
  static unsigned mask = 0;
  if (int rc = allow_big_lds(k_lin4_attn<WN>, lds, mask)) return rc;
  k_lin4_attn<WN><<<grid, 512, lds, st>>>(a);
  SF_CHECK_LAUNCH("lin4_attn");
  return SF_OK;
}

template <int WM, int WN, int EPT, bool POOL>
static int launch_fconv_pipe(const FConvArgs& a, uint32_t grid, uint32_t lds, hipStream_t st) {
  static unsigned mask = 0;
  if (int rc = allow_big_lds(k_conv_fused_pipe<WM, WN, EPT, SF_FCONV_WAVES, POOL>, lds, mask)) return rc;
  k_conv_fused_pipe<WM, WN, EPT, SF_FCONV_WAVES, POOL><<<grid, SF_FCONV_WAVES * 64, lds, st>>>(a);
  SF_CHECK_LAUNCH("conv_fused_pipe");
  return SF_OK;
}

template <int HL, int C, int TWL, int WM, int WN, bool POOL>
static int launch_conv3s(const FConvArgs& a, uint32_t grid, hipStream_t st) {
  static unsigned mask = 0;
  constexpr uint32_t lds = Conv3sGeom1<HL, C, TWL, WM, WN, POOL>::LDS_BYTES;
  if (int rc = allow_big_lds(k_conv3s<HL, C, TWL, WM, WN, POOL>, lds, mask)) return rc;
  k_conv3s<HL, C, TWL, WM, WN, POOL><<<grid, 512, lds, st>>>(a);
  SF_CHECK_LAUNCH("conv3s");
  return SF_OK;
}

template <int HL, int C1, int C2, int COUT, int TWL, int WM, int WN>
static int launch_conv3s_rc(const FConvArgs& a, uint32_t grid, hipStream_t st) {
  static unsigned mask = 0;
  constexpr uint32_t lds = Conv3sGeom<HL, C1, C2, COUT, TWL, WM, WN, false, true>::LDS_BYTES;
  if (int rc = allow_big_lds(k_conv3s_rc<HL, C1, C2, COUT, TWL, WM, WN>, lds, mask)) return rc;
  k_conv3s_rc<HL, C1, C2, COUT, TWL, WM, WN><<<grid, 512, lds, st>>>(a);
  SF_CHECK_LAUNCH("conv3s_rc");
  return SF_OK;
}

static int run_fconv(const sf_op& op, hipStream_t st) {
  FConvArgs a;
  int WM, WN;
  uint32_t grid, lds;
  if (fconv_setup(op, a, WM, WN, grid, lds, sf_err_buf, sizeof(sf_err_buf))) return SF_ERR_INVALID;
  if (op.flags & 32) {
    const int twl = conv3s_twl(op, a, WM, WN);            // r06: the recurring geometries on their own kernel (fused_conv3s.h)
    if (twl >= 0) {
#define SF_TRY3(hl_, c_, twl_, wm_, wn_) \
      if (a.H == (1 << hl_) && a.C == c_ && twl == twl_ && WM == wm_ && WN == wn_) \
        return a.weff ? launch_conv3s<hl_, c_, twl_, wm_, wn_, true>(a, grid, st) : launch_conv3s<hl_, c_, twl_, wm_, wn_, false>(a, grid, st);
      SF_CONV3S_VARIANTS(SF_TRY3)
#undef SF_TRY3
    }
    if (op.i[19] >> 2) SF_FAIL(SF_ERR_INVALID, "fconv pipe: no k_conv3s variant for the %d-wide tile %dx%d of a %d-channel %dx%d map", op.i[19] >> 2, WM, WN, a.C, a.H, a.W);
    const int EPT = fconv_pipe_ept(a);
#define SF_TRYP(wm, wn, ept) if (WM == wm && WN == wn && EPT == ept) return a.weff ? launch_fconv_pipe<wm, wn, ept, true>(a, grid, lds, st) : launch_fconv_pipe<wm, wn, ept, false>(a, grid, lds, st);
    SF_FCONV_PIPE_VARIANTS(SF_TRYP)
#undef SF_TRYP
    SF_FAIL(SF_ERR_INVALID, "fconv pipe: no kernel variant for tile %dx%d, %d staging elements", WM, WN, EPT);
  }
  if (const int cs4 = conv4_cs4(op, a, WM, WN)) {       // r05: the 4x4 level's GroupNorm-self conv on its own kernel (fused_conv4.h)
    if (const int nb = conv4_mb_setup(op, a, cs4, grid, lds)) {       // B >= 2: NB images per workgroup share its weight slice
#define SF_TRY4M(c4_, lz_, nb_) if (cs4 == c4_ && a.s1.mode == lz_ && nb == nb_) return launch_conv4_mb<c4_, lz_, nb_>(a, grid, lds, st);
      SF_CONV4_MB_VARIANTS(SF_TRY4M)
#undef SF_TRY4M
      SF_FAIL(SF_ERR_INVALID, "fconv: no k_conv4_gn_mb variant for Cs4 %d lazy %d x %d images", cs4, a.s1.mode, nb);
    }
    if (a.norm == FNORM_NONE) return launch_conv4<64, 0, false>(a, grid, lds, st);      // (conv4_nonorm: Cs = 256, plain source)
#define SF_TRY4(c4_, lz_) if (cs4 == c4_ && a.s1.mode == lz_) return launch_conv4<c4_, lz_>(a, grid, lds, st);
    SF_TRY4(64, 0) SF_TRY4(64, 1) SF_TRY4(64, 2) SF_TRY4(128, 0) SF_TRY4(128, 1) SF_TRY4(128, 2)
#undef SF_TRY4
  }
  if (const int c4t = lin4_c4t(op, a, WM, WN)) {       // r05: LayerNorm -> Linear of the 16-token map on its own kernel
#define SF_TRYL(c_, wn_) if (c4t == c_ && WN == wn_) return launch_lin4<c_, wn_>(a, grid, lds, st);
    SF_TRYL(8, 1) SF_TRYL(8, 2) SF_TRYL(16, 1) SF_TRYL(16, 2)
#undef SF_TRYL
  }
  if (const int wn = lin4_attn_wn(op, a, WM, WN))       // r06: the attention-prologue projection of the 16-token map on its own kernel
    return wn == 1 ? launch_lin4_attn<1>(a, grid, lds, st) : launch_lin4_attn<2>(a, grid, lds, st);
#define SF_TRY(wm, wn, d, nm_, lz_) \
  if (WM == wm && WN == wn && a.norm == nm_ && a.s1.mode == lz_) return launch_fconv<wm, wn, d, nm_, lz_>(a, grid, lds, st);
  SF_FCONV_VARIANTS(SF_TRY)
#undef SF_TRY
  SF_FAIL(SF_ERR_INVALID, "fconv: no kernel variant for tile %dx%d norm %d lazy %d", WM, WN, a.norm, a.s1.mode);
}

template <int WM, int WN, int D, int NORM, int LAZY>
static int launch_fconv_pair(const FConvPairArgs& p, uint32_t grid, uint32_t lds, hipStream_t st) {
  static unsigned mask = 0;
  if (int rc = allow_big_lds(k_conv_fused_pair<WM, WN, D, NORM, LAZY, SF_FCONV_WAVES>, lds, mask)) return rc;
  k_conv_fused_pair<WM, WN, D, NORM, LAZY, SF_FCONV_WAVES><<<grid, SF_FCONV_WAVES * 64, lds, st>>>(p);
  SF_CHECK_LAUNCH("conv_fused_pair");
  return SF_OK;
}

template <int WM, int WN, int EPT>
static int launch_fconv_pipe_pair(const FConvPairArgs& p, uint32_t grid, uint32_t lds, hipStream_t st) {
  static unsigned mask = 0;
  if (int rc = allow_big_lds(k_conv_fused_pipe_pair<WM, WN, EPT, SF_FCONV_WAVES>, lds, mask)) return rc;
  k_conv_fused_pipe_pair<WM, WN, EPT, SF_FCONV_WAVES><<<grid, SF_FCONV_WAVES * 64, lds, st>>>(p);
  SF_CHECK_LAUNCH("conv_fused_pipe_pair");
  return SF_OK;
}

template <int WM, int WN, int EPT>
static int launch_fconv_pipe_rc(const FConvArgs& a, uint32_t grid, uint32_t lds, hipStream_t st) {
  static unsigned mask = 0;
  if (int rc = allow_big_lds(k_conv_fused_pipe_rc<WM, WN, EPT, SF_FCONV_WAVES>, lds, mask)) return rc;
  k_conv_fused_pipe_rc<WM, WN, EPT, SF_FCONV_WAVES><<<grid, SF_FCONV_WAVES * 64, lds, st>>>(a);
  SF_CHECK_LAUNCH("conv_fused_pipe_rc");
  return SF_OK;
}

// op1 = an un-normalised fconv (a res_conv, flags & 16), op2 = the GlobalContext pooling op of the same block: one launch
static int run_pool_rc_pair(const sf_op& op1, const sf_op& op2, hipStream_t st) {
  FConvArgs b;
  int WM, WN;
  uint32_t gb, lds, gp;
  GcaPoolArgs pa;
  GcaNetArgs na;
  GcaGateArgs ga;
  if (fconv_setup(op1, b, WM, WN, gb, lds, sf_err_buf, sizeof(sf_err_buf)) || gca_setup(op2, pa, na, ga, gp, sf_err_buf, sizeof(sf_err_buf)))
    return SF_ERR_INVALID;
  if (op2.flags != 1 || b.norm != FNORM_NONE || b.s1.mode || WM != 1 || WN != 1 || (op1.flags & 32) || b.dbg)
    SF_FAIL(SF_ERR_INVALID, "pool || res_conv pair: a plain 16-pixel x 16-channel res_conv tile next to a pooling op required");
  static unsigned mask = 0;
  constexpr int dyn_max = SF_LDS_MAX - 16384;                    // the pooling body keeps 9 KB of static LDS in the same kernel
  if ((int)lds > dyn_max) SF_FAIL(SF_ERR_INVALID, "pool || res_conv pair: %u bytes of LDS", lds);
  if (int rc = allow_big_lds(k_gca_pool_rc<1, 1, 12, SF_FCONV_WAVES>, lds, mask, dyn_max)) return rc;
  k_gca_pool_rc<1, 1, 12, SF_FCONV_WAVES><<<gb + gp, SF_FCONV_WAVES * 64, lds, st>>>(pa, b, (int)gb);
  SF_CHECK_LAUNCH("gca_pool_rc");
  return SF_OK;
}

int sf_plan_fused_pair(const sf_op* op1, const sf_op* op2, void* stream) {
  if (op2->type == SF_OP_GCA) return run_pool_rc_pair(*op1, *op2, (hipStream_t)stream);
  FConvPairArgs p;
  int WM, WN;
  uint32_t grid, lds;
  if (fconv_pair_setup(*op1, *op2, p, WM, WN, grid, lds, sf_err_buf, sizeof(sf_err_buf))) return SF_ERR_INVALID;
  if (op1->flags & 32) {
    const int EPT = fconv_pipe_ept(p.a);
    {                                                     // r04: the res_conv inside conv1's workgroups where the tile allows it
      FConvArgs a1;
      int wm1, wn1;
      uint32_t g1, l1;
      if (!fconv_setup(*op1, a1, wm1, wn1, g1, l1, sf_err_buf, sizeof(sf_err_buf))) {       // r06: the B = 1 plan's pairs on k_conv3s_rc (fused_conv3s.h)
        const int twl = conv3s_rc_twl(*op1, a1, p.b, WM, WN);
        if (twl >= 0) {
          a1.rc_w = p.b.w; a1.rc_bias = p.b.bias; a1.rc_out = p.b.out;
#define SF_TRY3R(hl_, c1_, c2_, co_, twl_, wm_, wn_) \
          if (a1.H == (1 << hl_) && a1.s1.C == c1_ && a1.s2.C == c2_ && a1.Cout == co_ && twl == twl_ && WM == wm_ && WN == wn_) \
            return launch_conv3s_rc<hl_, c1_, c2_, co_, twl_, wm_, wn_>(a1, g1, (hipStream_t)stream);
          SF_CONV3S_RC_VARIANTS(SF_TRY3R)
#undef SF_TRY3R
          a1.rc_w = nullptr; a1.rc_bias = nullptr; a1.rc_out = nullptr;
        }
        if (op1->i[19] >> 2) SF_FAIL(SF_ERR_INVALID, "fconv pipe pair: no k_conv3s_rc variant for the %d-wide tile %dx%d of a %d + %d -> %d channel %dx%d map", op1->i[19] >> 2, WM, WN, a1.s1.C, a1.s2.C, a1.Cout, a1.H, a1.W);
      }
      if (!fconv_setup(*op1, a1, wm1, wn1, g1, l1, sf_err_buf, sizeof(sf_err_buf)) && fconv_pipe_rc_merge(a1, p.b, WM, WN, l1)) {
#define SF_TRYR(wm, wn, ept) if (WM == wm && WN == wn && EPT == ept) return launch_fconv_pipe_rc<wm, wn, ept>(a1, g1, l1, (hipStream_t)stream);
        SF_FCONV_PIPE_RC_VARIANTS(SF_TRYR)
#undef SF_TRYR
      }
    }
#define SF_TRYP(wm, wn, ept) if (WM == wm && WN == wn && EPT == ept) return launch_fconv_pipe_pair<wm, wn, ept>(p, grid, lds, (hipStream_t)stream);
    SF_FCONV_PIPE_VARIANTS(SF_TRYP)
#undef SF_TRYP
    SF_FAIL(SF_ERR_INVALID, "fconv pipe pair: no kernel variant for tile %dx%d, %d staging elements", WM, WN, EPT);
  }
#define SF_TRY(wm, wn, d, nm_, lz_) \
  if (WM == wm && WN == wn && p.a.norm == nm_ && p.a.s1.mode == lz_) return launch_fconv_pair<wm, wn, d, nm_, lz_>(p, grid, lds, (hipStream_t)stream);
  SF_FCONV_PAIR_VARIANTS(SF_TRY)
#undef SF_TRY
  SF_FAIL(SF_ERR_INVALID, "fconv pair: no kernel variant for tile %dx%d norm %d lazy %d", WM, WN, p.a.norm, p.a.s1.mode);
}

static int run_slots(const sf_op& op, hipStream_t st) {
  const int M = op.i[0], C = op.i[1], HW = op.i[2];
  const int groups = op.i[3], npad = op.i[4];
  if (M % 16 || C % 16 || (!op.p[0] && !op.p[5]) || !op.p[4] || (op.p[1] && (!op.p[2] || !op.p[3])))
    SF_FAIL(SF_ERR_INVALID, "slots: M, C must be multiples of 16; gate mode needs res and out");
  if (op.p[5] && (op.p[1] || !op.p[3] || groups < 1 || groups > 8 || npad % 4 || npad < C)) SF_FAIL(SF_ERR_INVALID, "slots: bad split-K source (1..8 slabs)");
  const uint32_t waves = (uint32_t)(M / 16) * (C / 16);
  k_slots<<<sf_div_up(waves, 4), 256, 0, st>>>((const float*)op.p[0], (const float*)op.p[1], (const float*)op.p[2], (float*)op.p[3],
                                                (float*)op.p[4], M, C, HW, (const float*)op.p[5], (const float*)op.p[6], groups, npad);
  SF_CHECK_LAUNCH("slots");
  return SF_OK;
}

static int run_gca(const sf_op& op, hipStream_t st) {
  GcaPoolArgs pa;
  GcaNetArgs na;
  GcaGateArgs ga;
  uint32_t grid;
  if (gca_setup(op, pa, na, ga, grid, sf_err_buf, sizeof(sf_err_buf))) return SF_ERR_INVALID;
  if (op.flags == 1) k_gca_pool<<<grid, 256, 0, st>>>(pa);
  else if (op.flags == 2) {
    // r05: compile-time (C, chunk capacity) for the canonical UNet's blocks (i[5] & 1: keep k_gca_net0, parity tests)
#define SF_TRYN(c_, n_) if (!(op.i[5] & 1) && na.C == c_ && na.Kp == c_ && na.chunks <= n_ && (n_ == 8 || na.chunks > n_ / 2)) { k_gca_net0_t<c_, n_><<<grid, 256, 0, st>>>(na); SF_CHECK_LAUNCH("gca_net0_t"); return SF_OK; }
    SF_TRYN(256, 64) SF_TRYN(256, 8) SF_TRYN(512, 16) SF_TRYN(512, 8) SF_TRYN(1024, 8)
#undef SF_TRYN
    if (na.chunks <= 8) k_gca_net0<8><<<grid, 256, 0, st>>>(na);
    else if (na.chunks <= 16) k_gca_net0<16><<<grid, 256, 0, st>>>(na);
    else if (na.chunks <= 32) k_gca_net0<32><<<grid, 256, 0, st>>>(na);
    else k_gca_net0<64><<<grid, 256, 0, st>>>(na);
  }
  else if (ga.HID == 128 && !(op.i[5] & 1)) k_gca_gate_t<128><<<grid, 256, 0, st>>>(ga);      // r05: compile-time hidden width (i[5] & 1: keep k_gca_gate, parity tests)
  else if (ga.HID == 256 && !(op.i[5] & 1)) k_gca_gate_t<256><<<grid, 256, 0, st>>>(ga);
  else if (ga.HID == 512 && !(op.i[5] & 1)) k_gca_gate_t<512><<<grid, 256, 0, st>>>(ga);
  else k_gca_gate<<<grid, 256, 0, st>>>(ga);
  SF_CHECK_LAUNCH("gca");
  return SF_OK;
}

int sf_plan_fused_op(const sf_op* op, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  switch (op->type) {
    case SF_OP_FCONV: return run_fconv(*op, st);
    case SF_OP_SLOTS: return run_slots(*op, st);
    case SF_OP_GCA: return run_gca(*op, st);
    default: SF_FAIL(SF_ERR_INVALID, "fused: unknown op type %d", op->type);
  }
}

// C entry for the instrumented build of this file (libsf_fused_timing.so, -DSF_FCONV_TIMING; tools/fconv_phases.py)
extern "C" int sf_fused_op_run(const sf_op* op, void* stream) { return sf_plan_fused_op(op, stream); }

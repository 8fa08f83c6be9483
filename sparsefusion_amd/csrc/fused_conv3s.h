// k_conv3s: the slot-GroupNorm 3x3 convs of the 8x8 / 16x16 / 32x32 levels (external/imagen_pytorch.py:641-662: GroupNorm(8) -> x * (scale + 1)
// + shift -> SiLU -> Conv2d 3x3) for the geometries that recur in every eval: one plain source, Cout = C in {256, 512, 1024}.  Same op,
// operands, outputs (values to the fp32 summation order, statistics slots, pooled GlobalContext fragments) as k_conv_fused_pipe
// (fused_pipe.h), the general kernel, which stays for every other shape and behind the planner attribute Unet.conv3s.  r06.
//
// Why.  In the replayed eval graph a 32x32 256 -> 256 launch of the general kernel costs 11.6 us: boundary 2.05, entry -> statistics 1.4,
// first chunk staged 1.2, "pipelined chunks" 4.1, epilogue 1.5 (profiles/r03_fconv_pipe_phases.log, r05_graph_ablate_b1.log).  The 4.1 us
// are two phases of ~2 us for 1.2 us of staging and 0.3 us of matrix work: the matrix waves' weight ring is ONE chunk deep, so every
// revolution waits a full L2 / HBM round trip (~1.3 us) for loads issued half a microsecond earlier, and a deeper ring did not fit the
// general kernel (221-244 VGPRs: run-time tile shapes keep ~60 registers of addressing alive).  Here, as in k_conv4_gn (r05):
//   * everything that shapes the instruction stream is a template parameter (map size, channels, tile, chunk count): per-tap LDS and
//     weight offsets are immediates, the chunk loop is unrolled, no run-time branch stands in front of a load;
//   * every load the launch can issue up front is requested in its first instructions: statistics slots, RD chunks of weights per
//     matrix wave (RD = 2: ALL weights of a 256-channel layer), NB chunks of activations + their affine operands per staging wave,
//     bias and residual rows of the finalising waves;
//   * the affine (rstd * gamma * (scale + 1), ...) lives in the staging thread's registers (its channel chunk is fixed): no LDS
//     table, ONE barrier between the statistics and the first staged chunk;
//   * K is split over the 4 matrix waves by 32-channel sub-chunk (wave w = sub-chunk w of every 128-channel chunk, all 9 taps), so
//     tap offsets are compile-time and the sub-chunk is one base register;
//   * the pixel tile may be 2-D (TW < W): a 4 x 8 tile of the 32x32 map stages a 6 x 10 frame (60 pixels) where the 1 x 32 strip stages
//     3 x 32 + zero columns (96): the staging VALU (SiLU: two quarter-rate transcendentals per element, every n-tile re-normalises the
//     pixels it needs) is what bounds these launches once the weights are there.  Slots and pooled fragments are indexed by
//     (tile, m-fragment): their consumers sum over all fragments of an image, whatever pixels a fragment holds.
//
// One hazard met on the way (r06): see the comment at the affine computation in `consume`.
#pragma once
#include "fused_kernels.h"

// C1 + C2 input channels (C2 > 0: the virtual concat of a second source, scaled by s2.scale = 2^-1/2 for a skip connection), COUT output channels;
// RC: the ResnetBlock's res_conv (1x1 conv of the RAW concat, imagen_pytorch.py:700-729) rides in the same workgroups, one more k-step per matrix
// wave and chunk on a raw operand copy of the tile's own pixels (as k_conv_fused_pipe_rc, fused_pipe.h).
template <int HL, int C1, int C2, int COUT, int TWL, int WM, int WN, bool POOL, bool RC>
struct Conv3sGeom {
  static constexpr int C = C1 + C2;
  static constexpr int H = 1 << HL, W = H, HW = H * W, TW = 1 << TWL, TH = 16 * WM / TW;
  static constexpr bool FULLW = TW == W;
  static constexpr int FR = TH + 2, FW = TW + 2;
  static constexpr int SW = FULLW ? TW : TW + 2;            // staged columns of a frame row (a full-width strip's side columns are always zero)
  static constexpr int NPX = FR * SW, EPT = (NPX + 7) / 8;  // staged pixels; elements (float4 of one pixel) per staging thread and chunk
  static constexpr int CC = 128, NCH = C / CC, KS = 9 * (C / 32);
  static constexpr int PSTR = 288;                          // bytes per frame pixel: 128 operand-type channels + 32 (stride = 32 mod 256)
  static constexpr int BUF = ((FR * FW + 1) * PSTR + 15) / 16 * 16;      // + 1 spare pixel (dead staging elements)
  static constexpr int RAWB = RC ? ((TH * FW + 1) * PSTR + 15) / 16 * 16 : 0;      // RC: raw operand of the tile's own rows in the frame's pixel order (+ 1 spare), per buffer
  static constexpr int F = WM * WN, FT = F + (POOL ? WM : 0) + (RC ? F : 0);
  // the K-slice reduction buffer of the epilogue lies OVER the frame buffers: it is written behind the chunk loop's last barrier, when no frame is read any more
  static constexpr int RED_OFF = 0, RED_BYTES = 4 * FT * 1024, MISC_OFF = (2 * BUF > RED_BYTES ? 2 * BUF : RED_BYTES), WEFF_OFF = MISC_OFF + 256;
  static constexpr int RAW_OFF = WEFF_OFF + (POOL ? KS * 64 : 0);
  static constexpr int LDS_BYTES = RAW_OFF + 2 * RAWB;
  static constexpr int NTILES = COUT / (16 * WN), MTI = (H / TH) * (W / TW);
  static constexpr int G = 8, CG = C / G, NCF = CG / 16, NMF = HW / 16, SCNT = NMF * NCF, NSL = (SCNT + 63) / 64;
  static constexpr int KR = 9 + (RC ? 1 : 0);                                       // ring steps per matrix wave and chunk
  static constexpr int RD = NCH < 2 ? 1 : ((WN >= 4 || (RC && WN >= 2) || (NCH >= 8 && WM * WN >= 4)) ? 1 : 2);   // weight ring depth in chunks (KR * WN fragments per chunk and wave; RC: the second set of accumulators takes the registers of the second ring set)
  static constexpr int NB0 = EPT <= 4 ? (RC ? 3 : 4) : ((EPT <= 8 && !(NCH >= 8 && WM >= 2)) ? 3 : 2), NB = NB0 < NCH ? NB0 : NCH;      // staging batches (chunks) in registers
  static constexpr int NPW = RD * KR * WN, NPS = NB * (EPT + 4), NP = NPW > NPS ? NPW : NPS;
  static_assert(TH * TW == 16 * WM && TH >= 1 && TH <= H && TW <= W && TW >= 4, "tile = 16 * WM pixels");
  static_assert(C1 % 128 == 0 && C2 % 128 == 0 && CG % 16 == 0 && NSL <= 4 && COUT % (16 * WN) == 0, "channels");
  static_assert(!(POOL && RC), "a block's conv1 has no pooling epilogue");
  static_assert(LDS_BYTES <= 163840, "LDS");
};

template <int HL, int C1, int C2, int COUT, int TWL, int WM, int WN, bool POOL, bool RC>
SF_DEV void conv3s_body(const FConvArgs& a, const int bid) {
  using Gm = Conv3sGeom<HL, C1, C2, COUT, TWL, WM, WN, POOL, RC>;
  constexpr int C = Gm::C, KR = Gm::KR;
  constexpr int H = Gm::H, W = Gm::W, HW = Gm::HW, TW = Gm::TW, TH = Gm::TH, FR = Gm::FR, FW = Gm::FW, SW = Gm::SW, NPX = Gm::NPX, EPT = Gm::EPT;
  constexpr int CC = Gm::CC, NCH = Gm::NCH, PSTR = Gm::PSTR, BUF = Gm::BUF, F = Gm::F, FT = Gm::FT, RD = Gm::RD, NB = Gm::NB, NP = Gm::NP;
  constexpr int NTILES = Gm::NTILES, MTI = Gm::MTI, CG = Gm::CG, NCF = Gm::NCF, NMF = Gm::NMF, SCNT = Gm::SCNT, NSL = Gm::NSL;
  constexpr bool FULLW = Gm::FULLW;
  constexpr int NT = 512, NWM = 4;
  SF_DYN_LDS(lds);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = sf_uniform(tid >> 6);
  const bool mx_role = wave < NWM;

  // ---- which tile: workgroups t, t + 8, ... share an XCD; XCD x owns the n-tiles == x (mod 8) of ALL pixel tiles (fconv_tile_of, R = 1)
  const int MT = a.B * MTI;
  int mt, nt;
  if (NTILES % 8 == 0) {
    const int x = bid & 7, j = bid >> 3;
    const int q = j / MT;
    mt = j - q * MT;
    nt = x + 8 * q;
  } else {
    nt = bid % NTILES;
    mt = bid / NTILES;
  }
  const int b = mt / MTI, ti = mt - b * MTI;
  const int trow = ti / (W / TW), tcol = ti - trow * (W / TW);
  const int row0 = trow * TH, col0 = tcol * TW;
  const int mb = b * HW;
  float* misc = reinterpret_cast<float*>(lds + Gm::MISC_OFF);       // [8][2] = (mean, rstd) of the image's groups

  // ---- (1) statistics slots of group `wave` of image b: the head of the critical path (slots -> statistics -> first staged chunk)
  f32x2 sl[NSL];
  float slsc[NSL];                                       // scale of the slot's source (sums x scale, sums of squares x scale^2)
#pragma unroll
  for (int u = 0; u < NSL; ++u) {
    int i = lane + u * 64;
    if (i > SCNT - 1) i = SCNT - 1;
    const int mf = i / NCF, cfa = wave * NCF + (i - mf * NCF);
    const bool first = C2 == 0 || cfa < C1 / 16;           // the group's 16-channel fragments may lie in either source (a group can straddle the seam)
    const float* sp = first ? a.s1.slots + ((long)(b * NMF + mf) * (C1 / 16) + cfa) * 2
                            : a.s2.slots + ((long)(b * NMF + mf) * ((C2 ? C2 : 16) / 16) + (cfa - C1 / 16)) * 2;
    sl[u] = *reinterpret_cast<const f32x2*>(sp);
    slsc[u] = first ? 1.0f : a.s2.scale;
  }

  // ---- (2) ONE register pool for both roles (declared apart, the compiler allocates their sum): the weight ring of a matrix wave,
  //          the staging batches (EPT activations + gamma, beta, scale, shift of the chunk) of a staging wave
  f32x4 pool[NP];
  // matrix wave w: sub-chunk w of every chunk; k-step (tap, chunk c, w) of fragment nf sits at ((nf * KS) + tap * C / 32 + 4 c + w) * 64 + lane
  const bf16x8* wbase[WN];
#pragma unroll
  for (int ni = 0; ni < WN; ++ni) wbase[ni] = a.w + ((long)(nt * WN + ni) * Gm::KS + (mx_role ? wave : 0)) * 64 + lane;
  // RC: k-step (chunk c, sub-chunk w) of the res_conv's fragment nf sits at (nf * C / 32 + 4 c + w) * 64 + lane
  const bf16x8* rcw[WN];
#pragma unroll
  for (int ni = 0; ni < WN; ++ni) rcw[ni] = RC ? a.rc_w + ((long)(nt * WN + ni) * (C / 32) + (mx_role ? wave : 0)) * 64 + lane : nullptr;
  auto rcload = [&](int c, int ni) -> f32x4 { return __builtin_bit_cast(f32x4, __builtin_nontemporal_load(&rcw[ni][4 * c * 64])); };
  auto wload = [&](int c, int tap, int ni) -> f32x4 {
    return __builtin_bit_cast(f32x4, __builtin_nontemporal_load(&wbase[ni][(tap * (C / 32) + 4 * c) * 64]));
  };
  // staging thread: float4 channel chunk tcx of every 128-channel chunk, frame pixels tp, tp + 8, ...
  const int ts = tid - NWM * 64;
  const int tcx = ts & 31, tp = ts >> 5;
  int spix[EPT];                                         // source pixel of element e (a safe pixel when dead)
  int loff[EPT];                                         // LDS byte offset of element e inside a frame buffer (the spare pixel when dead)
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int pi = tp + e * 8;
    const int fr = pi / SW, fxs = pi - fr * SW;
    const int r = row0 - 1 + fr, x = col0 + fxs - (FULLW ? 0 : 1);
    const bool in = pi < NPX && r >= 0 && r < H && x >= 0 && x < W;
    spix[e] = mb + (in ? r * W + x : row0 * W + col0);
    loff[e] = (in ? fr * FW + fxs + (FULLW ? 1 : 0) : FR * FW) * PSTR + tcx * 8;
  }
  const float* ssrow = a.ss ? a.ss + (long)b * a.ss_stride : a.gamma;      // any valid address when there is no scale / shift
  const int shoff = a.ss ? C : 0;
  const float sc2 = a.s2.scale;
  auto issue = [&](int c, int vo) {
#pragma unroll
    for (int e = 0; e < EPT; ++e)                           // (c is a compile-time constant after unrolling: the chunk lies in ONE source)
      pool[vo + e] = c * CC < C1 ? *reinterpret_cast<const f32x4*>(a.s1.p + ((uint32_t)spix[e] * (uint32_t)C1 + (uint32_t)(c * CC + tcx * 4)))
                                 : *reinterpret_cast<const f32x4*>(a.s2.p + ((uint32_t)spix[e] * (uint32_t)(C2 ? C2 : 4) + (uint32_t)(c * CC - C1 + tcx * 4)));      // unsigned 32-bit element offsets: scalar base + one VGPR per load (signed ones become 64-bit VGPR pairs kept alive across the chunks)
    const int cg = c * CC + tcx * 4;
    pool[vo + EPT + 0] = *reinterpret_cast<const f32x4*>(a.gamma + cg);
    pool[vo + EPT + 1] = *reinterpret_cast<const f32x4*>(a.beta + cg);
    pool[vo + EPT + 2] = *reinterpret_cast<const f32x4*>(ssrow + cg);
    pool[vo + EPT + 3] = *reinterpret_cast<const f32x4*>(ssrow + shoff + cg);
  };
  if (mx_role) {
#pragma unroll
    for (int d = 0; d < RD; ++d)
#pragma unroll
      for (int tap = 0; tap < KR; ++tap)
#pragma unroll
        for (int ni = 0; ni < WN; ++ni) pool[(d * KR + tap) * WN + ni] = tap < 9 ? wload(d, tap, ni) : rcload(d, ni);
  } else {
#pragma unroll
    for (int j = 0; j < NB; ++j) issue(j, j * (EPT + 4));
  }

  // ---- (3) epilogue operands of the finalising waves: fragment f = wave, wave + 8, ... (NFW per wave)
  constexpr int NFW = (F + 7) / 8;
  float bv[NFW], rv[NFW][4], rcb[NFW];
  int orow[NFW];                                         // first output pixel row (global m) of the lane's 4 rows of fragment f
#pragma unroll
  for (int q = 0; q < NFW; ++q) {
    const int f = wave + q * 8, fc = f < F ? f : F - 1;
    const int mi = fc / WN, ni = fc - mi * WN;
    const int n = (nt * WN + ni) * 16 + (lane & 15);
    const int p0 = mi * 16 + (lane >> 4) * 4;
    orow[q] = mb + (row0 + p0 / TW) * W + col0 + (p0 & (TW - 1));
    bv[q] = a.bias[n];
    rcb[q] = RC ? (a.rc_bias ? a.rc_bias : a.bias)[n] : 0.0f;
    const float* rp = a.resid ? a.resid : a.bias;        // unconditional loads from a selected address (a load under `if` drains the queue at the merge)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = rp[a.resid ? (long)(orow[q] + r) * COUT + n : n];
      rv[q][r] = a.resid ? v : 0.0f;
    }
  }

  // ---- (4) zero padding: frame pixels outside the image, in both buffers; 8 threads per pixel
  for (int q = tid >> 3; q < FR * FW; q += NT / 8) {
    const int fr = q / FW, fx = q - fr * FW;
    const int r = row0 - 1 + fr, x = col0 - 1 + fx;
    if (r < 0 || r >= H || x < 0 || x >= W) {
      char* dst = lds + q * PSTR;
      for (int c8 = (tid & 7); c8 < CC / 8; c8 += 8) {
        *reinterpret_cast<bf16x8*>(dst + c8 * 16) = sf_zero8();
        *reinterpret_cast<bf16x8*>(dst + BUF + c8 * 16) = sf_zero8();
      }
    }
  }
  if (POOL && mx_role) {                                 // the w_eff table (KS k-steps x 32 operand-type values) into LDS
    const bf16x8* src = reinterpret_cast<const bf16x8*>(a.weff);
    bf16x8* dst = reinterpret_cast<bf16x8*>(lds + Gm::WEFF_OFF);
    for (int i = tid; i < Gm::KS * 4; i += NWM * 64) dst[i] = src[i];
  }

  // ---- (5) statistics: wave g sums the producer's (sum, sum of squares) slots of group g (8 groups = 8 waves)
  {
    float sm = 0.0f, sq = 0.0f;
#pragma unroll
    for (int u = 0; u < NSL; ++u) {
      const bool live = lane + u * 64 < SCNT;
      sm += live ? sl[u][0] * slsc[u] : 0.0f;
      sq += live ? sl[u][1] * (slsc[u] * slsc[u]) : 0.0f;
    }
    sm = sf_wave_sum(sm);
    sq = sf_wave_sum(sq);
    if (lane == 0) {
      const double rn = a.inv_n;
      const double mean = (double)sm * rn;
      double var = (double)sq * rn - mean * mean;
      if (var < 0.0) var = 0.0;
      misc[2 * wave] = (float)mean;
      misc[2 * wave + 1] = sf_rsqrt((float)var + a.eps);
    }
  }
  sf_sync();

  // ---- (6) the pipeline: phase 0 stages chunk 0; phase c + 1 multiplies chunk c while chunk c + 1 is staged; one barrier per phase
  auto consume = [&](int c, int vo) {
    const int gi = (c * CC + tcx * 4) / CG;
    const float mean = misc[2 * gi], rstd = misc[2 * gi + 1];
    // y = x * A + Bv  ==  ((x - mean) * rstd * gamma + beta) * (scale + 1) + shift.  Component by component, every intermediate pinned to a register
    // of its own: the f32x4 form of these lines (v_pk_mul_f32 / v_pk_fma_f32 fed straight from the ds_read_b64 pair that holds mean / rstd, through
    // op_sel broadcasts) gave wrong EVEN components in lanes 16-31 / 48-63 of some staging waves ON THE GPU -- 1-3 % output error in 3 of the 20
    // variants, moving with register allocation and with what the matrix wave of the same SIMD was doing; exact in the lane emulation.  Found by
    // reading the staged frames back (tools/exp/conv3s_{probe,kmask,lds_dump}.py, commit cfdf51f): the batch registers and the statistics were
    // right, the affine was not.  Each of {this scalar form, the packed form on splat vectors, mean / rstd passed through an empty asm first,
    // affine operands loaded here instead of with the batch} is exact on all variants; not root-caused further.
    f32x4 A, Bv;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float aj = pool[vo + EPT][j] * rstd;
      SF_USE_FROM_HERE(aj);
      const float srcs = c * CC < C1 ? 1.0f : sc2;             // the affine applies to the SCALED source value: fold the scale into the slope
      float scj = a.ss ? pool[vo + EPT + 2][j] + 1.0f : 1.0f, shj = a.ss ? pool[vo + EPT + 3][j] : 0.0f;
      SF_USE_FROM_HERE(scj);
      SF_USE_FROM_HERE(shj);
      float bj = (pool[vo + EPT + 1][j] - aj * mean) * scj + shj;
      SF_USE_FROM_HERE(bj);
      aj = aj * scj;
      SF_USE_FROM_HERE(aj);
      aj = aj * srcs;
      SF_USE_FROM_HERE(aj);
      A[j] = aj;
      Bv[j] = bj;
    }
    char* buf = lds + (c & 1) * BUF;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      f32x4 y = pool[vo + e] * A + Bv;
      const f32x4 t = y * -1.4426950408889634f;
      f32x4 ex;
#pragma unroll
      for (int j = 0; j < 4; ++j) ex[j] = sf_exp2(t[j]);
      ex = ex + 1.0f;
#pragma unroll
      for (int j = 0; j < 4; ++j) ex[j] = sf_rcp(ex[j]);
      y = y * ex;
      bf16x4 o;
      o[0] = (sf_opnd)y[0]; o[1] = (sf_opnd)y[1]; o[2] = (sf_opnd)y[2]; o[3] = (sf_opnd)y[3];
      *reinterpret_cast<bf16x4*>(buf + loff[e]) = o;
      if (RC) {                                            // the res_conv's A operand: the raw (scaled) value of the tile's own pixels
        const f32x4 rw = pool[vo + e] * (c * CC < C1 ? 1.0f : sc2);
        bf16x4 q;
        q[0] = (sf_opnd)rw[0]; q[1] = (sf_opnd)rw[1]; q[2] = (sf_opnd)rw[2]; q[3] = (sf_opnd)rw[3];
        // frame rows 1 .. TH are the tile's own: the raw buffer keeps the frame's pixel order minus its first row; everything else -> the spare pixel
        const int ro = loff[e] - FW * PSTR;
        const bool own = ro >= 0 && ro < TH * FW * PSTR;
        *reinterpret_cast<bf16x4*>(lds + Gm::RAW_OFF + (c & 1) * Gm::RAWB + (own ? ro : TH * FW * PSTR + tcx * 8)) = q;
      }
    }
  };
  f32x4 acc[WM][WN];
#pragma unroll
  for (int mi = 0; mi < WM; ++mi)
#pragma unroll
    for (int ni = 0; ni < WN; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 acc2[RC ? WM : 1][RC ? WN : 1];                   // RC: the res_conv's accumulators
#pragma unroll
  for (int mi = 0; mi < (RC ? WM : 1); ++mi)
#pragma unroll
    for (int ni = 0; ni < (RC ? WN : 1); ++ni) acc2[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 accl[POOL ? WM : 1];                              // POOL: context-logit fragment of every m-fragment (column 0 = the logit)
#pragma unroll
  for (int mi = 0; mi < (POOL ? WM : 1); ++mi) accl[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (mx_role) {
    int abase[WM];
#pragma unroll
    for (int mi = 0; mi < WM; ++mi) {
      const int p = mi * 16 + (lane & 15);
      abase[mi] = ((p / TW) * FW + (p & (TW - 1))) * PSTR + (lane >> 4) * 16 + wave * 64;
    }
    const char* weffL = lds + Gm::WEFF_OFF + wave * 64 + (lane >> 4) * 16;
    const bool col0l = (lane & 15) == 0;
    sf_sync();                                            // phase 0: chunk 0 is being staged
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const char* buf = lds + (c & 1) * BUF;
      const int d = c % RD;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int ky = tap / 3, kx = tap - ky * 3;
        bf16x8 fa[WM];
#pragma unroll
        for (int mi = 0; mi < WM; ++mi) fa[mi] = *reinterpret_cast<const bf16x8*>(buf + abase[mi] + (ky * FW + kx) * PSTR);
#pragma unroll
        for (int mi = 0; mi < WM; ++mi)
#pragma unroll
          for (int ni = 0; ni < WN; ++ni)
            acc[mi][ni] = sf_mfma16(fa[mi], __builtin_bit_cast(bf16x8, pool[(d * KR + tap) * WN + ni]), acc[mi][ni]);
        if (c + RD < NCH) {                               // (compile-time after unrolling) refill the slot with chunk c + RD
#pragma unroll
          for (int ni = 0; ni < WN; ++ni) pool[(d * KR + tap) * WN + ni] = wload(c + RD, tap, ni);
        }
        if (POOL) {                                       // one more MFMA per m-fragment: B = w_eff of the k-step in column 0, zero elsewhere
          bf16x8 wl = *reinterpret_cast<const bf16x8*>(weffL + (tap * (C / 32) + 4 * c) * 64);
          if (!col0l) wl = sf_zero8();
#pragma unroll
          for (int mi = 0; mi < WM; ++mi) accl[mi] = sf_mfma16(fa[mi], wl, accl[mi]);
        }
      }
      if (RC) {                                            // k-step 10 of this wave: raw operand, sub-chunk `wave`, the res_conv's weights
        const char* rb = lds + Gm::RAW_OFF + (c & 1) * Gm::RAWB + wave * 64 + (lane >> 4) * 16;
        bf16x8 fr_[WM];
#pragma unroll
        for (int mi = 0; mi < WM; ++mi) {
          const int p = mi * 16 + (lane & 15);
          fr_[mi] = *reinterpret_cast<const bf16x8*>(rb + ((p / TW) * FW + (p & (TW - 1)) + 1) * PSTR);      // (+ 1: the frame's left column)
        }
#pragma unroll
        for (int mi = 0; mi < WM; ++mi)
#pragma unroll
          for (int ni = 0; ni < WN; ++ni) acc2[mi][ni] = sf_mfma16(fr_[mi], __builtin_bit_cast(bf16x8, pool[(d * KR + 9) * WN + ni]), acc2[mi][ni]);
        if (c + RD < NCH) {
#pragma unroll
          for (int ni = 0; ni < WN; ++ni) pool[(d * KR + 9) * WN + ni] = rcload(c + RD, ni);
        }
      }
      sf_sync();
    }
  } else {
    consume(0, 0);
    sf_sync();                                            // phase 0 done
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      if (c + 1 < NCH) {
        if (c + NB < NCH) issue(c + NB, (c % NB) * (EPT + 4));       // batch c % NB was consumed in the previous phase
        consume(c + 1, ((c + 1) % NB) * (EPT + 4));
      }
      sf_sync();
    }
  }

  // ---- (7) epilogue: the 4 K-slices meet in LDS; wave w finalises fragments w, w + 8, ...
  float* red = reinterpret_cast<float*>(lds + Gm::RED_OFF);         // [matrix wave][frag][r][lane]
  if (mx_role) {
#pragma unroll
    for (int mi = 0; mi < WM; ++mi)
#pragma unroll
      for (int ni = 0; ni < WN; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[((wave * FT + mi * WN + ni) * 4 + r) * 64 + lane] = acc[mi][ni][r];
    if (POOL) {
#pragma unroll
      for (int mi = 0; mi < WM; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[((wave * FT + F + mi) * 4 + r) * 64 + lane] = accl[mi][r];
    }
    if (RC) {
#pragma unroll
      for (int mi = 0; mi < WM; ++mi)
#pragma unroll
        for (int ni = 0; ni < WN; ++ni)
#pragma unroll
          for (int r = 0; r < 4; ++r) red[((wave * FT + F + mi * WN + ni) * 4 + r) * 64 + lane] = acc2[mi][ni][r];
    }
  }
  sf_sync();
#pragma unroll
  for (int q = 0; q < NFW; ++q) {
    const int f = wave + q * 8;
    if (f < F) {
      const int mi = f / WN, ni = f - mi * WN;
      const int nf = nt * WN + ni;
      const int n = nf * 16 + (lane & 15);
      float sm = 0.0f, sqv = 0.0f, y4[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int idx = (f * 4 + r) * 64 + lane;
        float sacc = 0.0f;
#pragma unroll
        for (int w = 0; w < NWM; ++w) sacc += red[idx + w * FT * 256];
        const float y = sacc + bv[q] + rv[q][r];
        a.out[(long)(orow[q] + r) * COUT + n] = y;
        if (RC) {                                          // res_conv output of this fragment: plain rows [M][COUT]
          const int idx2 = ((F + f) * 4 + r) * 64 + lane;
          float s2 = 0.0f;
#pragma unroll
          for (int w = 0; w < NWM; ++w) s2 += red[idx2 + w * FT * 256];
          a.rc_out[(long)(orow[q] + r) * COUT + n] = s2 + rcb[q];
        }
        y4[r] = y;
        sm += y;
        sqv = fmaf(y, y, sqv);
      }
      const int mfrag = (mb >> 4) + ti * WM + mi;         // fragment index of (image, tile, m-fragment): consumers sum over an image's fragments
      if (a.slots_out) {
        sm = sf_wave_sum(sm);
        sqv = sf_wave_sum(sqv);
        if (lane == 0) {
          float* slo = a.slots_out + ((long)mfrag * (COUT / 16) + nf) * 2;
          slo[0] = sm;
          slo[1] = sqv;
        }
      }
      if (POOL) {
        // GlobalContext pooling of this fragment (imagen_pytorch.py:916-941): its 16 pixels are one softmax chunk.  The logit fragment's
        // column 0 sits in lanes 0 / 16 / 32 / 48 (rows 4 g .. 4 g + 3): every lane fetches the logits of ITS rows from lane (lane & 48)
        float l[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int idx = ((F + mi) * 4 + r) * 64 + lane;
          float sacc = 0.0f;
#pragma unroll
          for (int w = 0; w < NWM; ++w) sacc += red[idx + w * FT * 256];
          l[r] = sf_shfl(sacc, lane & 48);
        }
        float mx = fmaxf(fmaxf(l[0], l[1]), fmaxf(l[2], l[3]));
        mx = fmaxf(mx, sf_shfl_xor(mx, 16));
        mx = fmaxf(mx, sf_shfl_xor(mx, 32));
        float es = 0.0f, pv = 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = sf_exp(l[r] - mx);
          es += e;
          pv = fmaf(e, y4[r], pv);
        }
        es += sf_shfl_xor(es, 16); es += sf_shfl_xor(es, 32);
        pv += sf_shfl_xor(pv, 16); pv += sf_shfl_xor(pv, 32);
        if (lane < 16) a.pool_part[(long)mfrag * COUT + n] = pv;
        if (lane == 0 && nf == 0) {
          float* ms = a.pool_part + (long)(a.M >> 4) * COUT + (long)mfrag * 2;
          ms[0] = mx;
          ms[1] = es;
        }
      }
    }
  }
}

template <int HL, int C, int TWL, int WM, int WN, bool POOL>
using Conv3sGeom1 = Conv3sGeom<HL, C, 0, C, TWL, WM, WN, POOL, false>;

template <int HL, int C, int TWL, int WM, int WN, bool POOL>
SF_KERNEL(512) void k_conv3s(FConvArgs a) {
  sf_touch_kernarg<(int)sizeof(FConvArgs)>();
  conv3s_body<HL, C, 0, C, TWL, WM, WN, POOL, false>(a, (int)blockIdx.x);
}

// conv1 of a ResnetBlock whose input is the concat of two sources, with the block's res_conv in the same workgroups
template <int HL, int C1, int C2, int COUT, int TWL, int WM, int WN>
SF_KERNEL(512) void k_conv3s_rc(FConvArgs a) {
  sf_touch_kernarg<(int)sizeof(FConvArgs)>();
  conv3s_body<HL, C1, C2, COUT, TWL, WM, WN, false, true>(a, (int)blockIdx.x);
}

// k_conv_fused_pipe: the slot-GroupNorm 3x3 convs of the 8x8 / 16x16 / 32x32 levels with the normalisation and the matrix
// work OVERLAPPED inside the workgroup (same op, operands and results as k_conv_fused<.., FNORM_GN_SLOTS, 0, ..>).
//
// Why: in k_conv_fused both halves of the launch are serial and each is bound by a different unit -- the prologue
// (fp32 -> GroupNorm affine -> SiLU -> bf16 into LDS) by VALU issue (a wave64 op occupies its SIMD for 4 cycles; 6-8 us at
// these sizes because every n-tile re-normalises the rows it needs), the main loop by the weight stream from L2 / HBM
// (3.5-7.5 us).  Here the input channels are cut into chunks of 128, the frame is double-buffered in LDS, and the 8 waves
// take roles:
//     waves 4..7  (one per SIMD)  stage chunk p into buffer p & 1
//     waves 0..3  (one per SIMD)  multiply chunk p - 1 out of buffer (p - 1) & 1, weights through a 9-step register ring
// with one workgroup barrier per chunk.  A SIMD then issues VALU for its staging wave while its matrix wave waits on
// weights or feeds the matrix pipe: the launch costs ~max(staging, main) instead of their sum.  Statistics, affine table,
// epilogue, slots and context logits are those of k_conv_fused.
//
// What bounds the chunk loop (r03, measurement builds of this source -- their #if branches were removed in r04 --,
// profiles/r03_stage_experiment.log): the CU's vector-memory path, 64 B / clk shared by all 8 waves.  A chunk moves 48 KB of
// fp32 activations (3 haloed rows x 128 channels) and 72 KB of weights (36 k-steps x WN KiB) through it = 0.8 us of its 1.3-1.9 us;
// without the SiLU arithmetic the 512-channel 32x32 layer's loop falls 7.4 -> 5.3 us, without ANY arithmetic 5.2, without the
// weight stream 3.7, with every activation load hitting L1 (same instruction count) only 4.9.  An L2 prefetch of the weight slab
// at kernel entry changes nothing (the loads are not waiting on HBM), nor does sharing rows instead of n-tiles per XCD.
#pragma once
#include "fused_kernels.h"
// Measured r04 (profiles/r04_unet_fusions_ab.log, B = 1 eval in the sampler, 1.256 ms): plain instead of non-temporal weight loads in
// THIS kernel alone (its weights are re-read by the m-tiles of one XCD): 1.279 ms -- nt stays; s_setprio 1 / 2 on the staging waves
// (MI355X_MICROARCH.md: the younger half of an 8-wave workgroup loses VALU arbitration): 1.259 / 1.260 ms -- no effect, not kept.
// Chunk 0 staged by ALL 8 waves (the matrix waves have nothing to multiply until it is in LDS; their half of the elements held
// behind the weight ring): 1.231 -> 1.245 ms at B = 1, 1.415 -> 1.421 at B = 2 (profiles/r04_pipe_all_waves_stage_chunk0_ab.log):
// +28 VGPRs and eight more loads in front of the matrix waves' weight ring cost more than the shorter first phase saves.

template <int WM, int WN, int EPT, int NW, bool POOL = false, bool RC = false>
SF_DEV void conv_fused_pipe_body(const FConvArgs& a, const int bid) {
  constexpr int NT = NW * 64, NWM = NW / 2;             // NWM matrix waves, NT - 64 * NWM staging threads
  constexpr int CC = 128;                               // input channels per pipeline chunk (4 k-steps per tap)
  constexpr int KPW = 9 * (CC / 32) / NWM;              // k-steps per matrix wave and chunk (9 with 4 matrix waves)
  static_assert(NW == 8 && KPW * NWM == 36, "4 matrix waves x 9 k-steps cover one chunk of a 3x3 conv");
  SF_DYN_LDS(lds);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef SF_FCONV_TIMING
#define FP_STAMP(k) do { if (a.dbg && tid == 0) a.dbg[(long)bid * 8 + (k)] = sf_clock(); } while (0)
#else
#define FP_STAMP(k) do { } while (0)
#endif
  FP_STAMP(0);
  // ---- which tile (S == 1)
  const int MT = a.B * a.mt_per_img;
  int mt, nt;
  fconv_tile_of(a, bid, MT, mt, nt);
  const int b = mt / a.mt_per_img;
  const int row0 = (mt - b * a.mt_per_img) * a.TR;
  const int FW = a.W + 2, FR = a.TR + 2;
  const int HW = a.H * a.W;
  const long mb = (long)b * HW;
  const float sc1 = a.s1.scale, sc2 = a.s2.scale;
  const int NCH = a.C / CC;
  const int pstr = a.pix_stride;
  const bool mx_role = wave < NWM;

  float* tabA = reinterpret_cast<float*>(lds + a.tab_off);
  float* tabB = tabA + a.C;
  float* misc = reinterpret_cast<float*>(lds + a.misc_off);

  // ---- matrix waves: weight ring.  Step i of this wave in ANY chunk is (tap_i, 32-channel sub-chunk ccl_i); the ring slot
  // i is refilled with the next chunk's step i right after its use, so one ring revolution = one chunk.
  const bf16x8* wbase[WN];
#pragma unroll
  for (int ni = 0; ni < WN; ++ni) {
    int nf = nt * WN + ni;
    if (nf > a.n_frags - 1) nf = a.n_frags - 1;
    wbase[ni] = a.w + (long)nf * a.KS * 64 + lane;
  }
  int toff[KPW], woff[KPW];
#pragma unroll
  for (int i = 0; i < KPW; ++i) {
    const int jj = (mx_role ? wave : 0) * KPW + i, tap = jj >> 2, ccl = jj & 3;
    const int ky = tap >= 6 ? 2 : (tap >= 3 ? 1 : 0), kx = tap - ky * 3;
    toff[i] = (ky * FW + kx) * pstr + ccl * 64;
    woff[i] = (tap * a.cchunks + ccl) * 64;
  }
  auto wload = [&](int c, int i, int ni) -> bf16x8 {
    return __builtin_nontemporal_load(&wbase[ni][woff[i] + c * (CC / 32) * 64]);
  };
  // ONE register pool for both roles (the ring of the matrix waves, the two staging batches of the others): declared as
  // separate arrays the compiler keeps both alive across the role-independent code and allocates their SUM
  constexpr int NB = EPT <= 4 ? 4 : (EPT <= 6 ? 3 : 2);       // staging batches (chunks) in registers: ~16-24 float4 loads in flight per thread
  constexpr int KR = KPW + (RC ? 1 : 0);                      // ring steps per matrix wave and chunk: RC = one more, on the raw operand
  constexpr int NP = (KR * WN > NB * EPT) ? KR * WN : NB * EPT;
  f32x4 pool[NP];

  // ---- staging threads: a fixed float4 channel chunk (tcx) of every chunk, pixel lanes tp, tp + 8, ...
  const int ts = tid - NWM * 64;
  const int tcx = ts & 31, tp = ts >> 5;
  const int npx = FR << a.logW;                          // == EPT * 8 (host-checked)
  const int M0 = (int)mb + (row0 - 1) * a.W;
  const int pi_safe = 1 << a.logW;                       // first own row: always inside the image
  int fpx[EPT];                                          // LDS pixel of element e (the spare pixel for dead elements)
  int mxo[EPT];                                          // source pixel of element e
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int pi = tp + e * 8;
    const int fr = pi >> a.logW;
    const int r = row0 - 1 + fr;
    const bool in = pi < npx && r >= 0 && r < a.H;
    mxo[e] = M0 + (in ? pi : pi_safe);
    fpx[e] = in ? (pi + 2 * fr + 1) : FR * FW;
  }
  auto issue = [&](int c, const int vo) {
    const int cg = c * CC + tcx * 4;
    const bool first = cg < a.s1.C;
    const float* srcp = first ? a.s1.p + cg : a.s2.p + (cg - a.s1.C);
    const int srcld = first ? a.s1.C : a.s2.C;
#pragma unroll
    for (int e = 0; e < EPT; ++e) pool[vo + e] = *reinterpret_cast<const f32x4*>(srcp + mxo[e] * srcld);
  };
  auto consume = [&](int c, const int vo) {
    const int cg = c * CC + tcx * 4;
    const float scale = cg < a.s1.C ? sc1 : sc2;
    const f32x4 A = *reinterpret_cast<const f32x4*>(tabA + cg) * scale;
    const f32x4 Bv = *reinterpret_cast<const f32x4*>(tabB + cg);
    char* buf = lds + (c & 1) * a.buf_bytes + tcx * 8;
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
      *reinterpret_cast<bf16x4*>(buf + (long)fpx[e] * pstr) = o;
      if (RC) {                                            // the res_conv's A operand: the tile's OWN pixels (frame rows 1 .. TR), raw x scale
        const int pi = tp + e * 8, fr = pi >> a.logW;
        const bool own = fpx[e] != FR * FW && fr >= 1 && fr <= a.TR;
        const f32x4 rw = pool[vo + e] * scale;
        bf16x4 q;
        q[0] = (sf_opnd)rw[0]; q[1] = (sf_opnd)rw[1]; q[2] = (sf_opnd)rw[2]; q[3] = (sf_opnd)rw[3];
        *reinterpret_cast<bf16x4*>(lds + a.rc_off + (c & 1) * a.rc_buf_bytes + tcx * 8 + (long)(own ? pi - a.W : 16 * WM) * pstr) = q;
      }
    }
  };

  // ---- the statistics slots first (the critical path: slots -> statistics -> table -> first normalised chunk)
  const int Cg = a.C / a.G;
  const int ngs = a.G;                                   // all groups (S == 1)
  const int n_mf = HW >> 4, n_cf = Cg >> 4, scnt = n_mf * n_cf;
  const int cf1 = a.s1.C >> 4, cf2 = a.s2.C >> 4;
  f32x2 sl[4];
  float slsc[4];
  auto slot_loads = [&](int gi, int i0) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int i = i0 + u * 64;
      const bool live = i < scnt;
      if (!live) i = scnt - 1;
      const int mf = (int)fdiv((uint32_t)i, a.d_ncf), cfa = ((gi * Cg) >> 4) + (i - mf * n_cf);
      const long mfg = (long)b * n_mf + mf;
      const bool f1 = cfa < cf1;
      const float* base = f1 ? a.s1.slots : a.s2.slots;
      const long off = f1 ? (mfg * cf1 + cfa) : (mfg * cf2 + (cfa - cf1));
      sl[u] = *reinterpret_cast<const f32x2*>(base + off * 2);
      slsc[u] = live ? (f1 ? sc1 : sc2) : 0.0f;
    }
  };
  if (wave < ngs) slot_loads(wave, lane);
  // ---- first loads of every role go out before anything waits
  // RC: matrix wave w multiplies the raw operand's 32-channel sub-chunk w of every chunk with the res_conv's weights
  const bf16x8* rcw[WN];
#pragma unroll
  for (int ni = 0; ni < WN; ++ni) {
    int nf = nt * WN + ni;
    if (nf > a.n_frags - 1) nf = a.n_frags - 1;
    rcw[ni] = RC ? a.rc_w + ((long)nf * a.cchunks + (mx_role ? wave : 0)) * 64 + lane : nullptr;
  }
  auto rcload = [&](int c, int ni) -> bf16x8 {
    return __builtin_nontemporal_load(&rcw[ni][(long)c * (CC / 32) * 64]);
  };
  if (mx_role) {
#pragma unroll
    for (int i = 0; i < KPW; ++i)
#pragma unroll
      for (int ni = 0; ni < WN; ++ni) pool[i * WN + ni] = __builtin_bit_cast(f32x4, wload(0, i, ni));
    if (RC) {
#pragma unroll
      for (int ni = 0; ni < WN; ++ni) pool[KPW * WN + ni] = __builtin_bit_cast(f32x4, rcload(0, ni));
    }
  } else {
#pragma unroll
    for (int j = 0; j < NB - 1; ++j) issue(j < NCH ? j : NCH - 1, j * EPT);
  }
  FP_STAMP(6);
  // GroupNorm parameters of the whole input (C <= 4 * NT channels), statistics slots of this wave's group
  constexpr int TABN = 4;
  float tg[TABN], tb[TABN], tsc[TABN], tsh[TABN];
  {
    const float* ssrow = a.ss ? a.ss + (long)b * a.ss_stride : a.gamma;
    const int shoff = a.ss ? a.C : 0;
#pragma unroll
    for (int k = 0; k < TABN; ++k) {
      const int cl = tid + k * NT, cc = cl < a.C ? cl : a.C - 1;
      tg[k] = a.gamma[cc];
      tb[k] = a.beta[cc];
      tsc[k] = ssrow[cc];
      tsh[k] = ssrow[shoff + cc];
    }
  }
  // epilogue operands of the finalising waves (matrix waves 0 .. F-1)
  constexpr int F = WM * WN;
  const long m0 = mb + (long)row0 * a.W;
  const int my_mi = wave / WN, my_ni = wave - my_mi * WN;
  const int my_nf = nt * WN + my_ni;
  const bool fin = wave < F && my_nf < a.n_frags;
  const int n = my_nf * 16 + (lane & 15);
  const long mrow = m0 + my_mi * 16 + (lane >> 4) * 4;
  float bv = 0.0f, rv[4] = {0.f, 0.f, 0.f, 0.f}, wkv = 0.0f;
  if (fin && n < a.Cout) {
    if (a.logit_part) wkv = a.wk[n];
    if (a.bias) bv = a.bias[n];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long o = (mrow + r) * a.ldc + a.co_off + n;
      if (a.resid) rv[r] = a.resid[o];
      if (a.accum) rv[r] += a.out[o];
    }
  }
  FP_STAMP(7);

  // ---- zero the frame pixels outside the image in BOTH buffers (conv zero padding); 8 threads per pixel
  {
    const int npix = FR * FW;
    for (int q = tid >> 3; q < npix; q += NT / 8) {
      const int fr = q / FW, fx = q - fr * FW;
      const int r = row0 - 1 + fr, x = fx - 1;
      if (r < 0 || r >= a.H || x < 0 || x >= a.W) {
        char* dst = lds + (long)q * pstr;
        for (int c8 = (tid & 7); c8 < CC / 8; c8 += 8) {
          *reinterpret_cast<bf16x8*>(dst + c8 * 16) = sf_zero8();
          *reinterpret_cast<bf16x8*>(dst + a.buf_bytes + c8 * 16) = sf_zero8();
        }
      }
    }
  }
  FP_STAMP(1);
  // ---- statistics: one wave per group sums the producer's (sum, sum of squares) slots of image b
  for (int gi = wave; gi < ngs; gi += NW) {
    float sm = 0.0f, sq = 0.0f;
    for (int i0 = lane; i0 < scnt; i0 += 256) {
      if (gi != wave || i0 != lane) slot_loads(gi, i0);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        sm = fmaf(sl[u][0], slsc[u], sm);
        sq = fmaf(sl[u][1], slsc[u] * slsc[u], sq);
      }
    }
    sm = sf_wave_sum(sm);
    sq = sf_wave_sum(sq);
    if (lane == 0) {
      const double rn = a.inv_n;
      const double mean = (double)sm * rn;
      double var = (double)sq * rn - mean * mean;
      if (var < 0.0) var = 0.0;
      misc[16 + 2 * gi] = (float)mean;
      misc[17 + 2 * gi] = sf_rsqrt((float)var + a.eps);
    }
  }
  sf_sync();
#pragma unroll
  for (int k = 0; k < TABN; ++k) {
    const int cl = tid + k * NT;
    if (cl < a.C) {
      const int gi = (int)fdiv((uint32_t)cl, a.d_cg);
      const float mean = misc[16 + 2 * gi], rstd = misc[17 + 2 * gi];
      const float A = rstd * tg[k], sc = a.ss ? tsc[k] + 1.0f : 1.0f, sh = a.ss ? tsh[k] : 0.0f;
      tabA[cl] = A * sc;
      tabB[cl] = (tb[k] - mean * A) * sc + sh;
    }
  }
  sf_sync();
  FP_STAMP(2);

  // ---- the pipeline: NCH + 1 phases, one barrier each
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
  f32x4 accl[POOL ? WM : 1];                            // POOL: context-logit fragment of every m-fragment (column 0 = the logit)
#pragma unroll
  for (int mi = 0; mi < (POOL ? WM : 1); ++mi) accl[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (mx_role) {
    int abase[WM];
#pragma unroll
    for (int mi = 0; mi < WM; ++mi) {
      const int p = mi * 16 + (lane & 15);
      const int ty = p >> a.logW, tx = p - (ty << a.logW);
      abase[mi] = (ty * FW + tx) * pstr + (lane >> 4) * 16;
    }
    // POOL: B fragment of step i = w_eff[k-step][8 * (lane >> 4) .. + 7] in column 0 (lanes with lane & 15 == 0), zero elsewhere
    const char* weffL = lds + a.weff_off + (lane >> 4) * 16;
    const bool col0 = (lane & 15) == 0;
    if (POOL) {                                          // the w_eff table (KS x 32 bf16) into LDS, by the matrix waves while they
      const bf16x8* src = reinterpret_cast<const bf16x8*>(a.weff);      // wait for chunk 0 anyway; visible behind the phase-0 barrier
      bf16x8* dst = reinterpret_cast<bf16x8*>(lds + a.weff_off);
      for (int i = tid; i < a.KS * 4; i += NWM * 64) dst[i] = src[i];
    }
    sf_sync();                                           // phase 0: chunk 0 is being staged
    for (int c = 0; c < NCH; ++c) {
      const char* buf = lds + (c & 1) * a.buf_bytes;
      const int cn = c + 1 < NCH ? c + 1 : NCH - 1;      // the last revolution reloads the last chunk: loads stay unconditional
#pragma unroll
      for (int i = 0; i < KPW; ++i) {
        bf16x8 fa[WM];
#pragma unroll
        for (int mi = 0; mi < WM; ++mi) fa[mi] = *reinterpret_cast<const bf16x8*>(buf + abase[mi] + toff[i]);
#pragma unroll
        for (int mi = 0; mi < WM; ++mi)
#pragma unroll
          for (int ni = 0; ni < WN; ++ni) acc[mi][ni] = sf_mfma16(fa[mi], __builtin_bit_cast(bf16x8, pool[i * WN + ni]), acc[mi][ni]);
#pragma unroll
        for (int ni = 0; ni < WN; ++ni) pool[i * WN + ni] = __builtin_bit_cast(f32x4, wload(cn, i, ni));
        if (POOL) {                                        // one more MFMA per m-fragment, BEHIND the ring refill (the weight stream
          bf16x8 wl = *reinterpret_cast<const bf16x8*>(weffL + (woff[i] + c * (CC / 32) * 64));   // is what this loop waits on);
          if (!col0) wl = sf_zero8();                                                               // woff = k-step * 64 bytes as well
#pragma unroll
          for (int mi = 0; mi < WM; ++mi) accl[mi] = sf_mfma16(fa[mi], wl, accl[mi]);
        }
      }
      if (RC) {                                            // k-step 10 of this wave: raw operand, sub-chunk `wave`, res_conv weights
        const char* rb = lds + a.rc_off + (c & 1) * a.rc_buf_bytes + wave * 64 + (lane >> 4) * 16;
        bf16x8 fr_[WM];
#pragma unroll
        for (int mi = 0; mi < WM; ++mi) fr_[mi] = *reinterpret_cast<const bf16x8*>(rb + (mi * 16 + (lane & 15)) * pstr);
#pragma unroll
        for (int mi = 0; mi < WM; ++mi)
#pragma unroll
          for (int ni = 0; ni < WN; ++ni) acc2[mi][ni] = sf_mfma16(fr_[mi], __builtin_bit_cast(bf16x8, pool[KPW * WN + ni]), acc2[mi][ni]);
#pragma unroll
        for (int ni = 0; ni < WN; ++ni) pool[KPW * WN + ni] = __builtin_bit_cast(f32x4, rcload(cn, ni));
      }
      sf_sync();
    }
  } else {
    // chunk c lives in register batch c % NB; the batch freed by chunk c - 1 is refilled with chunk c + NB - 1 before chunk c
    // is normalised: NB - 1 chunks of loads are in flight behind the VALU work (one staging wave per SIMD has no partner
    // to hide a cold L2 round trip behind)
    for (int c = 0; c < NCH; c += NB) {
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        if (c + j < NCH) {
          issue(c + j + NB - 1 < NCH ? c + j + NB - 1 : NCH - 1, ((j + NB - 1) % NB) * EPT);
          consume(c + j, j * EPT);
          sf_sync();
        }
      }
    }
    sf_sync();                                           // phase NCH: the matrix waves finish the last chunk
  }
  FP_STAMP(3);
  FP_STAMP(4);

  // ---- epilogue: the NWM K-slices meet in LDS; wave f < F finalises fragment f
  constexpr int FT = F + (POOL ? WM : 0) + (RC ? F : 0);          // POOL: + one context-logit fragment per m-fragment; RC: + the res_conv's tile
  float* red = reinterpret_cast<float*>(lds + a.red_off);         // [matrix wave][frag][r][lane]
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
          for (int r = 0; r < 4; ++r) red[((wave * FT + F + (POOL ? WM : 0) + mi * WN + ni) * 4 + r) * 64 + lane] = acc2[mi][ni][r];
    }
  }
  const float rcb = (RC && fin && n < a.Cout && a.rc_bias) ? a.rc_bias[n] : 0.0f;
  sf_sync();
  if (fin) {
    const int f = wave;
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int idx = (f * 4 + r) * 64 + lane;
      float sacc = 0.0f;
#pragma unroll
      for (int w = 0; w < NWM; ++w) sacc += red[idx + w * FT * 256];
      v[r] = sacc;
    }
    if (a.logit_part) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float lp = v[r] * wkv;
        lp += sf_shfl_xor(lp, 1); lp += sf_shfl_xor(lp, 2); lp += sf_shfl_xor(lp, 4); lp += sf_shfl_xor(lp, 8);
        if ((lane & 15) == 0) a.logit_part[(long)my_nf * a.M + mrow + r] = lp;
      }
    }
    if (RC && n < a.Cout) {                              // res_conv output of this fragment: plain rows [M][Cout]
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int idx = ((F + (POOL ? WM : 0) + f) * 4 + r) * 64 + lane;
        float sacc = 0.0f;
#pragma unroll
        for (int w = 0; w < NWM; ++w) sacc += red[idx + w * FT * 256];
        a.rc_out[(mrow + r) * a.Cout + n] = sacc + rcb;
      }
    }
    float sm = 0.0f, sq = 0.0f;
    float y4[4] = {0.f, 0.f, 0.f, 0.f};
    if (n < a.Cout) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float y = v[r] + bv + rv[r];
        if (a.out_gelu) y = sf_gelu(y);
        a.out[(mrow + r) * a.ldc + a.co_off + n] = y;
        y4[r] = y;
        sm += y;
        sq = fmaf(y, y, sq);
      }
    }
    if (a.slots_out) {
      sm = sf_wave_sum(sm);
      sq = sf_wave_sum(sq);
      if (lane == 0) {
        float* slo = a.slots_out + (((m0 >> 4) + my_mi) * (long)(a.ldc >> 4) + (a.co_off >> 4) + my_nf) * 2;
        slo[0] = sm;
        slo[1] = sq;
      }
    }
    if (POOL) {
      // GlobalContext pooling of this fragment (imagen_pytorch.py:916-941): its 16 pixels are one softmax chunk.  The logit
      // fragment's column 0 sits in lanes 0 / 16 / 32 / 48 (rows 4 g .. 4 g + 3): every lane fetches the logits of ITS rows
      // from lane (lane & 48); max and sums over the 16 pixels = over r and the four 16-lane groups.
      float l[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int idx = ((F + my_mi) * 4 + r) * 64 + lane;
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
      const long mfrag = (m0 >> 4) + my_mi;
      if (lane < 16 && n < a.Cout) a.pool_part[mfrag * a.Cout + n] = pv;
      if (lane == 0 && my_nf == 0) {
        float* ms = a.pool_part + (long)(a.M >> 4) * a.Cout + mfrag * 2;
        ms[0] = mx;
        ms[1] = es;
      }
    }
  }
  FP_STAMP(5);
#undef FP_STAMP
}

template <int WM, int WN, int EPT, int NW, bool POOL = false>
SF_KERNEL(NW * 64) void k_conv_fused_pipe(FConvArgs a) {
  sf_touch_kernarg<(int)sizeof(FConvArgs)>();
  conv_fused_pipe_body<WM, WN, EPT, NW, POOL>(a, (int)blockIdx.x);
}

// conv1 + res_conv of one ResnetBlock in the SAME workgroups (r04): see FConvArgs.rc_w.  Replaces k_conv_fused_pipe_pair -- two
// sets of workgroups, i.e. two rounds on the chip at one workgroup per CU -- where the tile has the registers for it (WM <= 2).
template <int WM, int WN, int EPT, int NW>
SF_KERNEL(NW * 64) void k_conv_fused_pipe_rc(FConvArgs a) {
  sf_touch_kernarg<(int)sizeof(FConvArgs)>();
  conv_fused_pipe_body<WM, WN, EPT, NW, false, true>(a, (int)blockIdx.x);
}

// conv1 (pipelined) || res_conv (plain 1x1, k_conv_fused body) of one ResnetBlock in one launch: see k_conv_fused_pair.
template <int WM, int WN, int EPT, int NW>
SF_KERNEL(NW * 64) void k_conv_fused_pipe_pair(FConvPairArgs p) {
  sf_touch_kernarg<(int)sizeof(FConvPairArgs)>();
  if ((int)blockIdx.x < p.grid_b) conv_fused_body<WM, WN, (WM * WN == 1 ? 12 : 8), FNORM_NONE, 0, NW>(p.b, (int)blockIdx.x);
  else conv_fused_pipe_body<WM, WN, EPT, NW>(p.a, (int)blockIdx.x - p.grid_b);
}

// A ResnetBlock's res_conv beside the (16-workgroup) GlobalContext pooling launch of the same block (r04, the 4x4 level): in the
// conv1 || res_conv pair of that level the 64 res_conv workgroups were a second round behind conv1's 256 AND evaluated the whole
// lazy split-K source (six loads per element, 458 KB per workgroup): 24 us per pair against 11.6 for conv1 alone.  The result is
// only needed by the gate launch, so the res_conv runs here, on the source conv1 has materialised meanwhile, on CUs the pooling
// leaves idle.  512-thread workgroups; a pooling workgroup retires its upper half at once (ended waves leave the barrier count).
#include "fused_gca.h"
template <int WM, int WN, int D, int NW>
SF_KERNEL(NW * 64) void k_gca_pool_rc(GcaPoolArgs pa, FConvArgs b, int grid_b) {
  sf_touch_kernarg<(int)(sizeof(GcaPoolArgs) + sizeof(FConvArgs))>();
  if ((int)blockIdx.x < grid_b) conv_fused_body<WM, WN, D, FNORM_NONE, 0, NW>(b, (int)blockIdx.x);
  else if (threadIdx.x < 256) gca_pool_body(pa, (int)blockIdx.x - grid_b);
}

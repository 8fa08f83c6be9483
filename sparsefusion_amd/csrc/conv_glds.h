// k_conv_glds: the large-M implicit GEMM of conv_lds.h (workgroup tile 128 pixels x 16*BNF channels; weight fragments in MFMA lane
// order = 1 KiB of conflict-free LDS each, activation fragments as swizzled 128-byte pixel rows) with its staging rebuilt around
// LDS-DMA and wave specialisation.
//
// Why: k_conv_lds prefetches ONE stage into registers, so every stage of 64 input channels ends in a full L2 round trip before
// its ds_write (measured 1.3 us per stage against 0.1 us of MFMA work at one workgroup per CU: 200-320 TFLOP/s on the SD-VAE
// layers).  Here a stage never touches a register: global_load_lds_dwordx4 writes each fragment straight into a ring of NST
// stage buffers, NST - 1 stages are in flight while one is multiplied.  8 waves, two per SIMD: waves 0..3 (2 x 2, 64 x 8*BNF
// each) only ds_read and multiply; waves 4..7 only compute addresses and issue the LDS-DMA, so their integer work runs beside the
// other wave's MFMAs instead of between them.  One raw barrier per stage:
//
//     loaders                                         matrix waves
//     wait  vmcnt(G * (NST - 2))   own fragments of stage s have landed          (G = LDS-DMA loads per loader wave and stage)
//     barrier  <------------------------------------>  barrier     (after lgkmcnt(0): the reads of stage s - 1 have returned)
//     issue stage s + NST - 1      into the slot of    ds_read + MFMA stage s
//                                  stage s - 1
//
// A operand: line-shaped loads + source-side swizzle (see the loader below); Cin must be a multiple of 64 (else k_conv_lds).
// The loads are asm statements hipcc does not count (sf_dev.h: sf_glds16); no compiler-visible vector load is in flight while they
// are (the loaders have no other loads; the epilogue operands belong to the matrix waves).  Zero padding: a lane whose tap falls
// outside the image (or whose pixel is past the end) reads its 16 bytes of the zero line sf_zero128 instead -- the source address is
// per lane, only the LDS destination is wave-linear.  bf16 (operand-type) activations only; fp32 inputs stay on k_conv_lds, which
// converts in registers.  Same fragment order, same accumulation order, same epilogue as k_conv_lds: the results are bit-identical
// (tests/test_hostemu_conv_lds.py, tests/test_gpu_unet_ops.py); the GroupNorm partial sums are taken in another (also fixed) order.
// Dynamic LDS: NST * (16 + 2 * BNF) KiB (128 KiB at BNF = 8, NST = 4: one workgroup per CU, the ring is the latency hiding).
// Bounds at this tile: per 32-deep k-step the matrix waves read 32 KiB of fragments (128 clk of the LDS port at ds_read_b128's
// 256 B/clk) and the DMA writes 16 KiB against 256 clk of MFMA time -- neither is what was measured to bound it: the A operand's
// 16 KiB per stage through the L2 -> L1 path is (profiles/r03_conv_glds_experiments.log), which is why 3x3 layers go to
// k_conv3_halo (conv_halo.h) and this kernel keeps the 1x1 / stride-2 layers.
#pragma once
#include "conv_lds.h"

// ---- epilogue of the 8-wave tiles (k_conv_glds, k_conv3_halo): the 128 x 16*BNF tile goes through LDS once so that every global
// access is a full float4 of one row (the fragment layout gives a lane one column of four rows: 64-byte pieces per store
// instruction, 64 stores per lane; measured 42 -> 33 us on the 128x128 256->256 layer).  All 8 waves write: thread t owns the
// float4 column c4 = t % (COLS / 4) of rows t / (COLS / 4) + k * (512 / (COLS / 4)).  pix(row) = the row's pixel index in the
// output tensor or -1.  Host-checked: Cout, ldc, co_off are multiples of 4.
template <int BNF, int LDS_BYTES, bool GN, class Pix>
SF_DEV void conv_tile_epilogue(const ConvArgs& a, char* lds, const f32x4 (&acc)[4][BNF / 2], const bool loader, const int wm, const int wn,
                               const int lane, const int nt, const int mt, double* __restrict__ gn_part, const int gn_cg, Pix pix,
                               const int grp = 0) {
  constexpr int WNF = BNF / 2;
  constexpr int COLS = 16 * BNF, F4 = COLS / 4, RPP = 512 / F4, PITCH = COLS + 4;      // pitch = 4 mod 8 floats: the four row groups of a
  static_assert(128 * PITCH * 4 <= LDS_BYTES, "the output tile fits the staging buffers");   // fragment store land on disjoint banks
  float* ot = reinterpret_cast<float*>(lds);
  sf_lds_barrier();                                // every read of the last stage has returned; no LDS-DMA is in flight
  if (!loader) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int n = 0; n < WNF; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          ot[(wm * 64 + i * 16 + (lane >> 4) * 4 + r) * PITCH + (wn * WNF + n) * 16 + (lane & 15)] = acc[i][n][r];
  }
  sf_sync();
  const int tid = threadIdx.x;
  const int c4 = tid % F4, r0 = tid / F4;
  const int col = nt * COLS + c4 * 4;
  const bool cok = col < a.Cout;
  if (a.groups > 1) {                              // split-K (r06, k_conv_glds only): the partial tile -> workspace [grp][m][npad], as k_conv_igemm
    const long Mrows = (long)a.B * a.Ho * a.Wo;    // leaves it (no bias: the reduction adds it); pix(row) is the row index m here
    if (col < a.npad) {
#pragma unroll 4
      for (int row = r0; row < 128; row += RPP) {
        const long m = pix(row);
        if (m >= 0) *reinterpret_cast<f32x4*>(a.ws + (grp * Mrows + m) * a.npad + col) = *reinterpret_cast<const f32x4*>(ot + row * PITCH + c4 * 4);
      }
    }
    return;
  }
  f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
  if (cok && a.bias) bv = *reinterpret_cast<const f32x4*>(a.bias + col);
  if (a.pixshuf) {                                 // SiLU + PixelShuffle(2) of an Upsample (conv_igemm.h's epilogue): the float4 = the 2 x 2 output
    if (cok) {                                     // pixels of channel col / 4; adjacent threads write adjacent channels
      const int hw = a.Ho * a.Wo;
#pragma unroll 4
      for (int row = r0; row < 128; row += RPP) {
        const long m = pix(row);
        if (m < 0) continue;
        const f32x4 v = *reinterpret_cast<const f32x4*>(ot + row * PITCH + c4 * 4) + bv;
        const int b = (int)(m / hw), rr = (int)(m - (long)b * hw);
        const int oy = rr / a.Wo, ox = rr - oy * a.Wo;
        float* o = a.out + (((long)b * (2 * a.Ho) + 2 * oy) * (2 * a.Wo) + 2 * ox) * a.ldc + a.co_off + (col >> 2);
        o[0] = sf_silu(v[0]);
        o[a.ldc] = sf_silu(v[1]);
        o[(long)2 * a.Wo * a.ldc] = sf_silu(v[2]);
        o[(long)(2 * a.Wo + 1) * a.ldc] = sf_silu(v[3]);
      }
    }
    return;
  }
  float gs = 0.0f, gq = 0.0f;                      // (GN) sums of this thread's four columns over its rows
#pragma unroll 4
  for (int row = r0; row < 128; row += RPP) {
    const long m = pix(row);
    if (!cok || m < 0) continue;
    f32x4 v = *reinterpret_cast<const f32x4*>(ot + row * PITCH + c4 * 4) + bv;
    const long o = m * a.ldc + a.co_off + col;
    if (a.resid) v += *reinterpret_cast<const f32x4*>(a.resid + o);
    if (a.accum) v += *reinterpret_cast<const f32x4*>(a.out + o);
    if (a.relu == 1) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.0f);
    } else if (a.relu == 2) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = 0.5f * v[j] * (1.0f + erff(v[j] * 0.70710678118654752f));
    }
    *reinterpret_cast<f32x4*>(a.out + o) = v;
    if (a.ws) {                                    // operand-type twin of the output, dense [pixel][Cout]: the A operand of a following conv
      bf16x4 tw;
      tw[0] = (sf_opnd)v[0]; tw[1] = (sf_opnd)v[1]; tw[2] = (sf_opnd)v[2]; tw[3] = (sf_opnd)v[3];
      *reinterpret_cast<bf16x4*>(reinterpret_cast<sf_opnd*>(a.ws) + m * a.Cout + col) = tw;
    }
    if (GN) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { gs += v[j]; gq = fmaf(v[j], v[j], gq); }
    }
  }
  if (GN) {
    // per (pixel tile, group) partial sums in a fixed order: the threads of one float4 column (same c4, RPP apart in t), then the
    // gn_cg / 4 float4 columns of the group -- no atomics, reproducible bit for bit
    SF_SHARED float red[2][512];
    red[0][tid] = gs; red[1][tid] = gq;
    sf_sync();
    const int gpt = COLS / gn_cg;                  // groups in this channel tile
    if (tid < gpt) {
      const int gcol = nt * COLS + tid * gn_cg;
      if (gcol < a.Cout) {
        double s = 0.0, q = 0.0;
        for (int f = tid * (gn_cg >> 2); f < (tid + 1) * (gn_cg >> 2); ++f)
          for (int rr = 0; rr < RPP; ++rr) { s += (double)red[0][rr * F4 + f]; q += (double)red[1][rr * F4 + f]; }
        double* o = gn_part + ((long)mt * (a.Cout / gn_cg) + gcol / gn_cg) * 2;
        o[0] = s; o[1] = q;
      }
    }
  }
}

template <int BNF, int NST, bool GN>
SF_DEV void conv_glds_body(const ConvArgs& a, double* __restrict__ gn_part, const int gn_cg) {
  static_assert(NST == 3 || NST == 4, "ring depth 3 or 4 (the tail of the counted waits is written out for these)");
  static_assert(BNF == 4 || BNF == 8, "64 or 128 output channels per workgroup");
  constexpr int WNF = BNF / 2;                  // n-fragments per matrix wave (arranged 2 x 2)
  constexpr int BLD = BNF / 4;                  // B fragments each loader wave stages per k-step
  constexpr int G = 4 + 2 * BLD;                // LDS-DMA loads per loader wave and stage
  constexpr int A_BYTES = 8 * 2 * 1024;         // [m-frag 8][k-step 2][lane 64] x 16 B
  constexpr int STAGE = A_BYTES + BNF * 2 * 1024;
  SF_DYN_LDS(lds);
  const int lane = threadIdx.x & 63, wave = sf_uniform((int)(threadIdx.x >> 6));
  const bool loader = wave >= 4;                // waves 4..7 stage, waves 0..3 multiply: one of each per SIMD
  const int wm = (wave >> 1) & 1, wn = wave & 1;
  // XCD-aware tile order (8 XCDs, workgroups are dealt round-robin)
  const int tiles = a.m_tiles * a.n_tiles;
  const int grp = a.groups > 1 ? sf_uniform((int)blockIdx.x / tiles) : 0;      // split-K (r06): group g takes stages [g S / groups, (g + 1) S / groups)
  const int bid = (int)blockIdx.x - grp * tiles;
  int t = bid;
  if (tiles % 8 == 0) t = (bid & 7) * (tiles >> 3) + (bid >> 3);
  const int nt = t % a.n_tiles, mt = t / a.n_tiles;
  const int M = a.B * a.Ho * a.Wo;
  const int S_all = a.KS >> 1;                    // stages of one tap x 64 channels (KS is even: Cin % 64 == 0)
  const int s_lo = a.groups > 1 ? (int)((long)grp * S_all / a.groups) : 0;
  const int S = (a.groups > 1 ? (int)((long)(grp + 1) * S_all / a.groups) : S_all) - s_lo;      // stages of THIS workgroup

  f32x4 acc[4][WNF];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int n = 0; n < WNF; ++n) acc[i][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (loader) {
    // loader wave lw stages m-fragments 2*lw, 2*lw+1 (16 pixels each) and BLD n-fragments of every stage.  A stage is ONE tap and
    // 64 input channels = one full 128-byte line per pixel: an LDS-DMA instruction covers 8 pixels x 128 B, eight adjacent lanes
    // per line (a fragment-shaped load -- 16 pixels x 64 B, adjacent lanes on different lines -- costs the texture path a tag
    // look-up per LANE: measured 3 500 clk per stage, 4x the line-shaped one).  The LDS image of an m-fragment is therefore
    // [16 pixels][8 chunks of 16 B], and because the DMA destination is lane-linear the bank swizzle sits on the SOURCE: the
    // lane writing chunk position c of pixel m fetches the line's chunk c ^ ((m >> 1) & 7); the matrix waves read chunk
    // (4u + kgroup) ^ ((m >> 1) & 7), which is conflict-free over ds_read_b128's four 16-lane groups (MI355X_MICROARCH.md, LDS).
    const int lw = wave - 4;
    int py[4], px[4];
    long pbase[4];
    bool pv[4];
    const int Hs = a.H >> a.ups, Ws = a.W >> a.ups;        // stored input dims (nearest x2 upsampling is folded into the addressing)
#pragma unroll
    for (int q = 0; q < 4; ++q) {                          // q = 2 * fragment + half: pixels (lane >> 3) + 8 * half of fragment 2*lw + (q >> 1)
      const int m = (mt * 8 + 2 * lw + (q >> 1)) * 16 + (q & 1) * 8 + (lane >> 3);
      pv[q] = m < M;
      const int mm = pv[q] ? m : 0;
      const int pb = mm / (a.Ho * a.Wo);
      const int r = mm - pb * (a.Ho * a.Wo);
      const int oy = r / a.Wo;
      py[q] = oy * a.stride - a.pad;
      px[q] = (r - oy * a.Wo) * a.stride - a.pad;
      pbase[q] = (long)pb * Hs * Ws;
    }
    int coff[2];                                           // channel offset of this lane's chunk inside the 64-channel line, per half
#pragma unroll
    for (int h = 0; h < 2; ++h) coff[h] = ((lane & 7) ^ (((lane >> 4) + 4 * h) & 7)) * 8;
    const sf_opnd* in = reinterpret_cast<const sf_opnd*>(a.in);
    const bf16x8* wbase[BLD];
#pragma unroll
    for (int j = 0; j < BLD; ++j) {
      const int nf = min(nt * BNF + BLD * lw + j, a.n_frags - 1);
      wbase[j] = a.w + (long)nf * a.KS * 64 + lane;
    }
    // stages are issued strictly in order: (tap, 64-channel chunk) and the ring slot advance incrementally
    const int cpairs = a.cchunks >> 1;                     // host-checked: Cin is a multiple of 64
    const int tap0 = s_lo / cpairs;
    int i_ks = 2 * s_lo, i_cc = s_lo - tap0 * cpairs, i_ky = tap0 / a.kw, i_kx = tap0 - (tap0 / a.kw) * a.kw, i_buf = 0;
    auto issue_next = [&]() {
      char* sb = lds + i_buf * STAGE;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int iy = py[q] + i_ky, ix = px[q] + i_kx;
        const bool ok = pv[q] & (iy >= 0) & (iy < a.H) & (ix >= 0) & (ix < a.W);          // no short circuit: one select, no branch
        const int cy = min(max(iy, 0), a.H - 1) >> a.ups, cx = min(max(ix, 0), a.W - 1) >> a.ups;
        const sf_opnd* p = in + ((pbase[q] + (long)cy * Ws + cx) * a.Cin + i_cc * 64 + coff[q & 1]);
        const void* src = ok ? static_cast<const void*>(p) : static_cast<const void*>(sf_zero128 + (lane & 7) * 4);
        sf_glds16(sb + (2 * lw) * 2048 + q * 1024, src);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int j = 0; j < BLD; ++j) sf_glds16(sb + A_BYTES + ((BLD * lw + j) * 2 + u) * 1024, wbase[j] + (long)(i_ks + u) * 64);
      i_ks += 2;
      if (++i_cc == cpairs) {
        i_cc = 0;
        if (++i_kx == a.kw) { i_kx = 0; ++i_ky; }
      }
      if (++i_buf == NST) i_buf = 0;
    };
    for (int p = 0; p < NST - 1 && p < S; ++p) issue_next();
    for (int s = 0; s < S; ++s) {
      // stages newer than s in flight here: min(S - 1 - s, NST - 2); everything older than those has to have landed
      const int newer = S - 1 - s;
      if (newer >= NST - 2) sf_vmcnt<G * (NST - 2)>();
      else if (NST == 4 && newer == 1) sf_vmcnt<G>();
      else sf_vmcnt<0>();
      sf_lds_barrier();                             // stage s is in LDS for everyone; the matrix waves are done with stage s - 1
      if (s + NST - 1 < S) issue_next();            // ... whose slot takes stage s + NST - 1
    }
    sf_vmcnt<0>();
    sf_glds_done();
  } else {
    int r_buf = 0;
    int aoff[2];                                    // this lane's 16 bytes of an A fragment: pixel lane & 15, swizzled chunk of k-step u
#pragma unroll
    for (int u = 0; u < 2; ++u) aoff[u] = (lane & 15) * 128 + (((u * 4 + (lane >> 4)) ^ (((lane & 15) >> 1) & 7)) * 16);
    SF_LGKM0();                                     // no scalar load pending into the loop: its LDS waits can then be counted ones
    for (int s = 0; s < S; ++s) {
      sf_lds_barrier();                             // (waits for this wave's reads of stage s - 1, then meets the loaders)
      const char* sb = lds + r_buf * STAGE;
      // fragment reads in the order the MFMAs need them (A0, all B, then the other A rows), k-step 1 behind k-step 0
      bf16x8 fa[2][4], fb[2][WNF];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        fa[u][0] = *reinterpret_cast<const bf16x8*>(sb + (wm * 4) * 2048 + aoff[u]);
#pragma unroll
        for (int n = 0; n < WNF; ++n) fb[u][n] = *reinterpret_cast<const bf16x8*>(sb + A_BYTES + ((wn * WNF + n) * 2 + u) * 1024 + lane * 16);
#pragma unroll
        for (int i = 1; i < 4; ++i) fa[u][i] = *reinterpret_cast<const bf16x8*>(sb + (wm * 4 + i) * 2048 + aoff[u]);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int n = 0; n < WNF; ++n) acc[i][n] = sf_mfma16(fa[u][i], fb[u][n], acc[i][n]);
      // schedule: four reads, then one read behind each MFMA -- the first MFMA starts after two fragments with two more on their
      // way (the compiler's own order waits for a batch of ten while all four matrix waves queue on the LDS port)
      constexpr int NRD = 2 * (4 + WNF);
      SF_SCHED_GROUP(0x100, 4);
#pragma unroll
      for (int q = 0; q < NRD - 4; ++q) { SF_SCHED_GROUP(0x008, 1); SF_SCHED_GROUP(0x100, 1); }
      SF_SCHED_GROUP(0x008, 8 * WNF - (NRD - 4));
      if (++r_buf == NST) r_buf = 0;
    }
  }
  conv_tile_epilogue<BNF, NST * STAGE, GN>(a, lds, acc, loader, wm, wn, lane, nt, mt, gn_part, gn_cg,
                                           [&](int row) -> long { const int m = mt * 128 + row; return m < M ? (long)m : -1L; }, grp);
}

template <int BNF, int NST, bool GN>
SF_KERNEL(512, 1) void k_conv_glds(ConvArgs a, double* __restrict__ gn_part, int gn_cg) {
  sf_touch_kernarg<(int)sizeof(ConvArgs)>();
  conv_glds_body<BNF, NST, GN>(a, gn_part, gn_cg);
}

static inline uint32_t conv_glds_lds_bytes(int bnf, int nst) { return (uint32_t)nst * (16 + 2 * bnf) * 1024; }

// k_conv3_halo: the 3x3 / stride 1 / pad 1 conv of the SD-VAE, LPIPS and EFT plans on the k_conv_glds machinery (conv_glds.h: LDS-DMA
// staging, loader / matrix wave specialisation, counted waits, LDS-transposed float4 epilogue) with the im2col re-reads taken out of
// the memory system.
//
// Why: k_conv_glds stages the A operand once per (tap, 64-channel chunk) -- the same pixels nine times, shifted -- and at one 128 x 128
// tile per CU the L2 -> L1 path is what bounds it (measured: 16 KiB of A per stage from L2 costs what 36 us of a 42 us layer are
// made of; the same loads hitting L1 leave 31 us; profiles/r03_conv_glds_experiments.log).  Here the pixel tile is 8 rows x 16
// columns, and per 64-channel chunk its 10 x 18 halo tile (23 KiB, 1.4x the tile instead of 9x) is staged ONCE; the nine taps
// read shifted windows of it.  The K loop therefore runs chunk-major (for chunk: for tap), the weights stream through the ring
// of conv_glds.h one (tap, chunk) stage at a time, and the fp32 accumulation order differs from k_conv_lds / k_conv_glds
// (tap-major): results agree to reassociation, not bit for bit; they are identical run to run.
//
// LDS: two halo tiles (184 pixel slots x 128 B each, double-buffered across chunks) + NST weight stages of 2*BNF KiB.
//   halo slot p = row * 18 + column holds the pixel's eight 16-byte chunks at positions chunk ^ (p & 7): conflict-free for
//   ds_read_b128 at every window offset (the 16-lane groups of MI355X_MICROARCH.md, checked exhaustively), and written by
//   line-shaped LDS-DMA (8 lanes per 128-byte line, the permutation on the source address).
// Waves: 0..3 multiply (2 x 2, four tile rows x 8*BNF channels each); 4, 5 stream the weight ring; 6, 7 stage the halo tile
// of the NEXT chunk while the nine taps of the current one run.  One raw barrier per stage.
// Needs: k = 3, stride 1, pad 1 (a nearest-x2 upsampled input view is fine), W % 16 == 0, H % 8 == 0, Cin % 64 == 0, operand-type activations.
#pragma once
#include "conv_glds.h"

template <int BNF, int NST, bool GN>
SF_DEV void conv_halo_body(const ConvArgs& a, double* __restrict__ gn_part, const int gn_cg) {
  static_assert(NST == 3 || NST == 4, "weight ring depth 3 or 4");
  static_assert(BNF == 4 || BNF == 8, "64 or 128 output channels per workgroup");
  constexpr int WNF = BNF / 2;
  constexpr int G = BNF;                        // LDS-DMA loads per weight-loader wave and stage (BNF fragments x 2 k-steps over 2 waves)
  constexpr int HW_ = 18, HSLOTS = 184;         // halo tile: 10 rows x 18 columns = 180 pixel slots, rounded up to 23 loads of 8
  constexpr int A_BYTES = HSLOTS * 128, B_STAGE = BNF * 2 * 1024;
  constexpr int LDS_BYTES = 2 * A_BYTES + NST * B_STAGE;
  SF_DYN_LDS(lds);
  const int lane = threadIdx.x & 63, wave = sf_uniform((int)(threadIdx.x >> 6));
  const bool loader = wave >= 4;
  const int wm = (wave >> 1) & 1, wn = wave & 1;
  const int tiles = a.m_tiles * a.n_tiles;
  int t = blockIdx.x;
  if (tiles % 8 == 0) t = (blockIdx.x & 7) * (tiles >> 3) + (blockIdx.x >> 3);
  const int nt = t % a.n_tiles, mt = t / a.n_tiles;
  const int tiles_x = a.W >> 4, tpi = tiles_x * (a.H >> 3);
  const int b = mt / tpi, tr = mt - b * tpi;
  const int ty = tr / tiles_x, y0 = ty * 8, x0 = (tr - ty * tiles_x) * 16;
  const int cpairs = a.cchunks >> 1;
  const int S = 9 * cpairs;

  f32x4 acc[4][WNF];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int n = 0; n < WNF; ++n) acc[i][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (wave >= 6) {
    // ---- halo loaders: wave 6 stages slots [0, 96), wave 7 slots [96, 184); one load = 8 slots x 128 B
    const int g0 = (wave - 6) * 12;
    const sf_opnd* src[12];
    int step[12];                                // 64 channels on per chunk for a pixel inside the image, 0 for the zero line
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      const int p = (g0 + j) * 8 + (lane >> 3);
      const int hr = p / HW_, hc = p - hr * HW_;
      const int y = y0 - 1 + hr, x = x0 - 1 + hc;
      const bool ok = (p < 10 * HW_) & (y >= 0) & (y < a.H) & (x >= 0) & (x < a.W);
      const int chunk = (lane & 7) ^ (p & 7);
      const int ys = (ok ? y : 0) >> a.ups, xs = (ok ? x : 0) >> a.ups;       // nearest x2 upsampling folded into the addressing: (H, W) are the
      const sf_opnd* in = reinterpret_cast<const sf_opnd*>(a.in) + (((long)b * (a.H >> a.ups) + ys) * (a.W >> a.ups) + xs) * a.Cin + chunk * 8;      // upsampled dims
      src[j] = ok ? in : reinterpret_cast<const sf_opnd*>(sf_zero128) + (lane & 7) * 8;
      step[j] = ok ? 64 : 0;
    }
    const int nld = wave == 6 ? 12 : 11;         // 23 loads of 8 slots
    auto issue_tile = [&](int h) {
      char* ab = lds + (h & 1) * A_BYTES + g0 * 1024;
#pragma unroll
      for (int j = 0; j < 12; ++j)
        if (j < nld) sf_glds16(ab + j * 1024, src[j] + (long)h * step[j]);
    };
    issue_tile(0);
    int tap = 0, h = 0;
    for (int s = 0; s < S; ++s) {
      if (tap == 0) sf_vmcnt<0>();               // the halo tile of chunk h (issued nine stages ago) has landed
      sf_lds_barrier();
      if (tap == 0 && h + 1 < cpairs) issue_tile(h + 1);      // into the buffer chunk h - 1 was read from
      if (++tap == 9) { tap = 0; ++h; }
    }
    sf_vmcnt<0>();
    sf_glds_done();
  } else if (wave >= 4) {
    // ---- weight loaders: the ring of conv_glds.h, stage (chunk h, tap) = k-steps (tap * cchunks + 2h, + 1)
    const int lw = wave - 4;
    const bf16x8* wbase[WNF];
#pragma unroll
    for (int j = 0; j < WNF; ++j) {
      const int nf = min(nt * BNF + WNF * lw + j, a.n_frags - 1);
      wbase[j] = a.w + (long)nf * a.KS * 64 + lane;
    }
    int i_tap = 0, i_h = 0, i_buf = 0;
    auto issue_next = [&]() {
      char* sb = lds + 2 * A_BYTES + i_buf * B_STAGE;
      const long kk = (long)i_tap * a.cchunks + 2 * i_h;
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int j = 0; j < WNF; ++j) sf_glds16(sb + ((WNF * lw + j) * 2 + u) * 1024, wbase[j] + (kk + u) * 64);
      if (++i_tap == 9) { i_tap = 0; ++i_h; }
      if (++i_buf == NST) i_buf = 0;
    };
    for (int p = 0; p < NST - 1 && p < S; ++p) issue_next();
    for (int s = 0; s < S; ++s) {
      const int newer = S - 1 - s;
      if (newer >= NST - 2) sf_vmcnt<G * (NST - 2)>();
      else if (NST == 4 && newer == 1) sf_vmcnt<G>();
      else sf_vmcnt<0>();
      sf_lds_barrier();
      if (s + NST - 1 < S) issue_next();
    }
    sf_vmcnt<0>();
    sf_glds_done();
  } else {
    // ---- matrix waves: tile rows wm*4 .. wm*4+3 (16 pixels each) x WNF n-fragments.  Window offsets into the halo tile for
    // the 6 halo rows this wave touches x 3 column shifts x 2 k-steps, computed once (the tap loop is unrolled: static indices)
    int aoff[6][3][2];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int p = (wm * 4 + r) * HW_ + kx + (lane & 15);
#pragma unroll
        for (int u = 0; u < 2; ++u) aoff[r][kx][u] = p * 128 + (((u * 4 + (lane >> 4)) ^ (p & 7)) << 4);
      }
    int r_buf = 0;
    SF_LGKM0();
    for (int h = 0; h < cpairs; ++h) {
      const char* ab = lds + (h & 1) * A_BYTES;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int ky = tap / 3, kx = tap - 3 * ky;
        sf_lds_barrier();
        const char* sb = lds + 2 * A_BYTES + r_buf * B_STAGE + lane * 16;
        bf16x8 fa[2][4], fb[2][WNF];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          fa[u][0] = *reinterpret_cast<const bf16x8*>(ab + aoff[ky][kx][u]);
#pragma unroll
          for (int n = 0; n < WNF; ++n) fb[u][n] = *reinterpret_cast<const bf16x8*>(sb + ((wn * WNF + n) * 2 + u) * 1024);
#pragma unroll
          for (int i = 1; i < 4; ++i) fa[u][i] = *reinterpret_cast<const bf16x8*>(ab + aoff[i + ky][kx][u]);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int n = 0; n < WNF; ++n) acc[i][n] = sf_mfma16(fa[u][i], fb[u][n], acc[i][n]);
        constexpr int NRD = 2 * (4 + WNF);
        SF_SCHED_GROUP(0x100, 4);
#pragma unroll
        for (int q = 0; q < NRD - 4; ++q) { SF_SCHED_GROUP(0x008, 1); SF_SCHED_GROUP(0x100, 1); }
        SF_SCHED_GROUP(0x008, 8 * WNF - (NRD - 4));
        if (++r_buf == NST) r_buf = 0;
      }
    }
  }
  conv_tile_epilogue<BNF, LDS_BYTES, GN>(a, lds, acc, loader, wm, wn, lane, nt, mt, gn_part, gn_cg, [&](int row) -> long {
    return ((long)b * a.H + y0 + (row >> 4)) * a.W + x0 + (row & 15);
  });
}

template <int BNF, int NST, bool GN>
SF_KERNEL(512, 1) void k_conv3_halo(ConvArgs a, double* __restrict__ gn_part, int gn_cg) {
  sf_touch_kernarg<(int)sizeof(ConvArgs)>();
  conv_halo_body<BNF, NST, GN>(a, gn_part, gn_cg);
}

static inline uint32_t conv_halo_lds_bytes(int bnf, int nst) { return 2u * 184 * 128 + (uint32_t)nst * bnf * 2 * 1024; }
static inline bool conv_halo_ok(const ConvArgs& a) {
  return a.kh == 3 && a.kw == 3 && a.stride == 1 && a.pad == 1 && a.Ho == a.H && a.Wo == a.W && a.W % 16 == 0 && a.H % 8 == 0 &&
         a.Cin % 64 == 0;
}

// The latent half of the UNet's init conv inside a sampler trajectory: x0 = base + CrossEmbed(x) where x is the (<= 4-channel)
// latent and `base` holds the conditioning half + bias, evaluated once per trajectory (external/imagen_pytorch.py:1017-1042:
// CrossEmbedLayer = three convs k = 3 / 7 / 15 into channel slices; the conv is linear in its input channels).
//
// The first plan ran this through the implicit-GEMM kernel: pack (NCHW -> NHWC padded to 32 channels), three launches with
// 7/8 of every MFMA k-step multiplying zeros, a split-K reduction -- five dependent launches, ~45 us, for 0.3 GFLOP.  A direct
// convolution on the vector units with scalar-loaded weights came out no faster (35 us: one LDS read and one scalar load per two
// FMAs).  This version keeps the matrix cores but needs no im2col and no channel padding: the haloed patch of an 8 x 8 pixel
// tile sits in LDS as bf16 [row][column][4 channels], so the 8 K-elements a lane feeds to v_mfma_f32_16x16x32_bf16 -- two
// horizontally adjacent taps x 4 channels -- are ONE contiguous 16-byte LDS read, and a k-step covers 8 taps of one kernel row:
//     k = 15: 15 rows x 2 half-rows (kx padded to 16)  = 30 k-steps;  k = 7: 7 k-steps (kx padded to 8);
//     k = 3 : 2 k-steps (two kernel rows each, kx padded to 4).
// One workgroup = one (tile, conv); wave w owns output-channel fragments {w, w + 4, ..} for the tile's 64 pixels.
// Written against sf_dev.h so that tests/hostemu runs the same source on CPU threads (tests/test_hostemu_initx.py).
#pragma once
#include "sf_dev.h"


typedef bf16x8 ix_bf16x8;
typedef bf16x4 ix_bf16x4;
typedef f32x4 ix_f32x4;

#define IX_TILE 8
#define IX_HALO 7
#define IX_PH (IX_TILE + 2 * IX_HALO)     /* 22 rows */
#define IX_PW 24                          /* 22 columns + 2 zero columns read by the padded kx = 15 tap */

struct InitXArgs {
  const float* x;            // [B][Cx][H][W]
  const float* base;         // [B*H*W][ld]
  const ix_bf16x8* w;        // conv i at w + woff[i] (in fragments of 64 lanes x 8): [k-step][n-frag][lane]
  float* out;                // [B*H*W][ld]
  float* slots;              // or null: [B*H*W/16][ld/16][2] (sum, sum of squares) of `out`, one slot per 16-pixel x 16-channel MFMA
                             // fragment (16 pixels = two rows of the 8x8 tile, slot row = (image tile, row pair): the consuming GroupNorm
                             // sums ALL slots of an (image, group), so any pixel partition with the right channel column serves)
  int B, H, W, Cx, ld;
  int cw[3], co[3], woff[3];
};

// haloed patch as bf16 [row][column][4 channels], zero outside the image / beyond the latent's channels / in the pad columns
SF_DEV void initx_stage(const InitXArgs& a, char* __restrict__ patch, int tid, int b, int y0, int x0) {
  constexpr int IT = (IX_PH * IX_PW + 255) / 256;
  // every load is issued before the first use, from a clamped (always valid) address; the mask is applied to the value
  // (a load under a branch is fenced by a full vmcnt(0) wait: 12 serial round trips instead of one)
  float v[IT][4];
  bool in[IT];
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int i = tid + it * 256;
    const int fy = i / IX_PW, fx = i - fy * IX_PW;
    const int yy = y0 - IX_HALO + fy, xx = x0 - IX_HALO + fx;
    in[it] = i < IX_PH * IX_PW && fx < IX_PH && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
    const int yc = min(max(yy, 0), a.H - 1), xc = min(max(xx, 0), a.W - 1);
#pragma unroll
    for (int c = 0; c < 4; ++c) v[it][c] = a.x[(((long)b * a.Cx + min(c, a.Cx - 1)) * a.H + yc) * a.W + xc];
  }
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int i = tid + it * 256;
    ix_bf16x4 o;
#pragma unroll
    for (int c = 0; c < 4; ++c) o[c] = (sf_opnd)((in[it] && c < a.Cx) ? v[it][c] : 0.0f);
    if (i < IX_PH * IX_PW) *reinterpret_cast<ix_bf16x4*>(patch + i * 8) = o;
  }
}

// One (tile, conv) workgroup.  Wave w owns the n-fragments {w, w + 4, ..} (NFW of them) for all 64 pixels (4 m-fragments):
// its STEPS x NFW weight fragments fit in registers and are fetched FIRST, before the patch is staged, so the one global-memory
// latency of the kernel overlaps the staging; the main loop is LDS reads and MFMAs only.
template <int K, int NFW>
SF_DEV void initx_conv(const InitXArgs& a, char* __restrict__ patch, int conv, int tid, int b, int y0, int x0, int slot_row0) {
  constexpr int OFF = IX_HALO - K / 2;
  constexpr int STEPS = K == 15 ? 30 : (K == 7 ? 7 : 2);
  const int lane = tid & 63, wave = sf_uniform(tid >> 6);
  const int NF = a.cw[conv] >> 4;
  const bool active = wave < NF;                              // a 32-channel slice keeps two waves for staging only
  const int m = lane & 15, g = lane >> 4, n = lane & 15;
  const ix_bf16x8* __restrict__ w = a.w + (long)a.woff[conv] * 64 + lane;
  ix_bf16x8 wf[STEPS][NFW];
  if (active) {
#pragma unroll
    for (int s = 0; s < STEPS; ++s)
#pragma unroll
      for (int j = 0; j < NFW; ++j) wf[s][j] = w[(long)(s * NF + wave + 4 * j) * 64];
  }
  const long m0 = (long)b * a.H * a.W + (long)y0 * a.W + x0;
  // the epilogue's `base` operand too: D[i = 4 * (lane >> 4) + r][n = lane & 15] = pixel i of m-fragment mf, channel (wave + 4 j) * 16 + n
  float bv[4][4][NFW];
  if (active) {
#pragma unroll
    for (int mf = 0; mf < 4; ++mf)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 4 * g + r;
        const long mm = m0 + (long)(2 * mf + (i >> 3)) * a.W + (i & 7);
#pragma unroll
        for (int j = 0; j < NFW; ++j) bv[mf][r][j] = a.base[mm * a.ld + a.co[conv] + (wave + 4 * j) * 16 + n];
      }
  }
  initx_stage(a, patch, tid, b, y0, x0);
  sf_sync();
  if (!active) return;
  ix_f32x4 acc[4][NFW];
#pragma unroll
  for (int mf = 0; mf < 4; ++mf)
#pragma unroll
    for (int j = 0; j < NFW; ++j) acc[mf][j] = ix_f32x4{0.f, 0.f, 0.f, 0.f};
  const int y = m >> 3, x = m & 7;                            // this lane's A row = pixel (2 mf + y, x) of the tile
  // The A fragment of (m-fragment mf, k-step s) depends only on the patch row 2 mf + row(s) and the column half of s, so the
  // loop runs over those (R x NH distinct fragments instead of 4 x STEPS) and feeds each to every (mf, s) that uses it.  The
  // 16 bytes of a lane start at an 8-byte-aligned column: two ds_read_b64, not one (misaligned) ds_read_b128.
  constexpr int NH = K == 15 ? 2 : 1;
  constexpr int RSTEP = K == 3 ? 2 : 1;                       // k = 3 packs two kernel rows per k-step (row inside the lane: g >> 1)
  constexpr int ROWS = K == 3 ? 3 : K;                        // row(s) in [0, ROWS) step RSTEP
  const int lrow = K == 3 ? (g >> 1) : 0;
  const int lcol = K == 3 ? (g & 1) * 2 : 2 * g;
  const char* pa0 = patch + ((y + lrow + OFF) * IX_PW + x + lcol + OFF) * 8;
#pragma unroll
  for (int r = 0; r < 6 + ROWS; r += RSTEP) {
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const char* pa = pa0 + (r * IX_PW + h * 8) * 8;
      const ix_bf16x4 lo = *reinterpret_cast<const ix_bf16x4*>(pa), hi = *reinterpret_cast<const ix_bf16x4*>(pa + 8);
      const ix_bf16x8 fa = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
      for (int mf = 0; mf < 4; ++mf) {
        const int row = r - 2 * mf;                           // kernel row (pair) of the k-step that meets this fragment at mf
        if (row < 0 || row >= ROWS) continue;
        const int st = K == 15 ? 2 * row + h : (K == 7 ? row : row / 2);
#pragma unroll
        for (int j = 0; j < NFW; ++j) acc[mf][j] = sf_mfma16(fa, wf[st][j], acc[mf][j]);
      }
    }
  }
#pragma unroll
  for (int mf = 0; mf < 4; ++mf) {
    float sm[NFW], sq[NFW];
#pragma unroll
    for (int j = 0; j < NFW; ++j) sm[j] = sq[j] = 0.0f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 4 * g + r;
      const long mm = m0 + (long)(2 * mf + (i >> 3)) * a.W + (i & 7);
#pragma unroll
      for (int j = 0; j < NFW; ++j) {
        const float v = bv[mf][r][j] + acc[mf][j][r];
        a.out[mm * a.ld + a.co[conv] + (wave + 4 * j) * 16 + n] = v;
        sm[j] += v;
        sq[j] = fmaf(v, v, sq[j]);
      }
    }
    if (a.slots) {                                            // the statistics slots of the next GroupNorm-fused conv (r04: was a k_slots launch)
#pragma unroll
      for (int j = 0; j < NFW; ++j) {
        const float s1 = sf_wave_sum(sm[j]), s2 = sf_wave_sum(sq[j]);
        if (lane == 0) {
          float* sl = a.slots + ((long)(slot_row0 + mf) * (a.ld >> 4) + (a.co[conv] >> 4) + wave + 4 * j) * 2;
          sl[0] = s1;
          sl[1] = s2;
        }
      }
    }
  }
}

SF_KERNEL(256) void k_init_x(InitXArgs a) {
  SF_SHARED __attribute__((aligned(16))) char patch[IX_PH * IX_PW * 8];
  const int tid = threadIdx.x;
  const int tiles_x = a.W / IX_TILE, tiles = tiles_x * (a.H / IX_TILE);
  const int conv = blockIdx.x % 3;
  const int bt = blockIdx.x / 3;
  const int b = bt / tiles, t = bt - b * tiles;
  const int ty = t / tiles_x, tx = t - ty * tiles_x;
  const int y0 = ty * IX_TILE, x0 = tx * IX_TILE;
  const int srow = bt * 4;                                    // 4 fragment rows (row pairs) per 8x8 tile: (B * tiles) * 4 = B * H * W / 16
  if (conv == 2) initx_conv<15, 1>(a, patch, 2, tid, b, y0, x0, srow);
  else if (conv == 1) initx_conv<7, 1>(a, patch, 1, tid, b, y0, x0, srow);
  else if (a.cw[0] == 128) initx_conv<3, 2>(a, patch, 0, tid, b, y0, x0, srow);
  else initx_conv<3, 1>(a, patch, 0, tid, b, y0, x0, srow);
}


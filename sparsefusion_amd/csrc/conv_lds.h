// ConvArgs (shared by k_conv_igemm in unet_ops.hip) and the LDS-tiled large-M implicit GEMM k_conv_lds.  Written against
// sf_dev.h so that tests/hostemu runs the same source on CPU threads (tests/test_hostemu_conv_lds.py).
#pragma once
#include "sf_dev.h"

struct ConvArgs {
  const void* in; const bf16x8* w; const float* bias; float* out; const float* resid; float* ws;
  int accum, npad;
  int B, H, W, Cin, Ho, Wo, Cout, ldc, co_off, kh, kw, stride, pad, groups;
  int KS, cchunks, m_frags, n_frags, m_tiles, n_tiles, steps_per_wave;
  int pixshuf, ups, relu;   // relu: 0 none, 1 ReLU, 2 GELU (erf)
};

// ---------------------------------------------------------------------------------------------
// Large-M implicit GEMM (VAE, LPIPS-VGG, B >= 4): workgroup tile 128 pixels x 16*BNF channels, staged through LDS.
// The weight-streaming kernel above keeps every fragment private to a wave (right when M is one or two tiles and the
// weights are read once); at M = 4 096 .. 131 072 the same fragments are needed by several waves, and private loads
// make the kernel L1-bandwidth bound at ~4 % of the MFMA peak.  Here the 4 waves of a workgroup load each A / B
// fragment ONCE per stage (two k-steps = 64 input channels) in MFMA lane order -- so a fragment is a contiguous,
// conflict-free 1 KiB of LDS -- and every wave re-reads the 4 + BNF/2 fragments of its 64 x (8*BNF) sub-tile with
// ds_read_b128.  Global loads of stage s+1 are issued into registers before the MFMAs of stage s (one barrier per
// stage).  The workgroup -> tile map is XCD-aware: consecutive workgroups go round-robin over the 8 XCDs, so each
// XCD is handed a contiguous range of tiles (neighbouring pixels and all channel tiles of a pixel tile share one
// L2 -- the im2col reuse of a 3x3 conv is 9x in A).
// No split-K, no pixel shuffle; epilogue = bias (+ residual) (+ accumulate) (+ ReLU).
// ---------------------------------------------------------------------------------------------
template <int BNF, bool A_FP32>
SF_KERNEL(256, 2) void k_conv_lds(ConvArgs a) {
  sf_touch_kernarg<(int)sizeof(ConvArgs)>();
  constexpr int WNF = BNF / 2;                  // n-fragments per wave (waves are arranged 2 x 2)
  constexpr int BLD = BNF / 4;                  // B fragments each wave loads per k-step
  SF_SHARED bf16x8 sA[2][8][2][64];            // [stage buffer][m-frag][k-step][lane]  2 x 16 KiB
  SF_SHARED bf16x8 sB[2][BNF][2][64];          //                                      2 x 2*BNF KiB
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile order (8 XCDs, workgroups are dealt round-robin)
  const int tiles = a.m_tiles * a.n_tiles;
  int t = blockIdx.x;
  if (tiles % 8 == 0) t = (blockIdx.x & 7) * (tiles >> 3) + (blockIdx.x >> 3);
  const int nt = t % a.n_tiles, mt = t / a.n_tiles;
  const int M = a.B * a.Ho * a.Wo;
  const int S = (a.KS + 1) >> 1;

  // loader geometry: this wave fetches m-fragments 2*wave, 2*wave+1 (rows = pixels) and BLD n-fragments
  int pb[2], py[2], px[2];
  bool pv[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int m = (mt * 8 + 2 * wave + j) * 16 + (lane & 15);
    pv[j] = m < M;
    const int mm = pv[j] ? m : 0;
    pb[j] = mm / (a.Ho * a.Wo);
    const int r = mm - pb[j] * (a.Ho * a.Wo);
    const int oy = r / a.Wo;
    py[j] = oy * a.stride - a.pad;
    px[j] = (r - oy * a.Wo) * a.stride - a.pad;
  }
  const int cgrp = (lane >> 4) * 8;
  const bf16x8* wbase[BLD];
#pragma unroll
  for (int j = 0; j < BLD; ++j) {
    const int nf = min(nt * BNF + BLD * wave + j, a.n_frags - 1);
    wbase[j] = a.w + (long)nf * a.KS * 64 + lane;
  }

  f32x4 ra32[2][2][2];                          // prefetch registers (fp32 A: two 16-byte halves per fragment)
  bf16x8 ra16[2][2], rb[BLD][2];
  bool rin[2][2];

  auto gload = [&](int s) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int ks = 2 * s + u;
      const bool kv = ks < a.KS;
      const int kk = kv ? ks : a.KS - 1;
      const int tap = kk / a.cchunks, cc = kk - tap * a.cchunks;
      const int ky = tap / a.kw, kx = tap - ky * a.kw;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int iy = py[j] + ky, ix = px[j] + kx;
        const bool in = kv && pv[j] && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        rin[j][u] = in;
        const long off = (((long)pb[j] * (a.H >> a.ups) + ((in ? iy : 0) >> a.ups)) * (a.W >> a.ups) + ((in ? ix : 0) >> a.ups)) *
                             a.Cin + cc * 32 + cgrp;
        if (A_FP32) {
          const float* p = reinterpret_cast<const float*>(a.in) + off;
          ra32[j][u][0] = *reinterpret_cast<const f32x4*>(p);
          ra32[j][u][1] = *reinterpret_cast<const f32x4*>(p + 4);
        } else {
          ra16[j][u] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const __bf16*>(a.in) + off);
        }
      }
#pragma unroll
      for (int j = 0; j < BLD; ++j) rb[j][u] = wbase[j][(long)kk * 64];
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        bf16x8 v;
        if (A_FP32) {
          const f32x4 lo = ra32[j][u][0], hi = ra32[j][u][1];
          v[0] = (__bf16)lo[0]; v[1] = (__bf16)lo[1]; v[2] = (__bf16)lo[2]; v[3] = (__bf16)lo[3];
          v[4] = (__bf16)hi[0]; v[5] = (__bf16)hi[1]; v[6] = (__bf16)hi[2]; v[7] = (__bf16)hi[3];
        } else {
          v = ra16[j][u];
        }
        sA[buf][2 * wave + j][u][lane] = rin[j][u] ? v : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
      }
#pragma unroll
      for (int j = 0; j < BLD; ++j) sB[buf][BLD * wave + j][u][lane] = rb[j][u];
    }
  };

  f32x4 acc[4][WNF];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int n = 0; n < WNF; ++n) acc[i][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  gload(0);
  lstore(0);
  sf_sync();
  for (int s = 0; s < S; ++s) {
    const int buf = s & 1;
    if (s + 1 < S) gload(s + 1);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      bf16x8 fa[4], fb[WNF];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = sA[buf][wm * 4 + i][u][lane];
#pragma unroll
      for (int n = 0; n < WNF; ++n) fb[n] = sB[buf][wn * WNF + n][u][lane];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int n = 0; n < WNF; ++n) acc[i][n] = sf_mfma16(fa[i], fb[n], acc[i][n]);
    }
    if (s + 1 < S) lstore(buf ^ 1);
    sf_sync();
  }

#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int n = 0; n < WNF; ++n) {
      const int nfr = nt * BNF + wn * WNF + n;
      const int col = nfr * 16 + (lane & 15);
      if (nfr >= a.n_frags || col >= a.Cout) continue;
      const float bv = a.bias ? a.bias[col] : 0.0f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = (mt * 8 + wm * 4 + i) * 16 + (lane >> 4) * 4 + r;
        if (m >= M) continue;
        const long o = (long)m * a.ldc + a.co_off + col;
        float v = acc[i][n][r] + bv;
        if (a.resid) v += a.resid[o];
        if (a.accum) v += a.out[o];
        if (a.relu == 1) v = fmaxf(v, 0.0f);
        else if (a.relu == 2) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
        a.out[o] = v;
      }
    }
  }
}


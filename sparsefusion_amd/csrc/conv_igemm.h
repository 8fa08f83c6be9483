// k_conv_igemm: the weight-streaming implicit GEMM (operand contract and packed weight layout: the comment block in unet_ops.hip
// above its include).  Written against sf_dev.h so that tests/hostemu runs the same source on CPU threads
// (tests/test_hostemu_conv_igemm.py).
#pragma once
#include "sf_dev.h"
#include "conv_lds.h"          // ConvArgs

template <int WM, int WN, bool A_FP32>
SF_KERNEL(256) void k_conv_igemm(ConvArgs a) {
  sf_touch_kernarg<(int)sizeof(ConvArgs)>();        // all kernel-argument lines in one scalar-cache round trip (sf_dev.h)
  SF_SHARED __attribute__((aligned(16))) float red[3][WM * WN * 4 * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile = blockIdx.x % (a.m_tiles * a.n_tiles);
  const int grp = blockIdx.x / (a.m_tiles * a.n_tiles);
  const int nt = tile % a.n_tiles, mt = tile / a.n_tiles;
  const int k0 = (grp * 4 + wave) * a.steps_per_wave;
  const int k1 = min(a.KS, k0 + a.steps_per_wave);
  const int M = a.B * a.Ho * a.Wo;

  // per m-fragment pixel coordinates of this lane's A row
  int pb[WM], py[WM], px[WM];
  bool pv[WM];
#pragma unroll
  for (int mi = 0; mi < WM; ++mi) {
    const int m = (mt * WM + mi) * 16 + (lane & 15);
    pv[mi] = m < M;
    const int mm = pv[mi] ? m : 0;
    pb[mi] = mm / (a.Ho * a.Wo);
    const int r = mm - pb[mi] * (a.Ho * a.Wo);
    const int oy = r / a.Wo;
    py[mi] = oy * a.stride - a.pad;
    px[mi] = (r - oy * a.Wo) * a.stride - a.pad;
  }
  f32x4 acc[WM][WN];
#pragma unroll
  for (int mi = 0; mi < WM; ++mi)
#pragma unroll
    for (int ni = 0; ni < WN; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int cgrp = (lane >> 4) * 8;
  int ks = k0;
  while (ks < k1) {
    const int tap = ks / a.cchunks;
    const int cc0 = ks - tap * a.cchunks;
    const int cc1 = min(a.cchunks, cc0 + (k1 - ks));
    const int ky = tap / a.kw, kx = tap - ky * a.kw;
    long aoff[WM];
    bool ain[WM];
#pragma unroll
    for (int mi = 0; mi < WM; ++mi) {
      const int iy = py[mi] + ky, ix = px[mi] + kx;
      ain[mi] = pv[mi] && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      // a.ups = 1: the conv reads a nearest-neighbour x2 upsampling of a stored [H/2, W/2] map (fused Upsample)
      aoff[mi] = (((long)pb[mi] * (a.H >> a.ups) + ((ain[mi] ? iy : 0) >> a.ups)) * (a.W >> a.ups) + ((ain[mi] ? ix : 0) >> a.ups)) *
                     a.Cin + cgrp;
    }
    const bf16x8* wp[WN];
#pragma unroll
    for (int ni = 0; ni < WN; ++ni) {
      const int nf = min(nt * WN + ni, a.n_frags - 1);
      wp[ni] = a.w + ((long)nf * a.KS + ks) * 64 + lane;
    }
    // Software pipeline over trips of U k-steps: the 16-byte fragment loads of trip t+1 are issued (branch
    // free, unconditionally) before the MFMAs of trip t, so a wave keeps U*(WM+WN) KiB in flight while the
    // matrix pipe works -- weight streaming on the small-M layers is latency x bytes-in-flight bound.
    // Out-of-image taps read a clamped in-bounds pixel and are zeroed by a select, never by a branch.
    const sf_opnd* abase16[WM];
    const float* abase32[WM];
#pragma unroll
    for (int mi = 0; mi < WM; ++mi) {
      abase16[mi] = reinterpret_cast<const sf_opnd*>(a.in) + aoff[mi];
      abase32[mi] = reinterpret_cast<const float*>(a.in) + aoff[mi];
    }
    auto load_a = [&](int mi, int cc) -> bf16x8 {
      bf16x8 v;
      if (A_FP32) {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(abase32[mi] + cc * 32);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(abase32[mi] + cc * 32 + 4);
        v[0] = (sf_opnd)lo[0]; v[1] = (sf_opnd)lo[1]; v[2] = (sf_opnd)lo[2]; v[3] = (sf_opnd)lo[3];
        v[4] = (sf_opnd)hi[0]; v[5] = (sf_opnd)hi[1]; v[6] = (sf_opnd)hi[2]; v[7] = (sf_opnd)hi[3];
      } else {
        v = *reinterpret_cast<const bf16x8*>(abase16[mi] + cc * 32);
      }
      return ain[mi] ? v : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    };
    constexpr int U = (WM + WN <= 2) ? 8 : (WM + WN <= 4 ? 4 : 2);
    const int n_full = (cc1 - cc0) / U;
    if (n_full > 0) {
      bf16x8 fa[2][U][WM], fb[2][U][WN];
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int ni = 0; ni < WN; ++ni) fb[0][u][ni] = wp[ni][(long)u * 64];
#pragma unroll
        for (int mi = 0; mi < WM; ++mi) fa[0][u][mi] = load_a(mi, cc0 + u);
      }
      for (int t = 0; t < n_full; t += 2) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          if (t + half < n_full) {
            const int tn = min(t + half + 1, n_full - 1);        // the last trip re-reads itself: no tail branch
            const int ccn = cc0 + tn * U;
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
              for (int ni = 0; ni < WN; ++ni) fb[half ^ 1][u][ni] = wp[ni][((long)tn * U + u) * 64];
#pragma unroll
              for (int mi = 0; mi < WM; ++mi) fa[half ^ 1][u][mi] = load_a(mi, ccn + u);
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
              for (int mi = 0; mi < WM; ++mi)
#pragma unroll
                for (int ni = 0; ni < WN; ++ni)
                  acc[mi][ni] = sf_mfma16(fa[half][u][mi], fb[half][u][ni], acc[mi][ni]);
          }
        }
      }
    }
    for (int cc = cc0 + n_full * U; cc < cc1; ++cc) {              // ragged tail (< U steps)
      bf16x8 ta[WM], tb[WN];
#pragma unroll
      for (int ni = 0; ni < WN; ++ni) tb[ni] = wp[ni][(long)(cc - cc0) * 64];
#pragma unroll
      for (int mi = 0; mi < WM; ++mi) ta[mi] = load_a(mi, cc);
#pragma unroll
      for (int mi = 0; mi < WM; ++mi)
#pragma unroll
        for (int ni = 0; ni < WN; ++ni)
          acc[mi][ni] = sf_mfma16(ta[mi], tb[ni], acc[mi][ni]);
    }
    ks += cc1 - cc0;
  }

  // reduce the 4 K-slices of this workgroup through LDS
  if (wave > 0) {
#pragma unroll
    for (int mi = 0; mi < WM; ++mi)
#pragma unroll
      for (int ni = 0; ni < WN; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave - 1][((mi * WN + ni) * 4 + r) * 64 + lane] = acc[mi][ni][r];
  }
  sf_sync();
  if (wave != 0) return;
#pragma unroll
  for (int mi = 0; mi < WM; ++mi)
#pragma unroll
    for (int ni = 0; ni < WN; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int idx = ((mi * WN + ni) * 4 + r) * 64 + lane;
        acc[mi][ni][r] += red[0][idx] + red[1][idx] + red[2][idx];
      }

#pragma unroll
  for (int mi = 0; mi < WM; ++mi) {
#pragma unroll
    for (int ni = 0; ni < WN; ++ni) {
      const int n = (nt * WN + ni) * 16 + (lane & 15);
      if (nt * WN + ni >= a.n_frags) continue;
      if (a.groups > 1) {                       // split-K partial tile -> workspace [grp][m][npad]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = (mt * WM + mi) * 16 + (lane >> 4) * 4 + r;
          if (m < M) a.ws[((long)grp * M + m) * a.npad + n] = acc[mi][ni][r];
        }
        continue;
      }
      if (a.pixshuf && a.slots_out) {             // (lanes beyond Cout / M contribute zeros; the host checks 16 | Cout / 4, 16 | M)
        float sm = 0.0f, sq = 0.0f;
        if (n < a.Cout) {
          const float bq = a.bias ? a.bias[n] : 0.0f;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int m = (mt * WM + mi) * 16 + (lane >> 4) * 4 + r;
            const float v = sf_silu(acc[mi][ni][r] + bq);
            if (m < M) { sm += v; sq = fmaf(v, v, sq); }
          }
        }
        sm = sf_wave_sum(sm);
        sq = sf_wave_sum(sq);
        if (lane == 0) {
          const int nf = nt * WN + ni;
          float* sl = a.slots_out + (((long)(mt * WM + mi) * 4 + (nf & 3)) * (a.ldc >> 4) + (a.co_off >> 4) + (nf >> 2)) * 2;
          sl[0] = sm;
          sl[1] = sq;
        }
      }
      if (n >= a.Cout) continue;
      const float bv = a.bias ? a.bias[n] : 0.0f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = (mt * WM + mi) * 16 + (lane >> 4) * 4 + r;
        if (m >= M) continue;
        float v = acc[mi][ni][r] + bv;
        if (a.pixshuf) {
          // out[b, 2*oy+i, 2*ox+j, c] = silu(conv[b, oy, ox, c*4 + i*2 + j])   (PixelShuffle(2), :578-606)
          v = sf_silu(v);
          const int b = m / (a.Ho * a.Wo);
          const int rr = m - b * (a.Ho * a.Wo);
          const int oy = rr / a.Wo, ox = rr - oy * a.Wo;
          const int c = n >> 2, ii = (n >> 1) & 1, jj = n & 1;
          a.out[(((long)b * (2 * a.Ho) + 2 * oy + ii) * (2 * a.Wo) + 2 * ox + jj) * a.ldc + a.co_off + c] = v;
        } else {
          const long o = (long)m * a.ldc + a.co_off + n;
          if (a.resid) v += a.resid[o];
          if (a.accum) v += a.out[o];
          if (a.relu == 1) v = fmaxf(v, 0.0f);
          else if (a.relu == 2) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
          a.out[o] = v;
        }
      }
    }
  }
}

// Fused UNet kernels for gfx950 (device code only; launchers in unet_fused.hip, CPU logic tests in tests/hostemu).
//
// k_conv_fused: [GroupNorm | LayerNorm | nothing] (+ scale/shift) (+ SiLU) -> conv / linear in ONE launch.
//   Replaces the reference's Block = GroupNorm -> x*(scale+1)+shift -> SiLU -> Conv2d (external/imagen_pytorch.py:641-662)
//   and the LayerNorm -> Linear pairs of the transformer blocks (:480-566, :944-1010) which the first-round plan ran as
//   three dependent launches (statistics, apply, conv).  At B = 1 every dependent launch costs ~4-6 us, more than the
//   kernels' own work, so the normalisation moves into the conv's A-operand prologue:
//     * a workgroup owns an output tile of 16*WM pixels (whole image rows) x 16*WN channels and, with S > 1, one of S
//       input-channel slices (split-K across workgroups, partial tiles go to a workspace slab);
//     * prologue: the (haloed) input rows of the tile are read ONCE as fp32 -- a virtual concat of two sources, the
//       first possibly "lazy" (un-reduced split-K slabs + bias + residual, or a gated residual h*gate + res, which this
//       kernel materialises for later consumers) -- normalised, activated and written to LDS as bf16 [pixel][channel];
//       GroupNorm statistics come either from the data itself (GN_SELF: the tile holds every pixel of whole groups,
//       the 4x4 level) or from per-(16 pixels x 16 channels) (sum, sum of squares) slots the producer left (GN_SLOTS);
//     * main loop: the 4 waves split the k-steps (tap x 32-channel chunk); weights stream from global memory in MFMA
//       fragment order (sf_conv_pack_weights) through a D-deep register ring that is filled BEFORE the prologue, so
//       the HBM / L2 weight stream runs under the normalisation; A fragments are 16-byte LDS reads of shifted pixels;
//     * epilogue: the 4 K-slices meet in LDS; final mode adds bias / residual / previous contents and leaves the
//       (sum, sum of squares) slots of the output for the next GroupNorm; partial mode stores the slab.
#pragma once
#include "sf_dev.h"

enum { FNORM_NONE = 0, FNORM_GN_SELF = 1, FNORM_GN_SLOTS = 2, FNORM_LN = 3 };

// fp32 NHWC source [M = B*HW, C].  mode 0: plain at p.  mode 1: v = b[c] + sum_g a[g][m][c (ld npad)] (+ r[m][c]).
// mode 2: v = a[m][c] * b[batch][c] + r[m][c].  Lazy sources are written to p by the workgroups that own the element.
struct FSrc {
  float* p;
  const float* a;
  const float* b;
  const float* r;
  const float* slots;     // [M/16][C/16][2] sums of the final values, or null
  int C, mode, groups, npad;
  float scale;            // applied after evaluation (skip connections: 2^-1/2); 1 for lazy sources
};

struct FConvArgs {
  FSrc s1, s2;            // s2.C == 0: no concat
  const bf16x8* w;
  const float* bias;
  float* out;
  const float* resid;
  float* ws;              // partial mode (S > 1): slabs [S][M][npad]
  float* slots_out;       // final mode: [M/16][ldc/16][2] or null
  const float* gamma;
  const float* beta;
  const float* ss;        // row b at ss + b*ss_stride: scale[C] then shift[C]; or null
  int ss_stride, norm, silu, pre_gelu, accum, G;
  float eps;
  int B, H, W, C, Cout, ldc, co_off, k;
  int TR, S, cps, cchunks, KS, n_frags, n_tiles, mt_per_img, npad, M;
  int pix_stride, xcd_map;
  int red_off, tab_off, misc_off;    // LDS byte offsets
};

SF_DEV f32x4 fsrc_load4(const FSrc& s, int M, int HW, long m, int c) {
  f32x4 v;
  if (s.mode == 1) {
    v = s.b ? *reinterpret_cast<const f32x4*>(s.b + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int g = 0; g < s.groups; ++g) v += *reinterpret_cast<const f32x4*>(s.a + ((long)g * M + m) * s.npad + c);
    if (s.r) v += *reinterpret_cast<const f32x4*>(s.r + m * s.C + c);
  } else if (s.mode == 2) {
    const f32x4 hv = *reinterpret_cast<const f32x4*>(s.a + m * s.C + c);
    const f32x4 gv = *reinterpret_cast<const f32x4*>(s.b + (m / HW) * s.C + c);
    const f32x4 rv = *reinterpret_cast<const f32x4*>(s.r + m * s.C + c);
    v = hv * gv + rv;
  } else {
    v = *reinterpret_cast<const f32x4*>(s.p + m * s.C + c);
  }
  return v;
}

// value of the (virtual) concat at pixel row m, concat channel c (c % 4 == 0); materialises lazy s1 elements when `own`
SF_DEV f32x4 fconv_load(const FConvArgs& a, long m, int c, bool own) {
  const int HW = a.H * a.W;
  if (c < a.s1.C) {
    f32x4 v = fsrc_load4(a.s1, a.M, HW, m, c);
    if (own && a.s1.mode) *reinterpret_cast<f32x4*>(a.s1.p + m * a.s1.C + c) = v;
    return v * a.s1.scale;
  }
  return fsrc_load4(a.s2, a.M, HW, m, c - a.s1.C) * a.s2.scale;
}

template <int WM, int WN, int D>
SF_KERNEL(256) void k_conv_fused(FConvArgs a) {
  SF_DYN_LDS(lds);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // ---- which tile
  const int MT = a.B * a.mt_per_img;
  const int tiles = MT * a.n_tiles;
  const int s = blockIdx.x / tiles;
  const int t = blockIdx.x - s * tiles;
  int mt, nt;
  if (a.xcd_map) {                       // workgroups b, b+8, ... run on one XCD: give them the m-tiles of ONE n-tile (weights hit in L2)
    const int x = t & 7, j = t >> 3;
    mt = j % MT;
    nt = (j / MT) * 8 + x;
  } else {
    nt = t % a.n_tiles;
    mt = t / a.n_tiles;
  }
  const int b = mt / a.mt_per_img;
  const int row0 = (mt - b * a.mt_per_img) * a.TR;
  const int h = a.k >> 1;
  const int FW = a.W + 2 * h, FR = a.TR + 2 * h;
  const int Cs = a.cps * 32, c0 = s * Cs, Cs4 = Cs >> 2;
  const int HW = a.H * a.W;
  const long mb = (long)b * HW;           // first pixel row of this image

  // ---- weight stream: this wave's k-steps [k0, k1) of the slice-local list (tap-major, then 32-channel chunk)
  const int KSl = a.k * a.k * a.cps;
  const int spw = (KSl + 3) >> 2;
  const int k0 = wave * spw;
  const int k1 = (KSl < k0 + spw) ? KSl : (k0 + spw);
  const bf16x8* wbase[WN];
#pragma unroll
  for (int ni = 0; ni < WN; ++ni) {
    int nf = nt * WN + ni;
    if (nf > a.n_frags - 1) nf = a.n_frags - 1;
    wbase[ni] = a.w + ((long)nf * a.KS + s * a.cps) * 64 + lane;
  }
  auto wload = [&](int j, int ni) -> bf16x8 {
    if (j > KSl - 1) j = KSl - 1;
    const int tap = j / a.cps, ccl = j - tap * a.cps;
    return wbase[ni][(long)(tap * a.cchunks + ccl) * 64];
  };
  bf16x8 fb[D][WN];
#pragma unroll
  for (int u = 0; u < D; ++u)
#pragma unroll
    for (int ni = 0; ni < WN; ++ni) fb[u][ni] = wload(k0 + u, ni);

  float* tabA = reinterpret_cast<float*>(lds + a.tab_off);
  float* tabB = tabA + Cs;
  float* misc = reinterpret_cast<float*>(lds + a.misc_off);      // [0..7] block-reduction scratch, [8..] group / row statistics

  // ---- (a) zero the frame pixels outside the image (conv zero padding); 8 threads per pixel
  if (h) {
    const int npix = FR * FW;
    for (int q = tid >> 3; q < npix; q += 32) {
      const int fr = q / FW, fx = q - fr * FW;
      const int r = row0 - h + fr, x = fx - h;
      if (r < 0 || r >= a.H || x < 0 || x >= a.W) {
        char* dst = lds + (long)q * a.pix_stride;
        for (int c8 = (tid & 7); c8 < (Cs >> 3); c8 += 8) *reinterpret_cast<bf16x8*>(dst + c8 * 16) = sf_zero8();
      }
    }
  }

  // ---- (b) normalisation parameters
  const int Cg = (a.norm == FNORM_GN_SELF || a.norm == FNORM_GN_SLOTS) ? a.C / a.G : 1;
  if (a.norm == FNORM_GN_SELF) {
    // the tile holds all HW pixels of image b and the slice holds whole groups: statistics from the data
    const int ngs = Cs / Cg, Cg4 = Cg >> 2, cnt = HW * Cg4;
    for (int gi = 0; gi < ngs; ++gi) {
      float sm = 0.0f, sq = 0.0f;
      for (int i = tid; i < cnt; i += 256) {
        const int p = i / Cg4, c = c0 + gi * Cg + (i - p * Cg4) * 4;
        const f32x4 v = fconv_load(a, mb + p, c, false);
        sm += (v[0] + v[1]) + (v[2] + v[3]);
        sq = fmaf(v[0], v[0], sq); sq = fmaf(v[1], v[1], sq); sq = fmaf(v[2], v[2], sq); sq = fmaf(v[3], v[3], sq);
      }
      sm = sf_wave_sum(sm);
      sq = sf_wave_sum(sq);
      if (lane == 0) { misc[wave] = sm; misc[4 + wave] = sq; }
      sf_sync();
      if (tid == 0) {
        const double n = (double)HW * Cg;
        const double mean = ((double)misc[0] + misc[1] + misc[2] + misc[3]) / n;
        double var = ((double)misc[4] + misc[5] + misc[6] + misc[7]) / n - mean * mean;
        if (var < 0.0) var = 0.0;
        misc[8 + 2 * gi] = (float)mean;
        misc[9 + 2 * gi] = sf_rsqrt((float)var + a.eps);
      }
      sf_sync();
    }
  } else if (a.norm == FNORM_GN_SLOTS) {
    // 32 lanes per group sum the producer's (sum, sum of squares) slots of image b
    const int ngs = Cs / Cg, gi = tid >> 5, li = tid & 31;
    const int n_mf = HW >> 4, n_cf = Cg >> 4, cnt = n_mf * n_cf;
    const int cf1 = a.s1.C >> 4, cf2 = a.s2.C >> 4;
    float sm = 0.0f, sq = 0.0f;
    if (gi < ngs) {
      for (int i = li; i < cnt; i += 32) {
        const int mf = i / n_cf, cfa = ((c0 + gi * Cg) >> 4) + (i - mf * n_cf);
        const long mfg = (long)b * n_mf + mf;
        if (cfa < cf1) {
          const float* sl = a.s1.slots + (mfg * cf1 + cfa) * 2;
          sm += sl[0] * a.s1.scale;
          sq += sl[1] * a.s1.scale * a.s1.scale;
        } else {
          const float* sl = a.s2.slots + (mfg * cf2 + (cfa - cf1)) * 2;
          sm += sl[0] * a.s2.scale;
          sq += sl[1] * a.s2.scale * a.s2.scale;
        }
      }
    }
    sm = sf_group_sum(sm, 32);
    sq = sf_group_sum(sq, 32);
    if (li == 0 && gi < ngs) {
      const double n = (double)HW * Cg;
      const double mean = (double)sm / n;
      double var = (double)sq / n - mean * mean;
      if (var < 0.0) var = 0.0;
      misc[8 + 2 * gi] = (float)mean;
      misc[9 + 2 * gi] = sf_rsqrt((float)var + a.eps);
    }
    sf_sync();
  } else if (a.norm == FNORM_LN) {
    // per-row statistics over all C channels (S == 1, k == 1): 256 / rows threads per row, two passes like nn.LayerNorm
    const int rows = 16 * WM, tpr = 256 / rows;
    const int row = tid / tpr, part = tid - row * tpr;
    const long m = mb + (long)row0 * a.W + row;
    float sm = 0.0f;
    for (int c4 = part; c4 < Cs4; c4 += tpr) {
      f32x4 v = fconv_load(a, m, c4 * 4, false);
      if (a.pre_gelu) { v[0] = sf_gelu(v[0]); v[1] = sf_gelu(v[1]); v[2] = sf_gelu(v[2]); v[3] = sf_gelu(v[3]); }
      sm += (v[0] + v[1]) + (v[2] + v[3]);
    }
    const float mean = sf_group_sum(sm, tpr) / (float)Cs;
    float sq = 0.0f;
    for (int c4 = part; c4 < Cs4; c4 += tpr) {
      f32x4 v = fconv_load(a, m, c4 * 4, false);
      if (a.pre_gelu) { v[0] = sf_gelu(v[0]); v[1] = sf_gelu(v[1]); v[2] = sf_gelu(v[2]); v[3] = sf_gelu(v[3]); }
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float d = v[j] - mean; sq = fmaf(d, d, sq); }
    }
    const float rstd = sf_rsqrt(sf_group_sum(sq, tpr) / (float)Cs + a.eps);
    if (part == 0) { misc[8 + 2 * row] = mean; misc[9 + 2 * row] = rstd; }
    sf_sync();
  }
  if (a.norm == FNORM_GN_SELF || a.norm == FNORM_GN_SLOTS) {
    // per-channel affine of this (image, slice): y = v * A + B  ==  ((v - mean) * rstd * gamma + beta) * (scale + 1) + shift
    for (int cl = tid; cl < Cs; cl += 256) {
      const int c = c0 + cl, gi = cl / Cg;
      const float mean = misc[8 + 2 * gi], rstd = misc[9 + 2 * gi];
      float A = rstd * a.gamma[c], Bv = a.beta[c] - mean * A;
      if (a.ss) {
        const float sc = a.ss[(long)b * a.ss_stride + c] + 1.0f, sh = a.ss[(long)b * a.ss_stride + a.C + c];
        A *= sc;
        Bv = Bv * sc + sh;
      }
      tabA[cl] = A;
      tabB[cl] = Bv;
    }
    sf_sync();
  }

  // ---- (c) stage the in-image frame rows: fp32 -> normalise -> activate -> bf16 [frame pixel][channel]
  {
    const int per_row = a.W * Cs4, cnt = FR * per_row;
    for (int i = tid; i < cnt; i += 256) {
      const int fr = i / per_row, rem = i - fr * per_row;
      const int x = rem / Cs4, c4 = rem - x * Cs4;
      const int r = row0 - h + fr;
      if (r < 0 || r >= a.H) continue;
      const int cl = c4 * 4;
      const long m = mb + (long)r * a.W + x;
      const bool own = (nt == 0) && fr >= h && fr < h + a.TR;
      f32x4 v = fconv_load(a, m, c0 + cl, own);
      if (a.norm == FNORM_GN_SELF || a.norm == FNORM_GN_SLOTS) {
        const f32x4 A = *reinterpret_cast<const f32x4*>(tabA + cl), Bv = *reinterpret_cast<const f32x4*>(tabB + cl);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaf(v[j], A[j], Bv[j]);
      } else if (a.norm == FNORM_LN) {
        const int row = (fr - h) * a.W + x;
        const float mean = misc[8 + 2 * row], rstd = misc[9 + 2 * row];
        const f32x4 g = *reinterpret_cast<const f32x4*>(a.gamma + c0 + cl);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float u = a.pre_gelu ? sf_gelu(v[j]) : v[j];
          v[j] = (u - mean) * rstd * g[j];
        }
        if (a.beta) {
          const f32x4 be = *reinterpret_cast<const f32x4*>(a.beta + c0 + cl);
          v += be;
        }
      }
      if (a.silu) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = sf_silu(v[j]);
      }
      bf16x4 o;
      o[0] = (__bf16)v[0]; o[1] = (__bf16)v[1]; o[2] = (__bf16)v[2]; o[3] = (__bf16)v[3];
      *reinterpret_cast<bf16x4*>(lds + (long)(fr * FW + x + h) * a.pix_stride + cl * 2) = o;
    }
  }
  sf_sync();

  // ---- main loop
  f32x4 acc[WM][WN];
#pragma unroll
  for (int mi = 0; mi < WM; ++mi)
#pragma unroll
    for (int ni = 0; ni < WN; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
  int abase[WM];
#pragma unroll
  for (int mi = 0; mi < WM; ++mi) {
    const int p = mi * 16 + (lane & 15);
    const int ty = p / a.W, tx = p - ty * a.W;
    abase[mi] = (ty * FW + tx) * a.pix_stride + (lane >> 4) * 16;
  }
  for (int j0 = k0; j0 < k1; j0 += D) {
#pragma unroll
    for (int u = 0; u < D; ++u) {
      const int j = j0 + u;
      if (j < k1) {
        const int tap = j / a.cps, ccl = j - tap * a.cps;
        const int ky = tap / a.k, kx = tap - ky * a.k;
        const int toff = (ky * FW + kx) * a.pix_stride + ccl * 64;
        bf16x8 fa[WM];
#pragma unroll
        for (int mi = 0; mi < WM; ++mi) fa[mi] = *reinterpret_cast<const bf16x8*>(lds + abase[mi] + toff);
#pragma unroll
        for (int mi = 0; mi < WM; ++mi)
#pragma unroll
          for (int ni = 0; ni < WN; ++ni) acc[mi][ni] = sf_mfma16(fa[mi], fb[u][ni], acc[mi][ni]);
#pragma unroll
        for (int ni = 0; ni < WN; ++ni) fb[u][ni] = wload(j + D, ni);     // refill the ring slot just consumed
      }
    }
  }

  // ---- epilogue: the 4 K-slices of the workgroup meet in LDS
  float* red = reinterpret_cast<float*>(lds + a.red_off);         // [wave][frag][r][lane]
#pragma unroll
  for (int mi = 0; mi < WM; ++mi)
#pragma unroll
    for (int ni = 0; ni < WN; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[((wave * (WM * WN) + mi * WN + ni) * 4 + r) * 64 + lane] = acc[mi][ni][r];
  sf_sync();
  const long m0 = mb + (long)row0 * a.W;
#pragma unroll
  for (int mi = 0; mi < WM; ++mi) {
#pragma unroll
    for (int ni = 0; ni < WN; ++ni) {
      const int f = mi * WN + ni;
      if ((f & 3) != wave) continue;
      const int nf = nt * WN + ni;
      if (nf >= a.n_frags) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int idx = (f * 4 + r) * 64 + lane;
        v[r] = (red[idx] + red[idx + WM * WN * 256]) + (red[idx + 2 * WM * WN * 256] + red[idx + 3 * WM * WN * 256]);
      }
      const int n = nf * 16 + (lane & 15);
      const long mrow = m0 + mi * 16 + (lane >> 4) * 4;
      if (a.S > 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) a.ws[((long)s * a.M + mrow + r) * a.npad + n] = v[r];
        continue;
      }
      float sm = 0.0f, sq = 0.0f;
      if (n < a.Cout) {
        const float bv = a.bias ? a.bias[n] : 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const long o = (mrow + r) * a.ldc + a.co_off + n;
          float y = v[r] + bv;
          if (a.resid) y += a.resid[o];
          if (a.accum) y += a.out[o];
          a.out[o] = y;
          sm += y;
          sq = fmaf(y, y, sq);
        }
      }
      if (a.slots_out) {
        sm = sf_wave_sum(sm);
        sq = sf_wave_sum(sq);
        if (lane == 0) {
          float* sl = a.slots_out + (((m0 >> 4) + mi) * (long)(a.ldc >> 4) + (a.co_off >> 4) + nf) * 2;
          sl[0] = sm;
          sl[1] = sq;
        }
      }
    }
  }
}

// (sum, sum of squares) slots of an fp32 NHWC tensor [M, C], one wave per 16 pixels x 16 channels; with `gate` the
// tensor is first formed as x = h * gate[batch] + res and written to `out` (GlobalContext gating + residual,
// imagen_pytorch.py:727-729, :936-941) so that the next GroupNorm-fused conv finds both the values and their sums.
SF_KERNEL(256) void k_slots(const float* __restrict__ x, const float* __restrict__ gate, const float* __restrict__ res,
                            float* __restrict__ out, float* __restrict__ slots, int M, int C, int HW) {
  const int lane = threadIdx.x & 63, gw = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int CF = C >> 4;
  if (gw >= (M >> 4) * CF) return;
  const int mf = gw / CF, cf = gw - mf * CF;
  const long m = (long)mf * 16 + (lane >> 2);
  const int c = cf * 16 + (lane & 3) * 4;
  f32x4 v = *reinterpret_cast<const f32x4*>(x + m * C + c);
  if (gate) {
    const f32x4 g = *reinterpret_cast<const f32x4*>(gate + (m / HW) * C + c);
    const f32x4 r = *reinterpret_cast<const f32x4*>(res + m * C + c);
    v = v * g + r;
    *reinterpret_cast<f32x4*>(out + m * C + c) = v;
  }
  float sm = (v[0] + v[1]) + (v[2] + v[3]);
  float sq = fmaf(v[0], v[0], fmaf(v[1], v[1], fmaf(v[2], v[2], v[3] * v[3])));
  sm = sf_wave_sum(sm);
  sq = sf_wave_sum(sq);
  if (lane == 0) { slots[(long)gw * 2] = sm; slots[(long)gw * 2 + 1] = sq; }
}

// k_conv4_gn: the GroupNorm-self 3x3 conv of the UNet's 4x4 level (external/imagen_pytorch.py:641-662 at 16 pixels), the same op,
// operands and LDS layout as k_conv_fused<1, 1, 12, FNORM_GN_SELF, LAZY, 8> (fused_kernels.h) for the one geometry every such layer
// of the plans has: H = W = 4, k = 3, a 16-pixel x 16-channel tile, S > 1 input-channel slices of TWO whole groups each
// (Cs = 256 | 512), split-K slabs out.  r05.
//
// Why a second kernel for one geometry.  A graph chain that imitates this launch -- 64 KB lazy slice, statistics, SiLU, LDS frame, 72 KB
// weight slab requested at entry, LDS reduction -- costs 6.9 us per launch on MI355X (tools/exp/weight_prefetch_chain.hip: boundary
// 2.05 + issue 0.5 + weight flight at 6.5 TB/s 2.9 + tail 1.5); the general kernel costs 11.3, and no reordering INSIDE it moved that by
// more than 0.5 us (profiles/r05_fconv4_order_lean_ab.log): its prologue is one source for every norm / lazy / tile / slice shape, with
// run-time selected paths -- and where two paths define the destination registers of loads in flight hipcc drains vmcnt(0) at the merge
// (seen in the ISA).  Here everything that shapes the instruction stream is a template parameter:
//   * every load of the launch is requested in the first ~100 instructions, in the order it is needed: the thread's 2 | 4 lazy elements
//     (its channel chunk is fixed: c4 = tid % CS4), the four float4 of ITS affine operands (gamma, beta, scale, shift: no LDS table),
//     then its whole share of the weight slice (9 | 18 k-steps: no ring refill);
//   * ONE barrier between the gather and the frame: segment sums by half-wave / wave shuffles, one partial per segment, and every
//     thread adds the 8 | 4 partials of its own group in segment order, in double (fixed order: run-to-run identical);
//   * main loop = 9 | 18 MFMAs per wave on LDS fragments, the 8 K-slices meet in LDS, wave 0 stores the slab tile (and the partial
//     context logits of a GlobalContext block's conv2).
// Values: the arithmetic of k_conv_fused up to the summation order of the group statistics (double either way).
#pragma once
#include "fused_kernels.h"

// CS4 = float4 chunks per slice (64: Cs = 256, 128: Cs = 512); LAZY = mode of source 1 (FSrc); NORM = false (r06): the same launch without the
// GroupNorm -- FNORM_NONE, the operand is the (scaled) source value, SiLU only if the op asks: the merged 3x3 + 1x1 conv of the last Downsample
// (imagen_pytorch.py:1294-1297), which ran on k_conv_igemm at 13.5 us where its GroupNorm-self neighbours take 7.5
template <int CS4, int LAZY, bool NORM = true>
SF_DEV void conv4_gn_body(const FConvArgs& a, const int bid) {
  constexpr int NT = 512, NE = CS4 / 32;              // 16 pixels x CS4 chunks over 512 threads: 2 | 4 elements per thread
  constexpr int CPS = CS4 / 8;                        // 32-channel k-chunks per slice: 8 | 16
  constexpr int KW = 9 * CPS / 8;                     // k-steps per wave: 9 | 18
  constexpr int W = CS4 / 2;                          // lanes per statistics segment = float4 chunks per group: 32 | 64
  constexpr int NSEG = NT / W;                        // 16 | 8 segments, segment sl belongs to group sl & 1
  constexpr int FW = 6;                               // frame: 6 x 6 pixels
  SF_DYN_LDS(lds);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles = a.B * a.n_tiles;
  const int s = bid / tiles, t = bid - s * tiles;
  const int nt = t % a.n_tiles, b = t / a.n_tiles;
  const int c0 = s * (CS4 * 4);
  const long mb = (long)b * 16;
  const float sc1 = a.s1.scale, sc2 = a.s2.scale;
  const int c4 = tid & (CS4 - 1), c = c0 + c4 * 4;
  const int px0 = tid / CS4;                          // element u is pixel px0 + u * (NT / CS4)

  // ---- (1) every load of the launch, in the order of need
  FGather<LAZY> gq[NE];
#pragma unroll
  for (int u = 0; u < NE; ++u) gq[u].issue(a, mb + px0 + u * (NT / CS4), c);
  f32x4 qg = f32x4{1.f, 1.f, 1.f, 1.f}, qb = f32x4{0.f, 0.f, 0.f, 0.f}, qsc = qb, qsh = qb;
  if constexpr (NORM) {
    const float* ssrow = a.ss ? a.ss + (long)b * a.ss_stride : a.gamma;      // any valid address when there is no scale / shift
    const int shoff = a.ss ? a.C : 0;
    qg = *reinterpret_cast<const f32x4*>(a.gamma + c);
    qb = *reinterpret_cast<const f32x4*>(a.beta + c);
    qsc = *reinterpret_cast<const f32x4*>(ssrow + c);
    qsh = *reinterpret_cast<const f32x4*>(ssrow + shoff + c);
  }
  const int nf = nt < a.n_frags ? nt : a.n_frags - 1;
  const int n = nf * 16 + (lane & 15);
  // context-logit weight of this output channel (wave 0 uses it): an UNCONDITIONAL load from a selected address, ahead of the weight
  // slice -- a load under `if (a.wk)` behind it would make the compiler drain the whole slice at the merge
  const float wkq = (a.wk ? a.wk : reinterpret_cast<const float*>(a.w))[a.wk && n < a.Cout ? n : 0];
  const float wkv = a.wk ? wkq : 0.0f;
  const bf16x8* wbase = a.w + ((long)nf * a.KS + s * CPS) * 64 + lane;
  bf16x8 fb[KW];
#pragma unroll
  for (int i = 0; i < KW; ++i) {
    const int j = wave * KW + i, tap = j / CPS, ccl = j - tap * CPS;
    fb[i] = __builtin_nontemporal_load(&wbase[(long)(tap * a.cchunks + ccl) * 64]);
  }

  // ---- (2) zero padding of the frame: the 20 border pixels, 8 threads per pixel
  for (int q = tid >> 3; q < FW * FW; q += NT / 8) {
    const int fr = q / FW, fx = q - fr * FW;
    if (fr == 0 || fr == FW - 1 || fx == 0 || fx == FW - 1) {
      char* dst = lds + (long)q * a.pix_stride;
      for (int c8 = (tid & 7); c8 < CS4 / 2; c8 += 8) *reinterpret_cast<bf16x8*>(dst + c8 * 16) = sf_zero8();
    }
  }

  // ---- (3) element values, segment sums (this thread's elements lie in ONE group: c4 is fixed)
  const bool first = c < a.s1.C;
  const float scl = first ? sc1 : sc2;
  f32x4 v[NE];
  float sm = 0.0f, sq = 0.0f;
#pragma unroll
  for (int u = 0; u < NE; ++u) {
    v[u] = gq[u].combine();
    const f32x4 w = v[u] * scl;
    sm += (w[0] + w[1]) + (w[2] + w[3]);
    sq = fmaf(w[0], w[0], sq); sq = fmaf(w[1], w[1], sq); sq = fmaf(w[2], w[2], sq); sq = fmaf(w[3], w[3], sq);
  }
  if constexpr (NORM) {
    sm = sf_group_sum(sm, W);
    sq = sf_group_sum(sq, W);
    float* misc = reinterpret_cast<float*>(lds + a.misc_off);
    float* part = misc + 160;                              // [NSEG][2]
    if ((lane & (W - 1)) == 0) { part[2 * (tid / W)] = sm; part[2 * (tid / W) + 1] = sq; }
    sf_sync();
    // ---- (4) (mean, rstd) of this thread's group: its NSEG / 2 partials in segment order (uniform addresses per half-wave: broadcasts)
    const int gi = c4 / W;                                 // 0 | 1
    double S = 0.0, Q = 0.0;
#pragma unroll
    for (int k = 0; k < NSEG / 2; ++k) {
      S += (double)part[2 * (2 * k + gi)];
      Q += (double)part[2 * (2 * k + gi) + 1];
    }
    const double mean_d = S * a.inv_n;
    double var = Q * a.inv_n - mean_d * mean_d;
    if (var < 0.0) var = 0.0;
    const float mean = (float)mean_d, rstd = sf_rsqrt((float)var + a.eps);
    // y = x * A + B  ==  ((x - mean) * rstd * gamma + beta) * (scale + 1) + shift   (x = the scaled source value)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float A = rstd * qg[j], scj = a.ss ? qsc[j] + 1.0f : 1.0f, shj = a.ss ? qsh[j] : 0.0f;
      qg[j] = A * scj;
      qb[j] = (qb[j] - mean * A) * scj + shj;
    }
  }
  // ---- (5) normalise, activate, bf16 into the frame; the workgroups of n-tile 0 materialise a lazy first source
#pragma unroll
  for (int u = 0; u < NE; ++u) {
    const int p = px0 + u * (NT / CS4), py = p >> 2, pxx = p & 3;
    if (LAZY && nt == 0 && first && a.s1.p) *reinterpret_cast<f32x4*>(a.s1.p + (mb + p) * a.s1.C + c) = v[u];
    f32x4 y = (v[u] * scl) * qg + qb;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float sv = sf_silu_fast(y[j]);
      y[j] = a.silu ? sv : y[j];
    }
    bf16x4 o;
    o[0] = (sf_opnd)y[0]; o[1] = (sf_opnd)y[1]; o[2] = (sf_opnd)y[2]; o[3] = (sf_opnd)y[3];
    *reinterpret_cast<bf16x4*>(lds + (long)((py + 1) * FW + pxx + 1) * a.pix_stride + c4 * 8) = o;
  }
  sf_sync();

  // ---- (6) main loop: this wave's k-steps (tap-major, then 32-channel chunk) on the frame
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  const int pA = lane & 15;
  const char* abase = lds + (long)((pA >> 2) * FW + (pA & 3)) * a.pix_stride + (lane >> 4) * 16;
#pragma unroll
  for (int i = 0; i < KW; ++i) {
    const int j = wave * KW + i, tap = j / CPS, ccl = j - tap * CPS;
    const int ky = tap / 3, kx = tap - ky * 3;
    const bf16x8 fa = *reinterpret_cast<const bf16x8*>(abase + (long)(ky * FW + kx) * a.pix_stride + ccl * 64);
    acc = sf_mfma16(fa, fb[i], acc);
  }
  // ---- (7) the 8 K-slices of the workgroup meet in LDS; wave 0 stores the slab tile
  float* red = reinterpret_cast<float*>(lds + a.red_off);         // [wave][r][lane]
#pragma unroll
  for (int r = 0; r < 4; ++r) red[(wave * 4 + r) * 64 + lane] = acc[r];
  sf_sync();
  if (wave == 0 && nt < a.n_frags) {
    const long mrow = mb + (lane >> 4) * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float sacc = 0.0f;
#pragma unroll
      for (int w = 0; w < 8; ++w) sacc += red[(w * 4 + r) * 64 + lane];
      if (a.logit_part) {                    // bias terms are the same for every pixel: they cancel in the softmax
        float lp = sacc * wkv;
        lp += sf_shfl_xor(lp, 1); lp += sf_shfl_xor(lp, 2); lp += sf_shfl_xor(lp, 4); lp += sf_shfl_xor(lp, 8);
        if ((lane & 15) == 0) a.logit_part[((long)s * a.n_frags + nt) * a.M + mrow + r] = lp;
      }
      a.ws[((long)s * a.M + mrow + r) * a.npad + n] = sacc;
    }
  }
}

template <int CS4, int LAZY, bool NORM = true>
SF_KERNEL(512) void k_conv4_gn(FConvArgs a) {
  sf_touch_kernarg<(int)sizeof(FConvArgs)>();
  conv4_gn_body<CS4, LAZY, NORM>(a, (int)blockIdx.x);
}

// k_conv4_gn_mb: the same op for NB = 2 | 4 images per workgroup (r05, B >= 2).  With one image per workgroup every image's 256 workgroups
// stream the layer's weights again (B = 4: 19.5 us per launch against 7.5 at B = 1, two to four rounds of workgroups); here the workgroup of
// (slice, n-tile) holds its weight share ONCE and the images ride side by side through every phase -- NB frames in LDS, one statistics
// barrier for all of them, each weight fragment meets NB A fragments (the images are extra rows of the MFMA's M side), waves 0 .. NB - 1
// finalise one image each.  A first batched form that WALKED the images (one chain after the other) was slower than one image per
// workgroup (24.3 us, profiles/r05_conv4_batch_ab.log): the chains must overlap, not queue.
// Registers bound how many images' lazy gathers are in flight together (a split-K element is 5 float4 loads + the thread's bias): SETS
// images at a time, the set of image ib is re-issued for image ib + SETS as soon as it is combined; the per-image scale / shift rows are
// requested behind the last image's gather, into the registers a combined set leaves.  LDS: frame ib at ib * a.buf_bytes, red = [NB][8][4][64].
template <int CS4, int LAZY, int NB>
SF_DEV void conv4_gn_mb_body(const FConvArgs& a, const int bid) {
  constexpr int NT = 512, NE = CS4 / 32, CPS = CS4 / 8, KW = 9 * CPS / 8, W = CS4 / 2, NSEG = NT / W, FW = 6;
  constexpr int K = LAZY == 1 ? 5 : (LAZY == 2 ? 3 : 1);   // loads per element (split-K: slabs 0..3 + residual; its bias is one load per THREAD)
  constexpr int SETS = (NB * NE * K * 4 <= 112) ? NB : ((2 * NE * K * 4 <= 112 && NB >= 2) ? 2 : 1);
  SF_DYN_LDS(lds);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles = (a.B / NB) * a.n_tiles;
  const int s = bid / tiles, t = bid - s * tiles;
  const int nt = t % a.n_tiles, b0 = (t / a.n_tiles) * NB;
  const int c0 = s * (CS4 * 4);
  const float sc1 = a.s1.scale, sc2 = a.s2.scale;
  const int c4 = tid & (CS4 - 1), c = c0 + c4 * 4;
  const int px0 = tid / CS4;

  // ---- (1) loads: the first SETS images' elements, gamma / beta, the weight share.  The element gather is FGather's (fused_kernels.h:
  // selected addresses, 0 | 1 weights, no control flow, the same summation order) with the per-THREAD parts hoisted: the channel chunk c
  // is fixed, so the weights and a split-K source's bias are loaded / formed once, not per element
  // A slice lies in ONE source (host: s1.C % Cs == 0), so `first` is uniform over the workgroup: the base pointers are scalar selects and
  // every load is (scalar base) + (one 32-bit element offset per lane) -- the four slabs of an element share their offset register.
  const bool first = c0 < a.s1.C;
  const unsigned cc = first ? (unsigned)c : (unsigned)(c - a.s1.C);
  const int gl = a.s1.groups - 1;
  const long gstride = (long)a.M * a.s1.npad;
  float gw[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) gw[g] = first ? (g <= gl ? 1.0f : 0.0f) : (g == 0 ? 1.0f : 0.0f);
  const float wbias = (LAZY == 1 && first && a.s1.b) ? 1.0f : 0.0f, wres = (LAZY == 1 && first && a.s1.r) ? 1.0f : 0.0f;
  const float* base[K];                                 // uniform
  unsigned ld[K];                                       // row stride of each load's tensor, in floats
  if constexpr (LAZY == 0) {
    base[0] = first ? a.s1.p : a.s2.p;
    ld[0] = first ? a.s1.C : a.s2.C;
  } else if constexpr (LAZY == 1) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      base[g] = first ? a.s1.a + (g < gl ? g : gl) * gstride : a.s2.p;
      ld[g] = first ? a.s1.npad : a.s2.C;
    }
    base[4] = first ? (a.s1.r ? a.s1.r : a.s1.a) : a.s2.p;
    ld[4] = first ? (a.s1.r ? a.s1.C : a.s1.npad) : a.s2.C;
  } else {
    base[0] = first ? a.s1.a : a.s2.p;
    base[1] = first ? a.s1.b : a.s2.p;
    base[2] = first ? a.s1.r : a.s2.p;
    ld[0] = ld[2] = first ? a.s1.C : a.s2.C;
    ld[1] = ld[0];
  }
  f32x4 gt[SETS][NE][K];
  auto issue = [&](int set, int u, unsigned m) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const unsigned row = (LAZY == 2 && k == 1 && first) ? (m >> 4) : m;        // the gate is one row per image
      gt[set][u][k] = *reinterpret_cast<const f32x4*>(base[k] + (row * ld[k] + cc));
    }
  };
  f32x4 qbias = f32x4{0.f, 0.f, 0.f, 0.f};
  if constexpr (LAZY == 1) qbias = *reinterpret_cast<const f32x4*>((first && a.s1.b) ? a.s1.b + cc : a.gamma + c);
  auto combine = [&](int set, int u) -> f32x4 {
    if constexpr (LAZY == 0) return gt[set][u][0];
    else if constexpr (LAZY == 1) {                       // order of k_splitk_reduce: bias, slab 0, 1, .., residual (second source: t[0] alone)
      f32x4 r = qbias * wbias;
#pragma unroll
      for (int g = 0; g < 4; ++g) r += gt[set][u][g] * gw[g];
      r += gt[set][u][4] * wres;
      return r;
    } else return first ? gt[set][u][0] * gt[set][u][1] + gt[set][u][2] : gt[set][u][0];
  };
#pragma unroll
  for (int ib = 0; ib < SETS; ++ib)
#pragma unroll
    for (int u = 0; u < NE; ++u) issue(ib, u, (unsigned)((b0 + ib) * 16 + px0 + u * (NT / CS4)));
  const int shoff = a.ss ? a.C : 0;
  f32x4 qg = *reinterpret_cast<const f32x4*>(a.gamma + c);
  f32x4 qb = *reinterpret_cast<const f32x4*>(a.beta + c);
  const int nf = nt < a.n_frags ? nt : a.n_frags - 1;
  const int n = nf * 16 + (lane & 15);
  float wkq = (a.wk ? a.wk : a.gamma)[a.wk && n < a.Cout ? n : 0];     // context-logit weight of this output channel: first used in (7)
  const bf16x8* wbase = a.w + ((long)nf * a.KS + s * CPS) * 64 + lane;
  bf16x8 fb[KW];
#pragma unroll
  for (int i = 0; i < KW; ++i) {
    const int j = wave * KW + i, tap = j / CPS, ccl = j - tap * CPS;
    fb[i] = __builtin_nontemporal_load(&wbase[(long)(tap * a.cchunks + ccl) * 64]);
  }

  // ---- (2) zero padding of the NB frames: 20 border pixels each, 8 threads per pixel
  for (int q = tid >> 3; q < NB * FW * FW; q += NT / 8) {
    const int ib = q / (FW * FW), qq = q - ib * (FW * FW);
    const int fr = qq / FW, fx = qq - fr * FW;
    if (fr == 0 || fr == FW - 1 || fx == 0 || fx == FW - 1) {
      char* dst = lds + (long)ib * a.buf_bytes + (long)qq * a.pix_stride;
      for (int c8 = (tid & 7); c8 < CS4 / 2; c8 += 8) *reinterpret_cast<bf16x8*>(dst + c8 * 16) = sf_zero8();
    }
  }

  // ---- (3) element values and segment sums, image by image; a combined set goes out again for image ib + SETS
  const float scl = first ? sc1 : sc2;
  f32x4 v[NB][NE], qsc[NB], qsh[NB];
  float* misc = reinterpret_cast<float*>(lds + a.misc_off);
  float* part = misc + 160;                              // [NB][NSEG][2]
#pragma unroll
  for (int ib = 0; ib < NB; ++ib) {
    float sm = 0.0f, sq = 0.0f;
#pragma unroll
    for (int u = 0; u < NE; ++u) {
      v[ib][u] = combine(ib % SETS, u);
      const f32x4 w = v[ib][u] * scl;
      sm += (w[0] + w[1]) + (w[2] + w[3]);
      sq = fmaf(w[0], w[0], sq); sq = fmaf(w[1], w[1], sq); sq = fmaf(w[2], w[2], sq); sq = fmaf(w[3], w[3], sq);
    }
    if (ib + SETS < NB) {
#pragma unroll
      for (int u = 0; u < NE; ++u) issue(ib % SETS, u, (unsigned)((b0 + ib + SETS) * 16 + px0 + u * (NT / CS4)));
    }
    if (ib == NB - 2) {
      // the images' scale / shift rows go out HERE, into the registers the combined set leaves, behind the last image's gather: they
      // arrive under its sums and the barrier (requested at entry they cost 8 NB registers at the kernel's peak: spills at NB = 4)
      const float* ssb = a.ss ? a.ss : a.gamma;                        // any valid address when there is no scale / shift
      const unsigned sstr = a.ss ? (unsigned)a.ss_stride : 0u;
#pragma unroll
      for (int jb = 0; jb < NB; ++jb) {
        qsc[jb] = *reinterpret_cast<const f32x4*>(ssb + ((unsigned)(b0 + jb) * sstr + (unsigned)c));
        qsh[jb] = *reinterpret_cast<const f32x4*>(ssb + ((unsigned)(b0 + jb) * sstr + (unsigned)(shoff + c)));
      }
    }
    sm = sf_group_sum(sm, W);
    sq = sf_group_sum(sq, W);
    if ((lane & (W - 1)) == 0) { part[2 * (ib * NSEG + tid / W)] = sm; part[2 * (ib * NSEG + tid / W) + 1] = sq; }
  }
  sf_sync();
  // ---- (4) + (5) per image: (mean, rstd) of this thread's group, the affine in registers, normalise / activate / bf16 into frame ib
  const int gi = c4 / W;
#pragma unroll
  for (int ib = 0; ib < NB; ++ib) {
    double S = 0.0, Q = 0.0;
#pragma unroll
    for (int k = 0; k < NSEG / 2; ++k) {
      S += (double)part[2 * (ib * NSEG + 2 * k + gi)];
      Q += (double)part[2 * (ib * NSEG + 2 * k + gi) + 1];
    }
    const double mean_d = S * a.inv_n;
    double var = Q * a.inv_n - mean_d * mean_d;
    if (var < 0.0) var = 0.0;
    const float mean = (float)mean_d, rstd = sf_rsqrt((float)var + a.eps);
    f32x4 A, Bv;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float A0 = rstd * qg[j], scj = a.ss ? qsc[ib][j] + 1.0f : 1.0f, shj = a.ss ? qsh[ib][j] : 0.0f;
      A[j] = A0 * scj;
      Bv[j] = (qb[j] - mean * A0) * scj + shj;
    }
    char* frame = lds + (long)ib * a.buf_bytes;
#pragma unroll
    for (int u = 0; u < NE; ++u) {
      const int p = px0 + u * (NT / CS4), py = p >> 2, pxx = p & 3;
      if (LAZY && nt == 0 && first && a.s1.p) *reinterpret_cast<f32x4*>(a.s1.p + ((long)(b0 + ib) * 16 + p) * a.s1.C + c) = v[ib][u];
      f32x4 y = (v[ib][u] * scl) * A + Bv;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float sv = sf_silu_fast(y[j]);
        y[j] = a.silu ? sv : y[j];
      }
      bf16x4 o;
      o[0] = (sf_opnd)y[0]; o[1] = (sf_opnd)y[1]; o[2] = (sf_opnd)y[2]; o[3] = (sf_opnd)y[3];
      *reinterpret_cast<bf16x4*>(frame + (long)((py + 1) * FW + pxx + 1) * a.pix_stride + c4 * 8) = o;
    }
  }
  sf_sync();

  // ---- (6) main loop: every weight fragment of this wave meets the NB images' A fragments
  f32x4 acc[NB];
#pragma unroll
  for (int ib = 0; ib < NB; ++ib) acc[ib] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int pA = lane & 15;
  const char* abase = lds + (long)((pA >> 2) * FW + (pA & 3)) * a.pix_stride + (lane >> 4) * 16;
#pragma unroll
  for (int i = 0; i < KW; ++i) {
    const int j = wave * KW + i, tap = j / CPS, ccl = j - tap * CPS;
    const int ky = tap / 3, kx = tap - ky * 3;
    const long off = (long)(ky * FW + kx) * a.pix_stride + ccl * 64;
    bf16x8 fa[NB];
#pragma unroll
    for (int ib = 0; ib < NB; ++ib) fa[ib] = *reinterpret_cast<const bf16x8*>(abase + (long)ib * a.buf_bytes + off);
#pragma unroll
    for (int ib = 0; ib < NB; ++ib) acc[ib] = sf_mfma16(fa[ib], fb[i], acc[ib]);
  }
  // ---- (7) the 8 K-slices meet in LDS; wave ib stores the slab tile of image ib
  float* red = reinterpret_cast<float*>(lds + a.red_off);         // [image][wave][r][lane]
#pragma unroll
  for (int ib = 0; ib < NB; ++ib)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[((ib * 8 + wave) * 4 + r) * 64 + lane] = acc[ib][r];
  sf_sync();
  SF_USE_FROM_HERE(wkq);
  const float wkv = a.wk ? wkq : 0.0f;
  if (wave < NB && nt < a.n_frags) {
    const long mrow = (long)(b0 + wave) * 16 + (lane >> 4) * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float sacc = 0.0f;
#pragma unroll
      for (int w = 0; w < 8; ++w) sacc += red[((wave * 8 + w) * 4 + r) * 64 + lane];
      if (a.logit_part) {
        float lp = sacc * wkv;
        lp += sf_shfl_xor(lp, 1); lp += sf_shfl_xor(lp, 2); lp += sf_shfl_xor(lp, 4); lp += sf_shfl_xor(lp, 8);
        if ((lane & 15) == 0) a.logit_part[((long)s * a.n_frags + nt) * a.M + mrow + r] = lp;
      }
      a.ws[((long)s * a.M + mrow + r) * a.npad + n] = sacc;
    }
  }
}

template <int CS4, int LAZY, int NB>
SF_KERNEL(512) void k_conv4_gn_mb(FConvArgs a) {
  sf_touch_kernarg<(int)sizeof(FConvArgs)>();
  conv4_gn_mb_body<CS4, LAZY, NB>(a, (int)blockIdx.x);
}

// k_lin4_ln: LayerNorm -> Linear on the 16-token map (the transformer blocks of the 4x4 level: merged q | k | v projection, ff1 with its
// GELU epilogue, ff2 with its residual; external/imagen_pytorch.py:480-566, :944-1010) -- the same op, operands and LDS layout as
// k_conv_fused<1, WN, ., FNORM_LN, 0, 8> for a plain source of C = 1024 | 2048 channels, built like k_conv4_gn: the row (8 | 16 float4 per
// thread, 32 threads per token), the gain (and bias) of the SAME chunks and the wave's whole weight share (4 | 8 k-steps x WN fragments)
// are requested in the first instructions; two-pass statistics like nn.LayerNorm by 32-lane shuffles; ONE barrier in front of the MFMAs.
// The general kernel fetched the gain after the statistics (a dependent L2 round trip) and half of its row loads were clamped dead
// elements at C = 1024.  C4T = float4 chunks per thread (C / 128).
template <int C4T, int WN>
SF_DEV void lin4_ln_body(const FConvArgs& a, const int bid) {
  constexpr int TPR = 32;                              // 512 threads = 16 tokens x 32 threads
  constexpr int KSL = C4T * 4;                         // 32-channel k-steps: C / 32 = 32 | 64
  constexpr int KW = KSL / 8;                          // per wave: 4 | 8
  constexpr int F = WN;
  SF_DYN_LDS(lds);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int MT = a.B;
  int mt, nt;
  fconv_tile_of(a, bid, MT, mt, nt);
  const long m0 = (long)mt * 16;
  const int row = tid / TPR, part = tid - row * TPR;
  const long m = m0 + row;
  // ---- (1) every load of the launch: the token's row, gain / bias of the same chunks, the weight share, the epilogue operands
  f32x4 v[C4T], g[C4T], bt[C4T];
  const float* xr = a.s1.p + m * a.s1.C;
#pragma unroll
  for (int u = 0; u < C4T; ++u) v[u] = *reinterpret_cast<const f32x4*>(xr + (part + u * TPR) * 4);
#pragma unroll
  for (int u = 0; u < C4T; ++u) g[u] = *reinterpret_cast<const f32x4*>(a.gamma + (part + u * TPR) * 4);
  const float* betap = a.beta ? a.beta : a.gamma;      // selected address, unconditional loads (no load under a run-time branch)
#pragma unroll
  for (int u = 0; u < C4T; ++u) bt[u] = *reinterpret_cast<const f32x4*>(betap + (part + u * TPR) * 4);
  const bf16x8* wbase[WN];
#pragma unroll
  for (int ni = 0; ni < WN; ++ni) {
    int nf = nt * WN + ni;
    if (nf > a.n_frags - 1) nf = a.n_frags - 1;
    wbase[ni] = a.w + (long)nf * a.KS * 64 + lane;
  }
  bf16x8 fb[KW][WN];
#pragma unroll
  for (int i = 0; i < KW; ++i)
#pragma unroll
    for (int ni = 0; ni < WN; ++ni) {
      fb[i][ni] = __builtin_nontemporal_load(&wbase[ni][(long)(wave * KW + i) * 64]);
    }
  const int my_nf = nt * WN + wave;                    // wave f < F finalises fragment f
  const bool fin = wave < F && my_nf < a.n_frags;
  const int n = (my_nf < a.n_frags ? my_nf : a.n_frags - 1) * 16 + (lane & 15);
  const int nc = n < a.Cout ? n : a.Cout - 1;
  const long mrow = m0 + (lane >> 4) * 4;
  const float bvq = (a.bias ? a.bias : a.gamma)[a.bias ? nc : 0];
  const float bv = a.bias ? bvq : 0.0f;
  float rv[4];
  {
    const float* rp = a.resid ? a.resid : a.out;        // selected address; out is always readable
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float q = rp[(mrow + r) * a.ldc + a.co_off + nc];
      rv[r] = a.resid ? q : 0.0f;
    }
    if (a.accum) {
#pragma unroll
      for (int r = 0; r < 4; ++r) rv[r] += a.out[(mrow + r) * a.ldc + a.co_off + nc];
    }
  }
  // ---- (2) two-pass statistics of the token over its 32 threads
  float sm = 0.0f;
#pragma unroll
  for (int u = 0; u < C4T; ++u) {
    if (a.pre_gelu) { v[u][0] = sf_gelu(v[u][0]); v[u][1] = sf_gelu(v[u][1]); v[u][2] = sf_gelu(v[u][2]); v[u][3] = sf_gelu(v[u][3]); }
    sm += (v[u][0] + v[u][1]) + (v[u][2] + v[u][3]);
  }
  const float mean = sf_group_sum(sm, TPR) / (float)(C4T * 128);
  float sq = 0.0f;
#pragma unroll
  for (int u = 0; u < C4T; ++u) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float d = v[u][j] - mean; sq = fmaf(d, d, sq); }
  }
  const float rstd = sf_rsqrt(sf_group_sum(sq, TPR) / (float)(C4T * 128) + a.eps);
  // ---- (3) normalise (gain, bias), optional SiLU, bf16 into the frame [token][channel]
#pragma unroll
  for (int u = 0; u < C4T; ++u) {
    f32x4 y;
#pragma unroll
    for (int j = 0; j < 4; ++j) y[j] = (v[u][j] - mean) * rstd * g[u][j];
    if (a.beta) y += bt[u];
    if (a.silu) { y[0] = sf_silu_fast(y[0]); y[1] = sf_silu_fast(y[1]); y[2] = sf_silu_fast(y[2]); y[3] = sf_silu_fast(y[3]); }
    bf16x4 o;
    o[0] = (sf_opnd)y[0]; o[1] = (sf_opnd)y[1]; o[2] = (sf_opnd)y[2]; o[3] = (sf_opnd)y[3];
    *reinterpret_cast<bf16x4*>(lds + (long)row * a.pix_stride + (part + u * TPR) * 8) = o;
  }
  sf_sync();
  // ---- (4) this wave's k-steps
  f32x4 acc[WN];
#pragma unroll
  for (int ni = 0; ni < WN; ++ni) acc[ni] = f32x4{0.f, 0.f, 0.f, 0.f};
  const char* abase = lds + (long)(lane & 15) * a.pix_stride + (lane >> 4) * 16;
#pragma unroll
  for (int i = 0; i < KW; ++i) {
    const bf16x8 fa = *reinterpret_cast<const bf16x8*>(abase + (wave * KW + i) * 64);
#pragma unroll
    for (int ni = 0; ni < WN; ++ni) acc[ni] = sf_mfma16(fa, fb[i][ni], acc[ni]);
  }
  // ---- (5) the 8 K-slices meet in LDS; wave f finalises fragment f: bias, residual, GELU, output, statistics slots
  float* red = reinterpret_cast<float*>(lds + a.red_off);         // [wave][frag][r][lane]
#pragma unroll
  for (int ni = 0; ni < WN; ++ni)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[((wave * F + ni) * 4 + r) * 64 + lane] = acc[ni][r];
  sf_sync();
  if (fin) {
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float sacc = 0.0f;
#pragma unroll
      for (int w = 0; w < 8; ++w) sacc += red[((w * F + wave) * 4 + r) * 64 + lane];
      if (n < a.Cout) {
        float y = sacc + bv + rv[r];
        if (a.out_gelu) y = sf_gelu(y);
        a.out[(mrow + r) * a.ldc + a.co_off + n] = y;
        s1 += y;
        s2 = fmaf(y, y, s2);
      }
    }
    if (a.slots_out) {
      s1 = sf_wave_sum(s1);
      s2 = sf_wave_sum(s2);
      if (lane == 0) {
        float* slo = a.slots_out + ((m0 >> 4) * (long)(a.ldc >> 4) + (a.co_off >> 4) + my_nf) * 2;
        slo[0] = s1;
        slo[1] = s2;
      }
    }
  }
}

template <int C4T, int WN>
SF_KERNEL(512) void k_lin4_ln(FConvArgs a) {
  sf_touch_kernarg<(int)sizeof(FConvArgs)>();
  lin4_ln_body<C4T, WN>(a, (int)blockIdx.x);
}

// k_lin4_attn (r06): the attention core in the prologue of its output projection (FNORM_ATTN: 8 heads x 64 = 512 inner channels, the 16 query
// tokens of one image, <= 24 keys; external/imagen_pytorch.py:480-566, :731-805) on k_lin4_ln's skeleton -- the same op, operands and LDS layout
// as k_conv_fused<1, WN, ., FNORM_ATTN, 0, 8>, whose prologue text this is, with what the general kernel keeps generic made static: 512
// channels = 16 k-steps = TWO per wave, so the wave's whole weight share (2 x WN fragments) is requested right behind the q / k / v loads
// (no ring), bias and residual rows of the finalising waves at entry, ONE barrier between the frame and the MFMAs.
template <int WN>
SF_DEV void lin4_attn_body(const FConvArgs& a, const int bid) {
  constexpr int KW = 2, F = WN, NW = 8;
  SF_DYN_LDS(lds);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int MT = a.B;
  int mt, nt;
  fconv_tile_of(a, bid, MT, mt, nt);
  const long m0 = (long)mt * 16;
  const int b = mt;
  const bf16x8* wbase[WN];
#pragma unroll
  for (int ni = 0; ni < WN; ++ni) {
    int nf = nt * WN + ni;
    if (nf > a.n_frags - 1) nf = a.n_frags - 1;
    wbase[ni] = a.w + (long)nf * a.KS * 64 + lane;
  }
  bf16x8 fb[KW][WN];
  const int my_nf = nt * WN + wave;                    // wave f < F finalises fragment f
  const bool fin = wave < F && my_nf < a.n_frags;
  const int n = (my_nf < a.n_frags ? my_nf : a.n_frags - 1) * 16 + (lane & 15);
  const int nc = n < a.Cout ? n : a.Cout - 1;
  const long mrow = m0 + (lane >> 4) * 4;
  float bv = 0.0f, rv[4] = {0.f, 0.f, 0.f, 0.f};
#define LIN4_ATTN_WEIGHTS() do { \
    _Pragma("unroll") for (int i = 0; i < KW; ++i) \
      _Pragma("unroll") for (int ni = 0; ni < WN; ++ni) fb[i][ni] = __builtin_nontemporal_load(&wbase[ni][(long)(wave * KW + i) * 64]); \
    const float bvq = (a.bias ? a.bias : reinterpret_cast<const float*>(a.w))[a.bias ? nc : 0]; \
    bv = a.bias ? bvq : 0.0f; \
    const float* rp = a.resid ? a.resid : a.out; \
    _Pragma("unroll") for (int r = 0; r < 4; ++r) { const float q_ = rp[(mrow + r) * a.ldc + a.co_off + nc]; rv[r] = a.resid ? q_ : 0.0f; } \
    if (a.accum) { _Pragma("unroll") for (int r = 0; r < 4; ++r) rv[r] += a.out[(mrow + r) * a.ldc + a.co_off + nc]; } \
  } while (0)
  // ---- (b1'') the attention core: wave = head (8 waves x 64 lanes = the 512 inner channels), the tile's 16 pixels = the 16
  // query tokens.  Keys / values are staged once in LDS (fp32), scores and P . V run on the matrix cores, the softmax on the D
  // fragments by shuffles; the result goes straight into the frame as the conv's A operand.
  const FAttn& at = a.attn;
  float* sp = reinterpret_cast<float*>(lds + a.attn_off);                 // [8 heads][16 queries][SF_ATTN_PSTRIDE]
  float* skv = sp + 8 * 16 * SF_ATTN_PSTRIDE;                             // keys [regions][J][KSTRIDE], then values
  const int J = at.J, nreg = at.per_head ? 8 : 1;
  // query A fragments straight from global memory: row (lane & 15), dims 32 ks + 8 (lane >> 4) .. + 7 of head `wave`
  const float* qp = at.q + ((long)b * 16 + (lane & 15)) * at.ldq + wave * 64 + 8 * (lane >> 4);
  f32x4 q[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) q[u] = *reinterpret_cast<const f32x4*>(qp + 32 * (u >> 1) + 4 * (u & 1));
  // key / value rows: per-head segments -> this wave stages the J rows of ITS head; shared head -> the 8 waves split the rows
  // (all of a wave's <= 4 rows are loaded before the first LDS store, from clamped addresses: a load inside the row loop was one
  // cold round trip per row -- the rows were written by the previous kernel on other XCDs -- 7.7 us of prologue instead of ~2)
  {
    const int j0 = at.per_head ? 0 : wave, jst = at.per_head ? 1 : NW;
    const int reg = at.per_head ? wave : 0;
    const int r0 = at.seg[0].rows, r1 = at.seg[1].rows;
    constexpr int NR = 4;                                            // per head: J <= 4; shared: ceil(24 / 8) = 3
    float kq[NR], vq[NR];
#pragma unroll
    for (int u = 0; u < NR; ++u) {
      const int j = j0 + u * jst < J ? j0 + u * jst : J - 1;
      const int si = j < r0 ? 0 : (j < r0 + r1 ? 1 : 2);
      const int r = j - (si == 0 ? 0 : (si == 1 ? r0 : r0 + r1));
      const float* kp = si == 0 ? at.seg[0].k : (si == 1 ? at.seg[1].k : at.seg[2].k);
      const int voff = si == 0 ? at.seg[0].v_off : (si == 1 ? at.seg[1].v_off : at.seg[2].v_off);
      const int rs = si == 0 ? at.seg[0].row_stride : (si == 1 ? at.seg[1].row_stride : at.seg[2].row_stride);
      const int bs = si == 0 ? at.seg[0].batch_stride : (si == 1 ? at.seg[1].batch_stride : at.seg[2].batch_stride);
      const int hs = si == 0 ? at.seg[0].head_stride : (si == 1 ? at.seg[1].head_stride : at.seg[2].head_stride);
      const long off = (long)b * bs + (long)r * rs + (long)wave * (at.per_head ? hs : 0) + lane;
      kq[u] = kp[off];
      vq[u] = kp[off + voff];
    }
    LIN4_ATTN_WEIGHTS();    // the weight share goes out BEHIND the q / k / v loads: loads return in order, and the first wait below is for k / v
#pragma unroll
    for (int u = 0; u < NR; ++u) {
      const int j = j0 + u * jst;
      if (j < J) {
        skv[(reg * J + j) * SF_ATTN_KSTRIDE + lane] = kq[u];
        skv[((nreg + reg) * J + j) * SF_ATTN_KSTRIDE + lane] = vq[u];
      }
    }
  }
  sf_sync();
  const float* kb = skv + (at.per_head ? wave : 0) * J * SF_ATTN_KSTRIDE;
  const float* vb = skv + (nreg + (at.per_head ? wave : 0)) * J * SF_ATTN_KSTRIDE;
  // Matrix-core form (the first r04 version ran scores and P.V on the vector units out of broadcast LDS reads: 3 us of LDS time
  // per workgroup).  fp32 operands are split into operand-type hi + lo parts and the lo x lo product is dropped (relative 2^-16):
  // S = Q K^T as 2 key blocks x 2 k-steps x 3 MFMAs, O = P V as 4 dim blocks x 3 MFMAs.  Fragment conventions (sf_dev.h):
  // A[m = lane & 15][k = 8 (lane >> 4) + j], B[k = 8 (lane >> 4) + j][n = lane & 15], D[m = 4 (lane >> 4) + r][n = lane & 15].
  auto split = [](const f32x4& a, const f32x4& c, bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      hi[e] = (sf_opnd)a[e]; lo[e] = (sf_opnd)(a[e] - (float)hi[e]);
      hi[4 + e] = (sf_opnd)c[e]; lo[4 + e] = (sf_opnd)(c[e] - (float)hi[4 + e]);
    }
  };
  const int g = lane >> 4, n16 = lane & 15;
  bf16x8 qh[2], ql[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) split(q[2 * ks], q[2 * ks + 1], qh[ks], ql[ks]);
  f32x4 sacc[2];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    const int j = 16 * cb + n16;
    const float* kr = kb + (j < J ? j : J - 1) * SF_ATTN_KSTRIDE + 8 * g;
    sacc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 kh, kl;
      split(*reinterpret_cast<const f32x4*>(kr + 32 * ks), *reinterpret_cast<const f32x4*>(kr + 32 * ks + 4), kh, kl);
      sacc[cb] = sf_mfma16(qh[ks], kh, sacc[cb]);
      sacc[cb] = sf_mfma16(qh[ks], kl, sacc[cb]);
      sacc[cb] = sf_mfma16(ql[ks], kh, sacc[cb]);
    }
  }
  // softmax of row i = 4 g + r over the keys: this lane holds columns n16 and 16 + n16; the other columns sit in the 16 lanes of its group
  float* prow = sp + wave * 16 * SF_ATTN_PSTRIDE;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float s0 = n16 < J ? sacc[0][r] * at.scale : -INFINITY;
    const float s1 = 16 + n16 < J ? sacc[1][r] * at.scale : -INFINITY;
    float mx = fmaxf(s0, s1);
    mx = fmaxf(mx, sf_shfl_xor(mx, 1)); mx = fmaxf(mx, sf_shfl_xor(mx, 2)); mx = fmaxf(mx, sf_shfl_xor(mx, 4)); mx = fmaxf(mx, sf_shfl_xor(mx, 8));
    const float e0 = n16 < J ? expf(s0 - mx) : 0.0f, e1 = 16 + n16 < J ? expf(s1 - mx) : 0.0f;
    float den = e0 + e1;
    den += sf_shfl_xor(den, 1); den += sf_shfl_xor(den, 2); den += sf_shfl_xor(den, 4); den += sf_shfl_xor(den, 8);
    const float inv = 1.0f / den;
    prow[(4 * g + r) * SF_ATTN_PSTRIDE + n16] = e0 * inv;               // zeros beyond the last key
    prow[(4 * g + r) * SF_ATTN_PSTRIDE + 16 + n16] = e1 * inv;
  }
  sf_wave_sync();
  bf16x8 ph, pl;
  {
    const float* pr = prow + n16 * SF_ATTN_PSTRIDE + 8 * g;             // A fragment: row n16, keys 8 g .. 8 g + 7
    split(*reinterpret_cast<const f32x4*>(pr), *reinterpret_cast<const f32x4*>(pr + 4), ph, pl);
  }
#pragma unroll
  for (int db = 0; db < 4; ++db) {
    f32x4 va, vc;                                                       // B fragment: keys 8 g + t (clamped: their P is 0), dim 16 db + n16
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int j0 = 8 * g + t, j1 = 8 * g + 4 + t;
      va[t] = vb[(j0 < J ? j0 : J - 1) * SF_ATTN_KSTRIDE + 16 * db + n16];
      vc[t] = vb[(j1 < J ? j1 : J - 1) * SF_ATTN_KSTRIDE + 16 * db + n16];
    }
    bf16x8 vh, vl;
    split(va, vc, vh, vl);
    f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
    o = sf_mfma16(ph, vh, o);
    o = sf_mfma16(ph, vl, o);
    o = sf_mfma16(pl, vh, o);
#pragma unroll
    for (int r = 0; r < 4; ++r)
      *reinterpret_cast<sf_opnd*>(lds + (long)(4 * g + r) * a.pix_stride + (wave * 64 + 16 * db + n16) * 2) = (sf_opnd)o[r];
  }

#undef LIN4_ATTN_WEIGHTS
  sf_sync();
  // ---- this wave's two k-steps on the frame [token][channel]
  f32x4 acc[WN];
#pragma unroll
  for (int ni = 0; ni < WN; ++ni) acc[ni] = f32x4{0.f, 0.f, 0.f, 0.f};
  const char* abase = lds + (long)(lane & 15) * a.pix_stride + (lane >> 4) * 16;
#pragma unroll
  for (int i = 0; i < KW; ++i) {
    const bf16x8 fa = *reinterpret_cast<const bf16x8*>(abase + (wave * KW + i) * 64);
#pragma unroll
    for (int ni = 0; ni < WN; ++ni) acc[ni] = sf_mfma16(fa, fb[i][ni], acc[ni]);
  }
  // ---- the 8 K-slices meet in LDS; wave f finalises fragment f: bias, residual, output, statistics slots
  float* red = reinterpret_cast<float*>(lds + a.red_off);         // [wave][frag][r][lane]
#pragma unroll
  for (int ni = 0; ni < WN; ++ni)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[((wave * F + ni) * 4 + r) * 64 + lane] = acc[ni][r];
  sf_sync();
  if (fin) {
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float sacc = 0.0f;
#pragma unroll
      for (int w = 0; w < 8; ++w) sacc += red[((w * F + wave) * 4 + r) * 64 + lane];
      if (n < a.Cout) {
        float y = sacc + bv + rv[r];
        if (a.out_gelu) y = sf_gelu(y);
        a.out[(mrow + r) * a.ldc + a.co_off + n] = y;
        s1 += y;
        s2 = fmaf(y, y, s2);
      }
    }
    if (a.slots_out) {
      s1 = sf_wave_sum(s1);
      s2 = sf_wave_sum(s2);
      if (lane == 0) {
        float* slo = a.slots_out + ((m0 >> 4) * (long)(a.ldc >> 4) + (a.co_off >> 4) + my_nf) * 2;
        slo[0] = s1;
        slo[1] = s2;
      }
    }
  }
}

template <int WN>
SF_KERNEL(512) void k_lin4_attn(FConvArgs a) {
  sf_touch_kernarg<(int)sizeof(FConvArgs)>();
  lin4_attn_body<WN>(a, (int)blockIdx.x);
}

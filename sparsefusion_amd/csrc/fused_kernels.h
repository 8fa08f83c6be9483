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

// n / d for small n (< 2^20) by one multiply: magic = floor(2^32 / d) + 1 (exact for the index ranges used here)
struct FDiv { uint32_t d, magic; };
SF_DEV uint32_t fdiv(uint32_t n, FDiv f) { return f.d == 1 ? n : (uint32_t)(((uint64_t)n * f.magic) >> 32); }

// Settled by measurement (the A/B switches these were live in r02-r05; their builds are in the history at c0adaa6 and the numbers in DESIGN.md):
// * staging batches of the fused prologue: 2 float4 loads per thread, two batches live (r02: 8 / 4 / 2 -> 1.79 / 1.65 / 1.62 ms per eval: the
//   prologue is instruction-issue bound, the shorter unrolled body wins);
// * weights are streamed with non-temporal loads everywhere (every byte is used once per eval; keeping it out of the L2's retained set
//   leaves the activations / slots there: r02 1.62 -> 1.54 ms per eval);
// * the weight ring of the slot / plain prologue is requested after the first staging batch.
constexpr int SF_STAGE_U = 2;
enum { FNORM_NONE = 0, FNORM_GN_SELF = 1, FNORM_GN_SLOTS = 2, FNORM_LN = 3, FNORM_ATTN = 4 };

// FNORM_ATTN: the A operand of an attention output projection IS the attention core's result, computed in the prologue
// (r04; was its own k_attn16 launch): 16 query tokens x 8 heads x 64 dims against up to 3 key / value segments (null k/v, time
// tokens, the 16 tokens' shared k/v head; external/imagen_pytorch.py:480-566, :731-805).  Key row r of segment s, head h, batch b:
// k[b * batch_stride + r * row_stride + h * head_stride + d], value = the same address + v_off.
#define SF_ATTN_MAX_KEYS 24
#define SF_ATTN_PSTRIDE 36          /* floats per probability row: 32 key columns (two MFMA blocks) + 4, 16-byte aligned */
#define SF_ATTN_KSTRIDE 68          /* floats per staged key / value row: 16-byte aligned, 4 consecutive rows on disjoint banks */
#define SF_ATTN_LDS_BYTES (8 * 16 * SF_ATTN_PSTRIDE * 4 + 2 * 8 * 4 * SF_ATTN_KSTRIDE * 4)   /* >= 2 * 24 shared rows as well */
struct FAttnSeg { const float* k; int v_off, rows, row_stride, batch_stride, head_stride; };
struct FAttn {
  const float* q;           // [B * 16][ldq], head h at columns [64 h, 64 h + 64)
  FAttnSeg seg[3];
  int ldq, J, per_head;     // J = total keys; per_head: some segment has its own k / v per head (then J <= 4)
  float scale;
};

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
  int ss_stride, norm, silu, pre_gelu, accum, G, out_gelu;   // out_gelu: GELU(erf) on the final output (ChanFeedForward, :953-961)
  float eps;
  int B, H, W, C, Cout, ldc, co_off, k;
  int TR, S, cps, cchunks, KS, n_frags, n_tiles, mt_per_img, npad, M;
  int pix_stride, xcd_map, logW;
  int det_w;                             // GN_SELF: lanes per deterministic reduction segment (0 = LDS-atomic fallback)
  FDiv d_ncf;                            // 16-channel fragments per group
  FDiv d_cs4, d_cg, d_cps, d_tc;         // Cs/4, channels per group, chunks per slice, min(Cs/4, threads)
  int red_off, tab_off, misc_off;    // LDS byte offsets
  double inv_n;                      // GroupNorm: 1 / (pixels per image * channels per group)
  int buf_bytes;                     // k_conv_fused_pipe: bytes of one of the two frame buffers (0 otherwise)
  const float* wk;                   // GlobalContext to_k weight [Cout] or null: the epilogue also emits partial context logits
  float* logit_part;                 //   logit_part[(s * n_frags + n_frag) * M + m] = sum over the fragment's 16 channels of value * wk
  long long* dbg;                    // optional [grid][8] phase timestamps (tools/fconv_phases.py), null in production
  // k_conv_fused_pipe<.., POOL = true> (r03): the GlobalContext softmax pooling of this conv's OUTPUT in its own epilogue.
  const sf_opnd* weff;                // [KS * 32] = (tap, channel) in k-step order: w_eff = sum_n wk[n] * W[n][channel][tap] -- the
                                     // context logit of a pixel is a 1-output-channel conv of the SAME staged input (bias terms cancel)
  float* pool_part;                  // [M / 16][Cout] un-normalised pooled fragments sum_p exp(l_p - max_frag) * out[p, n];
                                     // directly behind it [M / 16][2] = (max_frag, sum_p exp(l_p - max_frag)): one chunk = 16 pixels
  int weff_off;                      // LDS byte offset of the w_eff table
  // k_conv_fused_pipe<.., RC = true> (r04): the block's res_conv (1x1 conv of the RAW concat, imagen_pytorch.py:700-729) rides in the
  // SAME workgroups as conv1 -- one more k-step per matrix wave and chunk on a raw operand copy of the tile's own pixels
  const bf16x8* rc_w;                // packed 1x1 weights [Cout / 16][C / 32][lane][8], or null
  const float* rc_bias;
  float* rc_out;                     // [M][Cout]
  int rc_off, rc_buf_bytes;          // LDS: two raw-operand buffers of (16 * WM + 1) pixels x 128 channels
  FAttn attn;                        // FNORM_ATTN only
  int attn_off;                      //   LDS byte offset of its scratch (SF_ATTN_LDS_BYTES)
};

template <int MODE>
SF_DEV f32x4 fsrc_load4(const FSrc& s, int M, int HW, long m, int c) {
  f32x4 v;
  if (MODE == 1) {
    // EVERY load is issued before the first add, unconditionally (clamped slab index / any valid address, weight 0 or 1):
    // `v += load` in a loop or under `if (ptr)` makes the compiler wait for each load in turn (s_waitcnt vmcnt(0) after
    // every one of them: six serialised L2 round trips per element, ~2 us in front of every 4x4 conv).
    // Summation order = bias, slab 0, 1, ..., residual, as k_splitk_reduce.
    const float* ap = s.a + m * s.npad + c;
    const long gstride = (long)M * s.npad;
    const int gl = s.groups - 1;
    f32x4 t[8];
#pragma unroll
    for (int g = 0; g < 4; ++g) t[g] = *reinterpret_cast<const f32x4*>(ap + (g < gl ? g : gl) * gstride);
    const f32x4 bq = *reinterpret_cast<const f32x4*>(s.b ? s.b + c : ap);
    const f32x4 rq = *reinterpret_cast<const f32x4*>(s.r ? s.r + m * s.C + c : ap);
    if (s.groups > 4) {                                  // uniform: 5..8 slices
#pragma unroll
      for (int g = 4; g < 8; ++g) t[g] = *reinterpret_cast<const f32x4*>(ap + (g < gl ? g : gl) * gstride);
    }
    v = bq * (s.b ? 1.0f : 0.0f);
#pragma unroll
    for (int g = 0; g < 4; ++g) v += t[g] * (g <= gl ? 1.0f : 0.0f);
    if (s.groups > 4) {
#pragma unroll
      for (int g = 4; g < 8; ++g) v += t[g] * (g <= gl ? 1.0f : 0.0f);
    }
    v += rq * (s.r ? 1.0f : 0.0f);
  } else if (MODE == 2) {
    const f32x4 hv = *reinterpret_cast<const f32x4*>(s.a + m * s.C + c);
    const f32x4 gv = *reinterpret_cast<const f32x4*>(s.b + (m / HW) * s.C + c);
    const f32x4 rv = *reinterpret_cast<const f32x4*>(s.r + m * s.C + c);
    v = hv * gv + rv;
  } else {
    v = *reinterpret_cast<const f32x4*>(s.p + m * s.C + c);
  }
  return v;
}

// raw (unscaled) value of the (virtual) concat at pixel row m, concat channel c (c % 4 == 0); LAZY = mode of source 1
template <int LAZY>
SF_DEV f32x4 fconv_value(const FConvArgs& a, long m, int c) {
  const int HW = a.H * a.W;
  if (LAZY == 0) {       // plain sources: SELECT the address, one unconditional load (a load under `if (c < C1)` is fenced by vmcnt(0))
    const bool first = c < a.s1.C;
    const float* p1 = a.s1.p + m * a.s1.C + (first ? c : 0);
    const float* p2 = a.s2.C ? a.s2.p + m * a.s2.C + (first ? 0 : c - a.s1.C) : p1;
    return *reinterpret_cast<const f32x4*>(first ? p1 : p2);
  }
  if (c < a.s1.C) return fsrc_load4<LAZY>(a.s1, a.M, HW, m, c);
  return *reinterpret_cast<const f32x4*>(a.s2.p + m * a.s2.C + (c - a.s1.C));
}

template <int N>
struct FConst { static constexpr int value = N; };

// Branch-free two-phase evaluation of one float4 of the (virtual) concat for the register-resident 4x4 prologue: `issue`
// puts every load the element needs into t[0..K) without any control flow (addresses and weights are SELECTED per lane: a
// lane in the second source reads it K times with weight (1, 0, ..)), `combine` forms the value.  Keeping the loads of all
// elements of a thread in one straight-line block lets them travel together (one round trip); inside `if (i < cnt)` /
// `if (c < C1)` blocks each element's loads were fenced by s_waitcnt vmcnt(0).  K = loads per element of mode LAZY.
template <int LAZY>
struct FGather {
  static constexpr int K = LAZY == 1 ? 6 : (LAZY == 2 ? 3 : 1);       // mode 1: slabs 0..3, bias, residual (groups <= 4)
  f32x4 t[K];
  float w[K];
  SF_DEV void issue(const FConvArgs& a, long m, int c) {
    const bool first = c < a.s1.C;
    const float* p2 = a.s2.C ? a.s2.p + m * a.s2.C + (first ? 0 : c - a.s1.C) : nullptr;
    if (LAZY == 0) {
      const float* p1 = a.s1.p + m * a.s1.C + (first ? c : 0);
      t[0] = *reinterpret_cast<const f32x4*>(first ? p1 : p2);
      w[0] = 1.0f;
    } else if (LAZY == 1) {
      const int c1 = first ? c : 0;
      const float* ap = a.s1.a + m * a.s1.npad + c1;
      const long gstride = (long)a.M * a.s1.npad;
      const int gl = a.s1.groups - 1;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float* q = ap + (g < gl ? g : gl) * gstride;
        t[g] = *reinterpret_cast<const f32x4*>(first ? q : p2);
        w[g] = first ? (g <= gl ? 1.0f : 0.0f) : (g == 0 ? 1.0f : 0.0f);
      }
      const float* bp = a.s1.b ? a.s1.b + c1 : ap;
      const float* rp = a.s1.r ? a.s1.r + m * a.s1.C + c1 : ap;
      t[4] = *reinterpret_cast<const f32x4*>(first ? bp : p2);
      t[5] = *reinterpret_cast<const f32x4*>(first ? rp : p2);
      w[4] = (first && a.s1.b) ? 1.0f : 0.0f;
      w[5] = (first && a.s1.r) ? 1.0f : 0.0f;
    } else {
      const int c1 = first ? c : 0;
      const float* hp = a.s1.a + m * a.s1.C + c1;
      const float* gp = a.s1.b + (m / (a.H * a.W)) * a.s1.C + c1;
      const float* rp = a.s1.r + m * a.s1.C + c1;
      t[0] = *reinterpret_cast<const f32x4*>(first ? hp : p2);
      t[1] = *reinterpret_cast<const f32x4*>(first ? gp : p2);
      t[2] = *reinterpret_cast<const f32x4*>(first ? rp : p2);
      w[0] = first ? 1.0f : 0.0f;
    }
  }
  SF_DEV f32x4 combine() const {
    if (LAZY == 0) return t[0];
    if (LAZY == 1) {                       // order of k_splitk_reduce: bias, slab 0, 1, .., residual (second source: t[0] alone)
      f32x4 v = t[4] * w[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) v += t[g] * w[g];
      v += t[5] * w[5];
      return v;
    }
    return w[0] != 0.0f ? t[0] * t[1] + t[2] : t[0];
  }
};

// NORM / LAZY (normalisation kind, lazy mode of source 1) are compile-time: one straight-line prologue per variant
// (the all-in-one version was 49 k instructions and spilled 190 SGPRs).  NW = waves per workgroup: the prologue is VALU
// and latency bound (SiLU per element, one workgroup per CU because of the LDS frame), so it wants 2 waves per SIMD;
// the same 8 waves then split K eight ways and keep 8 x D KiB of weight fragments in flight per CU.
// Tile of workgroup t (within its K-slice).  Workgroups t, t + 8, ... run on one XCD (observed round-robin placement; only
// speed depends on it) and every XCD's L2 fetches what its workgroups read through the fabric: with R row groups x Q = 8 / R
// channel groups, XCD x = (rg = x % R, qg = x / R) owns the contiguous m-tiles [rg, rg + 1) * MT / R (neighbouring rows share
// their halo in L2) and the n-tiles == qg (mod Q): it pulls 1 / R of the activations and 1 / Q of the weights.  R = 1 is the
// r02 map (every XCD reads the whole activation map, each weight byte crosses the fabric once); the host picks R by bytes.
SF_DEV void fconv_tile_of(const FConvArgs& a, int t, int MT, int& mt, int& nt) {
  if (a.xcd_map) {
    const int R = a.xcd_map, Q = 8 / R;
    const int x = t & 7, j = t >> 3;
    const int rg = x % R, qg = x / R;
    const int MTg = MT / R;
    mt = rg * MTg + j % MTg;
    nt = qg + Q * (j / MTg);
  } else {
    nt = t % a.n_tiles;
    mt = t / a.n_tiles;
  }
}

template <int WM, int WN, int D, int NORM, int LAZY, int NW>
SF_DEV void conv_fused_body(const FConvArgs& a, const int bid) {
  constexpr int NT = NW * 64;
  SF_DYN_LDS(lds);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef SF_FCONV_TIMING      // tools/fconv_phases.py builds its own instrumented copy; a maybe-executed store would make every
#define FC_STAMP(k) do { if (a.dbg && tid == 0) a.dbg[(long)bid * 8 + (k)] = sf_clock(); } while (0)
#else                       // later s_waitcnt conservative, so the product has no trace of it
#define FC_STAMP(k) do { } while (0)
#endif
  FC_STAMP(0);
  // ---- which tile
  const int MT = a.B * a.mt_per_img;
  const int tiles = MT * a.n_tiles;
  const int s = bid / tiles;
  const int t = bid - s * tiles;
  int mt, nt;
  fconv_tile_of(a, t, MT, mt, nt);
  const int b = mt / a.mt_per_img;
  const int row0 = (mt - b * a.mt_per_img) * a.TR;
  const int h = a.k >> 1;
  const int FW = a.W + 2 * h, FR = a.TR + 2 * h;
  const int Cs = a.cps * 32, c0 = s * Cs, Cs4 = Cs >> 2;
  const int HW = a.H * a.W;
  const long mb = (long)b * HW;           // first pixel row of this image
  const float sc1 = a.s1.scale, sc2 = a.s2.scale;     // as values: a select between two kernarg FIELDS becomes a scratch array

  // ---- weight stream: this wave's k-steps [k0, k1) of the slice-local list (tap-major, then 32-channel chunk)
  const int KSl = a.k * a.k * a.cps;
  const int spw = (KSl + NW - 1) / NW;
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
    const int tap = (int)fdiv((uint32_t)j, a.d_cps), ccl = j - tap * a.cps;
    return __builtin_nontemporal_load(&wbase[ni][(long)(tap * a.cchunks + ccl) * 64]);
  };
  bf16x8 fb[D][WN];
  // fill the ring: the HBM / L2 weight stream runs under the rest of the prologue.  The loads are UNCONDITIONAL (indices
  // clamped into the wave's range): loads under a branch keep the compiler from counting what is in flight, and every
  // later s_waitcnt then drains the whole queue.
  const int klast = k1 > k0 ? k1 - 1 : (k0 < KSl ? k0 : KSl - 1);
  auto prefetch_weights = [&]() {
#pragma unroll
    for (int u = 0; u < D; ++u) {
      const int j = k0 + u < k1 ? k0 + u : klast;
#pragma unroll
      for (int ni = 0; ni < WN; ++ni) fb[u][ni] = wload(j, ni);
    }
  };

#define FC_PREFETCH() prefetch_weights()
  float* tabA = reinterpret_cast<float*>(lds + a.tab_off);
  float* tabB = tabA + Cs;
  float* misc = reinterpret_cast<float*>(lds + a.misc_off);      // [0..15] group sums, [16..] group / row (mean, rstd)

  // ---- GN_SELF: the tile's elements go out FIRST (statistics wait on them; everything below runs under their round trip)
  constexpr int NV = (NORM == FNORM_GN_SELF) ? 4096 / NT : 1;
  const int cnt = HW * Cs4;
  f32x4 v[NV];
  if (NORM == FNORM_GN_SELF) {
    const int nlive = (cnt + NT - 1) / NT;                           // elements per thread (uniform), 1..NV
    if (LAZY == 1 && a.s1.groups > 4) {                              // 5..8 slabs: the generic (per-element) path
#pragma unroll
      for (int u = 0; u < NV; ++u) {
        const int i = tid + u * NT;
        if (i < cnt) {
          const int p = (int)fdiv((uint32_t)i, a.d_cs4), c4 = i - p * Cs4;
          v[u] = fconv_value<LAZY>(a, mb + p, c0 + c4 * 4);
        }
      }
    } else {
      // batches of 4 elements: all their loads in one straight-line block, then the sums (dead elements read element cnt - 1)
#pragma unroll
      for (int u0 = 0; u0 < NV; u0 += 4) {
        if (u0 < nlive) {
          FGather<LAZY> gq[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            int i = tid + (u0 + u) * NT;
            if (i > cnt - 1) i = cnt - 1;
            const int p = (int)fdiv((uint32_t)i, a.d_cs4), c4 = i - p * Cs4;
            gq[u].issue(a, mb + p, c0 + c4 * 4);
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) v[u0 + u] = gq[u].combine();
        }
      }
    }
    FC_PREFETCH();          // right behind them: at the 4x4 level the launch is one HBM round trip of the whole weight slice
  }
  FC_STAMP(6);

  // ---- (a) zero the frame pixels outside the image (conv zero padding); 8 threads per pixel
  if (h) {
    const int npix = FR * FW;
    for (int q = tid >> 3; q < npix; q += NT / 8) {
      const int fr = q / FW, fx = q - fr * FW;
      const int r = row0 - h + fr, x = fx - h;
      if (r < 0 || r >= a.H || x < 0 || x >= a.W) {
        char* dst = lds + (long)q * a.pix_stride;
        for (int c8 = (tid & 7); c8 < (Cs >> 3); c8 += 8) *reinterpret_cast<bf16x8*>(dst + c8 * 16) = sf_zero8();
      }
    }
  }

  FC_STAMP(7);
  constexpr bool gn = NORM == FNORM_GN_SELF || NORM == FNORM_GN_SLOTS;
  const int Cg = gn ? a.C / a.G : 1;
  // per-channel affine of this (image, slice) from the group statistics in misc[16 + 2g], misc[17 + 2g]:
  //   y = v * A + B  ==  ((v - mean) * rstd * gamma + beta) * (scale + 1) + shift
  // gamma / beta / scale / shift are fetched into registers NOW (their global round trip runs under the statistics)
  constexpr int TABN = 4;                                          // Cs <= 2048 channels per slice at NT = 512
  float tg[TABN], tb[TABN], tsc[TABN], tsh[TABN];
  if (gn) {
    const float* ssrow = a.ss ? a.ss + (long)b * a.ss_stride : a.gamma;      // any valid address when there is no scale / shift
    const int shoff = a.ss ? a.C : 0;
#pragma unroll
    for (int k = 0; k < TABN; ++k) {
      const int cl = tid + k * NT, cc = c0 + (cl < Cs ? cl : Cs - 1);
      tg[k] = a.gamma[cc];
      tb[k] = a.beta[cc];
      tsc[k] = ssrow[cc];
      tsh[k] = ssrow[shoff + cc];
    }
  }
  auto build_table = [&]() {
#pragma unroll
    for (int k = 0; k < TABN; ++k) {
      const int cl = tid + k * NT;
      if (cl < Cs) {
        const int gi = (int)fdiv((uint32_t)cl, a.d_cg);
        const float mean = misc[16 + 2 * gi], rstd = misc[17 + 2 * gi];
        const float A = rstd * tg[k], sc = a.ss ? tsc[k] + 1.0f : 1.0f, sh = a.ss ? tsh[k] : 0.0f;
        tabA[cl] = A * sc;
        tabB[cl] = (tb[k] - mean * A) * sc + sh;
      }
    }
  };
  // normalise / activate one float4 of channels [cl, cl+4) and store it as bf16 at frame pixel fp.  (A, Bv) = the
  // thread's slice of the affine table (GroupNorm) or (gain, bias) (LayerNorm): loop invariants of the callers.
  auto finish = [&](f32x4 v, int fp, int cl, int row, const f32x4& A, const f32x4& Bv) {
    if (gn) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = fmaf(v[j], A[j], Bv[j]);
    } else if (NORM == FNORM_LN) {
      const float mean = misc[16 + 2 * row], rstd = misc[17 + 2 * row];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float u = a.pre_gelu ? sf_gelu(v[j]) : v[j];
        v[j] = fmaf((u - mean) * rstd, A[j], Bv[j]);
      }
    }
    // a select, not a branch: the U elements of a staging batch must stay in one basic block to interleave
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float sv = sf_silu_fast(v[j]);
      v[j] = a.silu ? sv : v[j];
    }
    bf16x4 o;
    o[0] = (sf_opnd)v[0]; o[1] = (sf_opnd)v[1]; o[2] = (sf_opnd)v[2]; o[3] = (sf_opnd)v[3];
    *reinterpret_cast<bf16x4*>(lds + (long)fp * a.pix_stride + cl * 2) = o;
  };
  auto affine_of = [&](int cl, f32x4& A, f32x4& Bv) {
    A = f32x4{1.f, 1.f, 1.f, 1.f};
    Bv = f32x4{0.f, 0.f, 0.f, 0.f};
    if (gn) {
      A = *reinterpret_cast<const f32x4*>(tabA + cl);
      Bv = *reinterpret_cast<const f32x4*>(tabB + cl);
    } else if (NORM == FNORM_LN) {
      A = *reinterpret_cast<const f32x4*>(a.gamma + c0 + cl);
      if (a.beta) Bv = *reinterpret_cast<const f32x4*>(a.beta + c0 + cl);
    }
  };

  if (NORM == FNORM_GN_SELF) {
    // ---- (b1) the 4x4 level: the tile holds all HW pixels of image b and the slice holds whole groups.  Every element is
    // loaded ONCE into registers (issued before the weight ring, so one round trip covers the whole prologue), group sums
    // meet in LDS, then the registers are normalised straight into the frame.
    FC_STAMP(1);
    if (a.det_w) {
      // deterministic group sums: a thread's elements share one channel chunk (NT % (Cs/4) == 0), hence one group; det_w
      // adjacent lanes lie in one group -> segment sums by shuffles, one partial per segment, fixed-order final sum.
      // (The LDS-atomic fallback below is order dependent: last-ulp noise in the statistics, visible after bf16 rounding.)
      const int W = a.det_w;
      float sm = 0.0f, sq = 0.0f;
      const int c4t = tid - (int)fdiv((uint32_t)tid, a.d_cs4) * Cs4;
      const float sc = (c0 + c4t * 4 < a.s1.C) ? sc1 : sc2;
#pragma unroll
      for (int u = 0; u < NV; ++u) {
        if (tid + u * NT < cnt) {
          const f32x4 w = v[u] * sc;
          sm += (w[0] + w[1]) + (w[2] + w[3]);
          sq = fmaf(w[0], w[0], sq); sq = fmaf(w[1], w[1], sq); sq = fmaf(w[2], w[2], sq); sq = fmaf(w[3], w[3], sq);
        }
      }
      sm = sf_group_sum(sm, W);
      sq = sf_group_sum(sq, W);
      float* part = misc + 160;                               // [NT / W][2]
      if ((lane & (W - 1)) == 0) { part[2 * (tid / W)] = sm; part[2 * (tid / W) + 1] = sq; }
      sf_sync();
      if (NT / W <= 64) {
        // wave g sums the segments of group g: lanes = segments, a fixed shuffle tree (deterministic); the serial loop of the
        // first version (one thread per group walking all segments) was ~1 us on the critical path of every 4x4 conv
        if (wave < Cs / Cg) {
          double S = 0.0, Q = 0.0;
          if (lane < NT / W) {
            const int t0 = lane * W, c4s = t0 - (int)fdiv((uint32_t)t0, a.d_cs4) * Cs4;
            if ((int)fdiv((uint32_t)(c4s * 4), a.d_cg) == wave) { S = (double)part[2 * lane]; Q = (double)part[2 * lane + 1]; }
          }
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) { S += sf_shfl_xor(S, o); Q += sf_shfl_xor(Q, o); }
          if (lane == 0) {
            const double rn = a.inv_n;                     // 1 / (HW * Cg) from the host: a double division is ~100 dependent instructions
            const double mean = S * rn;
            double var = Q * rn - mean * mean;
            if (var < 0.0) var = 0.0;
            misc[16 + 2 * wave] = (float)mean;
            misc[17 + 2 * wave] = sf_rsqrt((float)var + a.eps);
          }
        }
      } else if (tid < Cs / Cg) {
        double S = 0.0, Q = 0.0;
        for (int sl = 0; sl < NT / W; ++sl) {
          const int t0 = sl * W, c4s = t0 - (int)fdiv((uint32_t)t0, a.d_cs4) * Cs4;
          if ((int)fdiv((uint32_t)(c4s * 4), a.d_cg) == tid) { S += (double)part[2 * sl]; Q += (double)part[2 * sl + 1]; }
        }
        const double rn = a.inv_n;                     // 1 / (HW * Cg) from the host: a double division is ~100 dependent instructions
        const double mean = S * rn;
        double var = Q * rn - mean * mean;
        if (var < 0.0) var = 0.0;
        misc[16 + 2 * tid] = (float)mean;
        misc[17 + 2 * tid] = sf_rsqrt((float)var + a.eps);
      }
    } else {
      // slice shapes the segment scheme does not cover (NT % (Cs/4) != 0, odd group widths): still ORDER-INDEPENDENT --
      // per group, every thread's partial goes through the fixed wave shuffle tree, the <= 8 per-wave partials meet in LDS and
      // one thread adds them in wave order (r02 used LDS float atomics here: last-ulp run-to-run noise in the statistics).
      const int ng = (int)fdiv((uint32_t)Cs, a.d_cg);          // groups in this slice, <= 16 (host check)
      float* part = misc + 160;                               // [waves][16][2]
      for (int g = 0; g < ng; ++g) {
        float sm = 0.0f, sq = 0.0f;
#pragma unroll
        for (int u = 0; u < NV; ++u) {
          const int i = tid + u * NT;
          const int p = (int)fdiv((uint32_t)i, a.d_cs4), c4 = i - p * Cs4;
          if (i < cnt && (int)fdiv((uint32_t)(c4 * 4), a.d_cg) == g) {
            const f32x4 w = v[u] * ((c0 + c4 * 4 < a.s1.C) ? sc1 : sc2);
            sm += (w[0] + w[1]) + (w[2] + w[3]);
            sq = fmaf(w[0], w[0], sq); sq = fmaf(w[1], w[1], sq); sq = fmaf(w[2], w[2], sq); sq = fmaf(w[3], w[3], sq);
          }
        }
        sm = sf_wave_sum(sm);
        sq = sf_wave_sum(sq);
        if (lane == 0) { part[(wave * 16 + g) * 2] = sm; part[(wave * 16 + g) * 2 + 1] = sq; }
      }
      sf_sync();
      if (tid < ng) {
        double S = 0.0, Q = 0.0;
        for (int w = 0; w < NT / 64; ++w) { S += (double)part[(w * 16 + tid) * 2]; Q += (double)part[(w * 16 + tid) * 2 + 1]; }
        const double rn = a.inv_n;                     // 1 / (HW * Cg) from the host: a double division is ~100 dependent instructions
        const double mean = S * rn;
        double var = Q * rn - mean * mean;
        if (var < 0.0) var = 0.0;
        misc[16 + 2 * tid] = (float)mean;
        misc[17 + 2 * tid] = sf_rsqrt((float)var + a.eps);
      }
    }
    sf_sync();
    build_table();
    sf_sync();
    FC_STAMP(2);
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int i = tid + u * NT;
      if (i < cnt) {
        const int p = (int)fdiv((uint32_t)i, a.d_cs4), c4 = i - p * Cs4;
        const int py = p >> a.logW, px = p - (py << a.logW);
        const int c = c0 + c4 * 4;
        if (nt == 0 && LAZY && c < a.s1.C) *reinterpret_cast<f32x4*>(a.s1.p + (mb + p) * a.s1.C + c) = v[u];
        f32x4 A, Bv;
        affine_of(c4 * 4, A, Bv);
        finish(v[u] * (c < a.s1.C ? sc1 : sc2), (py + h) * FW + px + h, c4 * 4, 0, A, Bv);
      }
    }
  } else if (NORM == FNORM_LN) {
    // ---- (b1') LayerNorm over all C channels of each of the tile's 16*WM rows (S == 1, k == 1): NT / rows threads per
    // row keep the whole row in registers -- one load round trip, two-pass statistics like nn.LayerNorm, then the
    // registers are normalised straight into the frame.
    constexpr int NVL = 16;                                 // C <= 2048 at 32 threads per row
    const int rows = 16 * WM, tpr = NT / rows;
    const int row = tid / tpr, part = tid - row * tpr;
    const long m = mb + (long)row0 * a.W + row;
    f32x4 v[NVL];
#pragma unroll
    for (int u = 0; u < NVL; ++u) {
      const int c4 = part + u * tpr;
      v[u] = fconv_value<LAZY>(a, m, (c4 < Cs4 ? c4 : Cs4 - 1) * 4);
    }
    FC_PREFETCH();
    FC_STAMP(1);
    float sm = 0.0f;
#pragma unroll
    for (int u = 0; u < NVL; ++u) {
      if (part + u * tpr < Cs4) {
        if (LAZY && nt == 0) *reinterpret_cast<f32x4*>(a.s1.p + m * a.s1.C + (part + u * tpr) * 4) = v[u];
        if (a.pre_gelu) { v[u][0] = sf_gelu(v[u][0]); v[u][1] = sf_gelu(v[u][1]); v[u][2] = sf_gelu(v[u][2]); v[u][3] = sf_gelu(v[u][3]); }
        sm += (v[u][0] + v[u][1]) + (v[u][2] + v[u][3]);
      }
    }
    const float mean = sf_group_sum(sm, tpr) / (float)Cs;
    float sq = 0.0f;
#pragma unroll
    for (int u = 0; u < NVL; ++u)
      if (part + u * tpr < Cs4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float d = v[u][j] - mean; sq = fmaf(d, d, sq); }
      }
    const float rstd = sf_rsqrt(sf_group_sum(sq, tpr) / (float)Cs + a.eps);
    FC_STAMP(2);
#pragma unroll
    for (int u = 0; u < NVL; ++u) {
      const int c4 = part + u * tpr;
      if (c4 < Cs4) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(a.gamma + c4 * 4);
        f32x4 y;
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] = (v[u][j] - mean) * rstd * g[j];
        if (a.beta) y += *reinterpret_cast<const f32x4*>(a.beta + c4 * 4);
        if (a.silu) { y[0] = sf_silu_fast(y[0]); y[1] = sf_silu_fast(y[1]); y[2] = sf_silu_fast(y[2]); y[3] = sf_silu_fast(y[3]); }
        bf16x4 o;
        o[0] = (sf_opnd)y[0]; o[1] = (sf_opnd)y[1]; o[2] = (sf_opnd)y[2]; o[3] = (sf_opnd)y[3];
        *reinterpret_cast<bf16x4*>(lds + (long)row * a.pix_stride + c4 * 8) = o;
      }
    }
  } else if (NORM == FNORM_ATTN) {
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
      FC_PREFETCH();          // the weight ring goes out BEHIND the q / k / v loads: loads return in order, and the first wait below is for k / v
      FC_STAMP(1);
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
    FC_STAMP(2);
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
  } else {
    // ---- (b2) + (c): GroupNorm statistics from the producer's slots (or no normalisation), then the in-image frame rows
    // fp32 -> normalise -> activate -> bf16 [frame pixel][channel].
    // Thread layout: TC = min(Cs/4, NT) threads span the slice's float4 channel chunks (coalesced rows), NT / TC pixel
    // lanes; a thread keeps ONE channel chunk, so its affine (A, B) lives in registers and the per-element index math
    // is a shift and a mask.  The staging is software-pipelined: two register batches of U pixels, one in flight while
    // the other is in the VALU, and the FIRST batch is issued before the statistics -- the cold round trip of the
    // activations (written by the previous kernel on other XCDs) overlaps the one of the slots.  Loads are never
    // under a branch (dead elements read a safe pixel and land in a spare LDS pixel behind the frame).
    constexpr int U = (LAZY == 1) ? 2 : SF_STAGE_U;
    const int TC = Cs4 < NT ? Cs4 : NT;
    const int ppp = NT / TC;
    const int tp = (int)fdiv((uint32_t)tid, a.d_tc), tcx = tid - tp * TC;
    const int npx = FR << a.logW;
    const int stp = ppp * U;
    // frame pixel pi = fr * W + x of an in-image row is source pixel M0 + pi: addresses are LINEAR in pi
    const int M0 = (int)mb + (row0 - h) * a.W;
    const int pi_safe = h << a.logW;                                 // first own row: always inside the image
    int cl = 0, c = 0, srcld = 0;
    bool cact = false, first = true;
    const float* srcp = nullptr;
    auto chan = [&](int cb) {
      const int c4 = cb + tcx;
      cact = tp < ppp && c4 < Cs4;
      cl = (c4 < Cs4 ? c4 : Cs4 - 1) * 4;
      c = c0 + cl;
      first = c < a.s1.C;
      srcp = first ? a.s1.p + c : a.s2.p + (c - a.s1.C);             // plain sources: one unconditional load per element
      srcld = first ? a.s1.C : a.s2.C;
    };
    auto issue = [&](int p0, f32x4 (&v)[U], int (&fpx)[U], int (&mx)[U]) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int pi = p0 + u * ppp;
        const int fr = pi >> a.logW;
        const int r = row0 - h + fr;
        const bool in = pi < npx && cact && r >= 0 && r < a.H;
        mx[u] = M0 + (in ? pi : pi_safe);
        // fpx < 0: nothing to store; bit 30: this workgroup owns the element (materialises a lazy source)
        fpx[u] = in ? ((pi + 2 * h * fr + h) | ((nt == 0 && fr >= h && fr < h + a.TR) ? (1 << 30) : 0)) : -1;
        if (LAZY == 0) v[u] = *reinterpret_cast<const f32x4*>(srcp + mx[u] * srcld);
        else v[u] = fconv_value<LAZY>(a, mx[u], c);
      }
    };
    // (A, Bv) with the source's scale folded in: y = v * A + Bv; SILU is a compile-time copy of a.silu
    auto consume = [&](auto silu_c, f32x4 (&v)[U], int (&fpx)[U], int (&mx)[U], const f32x4& A, const f32x4& Bv) {
      constexpr bool SILU = decltype(silu_c)::value != 0;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (LAZY && fpx[u] >= 0 && (fpx[u] >> 30) && first && a.s1.p) *reinterpret_cast<f32x4*>(a.s1.p + (long)mx[u] * a.s1.C + c) = v[u];
        const int fp = fpx[u] < 0 ? FR * FW : (fpx[u] & 0x3fffffff);
        f32x4 y = gn ? v[u] * A + Bv : v[u] * A;
        if (SILU) {
          const f32x4 t = y * -1.4426950408889634f;
          f32x4 e;
#pragma unroll
          for (int j = 0; j < 4; ++j) e[j] = sf_exp2(t[j]);
          e = e + 1.0f;
#pragma unroll
          for (int j = 0; j < 4; ++j) e[j] = sf_rcp(e[j]);
          y = y * e;
        }
        bf16x4 o;
        o[0] = (sf_opnd)y[0]; o[1] = (sf_opnd)y[1]; o[2] = (sf_opnd)y[2]; o[3] = (sf_opnd)y[3];
        *reinterpret_cast<bf16x4*>(lds + (long)fp * a.pix_stride + cl * 2) = o;
      }
    };
    f32x4 va[U], vb[U];
    int fpa[U], mxa[U], fpb[U], mxb[U];
    f32x2 sl[4];
    float slsc[4];
    const int ngs = gn ? Cs / Cg : 0;
    const int n_mf = HW >> 4, n_cf = (Cg >> 4) > 0 ? (Cg >> 4) : 1, scnt = n_mf * n_cf;
    const int cf1 = a.s1.C >> 4, cf2 = a.s2.C >> 4;
    auto slot_loads = [&](int gi, int i0) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        int i = i0 + u * 64;
        const bool live = i < scnt;
        if (!live) i = scnt - 1;
        const int mf = (int)fdiv((uint32_t)i, a.d_ncf), cfa = ((c0 + gi * Cg) >> 4) + (i - mf * n_cf);
        const long mfg = (long)b * n_mf + mf;
        const bool f1 = cfa < cf1;
        const float* base = f1 ? a.s1.slots : a.s2.slots;
        const long off = f1 ? (mfg * cf1 + cfa) : (mfg * cf2 + (cfa - cf1));
        sl[u] = *reinterpret_cast<const f32x2*>(base + off * 2);
        slsc[u] = live ? (f1 ? sc1 : sc2) : 0.0f;
      }
    };
    // issue order: slots of this wave's first group, first staging batch, weight ring
    if (NORM == FNORM_GN_SLOTS && wave < ngs) slot_loads(wave, lane);
    chan(0);
    issue(tp, va, fpa, mxa);
    FC_PREFETCH();
    FC_STAMP(1);
    if (NORM == FNORM_GN_SLOTS) {
      // one wave per group sums the (sum, sum of squares) slots of image b: 4 independent slot loads in flight per lane
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
          const double rn = a.inv_n;                     // 1 / (HW * Cg) from the host: a double division is ~100 dependent instructions
          const double mean = (double)sm * rn;
          double var = (double)sq * rn - mean * mean;
          if (var < 0.0) var = 0.0;
          misc[16 + 2 * gi] = (float)mean;
          misc[17 + 2 * gi] = sf_rsqrt((float)var + a.eps);
        }
      }
      sf_sync();
      build_table();
      sf_sync();
    }
    FC_STAMP(2);
    auto run = [&](auto silu_c) {
      for (int cb = 0; cb < Cs4; cb += TC) {
        if (cb) { chan(cb); issue(tp, va, fpa, mxa); }
        f32x4 A, Bv;
        affine_of(cl, A, Bv);
        A = A * (first ? sc1 : sc2);
        for (int base = 0; base < npx; base += 2 * stp) {
          issue(base + stp + tp, vb, fpb, mxb);
          consume(silu_c, va, fpa, mxa, A, Bv);
          issue(base + 2 * stp + tp, va, fpa, mxa);
          consume(silu_c, vb, fpb, mxb, A, Bv);
        }
      }
    };
    if (a.silu) run(FConst<1>());
    else run(FConst<0>());
  }
  sf_sync();
  FC_STAMP(3);

  // ---- epilogue operands (bias, residual, previous contents, context-logit weight) are fetched NOW, ahead of the main loop:
  // issued at the epilogue they were a cold round trip (~1 us) on the critical path of every final-mode launch
  constexpr int F = WM * WN;
  const long m0 = mb + (long)row0 * a.W;
  const int my_mi = wave / WN, my_ni = wave - my_mi * WN;         // fragment of this wave (waves >= F only contribute partials)
  const int my_nf = nt * WN + my_ni;
  const bool fin = wave < F && my_nf < a.n_frags;
  const int n = my_nf * 16 + (lane & 15);
  const long mrow = m0 + my_mi * 16 + (lane >> 4) * 4;
  float bv = 0.0f, rv[4] = {0.f, 0.f, 0.f, 0.f}, wkv = 0.0f;
  if (fin && n < a.Cout) {
    if (a.logit_part) wkv = a.wk[n];
    if (a.S == 1) {
      if (a.bias) bv = a.bias[n];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long o = (mrow + r) * a.ldc + a.co_off + n;
        if (a.resid) rv[r] = a.resid[o];
        if (a.accum) rv[r] += a.out[o];
      }
    }
  }

  // ---- main loop: per k-step WM A fragments (16-byte LDS reads of shifted pixels, fetched one step ahead), WN weight
  // fragments from the ring, WM x WN MFMAs; the ring slot is refilled D steps ahead
  f32x4 acc[WM][WN];
#pragma unroll
  for (int mi = 0; mi < WM; ++mi)
#pragma unroll
    for (int ni = 0; ni < WN; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
  int abase[WM];
#pragma unroll
  for (int mi = 0; mi < WM; ++mi) {
    const int p = mi * 16 + (lane & 15);
    const int ty = p >> a.logW, tx = p - (ty << a.logW);
    abase[mi] = (ty * FW + tx) * a.pix_stride + (lane >> 4) * 16;
  }
  auto aoff = [&](int j) -> int {
    const int tap = (int)fdiv((uint32_t)j, a.d_cps), ccl = j - tap * a.cps;
    const int ky = (a.k == 3) ? (tap >= 6 ? 2 : (tap >= 3 ? 1 : 0)) : 0, kx = tap - ky * a.k;
    return (ky * FW + kx) * a.pix_stride + ccl * 64;
  };
  bf16x8 fa[WM];
  if (k0 < k1) {
    const int toff = aoff(k0);
#pragma unroll
    for (int mi = 0; mi < WM; ++mi) fa[mi] = *reinterpret_cast<const bf16x8*>(lds + abase[mi] + toff);
  }
  for (int j0 = k0; j0 < k1; j0 += D) {
#pragma unroll
    for (int u = 0; u < D; ++u) {
      const int j = j0 + u;
      if (j < k1) {
        bf16x8 fn[WM];
        const int toff = aoff(j + 1 < k1 ? j + 1 : j);
#pragma unroll
        for (int mi = 0; mi < WM; ++mi) fn[mi] = *reinterpret_cast<const bf16x8*>(lds + abase[mi] + toff);
#pragma unroll
        for (int mi = 0; mi < WM; ++mi)
#pragma unroll
          for (int ni = 0; ni < WN; ++ni) acc[mi][ni] = sf_mfma16(fa[mi], fb[u][ni], acc[mi][ni]);
        if (j + D < k1) {
#pragma unroll
          for (int ni = 0; ni < WN; ++ni) fb[u][ni] = wload(j + D, ni);     // refill the ring slot just consumed
        }
#pragma unroll
        for (int mi = 0; mi < WM; ++mi) fa[mi] = fn[mi];
      }
    }
  }

  FC_STAMP(4);
  // ---- epilogue: the NW K-slices of the workgroup meet in LDS; wave f finalises fragment f and fetches what it needs
  // for that (bias, residual, previous contents) BEFORE the reduction barrier
  float* red = reinterpret_cast<float*>(lds + a.red_off);         // [wave][frag][r][lane]
#pragma unroll
  for (int mi = 0; mi < WM; ++mi)
#pragma unroll
    for (int ni = 0; ni < WN; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[((wave * F + mi * WN + ni) * 4 + r) * 64 + lane] = acc[mi][ni][r];
  sf_sync();
  if (fin) {
    const int f = wave;
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int idx = (f * 4 + r) * 64 + lane;
      float sacc = 0.0f;
#pragma unroll
      for (int w = 0; w < NW; ++w) sacc += red[idx + w * F * 256];
      v[r] = sacc;
    }
    if (a.logit_part) {                      // bias terms are the same for every pixel: they cancel in the softmax
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float lp = v[r] * wkv;
        lp += sf_shfl_xor(lp, 1); lp += sf_shfl_xor(lp, 2); lp += sf_shfl_xor(lp, 4); lp += sf_shfl_xor(lp, 8);
        if ((lane & 15) == 0) a.logit_part[((long)s * a.n_frags + my_nf) * a.M + mrow + r] = lp;
      }
    }
    if (a.S > 1) {
#pragma unroll
      for (int r = 0; r < 4; ++r) a.ws[((long)s * a.M + mrow + r) * a.npad + n] = v[r];
    } else {
      float sm = 0.0f, sq = 0.0f;
      if (n < a.Cout) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float y = v[r] + bv + rv[r];
          if (a.out_gelu) y = sf_gelu(y);
          a.out[(mrow + r) * a.ldc + a.co_off + n] = y;
          sm += y;
          sq = fmaf(y, y, sq);
        }
      }
      if (a.slots_out) {
        sm = sf_wave_sum(sm);
        sq = sf_wave_sum(sq);
        if (lane == 0) {
          float* sl = a.slots_out + (((m0 >> 4) + my_mi) * (long)(a.ldc >> 4) + (a.co_off >> 4) + my_nf) * 2;
          sl[0] = sm;
          sl[1] = sq;
        }
      }
    }
  }
  FC_STAMP(5);
#undef FC_STAMP
#undef FC_PREFETCH
}

template <int WM, int WN, int D, int NORM, int LAZY, int NW>
SF_KERNEL(NW * 64) void k_conv_fused(FConvArgs a) {
  sf_touch_kernarg<(int)sizeof(FConvArgs)>();
  conv_fused_body<WM, WN, D, NORM, LAZY, NW>(a, (int)blockIdx.x);
}

// Two INDEPENDENT convs of one ResnetBlock in one launch: the block's first conv (GroupNorm + SiLU + 3x3, `a`) and its
// res_conv (1x1 on the raw concat, `b`; imagen_pytorch.py:700-729 reads the same input for both).  Workgroups
// [0, grid_b) run `b` -- the short ones first, so the CUs they free pick up the tail of `a` -- the rest run `a`.  `b`
// evaluates a lazy first source like `a` does but never materialises it (b.s1.p == null): `a` owns that.
struct FConvPairArgs {
  FConvArgs a, b;
  int grid_b;
};

template <int WM, int WN, int D, int NORM, int LAZY, int NW>
SF_KERNEL(NW * 64) void k_conv_fused_pair(FConvPairArgs p) {
  sf_touch_kernarg<(int)sizeof(FConvPairArgs)>();
  if ((int)blockIdx.x < p.grid_b) conv_fused_body<WM, WN, D, FNORM_NONE, LAZY, NW>(p.b, (int)blockIdx.x);
  else conv_fused_body<WM, WN, D, NORM, LAZY, NW>(p.a, (int)blockIdx.x - p.grid_b);
}

// (sum, sum of squares) slots of an fp32 NHWC tensor [M, C], one wave per 16 pixels x 16 channels; with `gate` the
// tensor is first formed as x = h * gate[batch] + res and written to `out` (GlobalContext gating + residual,
// imagen_pytorch.py:727-729, :936-941) so that the next GroupNorm-fused conv finds both the values and their sums.
// With `ws` the tensor is first reduced from split-K slabs: x = bias + sum_g ws[g][m][c (ld npad)], written to `out`
// (a deferred k_conv_igemm reduction and the slot pass in ONE launch instead of k_splitk_reduce + k_slots).
SF_KERNEL(256) void k_slots(const float* __restrict__ x, const float* __restrict__ gate, const float* __restrict__ res,
                            float* __restrict__ out, float* __restrict__ slots, int M, int C, int HW,
                            const float* __restrict__ ws, const float* __restrict__ bias, int groups, int npad) {
  const int lane = threadIdx.x & 63, gw = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int CF = C >> 4;
  if (gw >= (M >> 4) * CF) return;
  const int mf = gw / CF, cf = gw - mf * CF;
  const long m = (long)mf * 16 + (lane >> 2);
  const int c = cf * 16 + (lane & 3) * 4;
  f32x4 v;
  if (ws) {
    const float* ap = ws + m * npad + c;                 // all slab loads before the first add (see fsrc_load4<1>), 1..8 slabs
    const long gstride = (long)M * npad;
    const int gl = groups - 1;
    f32x4 t[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) t[g] = *reinterpret_cast<const f32x4*>(ap + (g < gl ? g : gl) * gstride);
    v = *reinterpret_cast<const f32x4*>(bias ? bias + c : ap) * (bias ? 1.0f : 0.0f);
#pragma unroll
    for (int g = 0; g < 8; ++g) v += t[g] * (g <= gl ? 1.0f : 0.0f);
    *reinterpret_cast<f32x4*>(out + m * C + c) = v;
  } else {
    v = *reinterpret_cast<const f32x4*>(x + m * C + c);
  }
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

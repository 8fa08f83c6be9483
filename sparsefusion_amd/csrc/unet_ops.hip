// UNet operator set for gfx950 + the plan executor (sf_plan_run).
//
// Covers the op inventory of the view-conditioned latent UNet (external/imagen_pytorch.py):
//   conv / linear          Conv2d k in {1,3,4,7,15}, nn.Linear on token maps        :641-662, :608-610, :1017-1042
//   GN_ACT                 nn.GroupNorm(8) [+ x*(scale+1)+shift] + SiLU  (Block)    :641-662
//   LN                     LayerNorm / ChanLayerNorm (gain only) / nn.LayerNorm     :301-329, :1214
//   GEMV                   time MLPs, GlobalContext net, context k/v projections    :682-686, :916-941, :1175-1190
//   ATTN                   16-query attention core (self multi-query / cross)        :480-566, :731-805
//   GCA_POOL               GlobalContext softmax pooling                             :930-941
//   ELTWISE                gating + residual, NCHW<->NHWC packing                    :727-729
//   TIME_EMB               LearnedSinusoidalPosEmb                                   :624-639
//
// Design (MI355X-first, not a translation of the torch op sequence):
//   * activations are NHWC; the residual stream and every reduction stay fp32, MFMA operands are
//     bf16 with fp32 accumulation (v_mfma_f32_16x16x32_bf16);
//   * conv = implicit GEMM, one wave per (16*WM x 16*WN) output tile and K slice.  Weights are
//     pre-packed in MFMA-fragment order so a wave streams its weight slice as contiguous 1 KiB
//     loads straight into registers (each weight byte is read once: no LDS round trip);
//     activations come from L2 as 16-byte fragments.  The 4 waves of a workgroup take 4 K slices
//     of the same tile, reduce through LDS, and one fp32 atomic add per element lands in a
//     pre-zeroed output (split-K keeps all 256 CUs streaming on the 4x4 / 8x8 layers where
//     M = 16..64 rows; several convs may accumulate into one output: res_conv, Parallel()).
//   * one op = one launch; the host builds the op list once and replays it (HIP-graph friendly:
//     static pointers, the only per-eval inputs are device buffers).
//
// Operand encodings are documented at each launcher (`run_*`).

#include "sf_common.h"
#include "plan_ops.h"
#include "sf_dev.h"
#include "gemm_rows.h"
#include "conv_halo.h"
#include "conv_halo_small.h"
#include "conv_igemm.h"
#include "attn_ln.h"
#include <math.h>

typedef __attribute__((ext_vector_type(8))) sf_opnd bf16x8;
typedef __attribute__((ext_vector_type(4))) sf_opnd bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(1))) float gfloat;

__device__ __forceinline__ void atomic_add_f32(float* p, float v) {
  (void)__builtin_amdgcn_global_atomic_fadd_f32((gfloat*)p, v);
}
__device__ __forceinline__ float silu_f(float v) { return v / (1.0f + __expf(-v)); }
__device__ __forceinline__ float sigmoid_f(float v) { return 1.0f / (1.0f + __expf(-v)); }
__device__ __forceinline__ float gelu_f(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---------------------------------------------------------------------------------------------
// CONV: implicit GEMM on MFMA.
//   p[0] in (NHWC, bf16 or f32 by flag), p[1] packed weights (bf16), p[2] bias f32 [Cout] or NULL,
//   p[3] out f32, p[4] residual f32 (same indexing as out) or NULL
//   p[5] split-K workspace f32 [groups][M][Npad] (groups > 1 only)
//   i[0..] = B, H, W, Cin_pad, Ho, Wo, Cout, ldc, co_off, kh, kw, stride, pad, ksplit_groups, tile (WM*16+WN), 0
//   flags: 1 = A is f32, 2 = epilogue SiLU + PixelShuffle(2) (no split-K), 4 = accumulate into out (out += ...),
//          8 = split-K partials stay in the workspace; the consumer reduces them (LazySrc mode 1),
//          16 = the input is a nearest x2 upsampling of a stored [B, H/2, W/2, Cin] map (H, W = upsampled dims),
//          32 = ReLU, 64 = GELU(erf) in the epilogue (after bias / residual / accumulate)
//          256 = the split-K reduction stores NCHW: out[(b * Cout + n) * Ho*Wo + pixel] (the plan's output tensor; groups > 1, not deferred)
//   p[7] = (sum, sum of squares) slots of the pixel-shuffled output for the next GroupNorm-fused conv, or NULL (flag 2 only)
// No atomics: with groups == 1 every output element is owned by one wave (plain store / read-modify-write);
// with groups > 1 each K-slice group stores its partial tile to the workspace and k_splitk_reduce sums them
// (fp32 L2 atomics top out at ~25 G lane-ops/s on MI355X, which made the atomic split-K epilogue 10x the
// weight-streaming time of the 4x4 layers).
// Packed weight layout: [n_frag = Cout_pad/16][ks = tap*(Cin_pad/32)+cc][lane 64][8] bf16 with
//   element (lane, j) = W[n = n_frag*16 + (lane&15)][tap][c = cc*32 + 8*(lane>>4) + j].
// ---------------------------------------------------------------------------------------------

// k_conv_igemm lives in conv_igemm.h (shared with the CPU-thread emulation of tests/hostemu)

// out[m][co_off+n] (+)= bias[n] + resid + sum_g ws[g][m][n]      (second half of a split-K conv)
__global__ __launch_bounds__(256) void k_splitk_reduce(const float* __restrict__ ws, const float* __restrict__ bias,
                                                       const float* __restrict__ resid, float* __restrict__ out, int M,
                                                       int Cout, int npad, int groups, int ldc, int co_off, int accum,
                                                       int relu, int nchw_hw) {
  const long total = (long)M * Cout;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    int m, n;
    if (nchw_hw) { n = (int)((i / nchw_hw) % Cout); m = (int)(i / ((long)nchw_hw * Cout)) * nchw_hw + (int)(i % nchw_hw); }   // i = NCHW index: coalesced stores
    else { m = (int)(i / Cout); n = (int)(i - (long)m * Cout); }
    float v = bias ? bias[n] : 0.0f;
    for (int g = 0; g < groups; ++g) v += ws[((long)g * M + m) * npad + n];
    const long o = nchw_hw ? i : (long)m * ldc + co_off + n;      // NCHW [B][Cout][hw]: the plan's output layout (r04: was k_unpack_out)
    if (resid) v += resid[o];
    if (accum) v += out[o];
    if (relu == 1) v = fmaxf(v, 0.0f);
    else if (relu == 2) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
    out[o] = v;
  }
}

// ---------------------------------------------------------------------------------------------
// GN_ACT: GroupNorm(G=8, eps) over a (virtual) channel concat of two NHWC f32 sources, optional
// x*(scale+1)+shift, SiLU, -> bf16 NHWC; optional raw bf16 copy of the concat.
//   p[0] src1 f32 [B,HW,C1], p[1] src2 f32 [B,HW,C2] or NULL, p[2] gamma [C], p[3] beta [C],
//   p[4] scale_shift f32 (row b at p[4] + b*ss_stride: scale[C] then shift[C]) or NULL,
//   p[5] out bf16 [B,HW,C], p[6] raw bf16 [B,HW,C] or NULL, p[7] stats f64 [B*8][2], zeroed by the caller
//   i = B, HW, C1, C2, ss_stride, lazy mode, groups, npad, G (0 = 8 groups) ; f = eps, src2_scale ; flags: 1 = no SiLU
//   p[8..10] lazy source operands of src1 (see LazySrc): k_gn_stats materialises src1 into p[0] while reading it
// Two launches so that a B=1 eval still fills the chip: k_gn_stats (grid B*8*slices; per-block fp32 partial
// sums, combined in f64 with one L2 atomic pair per block) and k_gn_apply (pure elementwise).
// ---------------------------------------------------------------------------------------------
#define GN_CHUNKS_PER_BLOCK 2048     // float4 chunks per stats workgroup (256 threads x 8)

// A "lazy" fp32 NHWC tensor: the producer left it un-materialised and the FIRST consumer (k_gn_stats or
// k_gca_logits) computes each element while reading it and stores it to its final address, which saves the
// producer's own elementwise launch (every dependent launch costs ~4 us at B = 1).
//   mode 1: split-K partials  v = bias[c] + sum_g ws[g][m][c] (+ resid[m][c])
//   mode 2: gated residual    v = h[m][c] * gate[b][c] + (res ? res[m][c] : dst[m][c])
struct LazySrc {
  int mode, groups, npad, M;
  const float* a;      // ws | h
  const float* b;      // bias or NULL | gate [B, C]
  const float* r;      // resid or NULL
};

__device__ __forceinline__ f32x4 lazy_load4(const LazySrc& L, float* __restrict__ dst, int b, long m, int c, int C) {
  float* d = dst + m * C + c;
  f32x4 v;
  if (L.mode == 1) {
    v = L.b ? *reinterpret_cast<const f32x4*>(L.b + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int g = 0; g < L.groups; ++g) v += *reinterpret_cast<const f32x4*>(L.a + ((long)g * L.M + m) * L.npad + c);
    if (L.r) v += *reinterpret_cast<const f32x4*>(L.r + m * C + c);
  } else {
    const f32x4 h = *reinterpret_cast<const f32x4*>(L.a + m * C + c);
    const f32x4 g = *reinterpret_cast<const f32x4*>(L.b + (long)b * C + c);
    const f32x4 r = *reinterpret_cast<const f32x4*>((L.r ? L.r : d) + (L.r ? m * C + c : 0));
    v = h * g + r;
  }
  *reinterpret_cast<f32x4*>(d) = v;
  return v;
}

__device__ __forceinline__ f32x4 gn_load(const float* __restrict__ s1, const float* __restrict__ s2, int b, int HW, int C1,
                                         int C2, int p, int c, float s2_scale) {
  if (c < C1) return *reinterpret_cast<const f32x4*>(s1 + ((long)b * HW + p) * C1 + c);
  f32x4 v = *reinterpret_cast<const f32x4*>(s2 + ((long)b * HW + p) * C2 + (c - C1));
  return v * s2_scale;
}

__global__ __launch_bounds__(256) void k_gn_stats(float* __restrict__ s1, const float* __restrict__ s2,
                                                  double* __restrict__ stats, int HW, int C1, int C2, int slices,
                                                  float s2_scale, LazySrc lz, int G) {
  __shared__ double red[8];
  const int C = C1 + C2, Cg = C / G, cg4 = Cg / 4;
  const int bg = blockIdx.x / slices, sl = blockIdx.x % slices;
  const int b = bg / G, g = bg % G;
  const int chunks = HW * cg4;
  const int per = (chunks + slices - 1) / slices;
  const int c0 = sl * per, c1 = min(chunks, c0 + per);
  float s = 0.0f, q = 0.0f;
  for (int ch = c0 + threadIdx.x; ch < c1; ch += 256) {
    const int p = ch / cg4, c = g * Cg + (ch - p * cg4) * 4;
    const f32x4 v = (lz.mode && c < C1) ? lazy_load4(lz, s1, b, (long)b * HW + p, c, C1)
                                        : gn_load(s1, s2, b, HW, C1, C2, p, c, s2_scale);
    s += (v[0] + v[1]) + (v[2] + v[3]);
    q = fmaf(v[0], v[0], q); q = fmaf(v[1], v[1], q); q = fmaf(v[2], v[2], q); q = fmaf(v[3], v[3], q);
  }
  double ds = (double)wave_sum(s), dq = (double)wave_sum(q);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) { red[wv] = ds; red[4 + wv] = dq; }
  __syncthreads();
  if (threadIdx.x == 0) {
    typedef __attribute__((address_space(1))) double gdouble;
    (void)__builtin_amdgcn_global_atomic_fadd_f64((gdouble*)(stats + bg * 2), red[0] + red[1] + red[2] + red[3]);
    (void)__builtin_amdgcn_global_atomic_fadd_f64((gdouble*)(stats + bg * 2 + 1), red[4] + red[5] + red[6] + red[7]);
  }
}

// Pixel-major statistics for narrow groups (VAE: 32 groups of 4..16 channels): a workgroup walks whole pixels
// (fully coalesced rows), each thread owns one fixed float4 channel chunk, per-group sums meet in LDS.
// Needs C/4 <= 256, 256 % (C/4) == 0 and (C/G) % 4 == 0.
__global__ __launch_bounds__(256) void k_gn_stats_px(const float* __restrict__ s1, double* __restrict__ stats, int HW, int C,
                                                     int G, int slabs) {
  __shared__ float ls[256], lq[256];
  const int b = blockIdx.x / slabs, sl = blockIdx.x % slabs;
  const int c4 = C / 4, ppi = 256 / c4;
  const int chunk = threadIdx.x % c4, pofs = threadIdx.x / c4;
  const int per = (HW + slabs - 1) / slabs;
  const int p0 = sl * per, p1 = min(HW, p0 + per);
  float s = 0.0f, q = 0.0f;
  for (int p = p0 + pofs; p < p1; p += ppi) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(s1 + ((long)b * HW + p) * C + chunk * 4);
    s += (v[0] + v[1]) + (v[2] + v[3]);
    q = fmaf(v[0], v[0], q); q = fmaf(v[1], v[1], q); q = fmaf(v[2], v[2], q); q = fmaf(v[3], v[3], q);
  }
  ls[threadIdx.x] = s; lq[threadIdx.x] = q;
  __syncthreads();
  if ((int)threadIdx.x < G) {
    const int cpg = c4 / G;
    double ds = 0.0, dq = 0.0;
    for (int po = 0; po < ppi; ++po)
      for (int k = 0; k < cpg; ++k) {
        const int idx = po * c4 + threadIdx.x * cpg + k;
        ds += (double)ls[idx]; dq += (double)lq[idx];
      }
    typedef __attribute__((address_space(1))) double gdouble;
    (void)__builtin_amdgcn_global_atomic_fadd_f64((gdouble*)(stats + ((long)b * G + threadIdx.x) * 2), ds);
    (void)__builtin_amdgcn_global_atomic_fadd_f64((gdouble*)(stats + ((long)b * G + threadIdx.x) * 2 + 1), dq);
  }
}

__global__ __launch_bounds__(256) void k_gn_apply(const float* __restrict__ s1, const float* __restrict__ s2,
                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                  const float* __restrict__ ss, const double* __restrict__ stats,
                                                  sf_opnd* __restrict__ out, sf_opnd* __restrict__ raw, int B, int HW, int C1,
                                                  int C2, int ss_stride, float eps, float s2_scale, int no_silu, int G) {
  const int C = C1 + C2, Cg = C / G, c4 = C / 4;
  const long total = (long)B * HW * c4;
  const double inv_n = 1.0 / ((double)HW * Cg);
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % c4) * 4;
    const long bp = i / c4;
    const int p = (int)(bp % HW), b = (int)(bp / HW);
    const int g = c / Cg;
    const double m = stats[(b * G + g) * 2] * inv_n;
    const double var = stats[(b * G + g) * 2 + 1] * inv_n - m * m;
    const float mean = (float)m, rstd = rsqrtf((float)(var > 0.0 ? var : 0.0) + eps);
    const f32x4 v = gn_load(s1, s2, b, HW, C1, C2, p, c, s2_scale);
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c), be = *reinterpret_cast<const f32x4*>(beta + c);
    bf16x4 o, r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float y = (v[j] - mean) * rstd * ga[j] + be[j];
      if (ss) y = y * (ss[(long)b * ss_stride + c + j] + 1.0f) + ss[(long)b * ss_stride + C + c + j];
      if (!no_silu) y = silu_f(y);
      o[j] = (sf_opnd)y;
      r[j] = (sf_opnd)v[j];
    }
    *reinterpret_cast<bf16x4*>(out + ((long)b * HW + p) * C + c) = o;
    if (raw) *reinterpret_cast<bf16x4*>(raw + ((long)b * HW + p) * C + c) = r;
  }
}

// k_gn_one (r06): statistics AND normalisation of one (image, group) in one workgroup, the slab held in registers between the two -- one
// launch and one read of the tensor instead of two of each (the large-batch plans run 30-42 GroupNorm passes per eval: 0.86 of a 5.66 ms
// B = 32 eval in k_gn_stats + k_gn_apply).  Same operands, lazy sources and outputs as the pair; the sums are taken per thread in fp32,
// over the workgroup in double (another -- also fixed -- order than k_gn_stats' per-block atomics: reproducible run to run).
// Taken when the grid B * G fills the chip (>= 256 workgroups) and a group's slab fits NT * NCH float4 registers (run_gn).
template <int NT, int NCH>
__global__ __launch_bounds__(NT) void k_gn_one(float* __restrict__ s1, const float* __restrict__ s2, const float* __restrict__ gamma,
                                               const float* __restrict__ beta, const float* __restrict__ ss, sf_opnd* __restrict__ out,
                                               sf_opnd* __restrict__ raw, int HW, int C1, int C2, int ss_stride, float eps, float s2_scale,
                                               int no_silu, LazySrc lz, int G) {
  __shared__ double red[2][NT / 64];
  const int C = C1 + C2, Cg = C / G, cg4 = Cg / 4;
  const int b = blockIdx.x / G, g = blockIdx.x - b * G;
  const int chunks = HW * cg4;
  f32x4 v[NCH];
  float s = 0.0f, q = 0.0f;
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int ch = threadIdx.x + k * NT;
    v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (ch < chunks) {
      const int p = ch / cg4, c = g * Cg + (ch - p * cg4) * 4;
      v[k] = (lz.mode && c < C1) ? lazy_load4(lz, s1, b, (long)b * HW + p, c, C1) : gn_load(s1, s2, b, HW, C1, C2, p, c, s2_scale);
      s += (v[k][0] + v[k][1]) + (v[k][2] + v[k][3]);
      q = fmaf(v[k][0], v[k][0], q); q = fmaf(v[k][1], v[k][1], q); q = fmaf(v[k][2], v[k][2], q); q = fmaf(v[k][3], v[k][3], q);
    }
  }
  const double ds = (double)wave_sum(s), dq = (double)wave_sum(q);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) { red[0][wv] = ds; red[1][wv] = dq; }
  __syncthreads();
  double ts = 0.0, tq = 0.0;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) { ts += red[0][w]; tq += red[1][w]; }
  const double inv_n = 1.0 / ((double)HW * Cg);
  const double m = ts * inv_n, var = tq * inv_n - m * m;
  const float mean = (float)m, rstd = rsqrtf((float)(var > 0.0 ? var : 0.0) + eps);
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int ch = threadIdx.x + k * NT;
    if (ch < chunks) {
      const int p = ch / cg4, c = g * Cg + (ch - p * cg4) * 4;
      const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c), be = *reinterpret_cast<const f32x4*>(beta + c);
      bf16x4 o, r;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float y = (v[k][j] - mean) * rstd * ga[j] + be[j];
        if (ss) y = y * (ss[(long)b * ss_stride + c + j] + 1.0f) + ss[(long)b * ss_stride + C + c + j];
        if (!no_silu) y = silu_f(y);
        o[j] = (sf_opnd)y;
        r[j] = (sf_opnd)v[k][j];
      }
      *reinterpret_cast<bf16x4*>(out + ((long)b * HW + p) * C + c) = o;
      if (raw) *reinterpret_cast<bf16x4*>(raw + ((long)b * HW + p) * C + c) = r;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// LN: per-row normalisation over C (biased variance), one 256-thread workgroup per row.
//   p[0] in f32 [R,C], p[1] gain [C], p[2] bias [C] or NULL, p[3] out (bf16 or f32), p[4] residual f32 or NULL
//   i = R, C ; f = eps ; flags: 1 = GELU(x) before normalising, 2 = f32 output (+ residual), else bf16 output
// ---------------------------------------------------------------------------------------------

// ---------------------------------------------------------------------------------------------
// GEMV: y[m][n] = act_out(bias[n] + sum_k W[n][k] * act_in(x[m][k])), M <= 8 rows, one wave per n.
//   p[0] x f32 (row stride ldx), p[1] W bf16 [N][Kp] (Kp = K padded to 8), p[2] bias f32 or NULL, p[3] y f32 (row stride ldy)
//   i = M, N, K, Kp, ldx, ldy ; flags: bit0 SiLU on input, bits 1-2 output act (0 none, 1 SiLU, 2 sigmoid)
// ---------------------------------------------------------------------------------------------
// GEMV_ROWS output rows per wave = that many independent 16-byte weight loads in flight per lane and step: 4 for the long
// layers (the 33 792-row time-MLP GEMV streams 72 MB), 1 for the short ones (N <= 2048: gca nets, time tokens), which are a
// single latency-bound phase and want as many workgroups as they have rows.
template <int GEMV_ROWS>
__global__ __launch_bounds__(256) void k_gemv(const float* __restrict__ x, const sf_opnd* __restrict__ W,
                                              const float* __restrict__ bias, float* __restrict__ y, int M, int N, int K,
                                              int Kp, int ldx, int ldy, int in_silu, int out_act) {
  const int n0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * GEMV_ROWS, lane = threadIdx.x & 63;
  if (n0 >= N) return;
  float acc[GEMV_ROWS][8];
#pragma unroll
  for (int r = 0; r < GEMV_ROWS; ++r)
#pragma unroll
    for (int m = 0; m < 8; ++m) acc[r][m] = 0.0f;
  for (int k = lane * 8; k < Kp; k += 512) {
    bf16x8 w[GEMV_ROWS];
#pragma unroll
    for (int r = 0; r < GEMV_ROWS; ++r) {
      const int n = min(n0 + r, N - 1);
      w[r] = *reinterpret_cast<const bf16x8*>(W + (long)n * Kp + k);
    }
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      if (m < M) {
        float xv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          xv[j] = (k + j < K) ? x[(long)m * ldx + k + j] : 0.0f;
          if (in_silu) xv[j] = silu_f(xv[j]);
        }
#pragma unroll
        for (int r = 0; r < GEMV_ROWS; ++r)
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[r][m] = fmaf((float)w[r][j], xv[j], acc[r][m]);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < GEMV_ROWS; ++r) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      if (m < M) {
        float v = wave_sum(acc[r][m]);
        if (lane == 0 && n0 + r < N) {
          if (bias) v += bias[n0 + r];
          if (out_act == 1) v = silu_f(v);
          else if (out_act == 2) v = sigmoid_f(v);
          y[(long)m * ldy + n0 + r] = v;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// ATTN: softmax(q k^T) v for 16 queries, head dim 64, <= 24 keys built from up to 3 segments.
//   p[0] q f32 [B*16, ldq] (head h at column h*64), p[1] out bf16 [B*16, 512]
//   segment s (s = 0..2): p[2+2s] keys, p[3+2s] values (f32); i[4+4s..] = rows, row_stride, batch_stride, head_stride
//   i[0] = B, i[1] = heads, i[2] = ldq ; f[0] = q scale ; flags: 1 = fp32 output.   One 256-thread workgroup per (b, head).
// ---------------------------------------------------------------------------------------------

// ---------------------------------------------------------------------------------------------
// GCA_POOL: pooled[b][c] = sum_p softmax_p(h[b,p,:] . wk + bk) * h[b,p,c]     (GlobalContext :930-941)
//   p[0] h f32 [B,HW,C], p[1] wk f32 [C], p[2] bk f32 [1], p[3] pooled f32 [B,C] ZEROED by the caller, p[4] logits scratch f32 [B*HW]
//   i = B, HW, C, lazy mode, groups, npad (p[8..10]: h may be un-reduced split-K partials).  Two launches: one wave per pixel for the logits, then one workgroup per (b, 32 channels).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gca_logits(float* __restrict__ h, const float* __restrict__ wk,
                                                    const float* __restrict__ bk, float* __restrict__ logit, int rows, int C,
                                                    LazySrc lz) {
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (p >= rows) return;
  float a = 0.0f;
  for (int c = lane * 4; c < C; c += 256) {
    const f32x4 x = lz.mode ? lazy_load4(lz, h, 0, (long)p, c, C) : *reinterpret_cast<const f32x4*>(h + (long)p * C + c);
    const f32x4 w = *reinterpret_cast<const f32x4*>(wk + c);
    a += x[0] * w[0] + x[1] * w[1] + x[2] * w[2] + x[3] * w[3];
  }
  a = wave_sum(a);
  if (lane == 0) logit[p] = a + bk[0];
}

// Small maps (<= 256 pixels): one workgroup per pixel so that the (lazy) row read is spread over 4x the lanes.
__global__ __launch_bounds__(256) void k_gca_logits_wg(float* __restrict__ h, const float* __restrict__ wk,
                                                       const float* __restrict__ bk, float* __restrict__ logit, int C, LazySrc lz) {
  __shared__ float red[4];
  const int p = blockIdx.x;
  float a = 0.0f;
  for (int c = threadIdx.x * 4; c < C; c += 1024) {
    const f32x4 x = lz.mode ? lazy_load4(lz, h, 0, (long)p, c, C) : *reinterpret_cast<const f32x4*>(h + (long)p * C + c);
    const f32x4 w = *reinterpret_cast<const f32x4*>(wk + c);
    a += x[0] * w[0] + x[1] * w[1] + x[2] * w[2] + x[3] * w[3];
  }
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) logit[p] = (red[0] + red[1]) + (red[2] + red[3]) + bk[0];
}

// grid (B, ceil(HW/32)): every workgroup re-derives the softmax normaliser from the <=1024 logits, then
// accumulates its 32 pixels for ALL channels (coalesced along C) and adds into the pre-zeroed pooled[b][:].
__global__ __launch_bounds__(256) void k_gca_pool(const float* __restrict__ h, const float* __restrict__ logit,
                                                  float* __restrict__ pooled, int HW, int C, int chunks, int csplit) {
  __shared__ float red[8];
  __shared__ float e[32];
  // grid = (b, pixel chunk, channel slab of 256): small maps (4x4, 8x8) have one or two pixel chunks only, the channel
  // slabs are what gives them enough workgroups to finish in one short phase
  const int cs = blockIdx.x % csplit, bp = blockIdx.x / csplit;
  const int b = bp / chunks, pc = bp % chunks;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const float* lg = logit + (long)b * HW;
  float mx = -INFINITY;
  for (int p = threadIdx.x; p < HW; p += 256) mx = fmaxf(mx, lg[p]);
  mx = wave_max(mx);
  if (lane == 0) red[wv] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float s = 0.0f;
  for (int p = threadIdx.x; p < HW; p += 256) s += expf(lg[p] - mx);
  s = wave_sum(s);
  if (lane == 0) red[4 + wv] = s;
  const int p0 = pc * 32;
  if (threadIdx.x < 32) e[threadIdx.x] = (p0 + (int)threadIdx.x < HW) ? expf(lg[p0 + threadIdx.x] - mx) : 0.0f;
  __syncthreads();
  const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
  const int np = min(32, HW - p0);
  const float* hb = h + ((long)b * HW + p0) * C;
  for (int c = cs * 256 + threadIdx.x; c < C; c += 256 * csplit) {
    float a = 0.0f;
    for (int p = 0; p < np; ++p) a = fmaf(e[p], hb[(long)p * C + c], a);
    atomic_add_f32(pooled + (long)b * C + c, a * inv);
  }
}

// ---------------------------------------------------------------------------------------------
// ELTWISE (flags = mode):
//   1 GATE_RES   out[b,p,c] = h[b,p,c]*gate[b,c] + res[b,p,c]   p0 h, p1 gate [B,C], p2 res or NULL (then out holds it), p3 out ; i = B,HW,C
//   2 PACK_IN    out NHWC f32 [B,HW,Cp] = concat(cond NCHW [B,Cc,HW], x NCHW [B,Cx,HW]), zero pad ; p0 cond, p1 x, p3 out ; i = B,HW,Cc,Cx,Cp
//   3 UNPACK_OUT out NCHW [B,C,HW] = in NHWC [B,HW,ldi] first C channels ; p0 in, p3 out ; i = B,HW,C,ldi
//   4 ADD        out[i] += p0[i] ; i[0] = n
//   5 PACK_ACT   p0 f32 [.., ld] -> p3 bf16 B-operand fragments ; i = N, K, ld, transpose
//   6 SOFTMAX    p3 bf16 [R,N] = softmax(f[0] * p0 f32 [R,N]) ; i = R, N
//   7 RELU_BWD / 8 SCALE_IN / 9 SCALE_IN_BWD: lpips_ops.hip
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gate_res(const float* __restrict__ h, const float* __restrict__ gate,
                                                  const float* __restrict__ res, float* __restrict__ out, int B, int HW,
                                                  int C) {
  const long n4 = (long)B * HW * C / 4;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const long e = i * 4;
    const int c = (int)(e % C);
    const int b = (int)(e / ((long)HW * C));
    const f32x4 hv = *reinterpret_cast<const f32x4*>(h + e);
    const f32x4 gv = *reinterpret_cast<const f32x4*>(gate + (long)b * C + c);
    const f32x4 rv = *reinterpret_cast<const f32x4*>((res ? res : out) + e);
    *reinterpret_cast<f32x4*>(out + e) = hv * gv + rv;
  }
}
__global__ __launch_bounds__(256) void k_pack_in(const float* __restrict__ cond, const float* __restrict__ x,
                                                 float* __restrict__ out, int B, int HW, int Cc, int Cx, int Cp) {
  const long n = (long)B * HW * Cp;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int c = (int)(i % Cp);
    const long bp = i / Cp;
    const int p = (int)(bp % HW), b = (int)(bp / HW);
    float v = 0.0f;
    if (c < Cc) v = cond[((long)b * Cc + c) * HW + p];
    else if (c < Cc + Cx) v = x[((long)b * Cx + (c - Cc)) * HW + p];
    out[i] = v;
  }
}
// PACK_ACT: an fp32 activation matrix becomes the B operand of k_conv_igemm (same fragment order as
// sf_conv_pack_weights): W[n][c] = src[n*ld + c] (T = 0) or src[c*ld + n] (T = 1), n < N, c < K, zero padded.
__global__ __launch_bounds__(256) void k_pack_act(const float* __restrict__ src, bf16x8* __restrict__ out, int N, int K, int ld,
                                                  int T) {
  const int kchunks = (K + 31) / 32, n_frags = (N + 15) / 16;
  const long total = (long)n_frags * kchunks * 64;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int lane = (int)(i & 63);
    const long r = i >> 6;
    const int ks = (int)(r % kchunks), nf = (int)(r / kchunks);
    const int n = nf * 16 + (lane & 15), c0 = ks * 32 + 8 * (lane >> 4);
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = c0 + j;
      float x = 0.0f;
      if (n < N && c < K) x = T ? src[(long)c * ld + n] : src[(long)n * ld + c];
      v[j] = (sf_opnd)x;
    }
    out[i] = v;
  }
}

// SOFTMAX_ROWS: out bf16 [R, N] = softmax(scale * in f32 [R, N]) per row; one workgroup per row.
__global__ __launch_bounds__(256) void k_softmax_rows(const float* __restrict__ in, sf_opnd* __restrict__ out, int N, float scale) {
  __shared__ float red[8];
  const float* x = in + (long)blockIdx.x * N;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float mx = -INFINITY;
  for (int i = threadIdx.x; i < N; i += 256) mx = fmaxf(mx, x[i] * scale);
  mx = wave_max(mx);
  if (lane == 0) red[wv] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float s = 0.0f;
  for (int i = threadIdx.x; i < N; i += 256) s += expf(x[i] * scale - mx);
  s = wave_sum(s);
  if (lane == 0) red[4 + wv] = s;
  __syncthreads();
  const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
  for (int i = threadIdx.x; i < N; i += 256) out[(long)blockIdx.x * N + i] = (sf_opnd)(expf(x[i] * scale - mx) * inv);
}

__global__ __launch_bounds__(256) void k_unpack_out(const float* __restrict__ in, float* __restrict__ out, int B, int HW,
                                                    int C, int ldi) {
  const long n = (long)B * C * HW;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int p = (int)(i % HW);
    const long bc = i / HW;
    const int c = (int)(bc % C), b = (int)(bc / C);
    out[i] = in[((long)b * HW + p) * ldi + c];
  }
}
__global__ __launch_bounds__(256) void k_add(const float* __restrict__ a, float* __restrict__ out, long n) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] += a[i];
}

// TIME_EMB: f[b] = [t, sin(2*pi*t*w_i) (i<half), cos(2*pi*t*w_i) (i<half)]   (LearnedSinusoidalPosEmb :624-639)
//   p0 t f32 [B], p1 w f32 [half], p3 out f32 [B, 1+2*half] ; i = B, half
__global__ void k_time_emb(const float* __restrict__ t, const float* __restrict__ w, float* __restrict__ out, int B,
                           int half) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * half) return;
  const int b = i / half, j = i - b * half;
  const float fr = t[b] * w[j] * 2.0f * 3.14159265358979323846f;
  float* o = out + (long)b * (1 + 2 * half);
  if (j == 0) o[0] = t[b];
  o[1 + j] = sinf(fr);
  o[1 + half + j] = cosf(fr);
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
template <int WM, int WN>
static void launch_conv(const ConvArgs& a, bool a_fp32, int blocks, hipStream_t st) {
  if (a_fp32) k_conv_igemm<WM, WN, true><<<blocks, 256, 0, st>>>(a);
  else k_conv_igemm<WM, WN, false><<<blocks, 256, 0, st>>>(a);
}

// k_conv_glds (conv_glds.h): the LDS-tiled conv for operand-type activations, staged by LDS-DMA through a ring of NST buffers.
// Ring depth 4 by default (3 measures the same: the ring is not latency bound; a tile code selects either explicitly, as the tests do).
static int conv_glds_depth() { return 4; }
template <int BNF, int NST, bool GN>
static int launch_conv_glds(const ConvArgs& a, int nblk, double* part, int cg, hipStream_t st) {
  static unsigned mask = 0;
  const uint32_t lds = conv_glds_lds_bytes(BNF, NST);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) SF_FAIL(SF_ERR_LAUNCH, "hipGetDevice failed");
  if (dev >= 32 || !(mask & (1u << dev))) {             // dynamic LDS above 64 KiB: per kernel and per device
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_glds<BNF, NST, GN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      SF_FAIL(SF_ERR_LAUNCH, "hipFuncSetAttribute(max dynamic LDS) failed");
    if (dev < 32) mask |= 1u << dev;
  }
  k_conv_glds<BNF, NST, GN><<<nblk, 512, lds, st>>>(a, part, cg);
  SF_CHECK_LAUNCH("conv_glds");
  return SF_OK;
}
// k_conv3_halo (conv_halo.h): 3x3 / stride 1 / pad 1 layers with the pixel tile's halo staged once per 64-channel chunk.
// (the default kernel of those layers; tile-code selectors 3 / 4 pick k_conv_glds, 1 k_conv_lds: tests compare all three)
static bool conv_halo_enabled() { return true; }
template <int BNF, int NST, bool GN>
static int launch_conv_halo(const ConvArgs& a, int nblk, double* part, int cg, hipStream_t st) {
  static unsigned mask = 0;
  const uint32_t lds = conv_halo_lds_bytes(BNF, NST);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) SF_FAIL(SF_ERR_LAUNCH, "hipGetDevice failed");
  if (dev >= 32 || !(mask & (1u << dev))) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3_halo<BNF, NST, GN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      SF_FAIL(SF_ERR_LAUNCH, "hipFuncSetAttribute(max dynamic LDS) failed");
    if (dev < 32) mask |= 1u << dev;
  }
  k_conv3_halo<BNF, NST, GN><<<nblk, 512, lds, st>>>(a, part, cg);
  SF_CHECK_LAUNCH("conv3_halo");
  return SF_OK;
}
template <int BNF, bool GN>
static int launch_conv_halo_nst(const ConvArgs& a, int nblk, double* part, int cg, int nst, hipStream_t st) {
  return nst == 3 ? launch_conv_halo<BNF, 3, GN>(a, nblk, part, cg, st) : launch_conv_halo<BNF, 4, GN>(a, nblk, part, cg, st);
}

// k_conv3_halo_sm (conv_halo_small.h): 3x3 / stride 1 / pad 1 layers on WHOLE 4x4 / 8x8 maps (8 / 2 maps per pixel tile), split-K over the 64-channel chunks
template <int BNF, int NST, int MAPL>
static int launch_conv_halo_sm(const ConvArgs& a, int nblk, hipStream_t st) {
  static unsigned mask = 0;
  const uint32_t lds = conv_halo_sm_lds_bytes(BNF, NST, MAPL);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) SF_FAIL(SF_ERR_LAUNCH, "hipGetDevice failed");
  if (dev >= 32 || !(mask & (1u << dev))) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3_halo_sm<BNF, NST, MAPL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      SF_FAIL(SF_ERR_LAUNCH, "hipFuncSetAttribute(max dynamic LDS) failed");
    if (dev < 32) mask |= 1u << dev;
  }
  k_conv3_halo_sm<BNF, NST, MAPL><<<nblk, 512, lds, st>>>(a);
  SF_CHECK_LAUNCH("conv3_halo_sm");
  return SF_OK;
}
template <int BNF>
static int launch_conv_halo_sm_nst(const ConvArgs& a, int nblk, int nst, hipStream_t st) {
  if (a.H == 4) return nst == 3 ? launch_conv_halo_sm<BNF, 3, 2>(a, nblk, st) : launch_conv_halo_sm<BNF, 4, 2>(a, nblk, st);
  return nst == 3 ? launch_conv_halo_sm<BNF, 3, 3>(a, nblk, st) : launch_conv_halo_sm<BNF, 4, 3>(a, nblk, st);
}

template <int BNF, bool GN>
static int launch_conv_glds_nst(const ConvArgs& a, int nblk, double* part, int cg, int nst, hipStream_t st) {
  return nst == 3 ? launch_conv_glds<BNF, 3, GN>(a, nblk, part, cg, st) : launch_conv_glds<BNF, 4, GN>(a, nblk, part, cg, st);
}

// The tail of a split-K conv launch: the reduction over the group slabs (bias, residual, accumulate, ReLU / GELU, NCHW form), unless deferred
static int conv_splitk_tail(const sf_op& op, const ConvArgs& a, int M, hipStream_t st) {
  if (a.groups <= 1) return SF_OK;
  if (!a.ws) SF_FAIL(SF_ERR_INVALID, "conv: split-K needs a workspace");
  if (op.flags & 8) return SF_OK;      // reduction deferred to the consumer (LazySrc mode 1)
  k_splitk_reduce<<<sf_grid_cap(sf_div_up((long)M * a.Cout, 256)), 256, 0, st>>>(a.ws, a.bias, a.resid, a.out, M, a.Cout, a.npad,
                                                                                a.groups, a.ldc, a.co_off, a.accum, a.relu, a.out_nchw_hw);
  SF_CHECK_LAUNCH("splitk_reduce");
  return SF_OK;
}

static int run_conv(const sf_op& op, hipStream_t st) {
  ConvArgs a;
  a.in = op.p[0]; a.w = (const bf16x8*)op.p[1]; a.bias = (const float*)op.p[2]; a.out = (float*)op.p[3];
  a.resid = (const float*)op.p[4]; a.ws = (float*)op.p[5];
  a.accum = (op.flags & 4) ? 1 : 0;
  a.B = op.i[0]; a.H = op.i[1]; a.W = op.i[2]; a.Cin = op.i[3]; a.Ho = op.i[4]; a.Wo = op.i[5]; a.Cout = op.i[6];
  a.ldc = op.i[7]; a.co_off = op.i[8]; a.kh = op.i[9]; a.kw = op.i[10]; a.stride = op.i[11]; a.pad = op.i[12];
  a.groups = op.i[13] > 0 ? op.i[13] : 1;
  const int tile = op.i[14];
  const int WM = tile / 16, WN = tile % 16;
  a.pixshuf = (op.flags & 2) ? 1 : 0;
  a.slots_out = (float*)op.p[7];
  a.out_nchw_hw = (op.flags & 256) ? op.i[4] * op.i[5] : 0;
  a.ups = (op.flags & 16) ? 1 : 0;
  a.relu = (op.flags & 32) ? 1 : ((op.flags & 64) ? 2 : 0);
  if (a.ups && ((a.H | a.W) & 1)) SF_FAIL(SF_ERR_INVALID, "conv: upsampled input dims must be even");
  if (a.Cin % 32) SF_FAIL(SF_ERR_INVALID, "conv: Cin_pad must be a multiple of 32");
  if (a.pixshuf && a.groups != 1) SF_FAIL(SF_ERR_INVALID, "conv: pixel-shuffle epilogue cannot be split-K");
  a.cchunks = a.Cin / 32;
  a.KS = a.kh * a.kw * a.cchunks;
  const int M = a.B * a.Ho * a.Wo;
  a.m_frags = (M + 15) / 16;
  a.n_frags = (a.Cout + 15) / 16;
  a.m_tiles = (a.m_frags + WM - 1) / WM;
  a.n_tiles = (a.n_frags + WN - 1) / WN;
  a.npad = a.n_frags * 16;
  if (a.pixshuf) a.groups = 1;
  if (a.slots_out && (!a.pixshuf || tile >= 256 || (a.Cout / 4) % 16 || a.ldc % 16 || a.co_off % 16 || M % 16 || (a.Ho * a.Wo) % 16))
    SF_FAIL(SF_ERR_INVALID, "conv: output slots ride on the pixel-shuffle epilogue only (16 | Cout / 4, ldc, co_off, Ho * Wo)");
  if (a.out_nchw_hw && (a.groups < 2 || (op.flags & (8 | 4)) || op.p[4] || tile >= 256 || a.co_off || a.ldc != a.Cout))
    SF_FAIL(SF_ERR_INVALID, "conv: NCHW output is written by the (non-deferred) split-K reduction of a plain conv only");
  a.steps_per_wave = (a.KS + a.groups * 4 - 1) / (a.groups * 4);
  const int blocks = a.m_tiles * a.n_tiles * a.groups;
  const bool f32 = (op.flags & 1) != 0;
  if (tile >= 256) {                                   // LDS-tiled large-M kernel: tile = 256 + 16 * sel + n-fragments per workgroup
    const int bnf = (tile - 256) & 15, sel = (tile - 256) >> 4;      // sel 0: the default kernel; 1: k_conv_lds; 3 / 4: k_conv_glds ring depth;
    if (sel != 0 && sel != 1 && sel != 3 && sel != 4 && sel != 6 && sel != 7 && sel != 8 && sel != 9)      // 6 / 7: k_conv3_halo with a 3 / 4 deep weight ring; 8 / 9: k_conv3_halo_sm
      SF_FAIL(SF_ERR_INVALID, "conv: unknown LDS kernel selector %d", sel);
    const bool glds_ok = !f32 && a.Cin % 64 == 0 && a.Cout % 4 == 0 && a.ldc % 4 == 0 && a.co_off % 4 == 0 && (bnf == 8 || bnf == 4);
    if (sel >= 3 && !glds_ok) SF_FAIL(SF_ERR_INVALID, "conv: k_conv_glds takes operand-type activations with Cin %% 64 == 0 and float4-aligned output rows only");
    // r06: k_conv_lds / k_conv_glds take split-K groups over their stage range (workspace [grp][m][npad], as k_conv_igemm leaves it) and the
    // SiLU + PixelShuffle(2) epilogue of an Upsample (the 4x4 level and the Upsample convs of the large-batch plans); k_conv3_halo does not
    const bool special = a.groups != 1 || a.pixshuf;      // (a split-K launch leaves residual / accumulate / ReLU to the reduction)
    if (special && ((op.flags & 128) || sel == 6 || sel == 7 || (a.groups != 1 && !a.ws) || (a.pixshuf && ((op.flags & 4) || op.p[4] || a.relu))))
      SF_FAIL(SF_ERR_INVALID, "conv: split-K / pixel shuffle on the LDS-tiled kernels: no GroupNorm partials, not k_conv3_halo; no residual / accumulate / ReLU under the shuffle");
    // whole 4x4 / 8x8 maps: k_conv3_halo_sm (the default for those layers; selectors 3 / 4 / 1 keep k_conv_glds / k_conv_lds for comparison)
    const bool sm_ok = glds_ok && conv_halo_sm_ok(a) && !a.pixshuf && !(op.flags & 128);
    if (sel >= 8 && !sm_ok) SF_FAIL(SF_ERR_INVALID, "conv: k_conv3_halo_sm takes 3x3 / stride 1 / pad 1 layers on 4x4 / 8x8 maps, groups <= Cin / 64");
    const int sm = sel >= 8 ? sel - 5 : ((sel == 0 && sm_ok) ? conv_glds_depth() : 0);      // ring depth, 0 = not taken
    if (!sm && a.groups > (((glds_ok && sel != 1) ? a.KS : a.KS + 1) >> 1)) SF_FAIL(SF_ERR_INVALID, "conv: more split-K groups than stages");
    if (a.groups == 1 && a.pixshuf) a.ws = nullptr;      // (ws of an LDS-tiled launch without split-K = the operand-type twin)
    a.m_tiles = (a.m_frags + 7) / 8;
    a.n_tiles = (a.n_frags + bnf - 1) / bnf;
    const int nblk = a.m_tiles * a.n_tiles * a.groups;
    const bool halo_ok = glds_ok && conv_halo_ok(a) && M % 128 == 0 && !special;
    if ((sel == 6 || sel == 7) && !halo_ok) SF_FAIL(SF_ERR_INVALID, "conv: k_conv3_halo takes 3x3 / stride 1 / pad 1 layers with W %% 16 == 0, H %% 8 == 0 only");
    if (sm) {
      if (int rc = bnf == 8 ? launch_conv_halo_sm_nst<8>(a, nblk, sm, st) : launch_conv_halo_sm_nst<4>(a, nblk, sm, st)) return rc;
      return conv_splitk_tail(op, a, M, st);
    }
    const int glds = (glds_ok && sel != 1 && sel < 6) ? (sel >= 3 ? sel : conv_glds_depth()) : 0;
    const int halo = (sel == 6 || sel == 7) ? sel - 3 : ((sel == 0 && halo_ok && glds && conv_halo_enabled()) ? glds : 0);      // ring depth of the halo kernel, 0 = not taken
    if (op.flags & 128) {                                // the epilogue also leaves GroupNorm partial sums (conv_lds.h)
      double* part = (double*)op.p[6];
      const int cg = op.i[15];
      if (!part || (cg != 4 && cg != 8 && cg != 16) || a.Cout % cg || a.co_off || a.ldc != a.Cout || (a.Ho * a.Wo) % 128)
        SF_FAIL(SF_ERR_INVALID, "conv: GroupNorm-partials epilogue needs whole rows, 128 | Ho*Wo, group width 4 / 8 / 16");
      if (halo) return bnf == 8 ? launch_conv_halo_nst<8, true>(a, nblk, part, cg, halo, st) : launch_conv_halo_nst<4, true>(a, nblk, part, cg, halo, st);
      if (glds) return bnf == 8 ? launch_conv_glds_nst<8, true>(a, nblk, part, cg, glds, st) : launch_conv_glds_nst<4, true>(a, nblk, part, cg, glds, st);
      if (bnf == 8) {
        if (f32) k_conv_lds_gn<8, true><<<nblk, 256, 0, st>>>(a, part, cg); else k_conv_lds_gn<8, false><<<nblk, 256, 0, st>>>(a, part, cg);
      } else if (bnf == 4) {
        if (f32) k_conv_lds_gn<4, true><<<nblk, 256, 0, st>>>(a, part, cg); else k_conv_lds_gn<4, false><<<nblk, 256, 0, st>>>(a, part, cg);
      } else {
        SF_FAIL(SF_ERR_INVALID, "conv: unsupported LDS tile %d", bnf);
      }
      SF_CHECK_LAUNCH("conv_lds_gn");
      return SF_OK;
    }
    if (halo) return bnf == 8 ? launch_conv_halo_nst<8, false>(a, nblk, nullptr, 0, halo, st) : launch_conv_halo_nst<4, false>(a, nblk, nullptr, 0, halo, st);
    if (glds) {
      if (int rc = bnf == 8 ? launch_conv_glds_nst<8, false>(a, nblk, nullptr, 0, glds, st) : launch_conv_glds_nst<4, false>(a, nblk, nullptr, 0, glds, st)) return rc;
      return conv_splitk_tail(op, a, M, st);
    }
    if (bnf == 8) {
      if (f32) k_conv_lds<8, true><<<nblk, 256, 0, st>>>(a); else k_conv_lds<8, false><<<nblk, 256, 0, st>>>(a);
    } else if (bnf == 4) {
      if (f32) k_conv_lds<4, true><<<nblk, 256, 0, st>>>(a); else k_conv_lds<4, false><<<nblk, 256, 0, st>>>(a);
    } else {
      SF_FAIL(SF_ERR_INVALID, "conv: unsupported LDS tile %d", bnf);
    }
    SF_CHECK_LAUNCH("conv_lds");
    return conv_splitk_tail(op, a, M, st);
  }
  switch (tile) {
    case 1 * 16 + 1: launch_conv<1, 1>(a, f32, blocks, st); break;
    case 1 * 16 + 2: launch_conv<1, 2>(a, f32, blocks, st); break;
    case 1 * 16 + 4: launch_conv<1, 4>(a, f32, blocks, st); break;
    case 2 * 16 + 1: launch_conv<2, 1>(a, f32, blocks, st); break;
    case 2 * 16 + 2: launch_conv<2, 2>(a, f32, blocks, st); break;
    case 2 * 16 + 4: launch_conv<2, 4>(a, f32, blocks, st); break;
    case 4 * 16 + 1: launch_conv<4, 1>(a, f32, blocks, st); break;
    case 4 * 16 + 2: launch_conv<4, 2>(a, f32, blocks, st); break;
    case 4 * 16 + 4: launch_conv<4, 4>(a, f32, blocks, st); break;
    default: SF_FAIL(SF_ERR_INVALID, "conv: unsupported wave tile %dx%d", WM, WN);
  }
  SF_CHECK_LAUNCH("conv_igemm");
  return conv_splitk_tail(op, a, M, st);
}

// Lazy-source descriptor of an op: i[ibase] = mode, i[ibase+1] = groups, i[ibase+2] = npad ; p[8] = a, p[9] = b, p[10] = r.
static int lazy_from_op(const sf_op& op, int ibase, int M, LazySrc& lz) {
  lz.mode = op.i[ibase]; lz.groups = op.i[ibase + 1]; lz.npad = op.i[ibase + 2]; lz.M = M;
  lz.a = (const float*)op.p[8]; lz.b = (const float*)op.p[9]; lz.r = (const float*)op.p[10];
  if (lz.mode < 0 || lz.mode > 2) SF_FAIL(SF_ERR_INVALID, "lazy source: unknown mode %d", lz.mode);
  if (lz.mode && !lz.a) SF_FAIL(SF_ERR_INVALID, "lazy source: missing operand");
  if (lz.mode == 1 && (lz.groups < 1 || lz.npad % 4)) SF_FAIL(SF_ERR_INVALID, "lazy source: bad split-K geometry");
  if (lz.mode == 2 && !lz.b) SF_FAIL(SF_ERR_INVALID, "lazy source: gate missing");
  return SF_OK;
}

static int run_gn(const sf_op& op, hipStream_t st) {
  const int B = op.i[0], HW = op.i[1], C1 = op.i[2], C2 = op.i[3];
  const int C = C1 + C2, G = op.i[8] > 0 ? op.i[8] : 8;
  if (C % (4 * G) || C1 % 4 || !op.p[7])
    SF_FAIL(SF_ERR_INVALID, "gn_act: unsupported shape HW=%d C=%d G=%d (or missing stats buffer)", HW, C, G);
  const int chunks = HW * (C / G) / 4;
  LazySrc lz;
  if (int rc = lazy_from_op(op, 5, B * HW, lz)) return rc;
  const int c4 = C / 4;
  // r06: one launch for statistics + normalisation when B * G workgroups are a fair share of the chip and a group's slab fits the registers of its
  // workgroup (the large-batch UNet plans: G = 8, B >= 32; flag 4 = the two-launch form, flag 8 = this form at any B, for comparison)
  if (!(op.flags & (2 | 4)) && (B * G >= 256 || (B * G >= 64 && (long)B * HW * C <= (1L << 19)) || (op.flags & 8)) && chunks <= 1024 * 16) {      // (... or a tensor of <= 2 MB on >= 64 workgroups: the 4x4 level of B >= 8)      // (measured: B = 32 eval 4.81 -> 4.66 ms; at 64 / 128 workgroups the pair wins: B = 8 2.46 vs 2.54 ms, B = 16 3.65 vs 3.68; flag 8 forces k_gn_one)
    float* s1 = (float*)op.p[0];
    const float *s2 = (const float*)op.p[1], *ga = (const float*)op.p[2], *be = (const float*)op.p[3], *ssp = (const float*)op.p[4];
    sf_opnd *o = (sf_opnd*)op.p[5], *rw = (sf_opnd*)op.p[6];
#define SF_GN_ONE(NT, NCH) k_gn_one<NT, NCH><<<B * G, NT, 0, st>>>(s1, s2, ga, be, ssp, o, rw, HW, C1, C2, op.i[4], op.f[0], op.f[1], op.flags & 1, lz, G)
    if (chunks <= 256 * 2) SF_GN_ONE(256, 2);
    else if (chunks <= 256 * 4) SF_GN_ONE(256, 4);
    else if (chunks <= 256 * 8) SF_GN_ONE(256, 8);
    else if (chunks <= 256 * 16) SF_GN_ONE(256, 16);
    else if (chunks <= 1024 * 8) SF_GN_ONE(1024, 8);
    else SF_GN_ONE(1024, 16);
#undef SF_GN_ONE
    SF_CHECK_LAUNCH("gn_one");
    return SF_OK;
  }
  if (op.flags & 2) {
    // the statistics are already in p[7] (k_gn_finalize over the producing conv's partial sums): no pass over the tensor
    if (op.p[1] || lz.mode) SF_FAIL(SF_ERR_INVALID, "gn_act: ready-made statistics need a plain single source");
  } else if (!op.p[1] && !lz.mode && C / G <= 16 && c4 <= 256 && 256 % c4 == 0) {
    const int ppi = 256 / c4;
    int slabs = HW / (ppi * 16);                       // >= 16 pixels per thread row before splitting further
    slabs = slabs < 1 ? 1 : (slabs > 1024 ? 1024 : slabs);
    k_gn_stats_px<<<B * slabs, 256, 0, st>>>((const float*)op.p[0], (double*)op.p[7], HW, C, G, slabs);
  } else {
    // a lazy split-K source multiplies the loads per chunk by `groups`: one chunk per thread then
    const int per_block = lz.mode == 1 ? 256 : GN_CHUNKS_PER_BLOCK;
    int slices = (chunks + per_block - 1) / per_block;
    // small maps: a (b, group) has too few chunks to fill its workgroup slice-wise -- aim for >= 128 workgroups in total,
    // down to one chunk per thread
    const int want = 128 / (B * G), max_slices = (chunks + 255) / 256;
    if (slices < want) slices = want < max_slices ? want : max_slices;
    if (slices < 1) slices = 1;
    k_gn_stats<<<B * G * slices, 256, 0, st>>>((float*)op.p[0], (const float*)op.p[1], (double*)op.p[7], HW, C1, C2, slices,
                                               op.f[1], lz, G);
  }
  SF_CHECK_LAUNCH("gn_stats");
  const long total = (long)B * HW * (C / 4);
  k_gn_apply<<<sf_grid_cap(sf_div_up(total, 256)), 256, 0, st>>>(
      (const float*)op.p[0], (const float*)op.p[1], (const float*)op.p[2], (const float*)op.p[3], (const float*)op.p[4],
      (const double*)op.p[7], (sf_opnd*)op.p[5], (sf_opnd*)op.p[6], B, HW, C1, C2, op.i[4], op.f[0], op.f[1], op.flags & 1, G);
  SF_CHECK_LAUNCH("gn_apply");
  return SF_OK;
}

static int run_ln(const sf_op& op, hipStream_t st) {
  const int R = op.i[0], C = op.i[1];
  if (C % 64 || C > 2048) SF_FAIL(SF_ERR_INVALID, "layernorm: C must be a multiple of 64 and <= 2048");
  if (C == 256 && R >= 1024) {                               // many short rows (EFT): one wave per row
    k_layernorm_w256<<<sf_div_up(R, 4), 256, 0, st>>>((const float*)op.p[0], (const float*)op.p[1], (const float*)op.p[2], op.p[3],
                                                      (const float*)op.p[4], R, op.f[0], op.flags & 1, (op.flags & 2) ? 1 : 0, (sf_opnd*)op.p[5]);
    SF_CHECK_LAUNCH("layernorm_w256");
    return SF_OK;
  }
  if (op.p[5]) SF_FAIL(SF_ERR_INVALID, "layernorm: the operand-type twin output exists for >= 1024 rows of 256 channels only");
  if (R <= 256 && (C == 512 || C == 1024 || C == 2048) && !(op.flags & 4) &&       // r05: few long rows (flag 4: keep k_layernorm, parity tests)
      !(((uintptr_t)op.p[0] | (uintptr_t)op.p[1] | (uintptr_t)op.p[2] | (uintptr_t)op.p[3] | (uintptr_t)op.p[4]) & 15)) {
    const uint32_t grid = sf_div_up(R, 4);
    if (C == 512) k_layernorm_wave<2><<<grid, 256, 0, st>>>((const float*)op.p[0], (const float*)op.p[1], (const float*)op.p[2], op.p[3], (const float*)op.p[4], R, op.f[0], op.flags & 1, (op.flags & 2) ? 1 : 0);
    else if (C == 1024) k_layernorm_wave<4><<<grid, 256, 0, st>>>((const float*)op.p[0], (const float*)op.p[1], (const float*)op.p[2], op.p[3], (const float*)op.p[4], R, op.f[0], op.flags & 1, (op.flags & 2) ? 1 : 0);
    else k_layernorm_wave<8><<<grid, 256, 0, st>>>((const float*)op.p[0], (const float*)op.p[1], (const float*)op.p[2], op.p[3], (const float*)op.p[4], R, op.f[0], op.flags & 1, (op.flags & 2) ? 1 : 0);
    SF_CHECK_LAUNCH("layernorm_wave");
    return SF_OK;
  }
  k_layernorm<<<R, 256, 0, st>>>((const float*)op.p[0], (const float*)op.p[1], (const float*)op.p[2], op.p[3],
                                              (const float*)op.p[4], R, C, op.f[0], op.flags & 1, (op.flags & 2) ? 1 : 0);
  SF_CHECK_LAUNCH("layernorm");
  return SF_OK;
}

static int run_gemv(const sf_op& op, hipStream_t st) {
  const int M = op.i[0], N = op.i[1];
  if (M > 64) SF_FAIL(SF_ERR_INVALID, "gemv: at most 64 rows");
  // (r06: from 8 rows on -- the GlobalContext MLPs of a B = 8 hybrid block ran on k_gemv at 12.9 us per launch, 18 launches per eval -- the
  // K-sliced MFMA form takes the op when N <= 4096; flag 8 keeps k_gemv (<= 8 rows) / k_gemm_rows (more))
  if (M > 8 || (M == 8 && N <= 4096 && !(op.flags & 8))) {     // many rows (a sampler's time table): rows on the MFMA M side, weights read once
    GemmRowsArgs a{(const float*)op.p[0], (const sf_opnd*)op.p[1], (const float*)op.p[2], (float*)op.p[3], M, N, op.i[2], op.i[3], op.i[4],
                   op.i[5], (int)(op.flags & 1), (int)((op.flags >> 1) & 3)};
    if (a.Kp < 8 || a.Kp % 8) SF_FAIL(SF_ERR_INVALID, "gemv: padded K must be a multiple of 8");
    if (N <= 4096 && !(op.flags & 8)) {                         // r06: the GlobalContext MLPs of a large batch -- N / 16 workgroups, K split over the waves (flag 8: the first form)
      const bool w8 = M <= 32 && a.K > 4 * GR_KC;                // 8 waves: every wave has ONE chunk of a 1024-column K (the slices hold 32 rows)
      static unsigned mask = 0;
      int dev = 0;
      if (hipGetDevice(&dev) != hipSuccess) SF_FAIL(SF_ERR_LAUNCH, "hipGetDevice failed");
      if (dev >= 32 || !(mask & (1u << dev))) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_rows_ks<4>), hipFuncAttributeMaxDynamicSharedMemorySize, GemmRowsKs<4>::LDS_BYTES) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_rows_ks<8>), hipFuncAttributeMaxDynamicSharedMemorySize, GemmRowsKs<8>::LDS_BYTES) != hipSuccess)
          SF_FAIL(SF_ERR_LAUNCH, "gemv: cannot raise the dynamic LDS limit of k_gemm_rows_ks");
        if (dev < 32) mask |= 1u << dev;
      }
      if (w8) k_gemm_rows_ks<8><<<sf_div_up(N, 16), 512, GemmRowsKs<8>::LDS_BYTES, st>>>(a);
      else k_gemm_rows_ks<4><<<sf_div_up(N, 16), 256, GemmRowsKs<4>::LDS_BYTES, st>>>(a);
      SF_CHECK_LAUNCH("gemm_rows_ks");
      return SF_OK;
    }
    k_gemm_rows<<<sf_div_up(N, 64), 256, 0, st>>>(a);
    SF_CHECK_LAUNCH("gemm_rows");
    return SF_OK;
  }
  if (N <= 2048)
    k_gemv<1><<<sf_div_up(N, 4), 256, 0, st>>>((const float*)op.p[0], (const sf_opnd*)op.p[1], (const float*)op.p[2], (float*)op.p[3], M,
                                              N, op.i[2], op.i[3], op.i[4], op.i[5], op.flags & 1, (op.flags >> 1) & 3);
  else
    k_gemv<4><<<sf_div_up(N, 16), 256, 0, st>>>((const float*)op.p[0], (const sf_opnd*)op.p[1], (const float*)op.p[2], (float*)op.p[3], M,
                                               N, op.i[2], op.i[3], op.i[4], op.i[5], op.flags & 1, (op.flags >> 1) & 3);
  SF_CHECK_LAUNCH("gemv");
  return SF_OK;
}

static int run_attn(const sf_op& op, hipStream_t st) {
  AttnSeg s[3];
  int J = 0;
  for (int k = 0; k < 3; ++k) {
    s[k].k = (const float*)op.p[2 + 2 * k]; s[k].v = (const float*)op.p[3 + 2 * k];
    s[k].rows = op.i[4 + 4 * k]; s[k].row_stride = op.i[5 + 4 * k]; s[k].batch_stride = op.i[6 + 4 * k];
    s[k].head_stride = op.i[7 + 4 * k];
    J += s[k].rows;
  }
  if (J < 1 || J > 24) SF_FAIL(SF_ERR_INVALID, "attn: 1..24 keys");
  k_attn16<<<op.i[0] * op.i[1], 256, 0, st>>>((const float*)op.p[0], op.p[1], s[0], s[1], s[2], op.i[1], op.i[2], op.f[0], op.flags & 1);
  SF_CHECK_LAUNCH("attn16");
  return SF_OK;
}

static int run_gca_pool(const sf_op& op, hipStream_t st) {
  const int B = op.i[0], HW = op.i[1], C = op.i[2];
  if (HW > 1024 || C % 32 || !op.p[4]) SF_FAIL(SF_ERR_INVALID, "gca_pool: HW <= 1024, C %% 32 == 0, logits scratch required");
  LazySrc lz;
  if (int rc = lazy_from_op(op, 3, B * HW, lz)) return rc;
  if (lz.mode == 2) SF_FAIL(SF_ERR_INVALID, "gca_pool: only split-K lazy sources");
  if (B * HW <= 256)
    k_gca_logits_wg<<<B * HW, 256, 0, st>>>((float*)op.p[0], (const float*)op.p[1], (const float*)op.p[2], (float*)op.p[4], C, lz);
  else
    k_gca_logits<<<sf_div_up(B * HW, 4), 256, 0, st>>>((float*)op.p[0], (const float*)op.p[1], (const float*)op.p[2],
                                                      (float*)op.p[4], B * HW, C, lz);
  SF_CHECK_LAUNCH("gca_logits");
  const int chunks = (HW + 31) / 32;
  const int csplit = chunks < 8 ? (C + 255) / 256 : 1;
  k_gca_pool<<<B * chunks * csplit, 256, 0, st>>>((const float*)op.p[0], (const float*)op.p[4], (float*)op.p[3], HW, C, chunks,
                                                  csplit);
  SF_CHECK_LAUNCH("gca_pool");
  return SF_OK;
}

static int run_eltwise(const sf_op& op, hipStream_t st) {
  switch (op.flags) {
    case 1: {
      const long n4 = (long)op.i[0] * op.i[1] * op.i[2] / 4;
      k_gate_res<<<sf_grid_cap(sf_div_up(n4, 256)), 256, 0, st>>>((const float*)op.p[0], (const float*)op.p[1],
                                                                 (const float*)op.p[2], (float*)op.p[3], op.i[0], op.i[1], op.i[2]);
      break;
    }
    case 2: {
      const long n = (long)op.i[0] * op.i[1] * op.i[4];
      k_pack_in<<<sf_grid_cap(sf_div_up(n, 256)), 256, 0, st>>>((const float*)op.p[0], (const float*)op.p[1], (float*)op.p[3],
                                                               op.i[0], op.i[1], op.i[2], op.i[3], op.i[4]);
      break;
    }
    case 3: {
      const long n = (long)op.i[0] * op.i[1] * op.i[2];
      k_unpack_out<<<sf_grid_cap(sf_div_up(n, 256)), 256, 0, st>>>((const float*)op.p[0], (float*)op.p[3], op.i[0], op.i[1],
                                                                  op.i[2], op.i[3]);
      break;
    }
    case 4: {
      const long n = (long)(uint32_t)op.i[0];
      k_add<<<sf_grid_cap(sf_div_up(n, 256)), 256, 0, st>>>((const float*)op.p[0], (float*)op.p[3], n);
      break;
    }
    case 5: {
      const long total = (long)((op.i[0] + 15) / 16) * ((op.i[1] + 31) / 32) * 64;
      k_pack_act<<<sf_grid_cap(sf_div_up(total, 256)), 256, 0, st>>>((const float*)op.p[0], (bf16x8*)op.p[3], op.i[0], op.i[1],
                                                                   op.i[2], op.i[3]);
      break;
    }
    case 6:
      k_softmax_rows<<<op.i[0], 256, 0, st>>>((const float*)op.p[0], (sf_opnd*)op.p[3], op.i[1], op.f[0]);
      break;
    case 7: case 8: case 9: return sf_plan_extra_op(&op, st);      // LPIPS helpers (lpips_ops.hip)
    default: SF_FAIL(SF_ERR_INVALID, "eltwise: unknown mode %d", op.flags);
  }
  SF_CHECK_LAUNCH("eltwise");
  return SF_OK;
}


static int plan_run_impl(const sf_op* ops, uint32_t n_ops, hipStream_t st_main, hipEvent_t* ev) {
  for (uint32_t k = 0; k < n_ops; ++k) {
    const sf_op& op = ops[k];
    int rc = SF_OK;
    hipStream_t st = st_main;
    if (ev && hipEventRecord(ev[k], st) != hipSuccess) SF_FAIL(SF_ERR_LAUNCH, "plan: hipEventRecord failed");
    switch (op.type) {
      case SF_OP_CONV: rc = run_conv(op, st); break;
      case SF_OP_GN_ACT: rc = run_gn(op, st); break;
      case SF_OP_LN: rc = run_ln(op, st); break;
      case SF_OP_GEMV: rc = run_gemv(op, st); break;
      case SF_OP_ATTN: rc = run_attn(op, st); break;
      case SF_OP_GCA_POOL: rc = run_gca_pool(op, st); break;
      case SF_OP_ELTWISE: rc = run_eltwise(op, st); break;
      case SF_OP_MEMSET:
        if (hipMemsetAsync(op.p[0], 0, (size_t)(uint32_t)op.i[0] * 4, st) != hipSuccess) SF_FAIL(SF_ERR_LAUNCH, "memset failed");
        break;
      case SF_OP_SPLITK_REDUCE: {
        const int M = op.i[0], Cout = op.i[1];
        if (!op.p[0] || !op.p[3] || op.i[3] < 1) SF_FAIL(SF_ERR_INVALID, "splitk_reduce: bad operands");
        k_splitk_reduce<<<sf_grid_cap(sf_div_up((long)M * Cout, 256)), 256, 0, st>>>(
            (const float*)op.p[0], (const float*)op.p[1], (const float*)op.p[2], (float*)op.p[3], M, Cout, op.i[2], op.i[3], Cout, 0, 0, 0, 0);
        if (hipGetLastError() != hipSuccess) SF_FAIL(SF_ERR_LAUNCH, "splitk_reduce launch failed");
        break;
      }
      case SF_OP_TIME_EMB:
        k_time_emb<<<sf_div_up(op.i[0] * op.i[1], 64), 64, 0, st>>>((const float*)op.p[0], (const float*)op.p[1], (float*)op.p[3],
                                                                  op.i[0], op.i[1]);
        if (hipGetLastError() != hipSuccess) SF_FAIL(SF_ERR_LAUNCH, "time_emb launch failed");
        break;
      case SF_OP_POOL:
      case SF_OP_LPIPS: rc = sf_plan_extra_op(&op, st); break;
      case SF_OP_EFT: rc = sf_plan_eft_op(&op, st); break;
      case SF_OP_INITX: rc = sf_plan_initx_op(&op, st); break;
      case SF_OP_GN_FINALIZE: {                            // p 0 partials  1 stats ; i 0 B  1 tiles per image  2 groups
        const int G = op.i[2];
        if (!op.p[0] || !op.p[1] || op.i[0] < 1 || op.i[1] < 1 || G < 1 || G > 256 || 256 % G)
          SF_FAIL(SF_ERR_INVALID, "gn_finalize: bad operands");
        k_gn_finalize<<<op.i[0] * G, 256, 0, st>>>((const double*)op.p[0], (double*)op.p[1], op.i[1], G);
        SF_CHECK_LAUNCH("gn_finalize");
        rc = SF_OK;
        break;
      }
      case SF_OP_FCONV:
        if (op.flags & 16) {                           // conv1 || res_conv of a ResnetBlock: one launch for this op and the next
          if (k + 1 >= n_ops) SF_FAIL(SF_ERR_INVALID, "plan: a paired fconv needs a successor");
          rc = sf_plan_fused_pair(&op, &ops[k + 1], st);
          if (!rc) {
            ++k;
            if (ev && hipEventRecord(ev[k], st) != hipSuccess) SF_FAIL(SF_ERR_LAUNCH, "plan: hipEventRecord failed");
          }
          break;
        }
        rc = sf_plan_fused_op(&op, st);
        break;
      case SF_OP_SLOTS:
      case SF_OP_GCA: rc = sf_plan_fused_op(&op, st); break;
      default: SF_FAIL(SF_ERR_INVALID, "plan: unknown op type %d at %u", op.type, k);
    }
    if (rc) {
      char tmp[400];
      snprintf(tmp, sizeof(tmp), "%s", sf_err_buf);
      snprintf(sf_err_buf, sizeof(sf_err_buf), "plan op %u (type %d): %s", k, op.type, tmp);
      return rc;
    }
  }
  if (ev && hipEventRecord(ev[n_ops], st_main) != hipSuccess) SF_FAIL(SF_ERR_LAUNCH, "plan: hipEventRecord failed");
  return SF_OK;
}

extern "C" int sf_plan_run(const sf_op* ops, uint32_t n_ops, void* stream) {
  return plan_run_impl(ops, n_ops, (hipStream_t)stream, nullptr);
}

// Same as sf_plan_run with a HIP event recorded before every op ON THE LAUNCH STREAM; h_ms[k] receives the
// elapsed milliseconds of op k (synchronises).  Used by bench.py for per-kernel roofline numbers.
extern "C" int sf_plan_profile(const sf_op* ops, uint32_t n_ops, void* stream, float* h_ms) {
  hipStream_t st = (hipStream_t)stream;
  hipEvent_t* ev = new hipEvent_t[n_ops + 1];
  uint32_t made = 0;
  int rc = SF_OK;
  for (; made <= n_ops; ++made)
    if (hipEventCreate(&ev[made]) != hipSuccess) { rc = SF_ERR_LAUNCH; break; }
  if (rc == SF_OK) rc = plan_run_impl(ops, n_ops, st, ev);
  else snprintf(sf_err_buf, sizeof(sf_err_buf), "plan_profile: hipEventCreate failed");
  if (rc == SF_OK && hipEventSynchronize(ev[n_ops]) != hipSuccess) rc = SF_ERR_LAUNCH;
  for (uint32_t k = 0; k < n_ops && rc == SF_OK; ++k)
    if (hipEventElapsedTime(&h_ms[k], ev[k], ev[k + 1]) != hipSuccess) rc = SF_ERR_LAUNCH;
  for (uint32_t k = 0; k < made; ++k) (void)hipEventDestroy(ev[k]);
  delete[] ev;
  return rc;
}

// ---------------------------------------------------------------------------------------------
// host-side weight packing (round-to-nearest-even bf16)
// ---------------------------------------------------------------------------------------------
#if SF_OPERAND_F16
static inline uint16_t f32_to_bf16_rne(float f) {            // operand = IEEE half in this build: the compiler's round-to-nearest-even
  const _Float16 h = (_Float16)f;
  uint16_t u;
  memcpy(&u, &h, 2);
  return u;
}
#else
static inline uint16_t f32_to_bf16_rne(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
#endif

extern "C" uint64_t sf_conv_packed_elems(uint32_t Cout, uint32_t cin_pad, uint32_t kh, uint32_t kw) {
  const uint64_t nfr = (Cout + 15) / 16;
  return nfr * kh * kw * (cin_pad / 32) * 64 * 8;
}

extern "C" int sf_conv_pack_weights(const float* h_w, uint32_t Cout, uint32_t Cin, uint32_t cin_pad, uint32_t kh,
                                    uint32_t kw, uint16_t* h_out) {
  if (cin_pad % 32 || cin_pad < Cin) SF_FAIL(SF_ERR_INVALID, "pack: cin_pad must be a multiple of 32 and >= Cin");
  const uint32_t nfr = (Cout + 15) / 16, cch = cin_pad / 32, taps = kh * kw;
  const uint64_t KS = (uint64_t)taps * cch;
  for (uint32_t nf = 0; nf < nfr; ++nf)
    for (uint32_t tap = 0; tap < taps; ++tap)
      for (uint32_t cc = 0; cc < cch; ++cc)
        for (uint32_t lane = 0; lane < 64; ++lane) {
          const uint32_t n = nf * 16 + (lane & 15);
          uint16_t* dst = h_out + (((uint64_t)nf * KS + (uint64_t)tap * cch + cc) * 64 + lane) * 8;
          for (uint32_t j = 0; j < 8; ++j) {
            const uint32_t c = cc * 32 + 8 * (lane >> 4) + j;
            float v = 0.0f;
            if (n < Cout && c < Cin) v = h_w[((uint64_t)n * Cin + c) * taps + tap];     // [Cout][Cin][kh][kw]
            dst[j] = f32_to_bf16_rne(v);
          }
        }
  return SF_OK;
}

// ---------------------------------------------------------------------------------------------
// PLMS latent updates (external/plms.py:122-214, imagen_pytorch.py:242-297)
// ---------------------------------------------------------------------------------------------
struct Coef6 { float alpha, sigma, alpha_next, c, noise_scale, clip; };
// x and x_prev may ALIAS (plms.py keeps the latents in place in the plan's input buffer): neither is __restrict__, and every
// element is read before it is written by the same thread.
__global__ __launch_bounds__(256) void k_plms_update(const float* x, const float* __restrict__ eps,
                                                     const float* __restrict__ noise, Coef6 k, long n,
                                                     float* x_prev, float* __restrict__ x0) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float xv = x[i];
    float s = __fdiv_rn(__fsub_rn(xv, __fmul_rn(k.sigma, eps[i])), fmaxf(k.alpha, 1e-8f));   // predict_start_from_noise
    s = fminf(fmaxf(s, -k.clip), k.clip);
    // q_posterior mean: alpha_next * (x_t * (1 - c) / alpha + c * x_start)
    const float mean = __fmul_rn(k.alpha_next, __fadd_rn(__fdiv_rn(__fmul_rn(xv, __fsub_rn(1.0f, k.c)), k.alpha), __fmul_rn(k.c, s)));
    float out = mean;
    if (noise) out = __fadd_rn(mean, __fmul_rn(k.noise_scale, noise[i]));
    x_prev[i] = out;
    if (x0) x0[i] = s;
  }
}
__global__ __launch_bounds__(256) void k_plms_combine(const float* __restrict__ e0, const float* __restrict__ e1,
                                                      const float* __restrict__ e2, const float* __restrict__ e3, float c0,
                                                      float c1, float c2, float c3, long n, float* __restrict__ out,
                                                      float* __restrict__ keep) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    if (keep) keep[i] = e0[i];                   // the eps history entry of this step (saves the sampler a copy launch)
    float v = c0 * e0[i];
    if (e1) v += c1 * e1[i];
    if (e2) v += c2 * e2[i];
    if (e3) v += c3 * e3[i];
    out[i] = v;
  }
}

// combine + update in ONE launch (the steady state of the sampler: every step after the first)
__global__ __launch_bounds__(256) void k_plms_step(const float* __restrict__ e0, const float* __restrict__ e1,
                                                   const float* __restrict__ e2, const float* __restrict__ e3, float c0, float c1,
                                                   float c2, float c3, float* __restrict__ keep, const float* x,
                                                   const float* __restrict__ noise, Coef6 k, long n, float* x_prev,   // x may alias x_prev
                                                   int step_blocks, const float* __restrict__ row_src, float* __restrict__ row_dst, long row_n4) {
  if ((int)blockIdx.x >= step_blocks) {          // the extra workgroups: the NEXT eval's time-block row into the plan (r04: was a copy launch per eval)
    const long nb = gridDim.x - step_blocks;
    for (long i = (blockIdx.x - step_blocks) * 256L + threadIdx.x; i < row_n4; i += nb * 256)
      reinterpret_cast<float4*>(row_dst)[i] = reinterpret_cast<const float4*>(row_src)[i];
    return;
  }
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)step_blocks * 256) {
    if (keep) keep[i] = e0[i];
    float v = c0 * e0[i];                        // same expression order as k_plms_combine
    if (e1) v += c1 * e1[i];
    if (e2) v += c2 * e2[i];
    if (e3) v += c3 * e3[i];
    const float xv = x[i];
    float s = __fdiv_rn(__fsub_rn(xv, __fmul_rn(k.sigma, v)), fmaxf(k.alpha, 1e-8f));
    s = fminf(fmaxf(s, -k.clip), k.clip);
    const float mean = __fmul_rn(k.alpha_next, __fadd_rn(__fdiv_rn(__fmul_rn(xv, __fsub_rn(1.0f, k.c)), k.alpha), __fmul_rn(k.c, s)));
    float out = mean;
    if (noise) out = __fadd_rn(mean, __fmul_rn(k.noise_scale, noise[i]));
    x_prev[i] = out;
  }
}

extern "C" int sf_plms_step(const float* e0, const float* e1, const float* e2, const float* e3, const float* h_c4, float* keep_e0,
                            const float* x, const float* noise, const float* h_coef6, uint64_t n, float* x_prev,
                            const float* row_src, float* row_dst, uint64_t row_n, void* stream) {
  if (!e0 || !h_c4 || !x || !h_coef6 || !x_prev) SF_FAIL(SF_ERR_INVALID, "plms_step: null argument");
  if (n == 0) return SF_OK;
  if (row_n && (!row_src || !row_dst || row_n % 4 || ((uintptr_t)row_src | (uintptr_t)row_dst) % 16)) SF_FAIL(SF_ERR_INVALID, "plms_step: the row copy wants 16-byte aligned rows of 4 n floats");
  Coef6 k{h_coef6[0], h_coef6[1], h_coef6[2], h_coef6[3], h_coef6[4], h_coef6[5]};
  const int sb = (int)sf_grid_cap(sf_div_up(n, 256));
  const int rb = row_n ? (int)sf_grid_cap(sf_div_up(row_n / 4, 256)) : 0;
  k_plms_step<<<sb + rb, 256, 0, (hipStream_t)stream>>>(e0, e1, e2, e3, h_c4[0], h_c4[1], h_c4[2], h_c4[3], keep_e0, x,
                                                        noise, k, (long)n, x_prev, sb, row_src, row_dst, (long)(row_n / 4));
  SF_CHECK_LAUNCH("plms_step");
  return SF_OK;
}

extern "C" int sf_plms_update(const float* x, const float* eps, const float* noise, const float* h_coef6, uint64_t n,
                              float* x_prev, float* x0, void* stream) {
  if (!x || !eps || !h_coef6 || !x_prev) SF_FAIL(SF_ERR_INVALID, "plms_update: null argument");
  if (n == 0) return SF_OK;
  Coef6 k{h_coef6[0], h_coef6[1], h_coef6[2], h_coef6[3], h_coef6[4], h_coef6[5]};
  k_plms_update<<<sf_grid_cap(sf_div_up(n, 256)), 256, 0, (hipStream_t)stream>>>(x, eps, noise, k, (long)n, x_prev, x0);
  SF_CHECK_LAUNCH("plms_update");
  return SF_OK;
}

extern "C" int sf_plms_combine(const float* e0, const float* e1, const float* e2, const float* e3, const float* h_c4,
                               uint64_t n, float* out, float* keep_e0, void* stream) {
  if (!e0 || !h_c4 || !out) SF_FAIL(SF_ERR_INVALID, "plms_combine: null argument");
  if (n == 0) return SF_OK;
  k_plms_combine<<<sf_grid_cap(sf_div_up(n, 256)), 256, 0, (hipStream_t)stream>>>(e0, e1, e2, e3, h_c4[0], h_c4[1], h_c4[2],
                                                                                  h_c4[3], (long)n, out, keep_e0);
  SF_CHECK_LAUNCH("plms_combine");
  return SF_OK;
}

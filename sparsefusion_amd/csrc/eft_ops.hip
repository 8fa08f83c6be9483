// Plan ops of the Epipolar Feature Transformer pre-pass (sparsefusion/eft.py).
// The GEMM-shaped work (resnet18 trunk convs with folded BatchNorm, every Linear of the three transformers) runs on
// the conv kernels of unet_ops.hip; this file holds the rest, all fp32, all HBM / latency bound:
//   SF_OP_EFT flags 0  RESIZE     bilinear, align_corners=True, NHWC -> channel slice of a wider NHWC map   eft.py:193-200
//              flags 1  GATHER     grid_sample(bilinear, border, align_corners=True) of the 512-ch pyramid and of the RGB
//                                  image at the projected sample points, written into the T1 input rows     eft.py:253-297
//              flags 2  HARMONIC   [sin(2^k x) | cos(2^k x) | x] embedding into a column range, with a row map that
//                                  broadcasts per-ray / per-sample quantities                               common_utils.py:145-155
//              flags 3  ATTN       single-head attention over short sequences (views or depths) that live as strided
//                                  rows of one [M, 768] q|k|v matrix (nn.TransformerEncoderLayer, seq-first)
//              flags 4  POOL       softmax(x.w + b) over the sequence, weighted sum (+ sigmoid colour head)  eft.py:425-449
//   SF_OP_POOL flags 2  3x3 stride-2 pad-1 max pooling (resnet stem)

#include "sf_common.h"
#include "plan_ops.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float eft_wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
__device__ __forceinline__ float eft_wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
  return v;
}

// ---- 3x3 / 2 max pooling --------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pool3_fwd(const float* __restrict__ in, float* __restrict__ out, int B, int H, int W,
                                                   int C, int Ho, int Wo) {
  const int c4 = C / 4;
  const long total = (long)B * Ho * Wo * c4;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % c4) * 4;
    long r = i / c4;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho);
    const int b = (int)(r / Ho);
    f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = 2 * oy - 1 + ky;
      if (iy < 0 || iy >= H) continue;
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = 2 * ox - 1 + kx;
        if (ix < 0 || ix >= W) continue;
        const f32x4 v = *reinterpret_cast<const f32x4*>(in + (((long)b * H + iy) * W + ix) * C + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) m[j] = fmaxf(m[j], v[j]);
      }
    }
    *reinterpret_cast<f32x4*>(out + (((long)b * Ho + oy) * Wo + ox) * C + c) = m;
  }
}

// ---- bilinear resize (align_corners) into a channel slice -----------------------------------------------------
__global__ __launch_bounds__(256) void k_resize_ac(const float* __restrict__ in, float* __restrict__ out, int B, int Hi, int Wi,
                                                   int C, int Ho, int Wo, int ldo, int co) {
  const int c4 = C / 4;
  const long total = (long)B * Ho * Wo * c4;
  const float rh = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.0f, rw = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.0f;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % c4) * 4;
    long r = i / c4;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho);
    const int b = (int)(r / Ho);
    const float h1r = rh * oy, w1r = rw * ox;
    const int h1 = (int)h1r, w1 = (int)w1r;
    const int h1p = h1 < Hi - 1 ? 1 : 0, w1p = w1 < Wi - 1 ? 1 : 0;
    const float h1l = h1r - h1, h0l = 1.0f - h1l, w1l = w1r - w1, w0l = 1.0f - w1l;
    const float* p = in + (((long)b * Hi + h1) * Wi + w1) * C + c;
    const f32x4 v00 = *reinterpret_cast<const f32x4*>(p), v01 = *reinterpret_cast<const f32x4*>(p + (long)w1p * C);
    const f32x4 v10 = *reinterpret_cast<const f32x4*>(p + (long)h1p * Wi * C);
    const f32x4 v11 = *reinterpret_cast<const f32x4*>(p + (long)h1p * Wi * C + (long)w1p * C);
    const f32x4 o = h0l * (w0l * v00 + w1l * v01) + h1l * (w0l * v10 + w1l * v11);
    *reinterpret_cast<f32x4*>(out + (((long)b * Ho + oy) * Wo + ox) * ldo + co + c) = o;
  }
}

// ---- grid_sample gather -----------------------------------------------------------------------------------------
// latent NHWC [NC, Hf, Wf, Cf], images NCHW [NC, 3, Hi, Wi], xy [NC, P, 2] (NDC of the sample points; the reference samples
// at -xy).  One wave per (camera, point): row (c*P + p) of `out` gets Cf feature channels at column co and 3 RGB after them.
__device__ __forceinline__ void gs_corners(float g, int size, int& i0, float& w0, float& w1) {
  float x = (g + 1.0f) * 0.5f * (float)(size - 1);            // align_corners=True unnormalisation
  x = fminf(fmaxf(x, 0.0f), (float)(size - 1));               // padding_mode='border'
  const float f = floorf(x);
  i0 = (int)f;
  w1 = x - f;
  w0 = 1.0f - w1;
}

__global__ __launch_bounds__(256) void k_grid_gather(const float* __restrict__ latent, const float* __restrict__ images,
                                                     const float* __restrict__ xy, float* __restrict__ out, int NC, long P, int Hf,
                                                     int Wf, int Cf, int Hi, int Wi, int ldo, int co) {
  const int lane = threadIdx.x & 63;
  const long rows = (long)NC * P;
  for (long q = blockIdx.x * 4L + (threadIdx.x >> 6); q < rows; q += (long)gridDim.x * 4) {
    const int c = (int)(q / P);
    const float gx = -xy[q * 2], gy = -xy[q * 2 + 1];
    int x0, y0;
    float wx0, wx1, wy0, wy1;
    gs_corners(gx, Wf, x0, wx0, wx1);
    gs_corners(gy, Hf, y0, wy0, wy1);
    const bool xin = x0 + 1 < Wf, yin = y0 + 1 < Hf;
    const float nw = wx0 * wy0, ne = wx1 * wy0, sw = wx0 * wy1, se = wx1 * wy1;
    const float* base = latent + (((long)c * Hf + y0) * Wf + x0) * Cf;
    float* o = out + q * ldo + co;
    for (int ch = lane * 4; ch < Cf; ch += 256) {
      f32x4 acc = nw * *reinterpret_cast<const f32x4*>(base + ch);
      if (xin) acc += ne * *reinterpret_cast<const f32x4*>(base + Cf + ch);
      if (yin) acc += sw * *reinterpret_cast<const f32x4*>(base + (long)Wf * Cf + ch);
      if (xin && yin) acc += se * *reinterpret_cast<const f32x4*>(base + (long)Wf * Cf + Cf + ch);
      o[ch] = acc[0]; o[ch + 1] = acc[1]; o[ch + 2] = acc[2]; o[ch + 3] = acc[3];      // row start is not 16-byte aligned
    }
    if (lane < 3) {                                            // RGB of the input view at the same NDC position
      int ix0, iy0;
      float ax0, ax1, ay0, ay1;
      gs_corners(gx, Wi, ix0, ax0, ax1);
      gs_corners(gy, Hi, iy0, ay0, ay1);
      const float* im = images + ((long)c * 3 + lane) * Hi * Wi;
      float v = ax0 * ay0 * im[(long)iy0 * Wi + ix0];
      if (ix0 + 1 < Wi) v += ax1 * ay0 * im[(long)iy0 * Wi + ix0 + 1];
      if (iy0 + 1 < Hi) v += ax0 * ay1 * im[(long)(iy0 + 1) * Wi + ix0];
      if (ix0 + 1 < Wi && iy0 + 1 < Hi) v += ax1 * ay1 * im[(long)(iy0 + 1) * Wi + ix0 + 1];
      o[Cf + lane] = v;
    }
  }
}

// ---- harmonic embedding -----------------------------------------------------------------------------------------
// out[r, co + ...] = [sin(2^k x_i) (i-major, k = 0..5) | cos(...) | x] of x = src[((r / div) % mod) * mul + add, 0..dim)
__global__ __launch_bounds__(256) void k_harmonic(const float* __restrict__ src, float* __restrict__ out, long rows, int dim, int ldo,
                                                  int co, long div, long mod, long mul, long add) {
  const long total = rows * dim;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / dim;
    const int d = (int)(i - r * dim);
    const long s = ((r / div) % mod) * mul + add;
    const float x = src[s * dim + d];
    float* o = out + r * ldo + co;
    float f = 1.0f;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const float e = x * f;
      o[d * 6 + k] = sinf(e);
      o[dim * 6 + d * 6 + k] = cosf(e);
      f *= 2.0f;
    }
    o[dim * 12 + d] = x;
  }
}

// ---- single-head attention over strided rows ----------------------------------------------------------------------
// qkv [M, 768] (q | k | v, E = 256).  Group g holds S rows: row(g, s) = g * gmul + s * stride.  out [M, 256].
#define EFT_E 256
#define EFT_SMAX 24
// out16 (r04): the result in MFMA operand type instead of fp32 -- its only consumer is the output projection, which rounds it anyway.
__global__ __launch_bounds__(256) void k_attn_small(const float* __restrict__ qkv, float* __restrict__ out, sf_opnd* __restrict__ out16, int S,
                                                    long stride, long gmul, float scale) {
  __shared__ float sq[EFT_SMAX][EFT_E], sk[EFT_SMAX][EFT_E];
  __shared__ float sp[EFT_SMAX][EFT_SMAX + 1];
  const long g = blockIdx.x;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int s = 0; s < S; ++s) {
    const float* row = qkv + (g * gmul + s * stride) * (3 * EFT_E);
    sq[s][threadIdx.x] = row[threadIdx.x];
    sk[s][threadIdx.x] = row[EFT_E + threadIdx.x];
  }
  __syncthreads();
  for (int pr = wv; pr < S * S; pr += 4) {                       // one wave per (query, key) pair
    const int s = pr / S, t = pr - s * S;
    const f32x4 a = *reinterpret_cast<const f32x4*>(&sq[s][lane * 4]), b = *reinterpret_cast<const f32x4*>(&sk[t][lane * 4]);
    const float d = eft_wave_sum(a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3]);
    if (lane == 0) sp[s][t] = d * scale;
  }
  __syncthreads();
  if ((int)threadIdx.x < S) {                                    // row softmax
    const int s = threadIdx.x;
    float m = -INFINITY;
    for (int t = 0; t < S; ++t) m = fmaxf(m, sp[s][t]);
    float sum = 0.0f;
    for (int t = 0; t < S; ++t) { const float e = expf(sp[s][t] - m); sp[s][t] = e; sum += e; }
    const float inv = 1.0f / sum;
    for (int t = 0; t < S; ++t) sp[s][t] *= inv;
  }
  __syncthreads();
  // thread e owns output channel e: out[s][e] = sum_t p[s][t] v[t][e]
  float acc[EFT_SMAX];
#pragma unroll
  for (int s = 0; s < EFT_SMAX; ++s) acc[s] = 0.0f;
  for (int t = 0; t < S; ++t) {
    const float v = qkv[(g * gmul + t * stride) * (3 * EFT_E) + 2 * EFT_E + threadIdx.x];
#pragma unroll
    for (int s = 0; s < EFT_SMAX; ++s)
      if (s < S) acc[s] = fmaf(sp[s][t], v, acc[s]);
  }
#pragma unroll
  for (int s = 0; s < EFT_SMAX; ++s)
    if (s < S) {
      if (out16) out16[(g * gmul + s * stride) * EFT_E + threadIdx.x] = (sf_opnd)acc[s];
      else out[(g * gmul + s * stride) * EFT_E + threadIdx.x] = acc[s];
    }
}

// ---- softmax pooling over the sequence (+ optional colour head) --------------------------------------------------
// x [M, 256]; group g: rows g * gmul + s * stride, s < S.  l_s = x_s . w + b ; p = softmax_s(l) ; out[g] = sum_s p_s x_s.
// head != NULL: rgb[g][j] = sigmoid(head_w[j] . out[g] + head_b[j]), j < 3.   One wave per group.
__global__ __launch_bounds__(256) void k_pool_softmax(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                      float* __restrict__ out, long G, int S, long stride, long gmul,
                                                      const float* __restrict__ head_w, const float* __restrict__ head_b,
                                                      float* __restrict__ rgb) {
  const int lane = threadIdx.x & 63;
  const f32x4 ww = *reinterpret_cast<const f32x4*>(w + lane * 4);
  for (long g = blockIdx.x * 4L + (threadIdx.x >> 6); g < G; g += (long)gridDim.x * 4) {
    float m = -INFINITY, sum = 0.0f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < S; ++s) {                                // online softmax: one pass over the rows
      const f32x4 v = *reinterpret_cast<const f32x4*>(x + (g * gmul + s * stride) * EFT_E + lane * 4);
      const float l = eft_wave_sum(v[0] * ww[0] + v[1] * ww[1] + v[2] * ww[2] + v[3] * ww[3]) + b[0];
      const float mn = fmaxf(m, l);
      const float c = expf(m - mn), e = expf(l - mn);
      acc = acc * c + e * v;
      sum = sum * c + e;
      m = mn;
    }
    acc = acc * (1.0f / sum);
    *reinterpret_cast<f32x4*>(out + g * EFT_E + lane * 4) = acc;
    if (head_w) {
      for (int j = 0; j < 3; ++j) {
        const f32x4 hw = *reinterpret_cast<const f32x4*>(head_w + j * EFT_E + lane * 4);
        const float d = eft_wave_sum(acc[0] * hw[0] + acc[1] * hw[1] + acc[2] * hw[2] + acc[3] * hw[3]) + head_b[j];
        if (lane == 0) rgb[g * 3 + j] = 1.0f / (1.0f + expf(-d));
      }
    }
  }
}

static inline long i64(const sf_op& op, int k) { return (long)(uint32_t)op.i[k] | ((long)op.i[k + 1] << 32); }

int sf_plan_eft_op(const sf_op* opp, void* stream) {
  const sf_op& op = *opp;
  hipStream_t st = (hipStream_t)stream;
  if (op.type == SF_OP_POOL) {                                   // flags 2: 3x3 stride 2 pad 1
    const int B = op.i[0], H = op.i[1], W = op.i[2], C = op.i[3];
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    if (C % 4 || !op.p[0] || !op.p[3]) SF_FAIL(SF_ERR_INVALID, "pool3: bad operands");
    k_pool3_fwd<<<sf_grid_cap(sf_div_up((long)B * Ho * Wo * (C / 4), 256)), 256, 0, st>>>((const float*)op.p[0], (float*)op.p[3], B, H,
                                                                                       W, C, Ho, Wo);
    SF_CHECK_LAUNCH("pool3");
    return SF_OK;
  }
  switch (op.flags) {
    case 0: {                                                    // RESIZE: i = B, Hi, Wi, C, Ho, Wo, ldo, co
      if (op.i[3] % 4 || op.i[6] % 4 || op.i[7] % 4 || !op.p[0] || !op.p[3]) SF_FAIL(SF_ERR_INVALID, "resize: bad operands");
      const long total = (long)op.i[0] * op.i[4] * op.i[5] * (op.i[3] / 4);
      k_resize_ac<<<sf_grid_cap(sf_div_up(total, 256)), 256, 0, st>>>((const float*)op.p[0], (float*)op.p[3], op.i[0], op.i[1], op.i[2],
                                                                     op.i[3], op.i[4], op.i[5], op.i[6], op.i[7]);
      break;
    }
    case 1: {                                                    // GATHER: i = NC, P(lo,hi), Hf, Wf, Cf, Hi, Wi, ldo, co
      const long P = i64(op, 1);
      if (op.i[5] % 4 || !op.p[0] || !op.p[1] || !op.p[2] || !op.p[3]) SF_FAIL(SF_ERR_INVALID, "gather: bad operands");
      k_grid_gather<<<sf_grid_cap(sf_div_up((uint64_t)op.i[0] * P, 4)), 256, 0, st>>>(
          (const float*)op.p[0], (const float*)op.p[1], (const float*)op.p[2], (float*)op.p[3], op.i[0], P, op.i[3], op.i[4], op.i[5],
          op.i[6], op.i[7], op.i[8], op.i[9]);
      break;
    }
    case 2: {                                                    // HARMONIC: i = rows(2), dim, ldo, co, div(2), mod(2), mul(2), add(2)
      const long rows = i64(op, 0);
      if (!op.p[0] || !op.p[3] || op.i[2] < 1) SF_FAIL(SF_ERR_INVALID, "harmonic: bad operands");
      k_harmonic<<<sf_grid_cap(sf_div_up((uint64_t)rows * op.i[2], 256)), 256, 0, st>>>(
          (const float*)op.p[0], (float*)op.p[3], rows, op.i[2], op.i[3], op.i[4], i64(op, 5), i64(op, 7), i64(op, 9), i64(op, 11));
      break;
    }
    case 3: {                                                    // ATTN: i = groups(2), S, stride(2), gmul(2) ; f[0] = scale
      const long G = i64(op, 0);
      if (op.i[2] < 1 || op.i[2] > EFT_SMAX || !op.p[0] || (!op.p[3] && !op.p[4])) SF_FAIL(SF_ERR_INVALID, "attn_small: sequence length 1..%d", EFT_SMAX);
      if (G > 0x7fffffffL) SF_FAIL(SF_ERR_INVALID, "attn_small: too many groups");
      k_attn_small<<<(uint32_t)G, 256, 0, st>>>((const float*)op.p[0], (float*)op.p[3], (sf_opnd*)op.p[4], op.i[2], i64(op, 3), i64(op, 5), op.f[0]);      // p[4]: operand-type output instead of p[3]
      break;
    }
    case 4: {                                                    // POOL: i = groups(2), S, stride(2), gmul(2)
      const long G = i64(op, 0);
      if (op.i[2] < 1 || !op.p[0] || !op.p[1] || !op.p[2] || !op.p[3]) SF_FAIL(SF_ERR_INVALID, "pool_softmax: bad operands");
      k_pool_softmax<<<sf_grid_cap(sf_div_up((uint64_t)G, 4)), 256, 0, st>>>((const float*)op.p[0], (const float*)op.p[1],
                                                                          (const float*)op.p[2], (float*)op.p[3], G, op.i[2],
                                                                          i64(op, 3), i64(op, 5), (const float*)op.p[4],
                                                                          (const float*)op.p[5], (float*)op.p[6]);
      break;
    }
    default: SF_FAIL(SF_ERR_INVALID, "eft: unknown sub-op %d", op.flags);
  }
  SF_CHECK_LAUNCH("eft_op");
  return SF_OK;
}

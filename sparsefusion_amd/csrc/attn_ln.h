// k_layernorm and k_attn16 (operand contracts: the comment blocks in unet_ops.hip).  Written against sf_dev.h so that
// tests/hostemu runs the same source on CPU threads (tests/test_hostemu_attn_ln.py).
#pragma once
#include "sf_dev.h"

SF_KERNEL(256) void k_layernorm(const float* __restrict__ in, const float* __restrict__ gain,
                                                   const float* __restrict__ bias, void* __restrict__ out,
                                                   const float* __restrict__ resid, int R, int C, float eps, int pre_gelu,
                                                   int out_f32) {
  SF_SHARED float red[8];
  const int row = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const float* x = in + (long)row * C;
  float v[8];                                   // C <= 2048 = 8 * 256
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = threadIdx.x + i * 256;
    v[i] = 0.0f;
    if (c < C) { const float t = x[c]; v[i] = pre_gelu ? sf_gelu(t) : t; s += v[i]; }
  }
  s = sf_wave_sum(s);
  if (lane == 0) red[wv] = s;
  sf_sync();
  const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)C;
  float q = 0.0f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (threadIdx.x + i * 256 < C) { const float d = v[i] - mean; q = fmaf(d, d, q); }
  q = sf_wave_sum(q);
  if (lane == 0) red[4 + wv] = q;
  sf_sync();
  const float rstd = sf_rsqrt((red[4] + red[5] + red[6] + red[7]) / (float)C + eps);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = threadIdx.x + i * 256;
    if (c < C) {
      float y = (v[i] - mean) * rstd * gain[c];
      if (bias) y += bias[c];
      if (out_f32) {
        if (resid) y += resid[(long)row * C + c];
        reinterpret_cast<float*>(out)[(long)row * C + c] = y;
      } else {
        reinterpret_cast<sf_opnd*>(out)[(long)row * C + c] = (sf_opnd)y;
      }
    }
  }
}

// Few long rows (r05; the post-attention LayerNorm + residual of the UNet's transformer blocks: 16 rows of 1024 channels, one launch per
// attention): one WAVE per row, NV float4 per lane (C = 256 NV), the row, the gain (bias) and the residual requested in the first
// instructions, two shuffle reductions, no block barrier, no load behind a branch.  k_layernorm above fetches scalars under `c < C`,
// the gain after the statistics, and crosses two barriers: 5.2 us per launch in the eval graph.  Same arithmetic (two-pass variance, fp32).
template <int NV>
SF_KERNEL(256) void k_layernorm_wave(const float* __restrict__ in, const float* __restrict__ gain, const float* __restrict__ bias,
                                     void* __restrict__ out, const float* __restrict__ resid, int R, float eps, int pre_gelu, int out_f32) {
  constexpr int C = 256 * NV;
  const int lane = threadIdx.x & 63;
  long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const bool live = row < R;
  if (!live) row = R - 1;                                  // (clamped: every wave runs the same instruction stream)
  f32x4 v[NV], g[NV], bq[NV], rq[NV];
  const float* x = in + row * C;
  const float* bp = bias ? bias : gain;
  const float* rp = resid ? resid + row * C : x;
#pragma unroll
  for (int u = 0; u < NV; ++u) v[u] = *reinterpret_cast<const f32x4*>(x + (lane + 64 * u) * 4);
#pragma unroll
  for (int u = 0; u < NV; ++u) g[u] = *reinterpret_cast<const f32x4*>(gain + (lane + 64 * u) * 4);
#pragma unroll
  for (int u = 0; u < NV; ++u) bq[u] = *reinterpret_cast<const f32x4*>(bp + (lane + 64 * u) * 4);
#pragma unroll
  for (int u = 0; u < NV; ++u) rq[u] = *reinterpret_cast<const f32x4*>(rp + (lane + 64 * u) * 4);
  float s = 0.0f;
#pragma unroll
  for (int u = 0; u < NV; ++u) {
    if (pre_gelu) { v[u][0] = sf_gelu(v[u][0]); v[u][1] = sf_gelu(v[u][1]); v[u][2] = sf_gelu(v[u][2]); v[u][3] = sf_gelu(v[u][3]); }
    s += (v[u][0] + v[u][1]) + (v[u][2] + v[u][3]);
  }
  const float mean = sf_wave_sum(s) / (float)C;
  float q = 0.0f;
#pragma unroll
  for (int u = 0; u < NV; ++u)
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float d = v[u][j] - mean; q = fmaf(d, d, q); }
  const float rstd = sf_rsqrt(sf_wave_sum(q) / (float)C + eps);
  if (!live) return;
#pragma unroll
  for (int u = 0; u < NV; ++u) {
    f32x4 y;
#pragma unroll
    for (int j = 0; j < 4; ++j) y[j] = (v[u][j] - mean) * rstd * g[u][j];
    if (bias) y += bq[u];
    if (out_f32) {
      if (resid) y += rq[u];
      *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out) + row * C + (lane + 64 * u) * 4) = y;
    } else {
      bf16x4 o;
      o[0] = (sf_opnd)y[0]; o[1] = (sf_opnd)y[1]; o[2] = (sf_opnd)y[2]; o[3] = (sf_opnd)y[3];
      *reinterpret_cast<bf16x4*>(reinterpret_cast<sf_opnd*>(out) + row * C + (lane + 64 * u) * 4) = o;
    }
  }
}

// Many short rows (the EFT transformers: 122 880 rows of 256 channels): one WAVE per row, four channels per lane, no block barrier --
// the block-per-row kernel above spends its time in two barriers and one element per thread (91 us for 252 MB moved, r04 trace); this one
// is one load, two shuffle reductions and one store per lane.  Same arithmetic (two-pass variance, fp32).  C == 256 only.
SF_KERNEL(256) void k_layernorm_w256(const float* __restrict__ in, const float* __restrict__ gain, const float* __restrict__ bias,
                                     void* __restrict__ out, const float* __restrict__ resid, int R, float eps, int pre_gelu, int out_f32,
                                     sf_opnd* __restrict__ twin) {      // twin: also an operand-type copy of the fp32 output (the next linear's A operand)
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  f32x4 v = *reinterpret_cast<const f32x4*>(in + row * 256 + lane * 4);
  const f32x4 g = *reinterpret_cast<const f32x4*>(gain + lane * 4);
  const f32x4 bq = bias ? *reinterpret_cast<const f32x4*>(bias + lane * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x4 rq = (resid && out_f32) ? *reinterpret_cast<const f32x4*>(resid + row * 256 + lane * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
  if (pre_gelu) { v[0] = sf_gelu(v[0]); v[1] = sf_gelu(v[1]); v[2] = sf_gelu(v[2]); v[3] = sf_gelu(v[3]); }
  const float mean = sf_wave_sum((v[0] + v[1]) + (v[2] + v[3])) / 256.0f;
  const f32x4 d = v - mean;
  const float rstd = sf_rsqrt(sf_wave_sum(fmaf(d[0], d[0], fmaf(d[1], d[1], fmaf(d[2], d[2], d[3] * d[3])))) / 256.0f + eps);
  f32x4 y;
#pragma unroll
  for (int j = 0; j < 4; ++j) y[j] = d[j] * rstd * g[j] + bq[j];
  if (out_f32) {
    y = y + rq;
    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out) + row * 256 + lane * 4) = y;
    if (twin) {
      bf16x4 o;
      o[0] = (sf_opnd)y[0]; o[1] = (sf_opnd)y[1]; o[2] = (sf_opnd)y[2]; o[3] = (sf_opnd)y[3];
      *reinterpret_cast<bf16x4*>(twin + row * 256 + lane * 4) = o;
    }
  } else {
    bf16x4 o;
    o[0] = (sf_opnd)y[0]; o[1] = (sf_opnd)y[1]; o[2] = (sf_opnd)y[2]; o[3] = (sf_opnd)y[3];
    *reinterpret_cast<bf16x4*>(reinterpret_cast<sf_opnd*>(out) + row * 256 + lane * 4) = o;
  }
}

struct AttnSeg { const float* k; const float* v; int rows, row_stride, batch_stride, head_stride; };
SF_KERNEL(256) void k_attn16(const float* __restrict__ q, void* __restrict__ out, AttnSeg s0, AttnSeg s1,
                                                AttnSeg s2, int heads, int ldq, float scale, int out_f32) {
  // 4 waves per (b, head): the kernel is a chain of dependent phases (load, q.k, softmax, p.v), so the only lever is to make
  // every phase short -- rows of q / k / v are fetched by different waves at once, the 16 x J scores and the 16 output rows
  // are spread over all 256 lanes
  SF_SHARED float sq[16][65];
  SF_SHARED float sk[24][65];
  SF_SHARED float sv[24][65];
  SF_SHARED float sim[16][25];
  const int b = blockIdx.x / heads, h = blockIdx.x % heads, t = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int i = wv; i < 16; i += 4) sq[i][t] = q[((long)b * 16 + i) * ldq + h * 64 + t] * scale;
  const AttnSeg segs[3] = {s0, s1, s2};
  int J = 0;
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    for (int r = wv; r < segs[s].rows; r += 4) {
      const long off = (long)b * segs[s].batch_stride + (long)r * segs[s].row_stride + (long)h * segs[s].head_stride + t;
      sk[J + r][t] = segs[s].k[off];
      sv[J + r][t] = segs[s].v[off];
    }
    J += segs[s].rows;
  }
  sf_sync();
  for (int e = threadIdx.x; e < 16 * J; e += 256) {
    const int i = e / J, j = e - i * J;
    float a = 0.0f;
#pragma unroll 8
    for (int d = 0; d < 64; ++d) a = fmaf(sq[i][d], sk[j][d], a);
    sim[i][j] = a;
  }
  sf_sync();
  if (threadIdx.x < 16) {
    const int i = threadIdx.x;
    float mx = -INFINITY;
    for (int j = 0; j < J; ++j) mx = fmaxf(mx, sim[i][j]);
    float den = 0.0f;
    for (int j = 0; j < J; ++j) { const float e = expf(sim[i][j] - mx); sim[i][j] = e; den += e; }
    const float inv = 1.0f / den;
    for (int j = 0; j < J; ++j) sim[i][j] *= inv;
  }
  sf_sync();
  for (int i = wv; i < 16; i += 4) {
    float a = 0.0f;
    for (int j = 0; j < J; ++j) a = fmaf(sim[i][j], sv[j][t], a);
    const long o = ((long)b * 16 + i) * (heads * 64) + h * 64 + t;
    if (out_f32) reinterpret_cast<float*>(out)[o] = a;          // consumed by a fused linear (fp32 A operand prologue)
    else reinterpret_cast<sf_opnd*>(out)[o] = (sf_opnd)a;
  }
}

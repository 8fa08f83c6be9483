// Fused GlobalContext (squeeze-excite with a learned softmax pooling, external/imagen_pytorch.py:916-941) + the gated
// residual of the ResnetBlock (:727-729) in three launches instead of five:
//     context logits  l[p] = h2[p, :] . wk           <- partial sums per 16-channel fragment come out of the producing
//                                                       conv's epilogue (k_conv_fused, FConvArgs.logit_part); the bias
//                                                       terms are the same for every pixel and cancel in the softmax
//     k_gca_pool      per (pixel chunk, 64-channel slab): local softmax numerators e = exp(l - max_chunk) and the
//                     un-normalised pooled slab sum_p e[p] * h2[p, c]; materialises h2 when it is still split-K slabs
//     k_gca_net0      merges the <= 8 chunks of an image (online-softmax merge), hid = SiLU(W0 . pooled + b0)
//     k_gca_gate      gate = sigmoid(W2 . hid + b2) for a 16-channel fragment, out = h2 * gate + res, and the
//                     (sum, sum of squares) slots of `out` for the next GroupNorm-fused conv
#pragma once
#include "sf_dev.h"

struct GcaPoolArgs {
  float* h2;                 // [M, C] final values (written when lazy)
  const float* ws;           // lazy: split-K slabs [groups][M][npad], else null
  const float* bias;         // lazy: conv bias [C] or null
  const float* logit_part;   // [nparts][M]
  float* part_pool;          // [B * chunks][C]
  float* part_ms;            // [B * chunks][2] (max, sum of exp)
  int M, C, HW, CH, chunks, nparts, groups, npad;
};

// grid = B * chunks * (C / 64); 256 threads = 16 channel float4 lanes x 16 pixel lanes
SF_KERNEL(256) void k_gca_pool(GcaPoolArgs a) {
  SF_SHARED float e[128];
  SF_SHARED float red[16][68];
  const int tid = threadIdx.x, lane = tid & 63;
  const int cslabs = a.C >> 6;
  const int cs = blockIdx.x % cslabs, bc = blockIdx.x / cslabs;      // bc = image * chunks + chunk
  const int b = bc / a.chunks, ch = bc - b * a.chunks;
  const long m0 = (long)b * a.HW + (long)ch * a.CH;
  // (1) logits of the chunk's pixels: 2 threads per pixel (CH <= 128), each sums half of the parts
  {
    const int p = tid >> 1, half = tid & 1;
    float l = 0.0f;
    if (p < a.CH)
      for (int k = half; k < a.nparts; k += 2) l += a.logit_part[(long)k * a.M + m0 + p];
    l += sf_shfl_xor(l, 1);
    if (p < a.CH && half == 0) e[p] = l;
  }
  sf_sync();
  // (2) chunk-local softmax numerators
  float mx = -INFINITY;
  for (int p = lane; p < a.CH; p += 64) mx = fmaxf(mx, e[p]);
  mx = sf_wave_max(mx);                      // every wave computes the same value
  float sm = 0.0f;
  for (int p = lane; p < a.CH; p += 64) sm += sf_exp(e[p] - mx);
  sm = sf_wave_sum(sm);
  sf_sync();
  if (tid < a.CH) e[tid] = sf_exp(e[tid] - mx);
  if (tid == 0 && cs == 0) { a.part_ms[(long)bc * 2] = mx; a.part_ms[(long)bc * 2 + 1] = sm; }
  sf_sync();
  // (3) un-normalised pooled slab; h2 is evaluated (and written back) from the slabs when lazy
  const int c4 = tid & 15, pl = tid >> 4;
  const int c = cs * 64 + c4 * 4;
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int p = pl; p < a.CH; p += 16) {
    const long m = m0 + p;
    f32x4 v;
    if (a.ws) {
      v = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + c) : f32x4{0.f, 0.f, 0.f, 0.f};
      for (int g = 0; g < a.groups; ++g) v += *reinterpret_cast<const f32x4*>(a.ws + ((long)g * a.M + m) * a.npad + c);
      *reinterpret_cast<f32x4*>(a.h2 + m * a.C + c) = v;
    } else {
      v = *reinterpret_cast<const f32x4*>(a.h2 + m * a.C + c);
    }
    acc += v * e[p];
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) red[pl][c4 * 4 + j] = acc[j];
  sf_sync();
  if (tid < 64) {
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += red[k][tid];
    a.part_pool[(long)bc * a.C + cs * 64 + tid] = s;
  }
}

struct GcaNetArgs {
  const float* part_pool;    // [B * chunks][C]
  const float* part_ms;      // [B * chunks][2]
  const __bf16* W0;          // [hid][Kp] (Kp = C padded to 8)
  const float* b0;
  float* hid;                // [B][hid]
  int B, C, Kp, HID, chunks;
};

// grid = B * ceil(HID / 16); 4 waves x 4 rows each
SF_KERNEL(256) void k_gca_net0(GcaNetArgs a) {
  SF_SHARED float pooled[2048];
  SF_SHARED float wgt[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rb = (a.HID + 15) / 16;
  const int b = blockIdx.x / rb, r0 = (blockIdx.x - b * rb) * 16;
  if (tid == 0) {                                         // online-softmax merge weights of the chunks
    float M = -INFINITY;
    for (int j = 0; j < a.chunks; ++j) M = fmaxf(M, a.part_ms[((long)b * a.chunks + j) * 2]);
    float Z = 0.0f;
    for (int j = 0; j < a.chunks; ++j) {
      const float w = sf_exp(a.part_ms[((long)b * a.chunks + j) * 2] - M);
      wgt[j] = w;
      Z += w * a.part_ms[((long)b * a.chunks + j) * 2 + 1];
    }
    const float inv = 1.0f / Z;
    for (int j = 0; j < a.chunks; ++j) wgt[j] *= inv;
  }
  sf_sync();
  for (int c = tid; c < a.C; c += 256) {
    float s = 0.0f;
    for (int j = 0; j < a.chunks; ++j) s = fmaf(wgt[j], a.part_pool[((long)b * a.chunks + j) * a.C + c], s);
    pooled[c] = s;
  }
  sf_sync();
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int r = r0 + wave * 4 + rr;
    if (r >= a.HID) break;                                // wave-uniform
    float acc = 0.0f;
    for (int k = lane * 8; k < a.C; k += 512) {
      const bf16x8 w = *reinterpret_cast<const bf16x8*>(a.W0 + (long)r * a.Kp + k);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc = fmaf((float)w[j], pooled[k + j], acc);
    }
    acc = sf_wave_sum(acc);
    if (lane == 0) a.hid[(long)b * a.HID + r] = sf_silu(acc + a.b0[r]);
  }
}

struct GcaGateArgs {
  const float* h2;
  const float* res;
  const float* hid;          // [B][HID]
  const __bf16* W2;          // [C][Kp2]
  const float* b2;
  float* out;
  float* slots;              // [M/16][C/16][2] or null
  int M, C, HW, HID, Kp2;
};

// one wave per (16 pixels, 16 channels); grid = ceil(M/16 * C/16 / 4)
SF_KERNEL(256) void k_gca_gate(GcaGateArgs a) {
  const int lane = threadIdx.x & 63, gw = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int CF = a.C >> 4;
  if (gw >= (a.M >> 4) * CF) return;
  const int mf = gw / CF, cf = gw - mf * CF;
  const int b = (mf * 16) / a.HW;
  // gate of channel cf*16 + (lane & 15): the 4 lanes with equal (lane & 15) split the hidden dimension
  const int ch = cf * 16 + (lane & 15), q = lane >> 4;
  float g = 0.0f;
  for (int k = q * 8; k < a.HID; k += 32) {
    const bf16x8 w = *reinterpret_cast<const bf16x8*>(a.W2 + (long)ch * a.Kp2 + k);
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (k + j < a.HID) g = fmaf((float)w[j], a.hid[(long)b * a.HID + k + j], g);
  }
  g += sf_shfl_xor(g, 16);
  g += sf_shfl_xor(g, 32);
  g = sf_sigmoid(g + a.b2[ch]);                           // lanes l, l+16, l+32, l+48 hold the gate of channel cf*16 + (l & 15)
  // the tile: lane -> pixel (lane >> 2), channels cf*16 + (lane & 3)*4 .. +3; fetch those 4 gates from their owner lanes
  const long m = (long)mf * 16 + (lane >> 2);
  const int c = cf * 16 + (lane & 3) * 4;
  f32x4 gv;
#pragma unroll
  for (int j = 0; j < 4; ++j) gv[j] = sf_shfl(g, (lane & 3) * 4 + j);
  const f32x4 hv = *reinterpret_cast<const f32x4*>(a.h2 + m * a.C + c);
  const f32x4 rv = *reinterpret_cast<const f32x4*>(a.res + m * a.C + c);
  const f32x4 v = hv * gv + rv;
  *reinterpret_cast<f32x4*>(a.out + m * a.C + c) = v;
  if (a.slots) {
    float sm = (v[0] + v[1]) + (v[2] + v[3]);
    float sq = fmaf(v[0], v[0], fmaf(v[1], v[1], fmaf(v[2], v[2], v[3] * v[3])));
    sm = sf_wave_sum(sm);
    sq = sf_wave_sum(sq);
    if (lane == 0) { a.slots[(long)gw * 2] = sm; a.slots[(long)gw * 2 + 1] = sq; }
  }
}

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

// grid = B * chunks * (C / 64); 256 threads = 16 channel float4 lanes x 16 pixel lanes.
// Every phase issues its loads as one independent batch: a dependent load per loop trip costs an L2 / fabric round trip
// (~0.5 us), and the first version of this kernel spent 30 us summing 256 logit parts one after the other.
SF_DEV void gca_pool_body(const GcaPoolArgs& a, const int bid) {
  SF_SHARED float e[128];
  SF_SHARED float red[16][132];
  const int tid = threadIdx.x, lane = tid & 63;
  const int cslabs = a.C >> 6;
  const int cs = bid % cslabs, bc = bid / cslabs;      // bc = image * chunks + chunk
  const int b = bc / a.chunks, ch = bc - b * a.chunks;
  const long m0 = (long)b * a.HW + (long)ch * a.CH;
  // the first trip of phase (3)'s operand loads goes out NOW: they do not depend on the logits, and issued after the softmax
  // they were a second cold round trip (the data was written by the previous kernel on other XCDs)
  const int c4 = tid & 15, pl = tid >> 4;
  const int c = cs * 64 + c4 * 4;
  const f32x4 bvec = (a.ws && a.bias) ? *reinterpret_cast<const f32x4*>(a.bias + c) : f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 v[4];
  auto load_trip = [&](int p0) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int p = p0 + u * 16;
      const long m = m0 + (p < a.CH ? p : a.CH - 1);
      if (a.ws) {                                 // all slab loads before the first add (see fsrc_load4<1>); order bias, slab 0, 1, ...
        const float* ap = a.ws + m * a.npad + c;
        const long gstride = (long)a.M * a.npad;
        const int gl = a.groups - 1;
        f32x4 t[8];
#pragma unroll
        for (int g = 0; g < 8; ++g) t[g] = *reinterpret_cast<const f32x4*>(ap + (g < gl ? g : gl) * gstride);
        v[u] = bvec;
#pragma unroll
        for (int g = 0; g < 8; ++g) v[u] += t[g] * (g <= gl ? 1.0f : 0.0f);
      } else {
        v[u] = *reinterpret_cast<const f32x4*>(a.h2 + m * a.C + c);
      }
    }
  };
  load_trip(pl);
  // (1) logits of the chunk's pixels: CH (power of two, 16..128) pixels x PL = 256 / CH part lanes
  {
    const int p = tid & (a.CH - 1), pl = tid / a.CH, PL = 256 / a.CH;
    float l = 0.0f;
    for (int k0 = pl; k0 < a.nparts; k0 += 8 * PL) {
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = k0 + u * PL;
        t[u] = a.logit_part[(long)(k < a.nparts ? k : a.nparts - 1) * a.M + m0 + p];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (k0 + u * PL < a.nparts) l += t[u];
    }
    red[pl][p] = l;
  }
  sf_sync();
  if (tid < a.CH) {
    const int PL = 256 / a.CH;
    float l = 0.0f;
    for (int k = 0; k < PL; ++k) l += red[k][tid];
    e[tid] = l;
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
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int p0 = pl; p0 < a.CH; p0 += 64) {   // 4 pixels per thread and trip, all their loads in flight together
    if (p0 != pl) load_trip(p0);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int p = p0 + u * 16;
      if (p < a.CH) {
        if (a.ws) *reinterpret_cast<f32x4*>(a.h2 + (m0 + p) * a.C + c) = v[u];
        acc += v[u] * e[p];
      }
    }
  }
  sf_sync();
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

SF_KERNEL(256) void k_gca_pool(GcaPoolArgs a) {
  sf_touch_kernarg<(int)sizeof(GcaPoolArgs)>();
  gca_pool_body(a, (int)blockIdx.x);
}

struct GcaNetArgs {
  const float* part_pool;    // [B * chunks][C]
  const float* part_ms;      // [B * chunks][2]
  const sf_opnd* W0;          // [hid][Kp] (Kp = C padded to 8)
  const float* b0;
  float* hid;                // [B][hid]
  int B, C, Kp, HID, chunks;
};

// grid = B * ceil(HID / 16); 4 waves x 4 rows each.  NCK = chunk capacity of the instantiation (8 | 16 | 32 | 64): every phase
// issues its loads as ONE batch -- all NCK pooled partials of a channel in flight together (a loop of 8-chunk trips would be up
// to 8 dependent round trips), those of the thread's first channel ahead of the merge weights they do not depend on.  More than 8
// chunks = the 16-pixel fragments the producing conv's epilogue pools (fused_pipe.h, POOL).
template <int NCK>
SF_DEV void gca_net0_body(const GcaNetArgs& a, float* pooled, float* wgt) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rb = (a.HID + 15) / 16;
  const int b = blockIdx.x / rb, r0 = (blockIdx.x - b * rb) * 16;
  // this wave's 4 weight rows: C <= 2048 -> at most 4 x 16-byte loads per row and lane, all issued before anything waits
  bf16x8 w[4][4];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int r = r0 + wave * 4 + rr;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int k = lane * 8 + it * 512;
      w[rr][it] = (r < a.HID && k < a.C) ? *reinterpret_cast<const bf16x8*>(a.W0 + (long)r * a.Kp + k) : sf_zero8();
    }
  }
  const float* pp = a.part_pool + (long)b * a.chunks * a.C;
  // ALL pooled partials this thread will merge are requested now (r04; the first version fetched those of its 2nd .. 4th channel
  // inside the merge loop: up to three more dependent round trips at C = 1024): CI channels per thread x NCK chunks <= 64 loads
  constexpr int CI = NCK <= 8 ? 4 : (NCK <= 16 ? 2 : 1);     // C <= 256 * CI is what the plans produce (host-checked; the loop below covers the rest)
  float pj0[CI][NCK];
#pragma unroll
  for (int ci = 0; ci < CI; ++ci) {
    const int cc = tid + ci * 256;
#pragma unroll
    for (int j = 0; j < NCK; ++j) pj0[ci][j] = pp[(long)(j < a.chunks ? j : a.chunks - 1) * a.C + (cc < a.C ? cc : 0)];
  }
  if (wave == 0) {                                        // online-softmax merge weights of the chunks (lanes = chunks, <= 64)
    const bool on = lane < a.chunks;
    const float mj = on ? a.part_ms[((long)b * a.chunks + lane) * 2] : -INFINITY;
    const float sj = on ? a.part_ms[((long)b * a.chunks + lane) * 2 + 1] : 0.0f;
    const float M = sf_wave_max(mj);
    const float wj = on ? sf_exp(mj - M) : 0.0f;
    const float Z = sf_wave_sum(wj * sj);
    wgt[lane] = on ? wj / Z : 0.0f;                       // 0 beyond the last chunk
  }
  sf_sync();
#pragma unroll
  for (int ci = 0; ci < CI; ++ci) {
    const int c = tid + ci * 256;
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < NCK; ++j) s = fmaf(wgt[j], pj0[ci][j], s);
    if (c < a.C) pooled[c] = s;
  }
  for (int c = tid + CI * 256; c < a.C; c += 256) {            // wider layers than the preload covers (not in the UNet's plans)
    float s = 0.0f;
    for (int j = 0; j < a.chunks; ++j) s = fmaf(wgt[j], pp[(long)j * a.C + c], s);
    pooled[c] = s;
  }
  sf_sync();
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int r = r0 + wave * 4 + rr;
    float acc = 0.0f;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int k = lane * 8 + it * 512;
      if (k < a.C) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc = fmaf((float)w[rr][it][j], pooled[k + j], acc);
      }
    }
    acc = sf_wave_sum(acc);
    if (lane == 0 && r < a.HID) a.hid[(long)b * a.HID + r] = sf_silu(acc + a.b0[r]);
  }
}

template <int NCK>
SF_KERNEL(256) void k_gca_net0(GcaNetArgs a) {
  sf_touch_kernarg<(int)sizeof(GcaNetArgs)>();
  SF_SHARED float pooled[2048];
  SF_SHARED float wgt[64];
  gca_net0_body<NCK>(a, pooled, wgt);
}

// k_gca_net0 with the channel count and the chunk capacity as template parameters (r05; C = 256 | 512 | 1024 with 8 | 16 | 64 chunks = every
// GlobalContext block of the canonical UNet).  Against k_gca_net0: the row biases are requested with everything else (they were a
// dependent load in front of the store: one more L2 round trip at the tail), every wave forms the chunks' softmax weights itself from
// part_ms (no wave-0 section, no block barrier in front of the merge: a per-wave LDS copy and a wave-level hand-off), weight loads are
// unconditional (clamped address, zero weight) instead of sixteen branches.  Same arithmetic.
template <int C, int NCK>
SF_KERNEL(256) void k_gca_net0_t(GcaNetArgs a) {
  sf_touch_kernarg<(int)sizeof(GcaNetArgs)>();
  SF_SHARED float pooled[C];
  SF_SHARED float wgt[4][64];
  constexpr int KIT = C > 512 ? C / 512 : 1;                 // 16-byte weight loads per row and lane
  constexpr int CI = C / 256;                                 // channels per thread in the merge
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rb = (a.HID + 15) / 16;
  const int b = blockIdx.x / rb, r0 = (blockIdx.x - b * rb) * 16;
  bf16x8 w[4][KIT];
  float bq[4];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int r = r0 + wave * 4 + rr, rc = r < a.HID ? r : a.HID - 1;
    bq[rr] = a.b0[rc];
#pragma unroll
    for (int it = 0; it < KIT; ++it) {
      const int k = lane * 8 + it * 512;
      w[rr][it] = *reinterpret_cast<const bf16x8*>(a.W0 + (long)rc * a.Kp + (k < C ? k : 0));
    }
  }
  const float* pp = a.part_pool + (long)b * a.chunks * C;
  float pj[CI][NCK];
#pragma unroll
  for (int ci = 0; ci < CI; ++ci)
#pragma unroll
    for (int j = 0; j < NCK; ++j) pj[ci][j] = pp[(long)(j < a.chunks ? j : a.chunks - 1) * C + tid + ci * 256];
  {
    const bool on = lane < a.chunks;
    const int jl = on ? lane : a.chunks - 1;
    const float mq = a.part_ms[((long)b * a.chunks + jl) * 2], sq = a.part_ms[((long)b * a.chunks + jl) * 2 + 1];
    const float mj = on ? mq : -INFINITY, sj = on ? sq : 0.0f;
    const float M = sf_wave_max(mj);
    const float wj = on ? sf_exp(mj - M) : 0.0f;
    const float Z = sf_wave_sum(wj * sj);
    wgt[wave][lane] = on ? wj / Z : 0.0f;                     // 0 beyond the last chunk; every wave keeps its own copy
  }
  sf_wave_sync();
#pragma unroll
  for (int ci = 0; ci < CI; ++ci) {
    float sacc = 0.0f;
#pragma unroll
    for (int j = 0; j < NCK; ++j) sacc = fmaf(wgt[wave][j], pj[ci][j], sacc);
    pooled[tid + ci * 256] = sacc;
  }
  sf_sync();
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int r = r0 + wave * 4 + rr;
    float acc = 0.0f;
#pragma unroll
    for (int it = 0; it < KIT; ++it) {
      const int k = lane * 8 + it * 512;
      if (k < C) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc = fmaf((float)w[rr][it][j], pooled[k + j], acc);
      }
    }
    acc = sf_wave_sum(acc);
    if (lane == 0 && r < a.HID) a.hid[(long)b * a.HID + r] = sf_silu(acc + bq[rr]);
  }
}

struct GcaGateArgs {
  const float* h2;
  const float* res;
  const float* hid;          // [B][HID]
  const sf_opnd* W2;          // [C][Kp2]
  const float* b2;
  float* out;
  float* slots;              // [M/16][C/16][2] or null
  int M, C, HW, HID, Kp2;
};

// one wave per (16 pixels, 16 channels); grid = ceil(M/16 * C/16 / 4)
SF_KERNEL(256) void k_gca_gate(GcaGateArgs a) {
  sf_touch_kernarg<(int)sizeof(GcaGateArgs)>();
  const int CF = a.C >> 4;
  const int lane = threadIdx.x & 63, gw = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (gw >= (a.M >> 4) * CF) return;
  const int mf = gw / CF, cf = gw - mf * CF;
  const int b = (mf * 16) / a.HW;
  // the tile's own operands first (independent of the gate): lane -> pixel (lane >> 2), channels cf*16 + (lane & 3)*4 .. +3
  const long m = (long)mf * 16 + (lane >> 2);
  const int c = cf * 16 + (lane & 3) * 4;
  const f32x4 hv = *reinterpret_cast<const f32x4*>(a.h2 + m * a.C + c);
  const f32x4 rv = *reinterpret_cast<const f32x4*>(a.res + m * a.C + c);
  // gate of channel cf*16 + (lane & 15): the 4 lanes with equal (lane & 15) split the hidden dimension; HID <= 1024
  // -> at most 4 trips of 8 k-steps, each trip's 8 weight + 16 hidden-vector loads in flight together
  const int ch = cf * 16 + (lane & 15), q = lane >> 4;
  const float bias = a.b2[ch];
  float g = 0.0f;
  for (int k0 = q * 8; k0 < a.HID; k0 += 256) {
    bf16x8 w[8];
    f32x4 h0[8], h1[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      int k = k0 + u * 32;
      if (k + 8 > a.Kp2) k = 0;                            // clamped, masked below
      w[u] = *reinterpret_cast<const bf16x8*>(a.W2 + (long)ch * a.Kp2 + k);
      int kh = k0 + u * 32;
      if (kh + 8 > a.HID) kh = a.HID >= 8 ? a.HID - 8 : 0;
      h0[u] = *reinterpret_cast<const f32x4*>(a.hid + (long)b * a.HID + kh);
      h1[u] = *reinterpret_cast<const f32x4*>(a.hid + (long)b * a.HID + kh + 4);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = k0 + u * 32;
      if (k + 8 <= a.HID) {
#pragma unroll
        for (int j = 0; j < 4; ++j) g = fmaf((float)w[u][j], h0[u][j], g);
#pragma unroll
        for (int j = 0; j < 4; ++j) g = fmaf((float)w[u][4 + j], h1[u][j], g);
      } else {
        for (int j = 0; j < 8; ++j)
          if (k + j < a.HID) g = fmaf((float)(reinterpret_cast<const sf_opnd*>(a.W2)[(long)ch * a.Kp2 + k + j]), a.hid[(long)b * a.HID + k + j], g);
      }
    }
  }
  g += sf_shfl_xor(g, 16);
  g += sf_shfl_xor(g, 32);
  g = sf_sigmoid(g + bias);                               // lanes l, l+16, l+32, l+48 hold the gate of channel cf*16 + (l & 15)
  f32x4 gv;
#pragma unroll
  for (int j = 0; j < 4; ++j) gv[j] = sf_shfl(g, (lane & 3) * 4 + j);
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

// k_gca_gate with the hidden width as a template parameter (r05; HID = 128 | 256 | 512 = every GlobalContext block of the canonical UNet):
// the trip loop of k_gca_gate has a run-time count with loads inside -- at HID = 512 two dependent round trips, at HID = 128 half of its
// 24 loads clamped -- here all W2 fragments and hidden-vector pieces of the lane are requested together with the tile's operands.
template <int HID>
SF_KERNEL(256) void k_gca_gate_t(GcaGateArgs a) {
  sf_touch_kernarg<(int)sizeof(GcaGateArgs)>();
  constexpr int NTR = (HID + 255) / 256, U = HID >= 256 ? 8 : HID / 32;
  const int CF = a.C >> 4;
  const int lane = threadIdx.x & 63, gw = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (gw >= (a.M >> 4) * CF) return;
  const int mf = gw / CF, cf = gw - mf * CF;
  const int b = (mf * 16) / a.HW;
  const long m = (long)mf * 16 + (lane >> 2);
  const int c = cf * 16 + (lane & 3) * 4;
  const f32x4 hv = *reinterpret_cast<const f32x4*>(a.h2 + m * a.C + c);
  const f32x4 rv = *reinterpret_cast<const f32x4*>(a.res + m * a.C + c);
  const int ch = cf * 16 + (lane & 15), q = lane >> 4;
  const float bias = a.b2[ch];
  bf16x8 w[NTR][U];
  f32x4 h0[NTR][U], h1[NTR][U];
  const sf_opnd* wr = a.W2 + (long)ch * a.Kp2 + q * 8;
  const float* hr = a.hid + (long)b * HID + q * 8;
#pragma unroll
  for (int t = 0; t < NTR; ++t)
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = t * 256 + u * 32;
      w[t][u] = *reinterpret_cast<const bf16x8*>(wr + k);
      h0[t][u] = *reinterpret_cast<const f32x4*>(hr + k);
      h1[t][u] = *reinterpret_cast<const f32x4*>(hr + k + 4);
    }
  float g = 0.0f;
#pragma unroll
  for (int t = 0; t < NTR; ++t)
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int j = 0; j < 4; ++j) g = fmaf((float)w[t][u][j], h0[t][u][j], g);
#pragma unroll
      for (int j = 0; j < 4; ++j) g = fmaf((float)w[t][u][4 + j], h1[t][u][j], g);
    }
  g += sf_shfl_xor(g, 16);
  g += sf_shfl_xor(g, 32);
  g = sf_sigmoid(g + bias);
  f32x4 gv;
#pragma unroll
  for (int j = 0; j < 4; ++j) gv[j] = sf_shfl(g, (lane & 3) * 4 + j);
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

// Measured and not kept (r04): net0 + gate as ONE launch on the C <= 512 levels, every workgroup recomputing the hidden vector
// (k_gca_ng: 64 KB of pooled partials + 64-256 KB of W0 per workgroup, then its 16 fragments).  Parity-green, 11 launches fewer,
// and SLOWER: B = 1 eval 1.243 -> 1.336 ms (+8.5 us per block; B = 4: 1.974 -> 2.052): the redundant matvec is a chain of
// L2-latency-bound batches on one 8-wave workgroup per CU and costs more than the 6.6 us launch it replaces -- the same outcome as
// r02's pool + net0 merge.  The all-to-all seams of GlobalContext stay kernel boundaries.
// Also measured and not kept (r04, profiles/r04_graph_ablate_b1_rejected_ln_gate.log vs r04_graph_ablate_b1_first.log): k_gca_gate with the hidden
// vector through a per-wave LDS copy and both trips' weight fragments requested up front -- +0.4 .. +1.1 us per launch (the copy is
// one more dependent step; the scattered loads it replaced travelled with the weight loads); the LayerNorm gain / bias of
// k_conv_fused<.., FNORM_LN> fetched with the row instead of after the statistics -- +2.4 us per launch (32 more vector loads per
// thread in front of the row statistics).  k_gca_net0's pooled partials all requested at entry: -0.7 us per launch, kept.

// Wave-per-ray merge + composite (r02; the thread-per-ray k_ngp_composite of ngp_render.hip spends 0.38 ms on 16 384 rays with one wave per CU).
// Same per-sample arithmetic as ngp_merge_composite; what changes is who does it:
//   rank   : lane l owns the samples l (coarse) and T + l (fine) of cat([coarse, fine]); its rank is the number of samples
//            that sort in front of it, counted against wave-uniform (v_readlane) copies of every sample's key = (sortable
//            z bits, index) -- a stable total order (coarse first on ties, NaNs last), no sorting network, no divergence;
//   scatter: each sample goes to its rank in a per-wave LDS image of the sorted ray;
//   scan   : lane l then owns sorted positions 2l, 2l + 1: deltas by one shuffle, transmittance by an exclusive product scan
//            (double, as the per-ray loop), weights, and butterfly sums for image / depth / opacity.
// Sums are taken in tree order instead of front to back (<= 1e-7 on weights that sum to <= 1; the tests hold 1e-5).
// Written against sf_dev.h so that tests/hostemu runs the same source on CPU threads (tests/test_hostemu_composite.py).
#pragma once
#include "sf_dev.h"
#include "ngp_device.h"

struct CompositeArgs {
  const float* z_c; const float* sig_c; const float* rgb_c;      // coarse samples [N][T], [N][T], [N][T][3]
  const float* z_f; const float* sig_f; const float* rgb_f;      // fine samples, same shapes
  const float* nears; const float* fars;
  uint32_t N, T;
  float bg;
  float* z_s; float* sig_s; float* rgb_s;                        // sorted ray [N][2T], [N][2T], [N][2T][3] (kept for the backward)
  float* image; float* depth; float* weights_sum;                // [N][3], [N], [N]
  uint32_t* perm;                                                // or null: [N][2T] index in cat([coarse, fine]) of every sorted sample
};                                                               // (the field backward finds a sample's cached features through it)

SF_DEV uint32_t ngp_sort_key(float z) {
  const uint32_t b = __builtin_bit_cast(uint32_t, z);
  return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}

SF_KERNEL(256) void k_ngp_composite_wave(CompositeArgs a) {
  SF_DYN_LDS(lds_raw);
  float* smem = reinterpret_cast<float*>(lds_raw);
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t n = blockIdx.x * 4 + wave;
  if (n >= a.N) return;                                        // whole waves leave together; no workgroup barrier below
  const uint32_t T = a.T, M = 2 * T;
  float* sz = smem + (size_t)wave * 5 * M;                     // [M] z | [M] sigma | [3 M] rgb
  float* ss = sz + M;
  float* sr = ss + M;
  // ---- this lane's two samples of cat([coarse, fine]) (lane < T)
  const bool have = lane < T;
  const size_t row = (size_t)n * T + (have ? lane : 0);
  float zv[2], sv[2], cv[2][3];
  zv[0] = a.z_c[row]; zv[1] = a.z_f[row];
  sv[0] = a.sig_c[row]; sv[1] = a.sig_f[row];
#pragma unroll
  for (int c = 0; c < 3; ++c) { cv[0][c] = a.rgb_c[row * 3 + c]; cv[1][c] = a.rgb_f[row * 3 + c]; }
  const uint32_t key0 = have ? ngp_sort_key(zv[0]) : 0xFFFFFFFFu;
  const uint32_t key1 = have ? ngp_sort_key(zv[1]) : 0xFFFFFFFFu;
  const uint64_t mine0 = ((uint64_t)key0 << 32) | lane;        // index in the concatenation: coarse l, fine T + l
  const uint64_t mine1 = ((uint64_t)key1 << 32) | (T + lane);
  uint32_t rank0 = 0, rank1 = 0;
  for (uint32_t j = 0; j < T; ++j) {
    const uint64_t oc = ((uint64_t)sf_readlane(key0, j) << 32) | j;
    const uint64_t of = ((uint64_t)sf_readlane(key1, j) << 32) | (T + j);
    rank0 += (uint32_t)(oc < mine0) + (uint32_t)(of < mine0);
    rank1 += (uint32_t)(oc < mine1) + (uint32_t)(of < mine1);
  }
  if (have) {
    sz[rank0] = zv[0]; ss[rank0] = sv[0]; sr[rank0 * 3 + 0] = cv[0][0]; sr[rank0 * 3 + 1] = cv[0][1]; sr[rank0 * 3 + 2] = cv[0][2];
    sz[rank1] = zv[1]; ss[rank1] = sv[1]; sr[rank1 * 3 + 0] = cv[1][0]; sr[rank1 * 3 + 1] = cv[1][1]; sr[rank1 * 3 + 2] = cv[1][2];
    if (a.perm) {
      a.perm[(size_t)n * M + rank0] = lane;
      a.perm[(size_t)n * M + rank1] = T + lane;
    }
  }
  sf_wave_sync();                                              // LDS writes of this wave are read back by other lanes of it
  // ---- sorted positions 2 lane, 2 lane + 1
  const uint32_t m0 = 2 * lane;
  const bool live = m0 < M;                                    // M is even: both positions or none
  const uint32_t q = live ? m0 : 0;
  const float z0 = sz[q], z1 = sz[q + 1], s0 = ss[q], s1 = ss[q + 1];
  float c0[3], c1[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) { c0[c] = sr[q * 3 + c]; c1[c] = sr[q * 3 + 3 + c]; }
  if (live) {
    float* zo = a.z_s + (size_t)n * M + m0;
    float* so = a.sig_s + (size_t)n * M + m0;
    float* ro = a.rgb_s + ((size_t)n * M + m0) * 3;
    zo[0] = z0; zo[1] = z1; so[0] = s0; so[1] = s1;
#pragma unroll
    for (int c = 0; c < 3; ++c) { ro[c] = c0[c]; ro[3 + c] = c1[c]; }
  }
  const float near = a.nears[n], far = a.fars[n];
  const float sample_dist = SF_DIV(SF_SUB(far, near), (float)T);
  const float span = SF_SUB(far, near);
  const float z_next = sf_shfl(z0, (int)((lane + 1) & 63));    // first sample of the next lane
  const float d0 = SF_SUB(z1, z0);
  const float d1 = (m0 + 2 < M) ? SF_SUB(z_next, z1) : sample_dist;
  const float a0 = live ? SF_SUB(1.0f, expf(SF_MUL(-d0, s0))) : 0.0f;
  const float a1 = live ? SF_SUB(1.0f, expf(SF_MUL(-d1, s1))) : 0.0f;
  const double f0 = live ? (double)SF_ADD(SF_SUB(1.0f, a0), 1e-15f) : 1.0;
  const double f1 = live ? (double)SF_ADD(SF_SUB(1.0f, a1), 1e-15f) : 1.0;
  double incl = f0 * f1;                                       // inclusive product scan over lanes
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const double up = sf_shfl(incl, (int)((lane - d) & 63));
    if ((int)lane >= d) incl *= up;
  }
  double excl = sf_shfl(incl, (int)((lane - 1) & 63));
  if (lane == 0) excl = 1.0;
  const float w0 = SF_MUL(a0, (float)excl);
  const float w1 = SF_MUL(a1, (float)(excl * f0));
  const float raw0 = SF_DIV(SF_SUB(z0, near), span), raw1 = SF_DIV(SF_SUB(z1, near), span);
  const float oz0 = (raw0 != raw0) ? raw0 : fminf(fmaxf(raw0, 0.0f), 1.0f);   // NaN (miss rays: 0/0) propagates like torch.clamp
  const float oz1 = (raw1 != raw1) ? raw1 : fminf(fmaxf(raw1, 0.0f), 1.0f);
  float acc[5];
  acc[0] = live ? SF_ADD(w0, w1) : 0.0f;
  acc[1] = live ? SF_ADD(SF_MUL(w0, oz0), SF_MUL(w1, oz1)) : 0.0f;
#pragma unroll
  for (int c = 0; c < 3; ++c) acc[2 + c] = live ? SF_ADD(SF_MUL(w0, c0[c]), SF_MUL(w1, c1[c])) : 0.0f;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
#pragma unroll
    for (int k = 0; k < 5; ++k) acc[k] = SF_ADD(acc[k], sf_shfl_xor(acc[k], d));
  }
  if (lane == 0) {
    const float rest = SF_MUL(SF_SUB(1.0f, acc[0]), a.bg);
    a.image[n * 3 + 0] = SF_ADD(acc[2], rest); a.image[n * 3 + 1] = SF_ADD(acc[3], rest); a.image[n * 3 + 2] = SF_ADD(acc[4], rest);
    a.depth[n] = acc[1];
    a.weights_sum[n] = acc[0];
  }
}

// Backward of the composite, one wave per ray (the thread-per-ray k_ngp_composite_bwd takes 0.105 ms for 16 384 rays): the
// arithmetic of ngp_composite_backward with lane l owning sorted positions 2l, 2l + 1 -- transmittance by the same exclusive
// product scan as the forward, the tail sums sum_{j>m} a_j w_j by an exclusive SUFFIX scan in double.
struct CompositeBwdArgs {
  const float* z_s; const float* sig_s; const float* rgb_s;      // sorted ray [N][2T], [N][2T], [N][2T][3]
  const float* nears; const float* fars;
  uint32_t N, T;
  float bg;
  const float* g_image; const float* g_ws;                       // [N][3], [N] or null
  float* dsig; float* drgb;                                      // [N][2T], [N][2T][3]
};

SF_KERNEL(256) void k_ngp_composite_bwd_wave(CompositeBwdArgs a) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t n = blockIdx.x * 4 + wave;
  if (n >= a.N) return;
  const uint32_t T = a.T, M = 2 * T;
  const uint32_t m0 = 2 * lane;
  const bool live = m0 < M;
  const size_t q = (size_t)n * M + (live ? m0 : 0);
  const float z0 = a.z_s[q], z1 = a.z_s[q + 1], s0 = a.sig_s[q], s1 = a.sig_s[q + 1];
  float c0[3], c1[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) { c0[c] = a.rgb_s[q * 3 + c]; c1[c] = a.rgb_s[q * 3 + 3 + c]; }
  const float gI[3] = {a.g_image[n * 3], a.g_image[n * 3 + 1], a.g_image[n * 3 + 2]};
  const float gW = a.g_ws ? a.g_ws[n] : 0.0f;
  const float near = a.nears[n], far = a.fars[n];
  const float sample_dist = SF_DIV(SF_SUB(far, near), (float)T);
  const float z_next = sf_shfl(z0, (int)((lane + 1) & 63));
  const float d0 = SF_SUB(z1, z0);
  const float d1 = (m0 + 2 < M) ? SF_SUB(z_next, z1) : sample_dist;
  const float e0 = expf(SF_MUL(-d0, s0)), e1 = expf(SF_MUL(-d1, s1));          // 1 - alpha
  const float a0 = live ? SF_SUB(1.0f, e0) : 0.0f, a1 = live ? SF_SUB(1.0f, e1) : 0.0f;
  const double f0 = live ? (double)SF_ADD(SF_SUB(1.0f, a0), 1e-15f) : 1.0;
  const double f1 = live ? (double)SF_ADD(SF_SUB(1.0f, a1), 1e-15f) : 1.0;
  double incl = f0 * f1;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const double up = sf_shfl(incl, (int)((lane - d) & 63));
    if ((int)lane >= d) incl *= up;
  }
  double excl = sf_shfl(incl, (int)((lane - 1) & 63));
  if (lane == 0) excl = 1.0;
  const float tr0 = (float)excl, tr1 = (float)(excl * f0);
  const float w0 = SF_MUL(a0, tr0), w1 = SF_MUL(a1, tr1);
  const float gsum = gI[0] + gI[1] + gI[2];
  const float av0 = gI[0] * c0[0] + gI[1] * c0[1] + gI[2] * c0[2] - a.bg * gsum + gW;
  const float av1 = gI[0] * c1[0] + gI[1] * c1[1] + gI[2] * c1[2] - a.bg * gsum + gW;
  const double t0 = live ? (double)av0 * (double)w0 : 0.0, t1 = live ? (double)av1 * (double)w1 : 0.0;
  double suf = t0 + t1;                                         // inclusive suffix sum over lanes
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const double dn = sf_shfl(suf, (int)((lane + d) & 63));
    if ((int)lane + d < 64) suf += dn;
  }
  double after = sf_shfl(suf, (int)((lane + 1) & 63));         // sum over the lanes behind this one
  if (lane == 63) after = 0.0;
  if (live) {
    const float one0 = SF_ADD(SF_SUB(1.0f, SF_SUB(1.0f, e0)), 1e-15f), one1 = SF_ADD(SF_SUB(1.0f, SF_SUB(1.0f, e1)), 1e-15f);
    const float da1 = av1 * tr1 - (float)(after / (double)one1);
    const float da0 = av0 * tr0 - (float)((after + t1) / (double)one0);
    float* ds = a.dsig + q;
    float* dr = a.drgb + q * 3;
    ds[0] = da0 * d0 * e0;
    ds[1] = da1 * d1 * e1;
#pragma unroll
    for (int c = 0; c < 3; ++c) { dr[c] = w0 * gI[c]; dr[3 + c] = w1 * gI[c]; }
  }
}

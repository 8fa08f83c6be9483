// Field backward of the fused Instant-NGP render on the matrix cores (gfx950), fp32 in / fp32 out.
//
// What it computes (per sample point, external/nerf/network_grid.py:14-33,77-88 backward through trunc_exp / sigmoid):
//   recompute feat -> h1 -> h2 -> out, then d(out) -> d(h2) -> d(h1) -> d(feat) and the weight / bias gradients
//   dW2 += dout^T h2, dW1 += dh2^T h1, dW0 += dh1^T feat.
// The first version (k_ngp_field_bwd) ran this as per-thread mat-vecs with LDS broadcast reads and LDS-staged outer
// products: 26 TFLOP/s of fp32 VALU, 3.9 ms per 2.1 M points.  All of it is GEMM over tiles of points, so here one WAVE
// owns 32 points per trip and runs six small GEMMs on v_mfma_f32_16x16x4_f32 (exact fp32 products and accumulation: same
// values as fmaf chains up to summation order -- the parity tolerances of tests/test_gpu_ngp.py are unchanged):
//     H1 = relu(F W0^T + b0)   [32x32][32x64]      H2 = relu(H1 W1^T + b1)   [32x64][64x64]
//     dW1 += dH2^T H1          [64x32][32x64]      dH1 = (dH2 W1) .* (H1>0)  [32x64][64x64]
//     dW0 += dH1^T F           [64x32][32x32]      dF  = dH1 W0              [32x64][64x32]
// (the 4-wide output layer and its gradients stay on the VALU).  Operands live in wave-private LDS tiles with odd row
// strides (conflict-free 4-byte fragment reads: lane (i, kq) reads element [i][4s + kq] or [4s + kq][i]); the weight
// gradients accumulate in registers over the whole kernel and are flushed once.  No workgroup barrier in the loop: the
// four waves of a workgroup never exchange data.
#pragma once
#include "sf_dev.h"
#include "ngp_device.h"

#define FB_PTS 32
#define FB_SF 33                                   // feat row stride (floats)
#define FB_SH 65                                   // hidden row stride
#define FB_WAVE_FLOATS (FB_PTS * FB_SF + 2 * FB_PTS * FB_SH + FB_PTS * 4 + FB_PTS)
#define FB_W0 0                                    // [64][33]
#define FB_W1 (FB_W0 + NGP_HID * FB_SF)            // [64][65]
#define FB_W2 (FB_W1 + NGP_HID * FB_SH)            // [4][64]
#define FB_B0 (FB_W2 + NGP_OUT * NGP_HID)
#define FB_B1 (FB_B0 + NGP_HID)
#define FB_B2 (FB_B1 + NGP_HID)
#define FB_WTOT (FB_B2 + 8)
#define FB_LDS_FLOATS (FB_WTOT + 4 * FB_WAVE_FLOATS)

struct FBArgs {
  const float* table; const float* w0; const float* b0; const float* w1; const float* b1; const float* w2; const float* b2;
  float bound;
  float* g_w0; float* g_b0; float* g_w1; float* g_b1; float* g_w2; float* g_b2;
  NgpLevels lv;
  const float* rays_o; const float* rays_d; const float* aabb; const float* z_s; const float* dsig; const float* drgb;
  float* dfeat_out;          // level-major [L][dfeat_P][2] or null (table frozen)
  uint32_t P, T2;
  uint32_t dfeat_P, p_off;   // a launch may cover the chunk [p_off, p_off + P) of a larger point set (ray / sample pointers are
                             // pre-offset by the host): dfeat keeps the layout of the WHOLE set, level stride dfeat_P
  // r03: features cached by the forward (sf_ngp_render_forward's field_cache) instead of the re-gather of 16 levels x 8 corners
  // per point (measured 0.73 of the 2.7 ms of this kernel per render: 0.5 ms hashing, 0.2 ms table reads).  All three null =
  // recompute.  feat_c / feat_f: [N * T][32] of the coarse / fine samples in sampling order, perm: [N][2T] sorted position ->
  // index in cat([coarse, fine]); all indexed with the WHOLE set's point index p_off + p.
  const float* feat_c; const float* feat_f; const uint32_t* perm;
};

SF_KERNEL(256) void k_ngp_field_bwd_mfma(FBArgs a) {
  SF_DYN_LDS(lds_raw);
  float* W = reinterpret_cast<float*>(lds_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* F = W + FB_WTOT + wave * FB_WAVE_FLOATS;    // [32][33] features
  float* H1 = F + FB_PTS * FB_SF;                     // [32][65] h1, later dh1
  float* H2 = H1 + FB_PTS * FB_SH;                    // [32][65] h2, later dh2
  float* DO = H2 + FB_PTS * FB_SH;                    // [32][4]  d(out)
  float* INS = DO + FB_PTS * 4;                       // [32]     1 = point inside the grid and live
  // padded weight image
  for (int i = tid; i < NGP_HID * NGP_FEAT; i += 256) W[FB_W0 + (i >> 5) * FB_SF + (i & 31)] = a.w0[i];
  for (int i = tid; i < NGP_HID * NGP_HID; i += 256) W[FB_W1 + (i >> 6) * FB_SH + (i & 63)] = a.w1[i];
  for (int i = tid; i < NGP_OUT * NGP_HID; i += 256) W[FB_W2 + i] = a.w2[i];
  if (tid < NGP_HID) { W[FB_B0 + tid] = a.b0[tid]; W[FB_B1 + tid] = a.b1[tid]; }
  if (tid < NGP_OUT) W[FB_B2 + tid] = a.b2[tid];
  float box[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) box[i] = a.aabb[i];
  sf_sync();

  const int li = lane & 15, kq = lane >> 4;            // MFMA fragment coordinates of this lane
  f32x4 acc1[4][4], acc0[4][2];                        // dW1[jt][kt], dW0[jt][ft] tiles (rows j = jt*16 + 4*kq + r, cols li)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    acc0[i][0] = f32x4{0.f, 0.f, 0.f, 0.f};
    acc0[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  float acc2[4] = {0.f, 0.f, 0.f, 0.f};                // dW2[o = kq][k = li*4 + i]
  float accb2 = 0.0f, accb1 = 0.0f, accb0 = 0.0f;      // db2[kq] (lanes li == 0), db1[lane], db0[lane]

  const uint32_t n_trips = (a.P + FB_PTS - 1) / FB_PTS;
  for (uint32_t trip = blockIdx.x * 4 + wave; trip < n_trips; trip += gridDim.x * 4) {
    const uint32_t p0 = trip * FB_PTS;
    // ---- A: positions + hash-grid features; lane = (point pl, half): each half encodes 8 of the 16 levels
    const int pl = lane & 31, half = lane >> 5;
    const uint32_t p = p0 + pl;
    const bool live = p < a.P;
    float x[3] = {0.f, 0.f, 0.f}, x01[3] = {0.f, 0.f, 0.f};
    bool inside = false;
    if (live) {
      const uint32_t n = p / a.T2;
      const float o[3] = {a.rays_o[n * 3], a.rays_o[n * 3 + 1], a.rays_o[n * 3 + 2]};
      const float d[3] = {a.rays_d[n * 3], a.rays_d[n * 3 + 1], a.rays_d[n * 3 + 2]};
      ngp_point(o, d, a.z_s[p], box, x);
      inside = ngp_unit(x, a.bound, x01);
    }
    if (a.feat_c) {
      // cached features: this lane's 16 floats (levels half*8 .. half*8+7) of the sample's row
      f32x4 fv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) fv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (live) {
        const uint32_t pg = a.p_off + p, n = pg / a.T2, T = a.T2 >> 1;
        const uint32_t src = a.perm[pg];
        const float* row = (src < T ? a.feat_c + ((size_t)n * T + src) * NGP_FEAT : a.feat_f + ((size_t)n * T + (src - T)) * NGP_FEAT) + half * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j) fv[j] = *reinterpret_cast<const f32x4*>(row + 4 * j);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) F[pl * FB_SF + half * 16 + 4 * j + e] = fv[j][e];
    } else {
#pragma unroll
      for (int ll = 0; ll < 8; ++ll) {
        const uint32_t l = half * 8 + ll;
        float r0 = 0.0f, r1 = 0.0f;
        if (inside && l < a.lv.L) {
          NgpCell c;
          ngp_cell(a.lv, l, x01, c);
          const float* tab = a.table + (size_t)a.lv.offset[l] * 2;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const f32x2 fv = *reinterpret_cast<const f32x2*>(tab + (size_t)c.row[i] * 2);
            r0 = fmaf(c.w[i], fv[0], r0);
            r1 = fmaf(c.w[i], fv[1], r1);
          }
        }
        F[pl * FB_SF + 2 * l] = r0;
        F[pl * FB_SF + 2 * l + 1] = r1;
      }
    }
    if (half == 0) INS[pl] = inside ? 1.0f : 0.0f;
    sf_wave_sync();

    // ---- B1: H1 = relu(F W0^T + b0)
    {
      f32x4 c[2][4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const float bv = W[FB_B0 + nt * 16 + li];
        c[0][nt] = f32x4{bv, bv, bv, bv};
        c[1][nt] = c[0][nt];
      }
#pragma unroll
      for (int s = 0; s < NGP_FEAT / 4; ++s) {
        const float a0 = F[li * FB_SF + 4 * s + kq], a1 = F[(16 + li) * FB_SF + 4 * s + kq];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const float b = W[FB_W0 + (nt * 16 + li) * FB_SF + 4 * s + kq];
          c[0][nt] = sf_mfma4(a0, b, c[0][nt]);
          c[1][nt] = sf_mfma4(a1, b, c[1][nt]);
        }
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) H1[(mt * 16 + 4 * kq + r) * FB_SH + nt * 16 + li] = fmaxf(c[mt][nt][r], 0.0f);
    }
    sf_wave_sync();
    // ---- B2: H2 = relu(H1 W1^T + b1)
    {
      f32x4 c[2][4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const float bv = W[FB_B1 + nt * 16 + li];
        c[0][nt] = f32x4{bv, bv, bv, bv};
        c[1][nt] = c[0][nt];
      }
#pragma unroll
      for (int s = 0; s < NGP_HID / 4; ++s) {
        const float a0 = H1[li * FB_SH + 4 * s + kq], a1 = H1[(16 + li) * FB_SH + 4 * s + kq];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const float b = W[FB_W1 + (nt * 16 + li) * FB_SH + 4 * s + kq];
          c[0][nt] = sf_mfma4(a0, b, c[0][nt]);
          c[1][nt] = sf_mfma4(a1, b, c[1][nt]);
        }
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) H2[(mt * 16 + 4 * kq + r) * FB_SH + nt * 16 + li] = fmaxf(c[mt][nt][r], 0.0f);
    }
    sf_wave_sync();
    // ---- C: output layer (4 wide, VALU) and d(out): lane = (point pl, half) owns outputs 2*half, 2*half + 1
    {
      float o0 = W[FB_B2 + 2 * half], o1 = W[FB_B2 + 2 * half + 1];
#pragma unroll 8
      for (int k = 0; k < NGP_HID; ++k) {
        const float hk = H2[pl * FB_SH + k];
        o0 = fmaf(W[FB_W2 + (2 * half) * NGP_HID + k], hk, o0);
        o1 = fmaf(W[FB_W2 + (2 * half + 1) * NGP_HID + k], hk, o1);
      }
      float d0 = 0.0f, d1 = 0.0f;
      if (live) {
        if (half == 0) {
          const float pre = o0 + ngp_blob(x);
          d0 = a.dsig[p] * expf(fminf(fmaxf(pre, -15.0f), 15.0f));            // trunc_exp backward
          const float sg = ngp_sigmoid(o1);
          d1 = a.drgb[p * 3 + 0] * sg * (1.0f - sg);
        } else {
          const float s0 = ngp_sigmoid(o0), s1 = ngp_sigmoid(o1);
          d0 = a.drgb[p * 3 + 1] * s0 * (1.0f - s0);
          d1 = a.drgb[p * 3 + 2] * s1 * (1.0f - s1);
        }
      }
      DO[pl * 4 + 2 * half] = d0;
      DO[pl * 4 + 2 * half + 1] = d1;
    }
    sf_wave_sync();
    // ---- D: dW2[o = kq][k = 4*li .. +3] += sum_p dout[p][o] h2[p][k];  db2
    {
      float b = 0.0f;
#pragma unroll 4
      for (int q = 0; q < FB_PTS; ++q) {
        const float dv = DO[q * 4 + kq];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc2[i] = fmaf(dv, H2[q * FB_SH + 4 * li + i], acc2[i]);
        b += dv;
      }
      if (li == 0) accb2 += b;
    }
    sf_wave_sync();
    // ---- E: dh2[p][k = lane] = (W2^T dout[p])[k] masked by h2 > 0, written over h2;  db1
    {
      const float w0 = W[FB_W2 + lane], w1 = W[FB_W2 + NGP_HID + lane], w2 = W[FB_W2 + 2 * NGP_HID + lane],
                  w3 = W[FB_W2 + 3 * NGP_HID + lane];
      float b = 0.0f;
#pragma unroll 4
      for (int q = 0; q < FB_PTS; ++q) {
        const f32x4 dv = *reinterpret_cast<const f32x4*>(DO + q * 4);
        const float g = fmaf(w3, dv[3], fmaf(w2, dv[2], fmaf(w1, dv[1], w0 * dv[0])));
        const float m = H2[q * FB_SH + lane] > 0.0f ? g : 0.0f;
        H2[q * FB_SH + lane] = m;
        b += m;
      }
      accb1 += b;
    }
    sf_wave_sync();
    // ---- F: dW1[j][k] += sum_p dh2[p][j] h1[p][k]   (A = dh2^T, B = h1; K = the 32 points)
#pragma unroll
    for (int s = 0; s < FB_PTS / 4; ++s) {
      float av[4], bv[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        av[t] = H2[(4 * s + kq) * FB_SH + t * 16 + li];
        bv[t] = H1[(4 * s + kq) * FB_SH + t * 16 + li];
      }
#pragma unroll
      for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) acc1[jt][kt] = sf_mfma4(av[jt], bv[kt], acc1[jt][kt]);
    }
    // ---- G: dh1 = (dh2 W1) masked by h1 > 0, written over h1 (phase F has read h1 already: same wave, program order)
    {
      f32x4 c[2][4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) { c[0][nt] = f32x4{0.f, 0.f, 0.f, 0.f}; c[1][nt] = c[0][nt]; }
#pragma unroll
      for (int s = 0; s < NGP_HID / 4; ++s) {
        const float a0 = H2[li * FB_SH + 4 * s + kq], a1 = H2[(16 + li) * FB_SH + 4 * s + kq];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const float b = W[FB_W1 + (4 * s + kq) * FB_SH + nt * 16 + li];
          c[0][nt] = sf_mfma4(a0, b, c[0][nt]);
          c[1][nt] = sf_mfma4(a1, b, c[1][nt]);
        }
      }
      sf_wave_sync();
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* hp = H1 + (mt * 16 + 4 * kq + r) * FB_SH + nt * 16 + li;
            *hp = *hp > 0.0f ? c[mt][nt][r] : 0.0f;
          }
    }
    sf_wave_sync();
    {  // db0[j = lane]
      float b = 0.0f;
#pragma unroll 8
      for (int q = 0; q < FB_PTS; ++q) b += H1[q * FB_SH + lane];
      accb0 += b;
    }
    // ---- H: dW0[j][f] += sum_p dh1[p][j] feat[p][f]
#pragma unroll
    for (int s = 0; s < FB_PTS / 4; ++s) {
      float av[4], bv[2];
#pragma unroll
      for (int t = 0; t < 4; ++t) av[t] = H1[(4 * s + kq) * FB_SH + t * 16 + li];
      bv[0] = F[(4 * s + kq) * FB_SF + li];
      bv[1] = F[(4 * s + kq) * FB_SF + 16 + li];
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) {
        acc0[jt][0] = sf_mfma4(av[jt], bv[0], acc0[jt][0]);
        acc0[jt][1] = sf_mfma4(av[jt], bv[1], acc0[jt][1]);
      }
    }
    // ---- I: d(feat) = dh1 W0 -> level-major [L][P][2] (zero for points outside the grid)
    if (a.dfeat_out) {
      f32x4 c[2][2];
      c[0][0] = c[0][1] = c[1][0] = c[1][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < NGP_HID / 4; ++s) {
        const float a0 = H1[li * FB_SH + 4 * s + kq], a1 = H1[(16 + li) * FB_SH + 4 * s + kq];
        const float b0 = W[FB_W0 + (4 * s + kq) * FB_SF + li], b1 = W[FB_W0 + (4 * s + kq) * FB_SF + 16 + li];
        c[0][0] = sf_mfma4(a0, b0, c[0][0]);
        c[0][1] = sf_mfma4(a0, b1, c[0][1]);
        c[1][0] = sf_mfma4(a1, b0, c[1][0]);
        c[1][1] = sf_mfma4(a1, b1, c[1][1]);
      }
      // The fragment layout gives a lane one feature of four points: stored directly that is 32 eight-byte pieces per store
      // instruction, scattered over 8 level planes.  Through the (now free) F tile instead: lane = (level lane >> 2, quarter
      // lane & 3) owns 8 consecutive points of one level = 64 contiguous bytes of the level-major [L][P][2] image.
      sf_wave_sync();                                    // phase H has read F
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int ft = 0; ft < 2; ++ft)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int q = mt * 16 + 4 * kq + r, f = ft * 16 + li;
            F[q * FB_SF + f] = INS[q] != 0.0f ? c[mt][ft][r] : 0.0f;
          }
      sf_wave_sync();
      {
        const uint32_t l = lane >> 2, q0 = (lane & 3) * 8;
        if (l < a.lv.L) {
          float* dst = a.dfeat_out + ((size_t)l * a.dfeat_P + a.p_off + p0 + q0) * 2;
          if (p0 + FB_PTS <= a.P && !((a.dfeat_P | a.p_off) & 1)) {      // whole trip, 16-byte aligned rows
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              f32x4 v;
              v[0] = F[(q0 + 2 * j) * FB_SF + 2 * l];     v[1] = F[(q0 + 2 * j) * FB_SF + 2 * l + 1];
              v[2] = F[(q0 + 2 * j + 1) * FB_SF + 2 * l]; v[3] = F[(q0 + 2 * j + 1) * FB_SF + 2 * l + 1];
              *reinterpret_cast<f32x4*>(dst + 4 * j) = v;
            }
          } else {                                         // ragged last trip (or an odd point count)
            for (int j = 0; j < 8; ++j)
              if (p0 + q0 + j < a.P) { dst[2 * j] = F[(q0 + j) * FB_SF + 2 * l]; dst[2 * j + 1] = F[(q0 + j) * FB_SF + 2 * l + 1]; }
          }
        }
      }
    }
    sf_wave_sync();                                      // before the next trip restages F / H1 / H2 / DO / INS
  }

  // ---- flush this wave's gradient tiles (row j = jt*16 + 4*kq + r, column li: 16 lanes = one 64-byte line per atomic request)
#pragma unroll
  for (int jt = 0; jt < 4; ++jt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = jt * 16 + 4 * kq + r;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) sf_global_add(a.g_w1 + j * NGP_HID + kt * 16 + li, acc1[jt][kt][r]);
      sf_global_add(a.g_w0 + j * NGP_FEAT + li, acc0[jt][0][r]);
      sf_global_add(a.g_w0 + j * NGP_FEAT + 16 + li, acc0[jt][1][r]);
    }
#pragma unroll
  for (int i = 0; i < 4; ++i) sf_global_add(a.g_w2 + kq * NGP_HID + 4 * li + i, acc2[i]);
  if (li == 0) sf_global_add(a.g_b2 + kq, accb2);
  sf_global_add(a.g_b1 + lane, accb1);
  sf_global_add(a.g_b0 + lane, accb0);
}

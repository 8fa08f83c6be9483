// Field forward of the fused Instant-NGP render on the matrix cores (gfx950), fp32 in / fp32 out.  EXPERIMENTAL, selected with
// SF_NGP_FWD_MFMA=1: parity-green (CPU threads: tests/test_hostemu_ngp_fwd.py; GPU: the reference-golden render and density tests
// of tests/test_gpu_ngp.py) but SLOWER than the default k_ngp_field in its first shape -- render forward 1.34 vs 1.12 ms at
// 16 384 rays (r02, one run): 94 KB of LDS per workgroup leave one wave per SIMD, so the 64 table gathers per lane are exposed,
// and the forward has no wgrad / dgrad GEMMs to amortise them over as the backward does.  Next: 16-point tiles (8 waves per
// workgroup on one weight image), or the VALU kernel's occupancy with the hidden layers on MFMA through registers.
//
// What it computes per sample point (external/nerf/network_grid.py:77-104 through common_forward): hash-grid features ->
// h1 = relu(W0 f + b0) -> h2 = relu(W1 h1 + b1) -> out = W2 h2 + b2; sigma = trunc_exp(out0 + blob), albedo = sigmoid(out1..3).
// Same decomposition as the backward kernel (ngp_bwd_mfma.h, whose first half this is): one WAVE owns 32 points per trip, lane =
// (point, half) encodes 8 of the 16 levels into a wave-private LDS tile, the two hidden layers are fp32 GEMMs on
// v_mfma_f32_16x16x4_f32 (exact fp32 products; results equal the fmaf chains of ngp_mlp_forward up to summation order), the
// 4-wide output layer and the activations stay on the VALU.  Sample depths are produced by the same ngp_coarse_z expression as
// k_ngp_field<0>, so positions -- and cell indices -- stay bit-exact.
#pragma once
#include "sf_dev.h"
#include "ngp_device.h"
#include "ngp_bwd_mfma.h"          // FB_* : padded LDS image of the MLP weights, tile strides

#define FF_WAVE_FLOATS (2 * FB_PTS * FB_SH + FB_PTS)     // H1 [32][65] | F [32][33] then H2 [32][65] in the same tile | inside flags
#define FF_LDS_FLOATS (FB_WTOT + 4 * FF_WAVE_FLOATS)

struct FFArgs {
  const float* table; const float* w0; const float* b0; const float* w1; const float* b1; const float* w2; const float* b2;
  float bound;
  NgpLevels lv;
  const float* rays_o; const float* rays_d; const float* aabb; const float* nears; const float* fars;
  const float* lin; const float* u;      // mode 0: stratified coarse rule (u null: midpoints)
  const float* z_in;                     // mode 1: depths given
  uint32_t P, T;
  int mode;
  float* z_out;                          // mode 0: the depths it used
  float* sigma; float* rgb;              // [P], [P][3]
};

SF_KERNEL(256) void k_ngp_field_fwd_mfma(FFArgs a) {
  SF_DYN_LDS(lds_raw);
  float* W = reinterpret_cast<float*>(lds_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* H1 = W + FB_WTOT + wave * FF_WAVE_FLOATS;    // [32][65]
  float* F = H1 + FB_PTS * FB_SH;                      // [32][33] features ...
  float* H2 = F;                                       // ... overwritten by h2 [32][65] once h1 exists
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
  const uint32_t n_trips = (a.P + FB_PTS - 1) / FB_PTS;
  for (uint32_t trip = blockIdx.x * 4 + wave; trip < n_trips; trip += gridDim.x * 4) {
    const uint32_t p0 = trip * FB_PTS;
    // ---- A: depth, position, hash-grid features; lane = (point pl, half): each half encodes 8 of the 16 levels
    const int pl = lane & 31, half = lane >> 5;
    const uint32_t p = p0 + pl;
    const bool live = p < a.P;
    float x[3] = {0.f, 0.f, 0.f}, x01[3] = {0.f, 0.f, 0.f};
    bool inside = false;
    if (live) {
      const uint32_t n = p / a.T, k = p - n * a.T;
      float z;
      if (a.mode == 0) {
        z = ngp_coarse_z(a.nears[n], a.fars[n], a.lin[k], a.u ? a.u[p] : -1.0f, a.T);
        if (half == 0) a.z_out[p] = z;
      } else {
        z = a.z_in[p];
      }
      const float o[3] = {a.rays_o[n * 3], a.rays_o[n * 3 + 1], a.rays_o[n * 3 + 2]};
      const float d[3] = {a.rays_d[n * 3], a.rays_d[n * 3 + 1], a.rays_d[n * 3 + 2]};
      ngp_point(o, d, z, box, x);
      inside = ngp_unit(x, a.bound, x01);
    }
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
    // ---- B2: H2 = relu(H1 W1^T + b1)   (written over the feature tile)
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
    // ---- C: output layer (4 wide, VALU) + activations: lane = (point pl, half) owns outputs 2*half, 2*half + 1
    {
      float o0 = W[FB_B2 + 2 * half], o1 = W[FB_B2 + 2 * half + 1];
#pragma unroll 8
      for (int k = 0; k < NGP_HID; ++k) {
        const float hk = H2[pl * FB_SH + k];
        o0 = fmaf(W[FB_W2 + (2 * half) * NGP_HID + k], hk, o0);
        o1 = fmaf(W[FB_W2 + (2 * half + 1) * NGP_HID + k], hk, o1);
      }
      if (live) {
        if (half == 0) {
          a.sigma[p] = expf(o0 + ngp_blob(x));
          a.rgb[p * 3 + 0] = ngp_sigmoid(o1);
        } else {
          a.rgb[p * 3 + 1] = ngp_sigmoid(o0);
          a.rgb[p * 3 + 2] = ngp_sigmoid(o1);
        }
      }
    }
    sf_wave_sync();                                     // the next trip overwrites the tiles
  }
}

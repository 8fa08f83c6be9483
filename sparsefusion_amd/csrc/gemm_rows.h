// y[M, N] = act(x[M, K] W^T + bias) for 9..64 rows: the sampler's time table (Unet.time_table: 51 log-snr rows through the time
// MLPs, the 27 time_mlp Linears batched into one [33 792 x 1 024] matrix, and the k / v of the time tokens:
// external/imagen_pytorch.py:1514-1604).  k_gemv takes 8 rows per launch and re-streams the weights for every 8: 7 x 110 us for
// the big matrix of a 51-row table.  Here the rows are the M side of v_mfma_f32_16x16x32_bf16 and the weights are read ONCE:
//   * W is [N][Kp] bf16 row-major, which IS the B-fragment layout (lane (n, g) needs W[n0 + n][k0 + 8 g .. + 7]: 16 contiguous bytes);
//   * x stays fp32-accurate: it is split into bf16 hi + lo parts in LDS (x = hi + lo to 16 mantissa bits) and each k-step runs
//     two MFMAs -- the products match k_gemv's fp32 x times bf16 w to ~1e-5, so the table is the same whichever kernel made it;
//   * one workgroup = 64 output columns (one 16-column fragment per wave) x all rows, K in chunks of 128 through LDS.
// Written against sf_dev.h so that tests/hostemu runs the same source on CPU threads (tests/test_hostemu_gemm_rows.py).
#pragma once
#include "sf_dev.h"

#define GR_KC 128
#define GR_LD (GR_KC + 8)          // bf16 elements per LDS row: 272 B = 17 x 16 B, conflict-free 16-byte fragment reads; 2 x 17 KB of LDS

struct GemmRowsArgs {
  const float* x;                  // [M][ldx]
  const sf_opnd* W;                 // [N][Kp], Kp = K rounded up to 8, zero padded
  const float* bias;               // [N] or null
  float* y;                        // [M][ldy]
  int M, N, K, Kp, ldx, ldy, in_silu, out_act;     // out_act: 0 none, 1 SiLU, 2 sigmoid
};

SF_KERNEL(256) void k_gemm_rows(GemmRowsArgs a) {
  SF_SHARED __attribute__((aligned(16))) sf_opnd hi[64 * GR_LD];
  SF_SHARED __attribute__((aligned(16))) sf_opnd lo[64 * GR_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, g = lane >> 4;
  const int ncol = blockIdx.x * 64 + wave * 16 + n;
  const sf_opnd* __restrict__ wrow = a.W + (long)min(ncol, a.N - 1) * a.Kp;
  const int MF = (a.M + 15) >> 4;
  f32x4 acc[4];
#pragma unroll
  for (int mf = 0; mf < 4; ++mf) acc[mf] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int kc = 0; kc < a.K; kc += GR_KC) {
    // ---- stage rows [0, 64) x columns [kc, kc + 128) as bf16 hi / lo; loads from clamped addresses, mask on the value
#pragma unroll 2
    for (int it = 0; it < 64 * GR_KC / 8 / 256; ++it) {
      const int idx = tid + it * 256;
      const int m = idx / (GR_KC / 8), k8 = (idx % (GR_KC / 8)) * 8;
      const float* __restrict__ xr = a.x + (long)min(m, a.M - 1) * a.ldx;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = xr[min(kc + k8 + j, a.K - 1)];
      bf16x8 h, l;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float t = (m < a.M && kc + k8 + j < a.K) ? v[j] : 0.0f;
        if (a.in_silu) t = sf_silu(t);
        const sf_opnd hb = (sf_opnd)t;
        h[j] = hb;
        l[j] = (sf_opnd)(t - (float)hb);
      }
      *reinterpret_cast<bf16x8*>(&hi[m * GR_LD + k8]) = h;
      *reinterpret_cast<bf16x8*>(&lo[m * GR_LD + k8]) = l;
    }
    sf_sync();
    const int left = a.K - kc;
    const int steps = left >= GR_KC ? GR_KC / 32 : (left + 31) / 32;
    for (int ks = 0; ks < steps; ++ks) {
      const int k0 = kc + ks * 32 + 8 * g;
      bf16x8 b = *reinterpret_cast<const bf16x8*>(wrow + min(k0, a.Kp - 8));
      if (k0 >= a.Kp) b = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
      const int col = ks * 32 + 8 * g;
#pragma unroll
      for (int mf = 0; mf < 4; ++mf) {
        if (mf < MF) {
          const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&hi[(mf * 16 + n) * GR_LD + col]);
          const bf16x8 al = *reinterpret_cast<const bf16x8*>(&lo[(mf * 16 + n) * GR_LD + col]);
          acc[mf] = sf_mfma16(ah, b, acc[mf]);
          acc[mf] = sf_mfma16(al, b, acc[mf]);
        }
      }
    }
    sf_sync();
  }
  // D[m = 4 g + r][n]: row mf * 16 + 4 g + r, column ncol
  if (ncol < a.N) {
    const float bv = a.bias ? a.bias[ncol] : 0.0f;
#pragma unroll
    for (int mf = 0; mf < 4; ++mf)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = mf * 16 + 4 * g + r;
        if (m < a.M) {
          float v = acc[mf][r] + bv;
          if (a.out_act == 1) v = sf_silu(v);
          else if (a.out_act == 2) v = sf_sigmoid(v);
          a.y[(long)m * a.ldy + ncol] = v;
        }
      }
  }
}

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

// k_gemm_rows_ks (r06): the same product for the GlobalContext MLPs of the large-batch plans (rows = the 9 .. 64 images of the batch, N and K
// 128 .. 1024).  k_gemm_rows runs N / 64 workgroups -- 8 for a 1024 -> 512 layer -- that each walk K in 128-column chunks behind two
// workgroup barriers per chunk: 23 us for [32 x 1024] x [1024 x 512], 26 of them in a B = 32 eval.  Here a workgroup owns 16 output columns
// (N / 16 workgroups) and its 4 waves split K by chunk (wave w: chunks w, w + 4, ...): a wave stages ITS chunk for itself in a wave-private LDS
// slice (same-wave DS order, no workgroup barrier inside the K loop), so the dependent chain is 4x shorter and every staging load of a chunk is
// in flight at once; one LDS reduction over the 4 waves at the end.  Only the rows that exist are staged.  Same arithmetic per product as
// k_gemm_rows (hi + lo split of x, bf16 weights), another summation order over K (per wave, then over waves).
// NW waves per workgroup: 4 (up to 64 rows) or 8 (up to 32 rows: a 1024-column K is then ONE chunk per wave -- no second round trip behind the
// first chunk's MFMAs; the slices hold 32 rows).  Dynamic LDS: NW hi + lo slices of ROWS rows + the reduction buffer [NW - 1][MFMAX x 4][64].
template <int NW>
struct GemmRowsKs {
  static constexpr int ROWS = NW == 8 ? 32 : 64, MFMAX = ROWS / 16;
  static constexpr int WAVE_ELEMS = 2 * ROWS * GR_LD;                      // hi + lo slices of one wave, operand-type elements
  static constexpr int RED = MFMAX * 4 * 64;                               // floats per wave in the reduction buffer
  static constexpr int LDS_BYTES = NW * WAVE_ELEMS * 2 + (NW - 1) * RED * 4;
  static_assert(LDS_BYTES <= 163840, "LDS");
};

template <int NW>
SF_KERNEL(NW * 64) void k_gemm_rows_ks(GemmRowsArgs a) {
  using Gk = GemmRowsKs<NW>;
  constexpr int MFMAX = Gk::MFMAX;
  SF_DYN_LDS(lds);
  const int tid = threadIdx.x, lane = tid & 63, wave = sf_uniform(tid >> 6);
  sf_opnd* hi = reinterpret_cast<sf_opnd*>(lds) + wave * Gk::WAVE_ELEMS;
  sf_opnd* lo = hi + Gk::ROWS * GR_LD;
  float* red = reinterpret_cast<float*>(lds + NW * Gk::WAVE_ELEMS * 2);
  const int n = lane & 15, g = lane >> 4;
  const int ncol = blockIdx.x * 16 + n;
  const sf_opnd* __restrict__ wrow = a.W + (long)min(ncol, a.N - 1) * a.Kp;
  const int MF = (a.M + 15) >> 4;
  const bool vec = (a.ldx & 3) == 0 && (a.K & 7) == 0 && (reinterpret_cast<uintptr_t>(a.x) & 15) == 0;      // float4 loads of whole 8-column pieces
  const float bv = (a.bias ? a.bias : reinterpret_cast<const float*>(a.W))[a.bias ? min(ncol, a.N - 1) : 0];      // requested at entry (an unconditional load from a selected address), used by wave 0's epilogue
  f32x4 acc[4];
#pragma unroll
  for (int mf = 0; mf < 4; ++mf) acc[mf] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int kc = wave * GR_KC; kc < a.K; kc += NW * GR_KC) {
    bf16x8 bw[GR_KC / 32];                          // the chunk's weight fragments, requested before its x rows are staged
#pragma unroll
    for (int ks = 0; ks < GR_KC / 32; ++ks) {
      const int k0 = kc + ks * 32 + 8 * g;
      bw[ks] = *reinterpret_cast<const bf16x8*>(wrow + min(k0, a.Kp - 8));
      if (k0 >= a.Kp) bw[ks] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
    const int lim = MF * 16 * (GR_KC / 8);
    if (vec && kc + GR_KC <= a.K) {
      // a whole chunk, float4 rows: ALL loads of the chunk first (a rolled loop waits for each pair before the next goes out: 8 L2 round
      // trips per chunk at 32 rows -- measured 11 us per launch), then the hi / lo split
      f32x4 xv[4 * MFMAX][2];
#pragma unroll
      for (int it = 0; it < 4 * MFMAX; ++it) {
        const int idx = lane + it * 64;
        if (idx < lim) {
          const int m = idx / (GR_KC / 8), k8 = (idx % (GR_KC / 8)) * 8;
          const float* __restrict__ xr = a.x + (long)min(m, a.M - 1) * a.ldx + kc + k8;
          xv[it][0] = *reinterpret_cast<const f32x4*>(xr);
          xv[it][1] = *reinterpret_cast<const f32x4*>(xr + 4);
        }
      }
#pragma unroll
      for (int it = 0; it < 4 * MFMAX; ++it) {
        const int idx = lane + it * 64;
        if (idx < lim) {
          const int m = idx / (GR_KC / 8), k8 = (idx % (GR_KC / 8)) * 8;
          bf16x8 h, l;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float t = m < a.M ? xv[it][j >> 2][j & 3] : 0.0f;
            if (a.in_silu) t = sf_silu(t);
            const sf_opnd hb = (sf_opnd)t;
            h[j] = hb;
            l[j] = (sf_opnd)(t - (float)hb);
          }
          *reinterpret_cast<bf16x8*>(&hi[m * GR_LD + k8]) = h;
          *reinterpret_cast<bf16x8*>(&lo[m * GR_LD + k8]) = l;
        }
      }
    } else
    for (int idx = lane; idx < MF * 16 * (GR_KC / 8); idx += 64) {
      const int m = idx / (GR_KC / 8), k8 = (idx % (GR_KC / 8)) * 8;
      const float* __restrict__ xr = a.x + (long)min(m, a.M - 1) * a.ldx;
      float v[8];
      if (vec && kc + k8 + 8 <= a.K) {
        const f32x4 p = *reinterpret_cast<const f32x4*>(xr + kc + k8), q = *reinterpret_cast<const f32x4*>(xr + kc + k8 + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = p[j]; v[4 + j] = q[j]; }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = xr[min(kc + k8 + j, a.K - 1)];
      }
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
    sf_wave_sync();
    const int left = a.K - kc;
    const int steps = left >= GR_KC ? GR_KC / 32 : (left + 31) / 32;
#pragma unroll
    for (int ks = 0; ks < GR_KC / 32; ++ks) {
      if (ks < steps) {
        const bf16x8 b = bw[ks];
        const int col = ks * 32 + 8 * g;
#pragma unroll
        for (int mf = 0; mf < MFMAX; ++mf) {
          if (mf < MF) {
            const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&hi[(mf * 16 + n) * GR_LD + col]);
            const bf16x8 al = *reinterpret_cast<const bf16x8*>(&lo[(mf * 16 + n) * GR_LD + col]);
            acc[mf] = sf_mfma16(ah, b, acc[mf]);
            acc[mf] = sf_mfma16(al, b, acc[mf]);
          }
        }
      }
    }
    sf_wave_sync();                                 // this wave's fragment reads are done before its next chunk overwrites the slice
  }
  if (wave > 0) {
#pragma unroll
    for (int mf = 0; mf < MFMAX; ++mf)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(wave - 1) * Gk::RED + (mf * 4 + r) * 64 + lane] = acc[mf][r];
  }
  sf_sync();
  if (wave != 0 || ncol >= a.N) return;
#pragma unroll
  for (int mf = 0; mf < MFMAX; ++mf)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = mf * 16 + 4 * g + r;
      if (m < a.M) {
        const int o = (mf * 4 + r) * 64 + lane;
        float v = acc[mf][r];
#pragma unroll
        for (int w = 0; w < NW - 1; ++w) v += red[w * Gk::RED + o];
        v += a.bias ? bv : 0.0f;
        if (a.out_act == 1) v = sf_silu(v);
        else if (a.out_act == 2) v = sf_sigmoid(v);
        a.y[(long)m * a.ldy + ncol] = v;
      }
    }
}

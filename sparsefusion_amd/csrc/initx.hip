// The latent half of the UNet's init conv inside a sampler trajectory: x0 = base + CrossEmbed(x) where x is the 4-channel
// latent and `base` holds the conditioning half + bias, evaluated once per trajectory (external/imagen_pytorch.py:1017-1042:
// CrossEmbedLayer = three convs k = 3 / 7 / 15 into channel slices; the conv is linear in its input channels).
//
// The first plan ran this through the implicit-GEMM kernel: pack (NCHW -> NHWC padded to 32 channels), three launches with
// 7/8 of every MFMA k-step multiplying zeros, a split-K reduction -- five dependent launches, ~45 us, for 0.3 GFLOP.  It is a
// direct convolution on the vector units instead: one workgroup = four waves on one 8 x 8 pixel tile, whose haloed 4-channel
// patch (22 x 22 x 4 floats) sits in LDS; a wave = one (conv, QC output channels) unit: lane <-> pixel, QC accumulators per
// lane, the weights of the unit are WAVE-UNIFORM (scalar loads, one contiguous [tap][QC] block per unit) -- per tap one LDS read and QC
// v_fmac with a scalar operand.  Units are sized to equal work (k = 15: 2 channels, k = 7: 8, k = 3: 32).  fp32 throughout
// (closer to the fp32 reference than the bf16 MFMA path it replaces; same tolerance in the tests).
#include "sf_common.h"

#define IX_TILE 8
#define IX_HALO 7
#define IX_PW (IX_TILE + 2 * IX_HALO)     /* 22 */

struct InitXArgs {
  const float* x;        // [B][Cx][H][W]
  const float* base;     // [B*H*W][ld]
  const float* w;        // conv i at w + woff[i]: [cw_i / QC_i units][Cx * k_i * k_i taps][QC_i] fp32, QC = 32 / 8 / 2 for k = 3 / 7 / 15
  float* out;            // [B*H*W][ld]
  int B, H, W, Cx, ld;
  int cw[3], co[3], woff[3];
  int units[3];          // wave-units per tile of each conv = cw / QC
  int wgs_per_tile;
};

template <int K, int QC>
__device__ __forceinline__ void initx_unit(const InitXArgs& a, const float* __restrict__ patch, int conv, int q0, int lane, long m0) {
  const int py = lane >> 3, px = lane & 7;
  // the unit's weights are ONE contiguous block [tap][QC] (a [tap][all channels] table made every scalar load of a 2-channel
  // unit touch its own cache line: 3600 lines per unit through a 16 KB scalar cache)
  const float* __restrict__ w = a.w + a.woff[conv] + (long)(q0 / QC) * (a.Cx * K * K * QC);
  float acc[QC];
#pragma unroll
  for (int q = 0; q < QC; ++q) acc[q] = 0.0f;
  constexpr int OFF = IX_HALO - K / 2;
  for (int ci = 0; ci < a.Cx; ++ci) {
    const float* pc = patch + ci * IX_PW * IX_PW + (py + OFF) * IX_PW + px + OFF;
    for (int ky = 0; ky < K; ++ky) {
#pragma unroll
      for (int kx = 0; kx < K; ++kx) {
        const float xv = pc[ky * IX_PW + kx];
        const float* __restrict__ wt = w + ((ci * K + ky) * K + kx) * QC;             // wave-uniform: scalar loads
#pragma unroll
        for (int q = 0; q < QC; ++q) acc[q] = fmaf(xv, wt[q], acc[q]);
      }
    }
  }
  const long m = m0 + (long)py * a.W + px;
  const float* __restrict__ bp = a.base + m * a.ld + a.co[conv] + q0;
  float* __restrict__ op = a.out + m * a.ld + a.co[conv] + q0;
  float2 b2[QC / 2];                 // all loads before the first store: `out` may alias `base` as far as the compiler knows
#pragma unroll
  for (int q = 0; q < QC; q += 2) b2[q / 2] = *reinterpret_cast<const float2*>(bp + q);
#pragma unroll
  for (int q = 0; q < QC; q += 2) *reinterpret_cast<float2*>(op + q) = make_float2(b2[q / 2].x + acc[q], b2[q / 2].y + acc[q + 1]);
}

__global__ __launch_bounds__(256) void k_init_x(InitXArgs a) {
  __shared__ float patch[8 * IX_PW * IX_PW];            // up to 8 latent channels
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // in an SGPR: the unit's weight addresses are then scalar loads
  const int tiles_x = a.W / IX_TILE, tiles = tiles_x * (a.H / IX_TILE);
  const int g = blockIdx.x % a.wgs_per_tile;
  const int bt = blockIdx.x / a.wgs_per_tile;
  const int b = bt / tiles, t = bt - b * tiles;
  const int ty = t / tiles_x, tx = t - ty * tiles_x;
  const int y0 = ty * IX_TILE, x0 = tx * IX_TILE;
  // haloed patch, zero outside the image
  for (int i = tid; i < a.Cx * IX_PW * IX_PW; i += 256) {
    const int ci = i / (IX_PW * IX_PW), r = i - ci * IX_PW * IX_PW;
    const int fy = r / IX_PW, fx = r - fy * IX_PW;
    const int y = y0 - IX_HALO + fy, x = x0 - IX_HALO + fx;
    patch[i] = (y >= 0 && y < a.H && x >= 0 && x < a.W) ? a.x[(((long)b * a.Cx + ci) * a.H + y) * a.W + x] : 0.0f;
  }
  __syncthreads();
  // this wave's unit: units are listed conv 2 (k = 15) first, then conv 1, then conv 0
  int u = g * 4 + wave;
  const long m0 = (long)b * a.H * a.W + (long)y0 * a.W + x0;
  if (u < a.units[2]) { initx_unit<15, 2>(a, patch, 2, u * 2, lane, m0); return; }
  u -= a.units[2];
  if (u < a.units[1]) { initx_unit<7, 8>(a, patch, 1, u * 8, lane, m0); return; }
  u -= a.units[1];
  if (u < a.units[0]) initx_unit<3, 32>(a, patch, 0, u * 32, lane, m0);
}

// op: p 0 x  1 base  2 weights  3 out ; i 0 B  1 H  2 W  3 Cx  4 ld  5..7 cw  8..10 channel offsets  11..13 weight offsets (floats)
int sf_plan_initx_op(const sf_op* op, void* stream) {
  InitXArgs a;
  a.x = (const float*)op->p[0]; a.base = (const float*)op->p[1]; a.w = (const float*)op->p[2]; a.out = (float*)op->p[3];
  a.B = op->i[0]; a.H = op->i[1]; a.W = op->i[2]; a.Cx = op->i[3]; a.ld = op->i[4];
  for (int k = 0; k < 3; ++k) { a.cw[k] = op->i[5 + k]; a.co[k] = op->i[8 + k]; a.woff[k] = op->i[11 + k]; }
  if (!a.x || !a.base || !a.w || !a.out || a.B < 1) SF_FAIL(SF_ERR_INVALID, "init_x: missing operand");
  if (a.H % IX_TILE || a.W % IX_TILE || a.Cx < 1 || a.Cx > 8) SF_FAIL(SF_ERR_INVALID, "init_x: H, W multiples of 8 and 1..8 latent channels");
  if (a.cw[0] % 32 || a.cw[1] % 8 || a.cw[2] % 2 || a.ld % 2 || (a.co[0] | a.co[1] | a.co[2]) % 2)
    SF_FAIL(SF_ERR_INVALID, "init_x: channel slices must be multiples of 32 / 8 / 2 (k = 3 / 7 / 15)");
  a.units[0] = a.cw[0] / 32; a.units[1] = a.cw[1] / 8; a.units[2] = a.cw[2] / 2;
  const int units = a.units[0] + a.units[1] + a.units[2];
  a.wgs_per_tile = (units + 3) / 4;
  const uint32_t grid = (uint32_t)a.B * (a.H / IX_TILE) * (a.W / IX_TILE) * a.wgs_per_tile;
  k_init_x<<<grid, 256, 0, (hipStream_t)stream>>>(a);
  SF_CHECK_LAUNCH("init_x");
  return SF_OK;
}

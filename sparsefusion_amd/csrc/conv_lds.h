// ConvArgs (shared by k_conv_igemm in unet_ops.hip) and the LDS-tiled large-M implicit GEMM k_conv_lds.  Written against
// sf_dev.h so that tests/hostemu runs the same source on CPU threads (tests/test_hostemu_conv_lds.py).
#pragma once
#include "sf_dev.h"

struct ConvArgs {
  const void* in; const bf16x8* w; const float* bias; float* out; const float* resid; float* ws;
  int accum, npad;
  int B, H, W, Cin, Ho, Wo, Cout, ldc, co_off, kh, kw, stride, pad, groups;
  int KS, cchunks, m_frags, n_frags, m_tiles, n_tiles, steps_per_wave;
  int pixshuf, ups, relu;   // relu: 0 none, 1 ReLU, 2 GELU (erf)
  float* slots_out;         // k_conv_igemm pixel-shuffle epilogue only, or null: (sum, sum of squares) slots [Mout/16][ldc/16][2] of the
                            // written (shuffled, SiLU'd) tensor for the next GroupNorm-fused conv -- input fragment (m-frag, n-frag) -> slot
                            // row 4 * m-frag + (n-frag & 3), column (co_off / 16) + n-frag / 4: its 16 conv channels are 4 output channels of
                            // ONE 16-channel column, its pixels stay inside one image; the consumer sums all slots of an (image, group)
  int out_nchw_hw;          // > 0: the split-K reduction writes out[(b * Cout + n) * hw + p] (plan output, NCHW) instead of rows
};

// ---------------------------------------------------------------------------------------------
// Large-M implicit GEMM (VAE, LPIPS-VGG, B >= 4): workgroup tile 128 pixels x 16*BNF channels, staged through LDS.
// The weight-streaming kernel above keeps every fragment private to a wave (right when M is one or two tiles and the
// weights are read once); at M = 4 096 .. 131 072 the same fragments are needed by several waves, and private loads
// make the kernel L1-bandwidth bound at ~4 % of the MFMA peak.  Here the 4 waves of a workgroup load each A / B
// fragment ONCE per stage (two k-steps = 64 input channels) in MFMA lane order -- so a fragment is a contiguous,
// conflict-free 1 KiB of LDS -- and every wave re-reads the 4 + BNF/2 fragments of its 64 x (8*BNF) sub-tile with
// ds_read_b128.  Global loads of stage s+1 are issued into registers before the MFMAs of stage s (one barrier per
// stage).  The workgroup -> tile map is XCD-aware: consecutive workgroups go round-robin over the 8 XCDs, so each
// XCD is handed a contiguous range of tiles (neighbouring pixels and all channel tiles of a pixel tile share one
// L2 -- the im2col reuse of a 3x3 conv is 9x in A).
// No split-K, no pixel shuffle; epilogue = bias (+ residual) (+ accumulate) (+ ReLU).
// ---------------------------------------------------------------------------------------------
#define CONV_LDS_NAME k_conv_lds
#define CONV_LDS_GN 0
#include "conv_lds_body.inc"
#undef CONV_LDS_NAME
#undef CONV_LDS_GN

// k_conv_lds_gn (the VAE default since r03; SF_VAE_GN_EPI=0 plans the statistics pass instead): the same kernel whose epilogue also leaves per-(pixel tile, GroupNorm group) partial sums (sum, sum of squares) of the values it
// writes, in double, at gn_part[(mt * (Cout / gn_cg) + group) * 2] -- no atomics; k_gn_finalize adds the tiles of an image up
// into the statistics k_gn_apply reads, which makes the separate statistics pass over the tensor (k_gn_stats_px) unnecessary.
// Needs: the conv writes whole rows of the tensor (co_off = 0, Cout = ldc), gn_cg in {4, 8, 16}, 128 | Ho * Wo.
#define CONV_LDS_NAME k_conv_lds_gn
#define CONV_LDS_GN 1
#include "conv_lds_body.inc"
#undef CONV_LDS_NAME
#undef CONV_LDS_GN

// stats[b][g] = sum over the pixel tiles of image b of the partials above: one workgroup per (image, group), thread = tile (one or two
// loads each), then a fixed-order tree -- reproducible, and a launch-sized kernel instead of one workgroup walking 512 tiles x 32
// groups (measured ~10 us per GroupNorm, 48 of them in the SD-VAE).
SF_KERNEL(256) void k_gn_finalize(const double* __restrict__ part, double* __restrict__ stats, int tiles_per_image, int G) {
  SF_SHARED double red[2][256];
  const int b = blockIdx.x / G, g = blockIdx.x - b * G, tid = threadIdx.x;
  double s = 0.0, q = 0.0;
  for (int t = tid; t < tiles_per_image; t += 256) {
    const double* p = part + (((long)b * tiles_per_image + t) * G + g) * 2;
    s += p[0]; q += p[1];
  }
  red[0][tid] = s; red[1][tid] = q;
  sf_sync();
  for (int w = 128; w > 0; w >>= 1) {
    if (tid < w) { red[0][tid] += red[0][tid + w]; red[1][tid] += red[1][tid + w]; }
    sf_sync();
  }
  if (tid == 0) {
    stats[((long)b * G + g) * 2] = red[0][0];
    stats[((long)b * G + g) * 2 + 1] = red[1][0];
  }
}

// Launcher of SF_OP_INITX; the kernel lives in initx.h (shared with the CPU-thread emulation of tests/hostemu).
#include "sf_common.h"
#include "initx.h"

// op: p 0 x  1 base  2 weight fragments  3 out  4 statistics slots of out or null ; i 0 B  1 H  2 W  3 Cx  4 ld  5..7 cw  8..10 channel offsets  11..13 weight
// offsets (fragments).  Channel slices: cw[0] in {64, 128}, cw[1], cw[2] in {32, 64} (the CrossEmbed split of dim 128 / 256).
int sf_plan_initx_op(const sf_op* op, void* stream) {
  InitXArgs a;
  a.x = (const float*)op->p[0]; a.base = (const float*)op->p[1]; a.w = (const ix_bf16x8*)op->p[2]; a.out = (float*)op->p[3];
  a.slots = (float*)op->p[4];
  a.B = op->i[0]; a.H = op->i[1]; a.W = op->i[2]; a.Cx = op->i[3]; a.ld = op->i[4];
  for (int k = 0; k < 3; ++k) { a.cw[k] = op->i[5 + k]; a.co[k] = op->i[8 + k]; a.woff[k] = op->i[11 + k]; }
  if (!a.x || !a.base || !a.w || !a.out || a.B < 1) SF_FAIL(SF_ERR_INVALID, "init_x: missing operand");
  if (a.H % IX_TILE || a.W % IX_TILE || a.Cx < 1 || a.Cx > 4) SF_FAIL(SF_ERR_INVALID, "init_x: H, W multiples of 8 and 1..4 latent channels");
  if ((a.cw[0] != 64 && a.cw[0] != 128) || (a.cw[1] != 32 && a.cw[1] != 64) || (a.cw[2] != 32 && a.cw[2] != 64))
    SF_FAIL(SF_ERR_INVALID, "init_x: channel slices (%d, %d, %d) not instantiated", a.cw[0], a.cw[1], a.cw[2]);
  if (a.slots && (a.ld % 16 || a.co[0] % 16 || a.co[1] % 16 || a.co[2] % 16)) SF_FAIL(SF_ERR_INVALID, "init_x: slots need 16-aligned channel slices");
  const uint32_t grid = (uint32_t)a.B * (a.H / IX_TILE) * (a.W / IX_TILE) * 3;
  k_init_x<<<grid, 256, 0, (hipStream_t)stream>>>(a);
  SF_CHECK_LAUNCH("init_x");
  return SF_OK;
}

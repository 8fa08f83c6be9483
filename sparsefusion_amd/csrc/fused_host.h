// Host-side decoding of the fused ops (SF_OP_FCONV / SF_OP_SLOTS): operand checks, tile geometry, LDS layout.
// Shared by the gfx950 launchers (unet_fused.hip) and the CPU kernel-logic harness (tests/hostemu/fused_emu.cpp).
//
// SF_OP_FCONV operands
//   p: 0 s1.p  1 s1.a  2 s1.b  3 s1.r  4 s1.slots  5 s2.p  6 s2.slots  7 packed weights  8 bias  9 out  10 resid
//      11 split-K slabs (S > 1)  12 slots_out  13 gamma / LN gain  14 beta / LN bias  15 scale_shift
//      16 debug: [grid][8] int64 phase timestamps (100 MHz), normally null
//      17 GlobalContext to_k weight [Cout]  18 partial context logits [S * n_frags][M]   (both or neither)
//   i: 0 B  1 H  2 W  3 C1  4 C2  5 Cout  6 ldc  7 co_off  8 k (1 | 3)  9 s1.mode  10 s1.groups  11 s1.npad
//      12 norm (FNORM_*)  13 G  14 TR (image rows per tile)  15 WM  16 WN  17 S (input-channel slices)  18 ss_stride
//   flags: 1 SiLU after the norm, 2 GELU before the LayerNorm, 4 accumulate into out, 8 GELU on the final output,
//          32 pipelined kernel (k_conv_fused_pipe: slot GroupNorm, k = 3, plain source, C % 128 == 0),
//          16 pair: the NEXT op (an un-normalised fconv of the same tile shape) runs in the same launch (k_conv_fused_pair)
//          64 GlobalContext pooling in the epilogue (k_conv_fused_pipe<.., POOL>), 128 keep the general kernel where k_conv4_gn would take the op
//   f: 0 eps  1 s1.scale  2 s2.scale
//   norm == FNORM_ATTN (the attention core as the prologue of its output projection; k = 1, 4x4 map, C1 = 512 = 8 heads x 64):
//      p: 0 q rows [B * 16][ldq] (instead of a source tensor)  19..21 key pointers of the <= 3 key / value segments
//      i: 19 ldq (other norms: bit 0 = keep one image per workgroup where k_conv4_gn_mb would take the op)  20 + 4 s .. 23 + 4 s: rows, row_stride, batch_stride, head_stride of segment s (rows = 0: unused)
//      f: 3 + s: value offset of segment s in floats (v = k + offset)  6: softmax scale
// SF_OP_SLOTS operands
//   p: 0 x (or h)  1 gate [B, C] or null  2 res  3 out (gate / split-K mode)  4 slots  5 split-K slabs or null  6 conv bias or null
//   i: 0 M  1 C  2 HW  3 slab groups  4 slab row stride (npad)
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../include/sparsefusion_hip.h"
#include "fused_kernels.h"
#include "fused_pipe.h"
#include "fused_gca.h"
#include "fused_conv4.h"
#include "fused_conv3s.h"

#define SF_LDS_MAX 163840
#ifndef SF_FCONV_WAVES
#define SF_FCONV_WAVES 8          /* waves per k_conv_fused workgroup */
#endif

// The instantiated variants of k_conv_fused: (WM, WN, D, NORM, LAZY); everything the planner emits (unet.py::fused_geometry).
#define SF_FCONV_VARIANTS(X) \
  X(1, 1, 12, FNORM_GN_SELF, 0) \
  X(1, 1, 12, FNORM_GN_SELF, 1) \
  X(1, 1, 12, FNORM_GN_SELF, 2) \
  X(1, 1, 12, FNORM_GN_SLOTS, 0) \
  X(1, 2, 8, FNORM_GN_SLOTS, 0) \
  X(2, 2, 8, FNORM_GN_SLOTS, 0) \
  X(2, 1, 12, FNORM_GN_SLOTS, 0) \
  X(4, 1, 12, FNORM_GN_SLOTS, 0) \
  X(4, 2, 8, FNORM_GN_SLOTS, 0) \
  X(4, 1, 12, FNORM_NONE, 0) \
  X(4, 2, 8, FNORM_NONE, 0) \
  X(2, 1, 12, FNORM_NONE, 0) \
  X(1, 1, 12, FNORM_NONE, 0) \
  X(1, 2, 8, FNORM_NONE, 0) \
  X(2, 2, 8, FNORM_NONE, 0) \
  X(1, 1, 12, FNORM_LN, 0) \
  X(1, 1, 12, FNORM_LN, 1) \
  X(1, 2, 8, FNORM_LN, 0) \
  X(1, 1, 12, FNORM_ATTN, 0) \
  X(1, 2, 8, FNORM_ATTN, 0)

// Pipelined slot-GroupNorm 3x3 convs (k_conv_fused_pipe, op flag 32): (WM, WN, EPT = (TR + 2) * W / 8 staging elements per thread and chunk)
// (2, *, 8), (2, *, 6) and (4, *, 16): the 32-pixel tiles of the 16x16 (TR = 2) and 8x8 (TR = 4) maps and the 64-pixel tile of the
// 32x32 map (TR = 2) that the planner picks
// from B = 4 on -- 32 pixels per workgroup re-read each weight byte half as often as 16 (the chunk loop is bound by the CU's
// vector-memory path, fused_pipe.h), and there are still >= 256 workgroups.  Measured and not kept (r04,
// profiles/r04_huge_tiles_b4_ab.log): 64-pixel tiles at 16x16 (TR = 4) / 8x8 (TR = 8) for B = 4 -- one round of 256 workgroups
// instead of two rounds of 512 -- 1.862 vs 1.860 ms: what the single round saves, the lost conv1 + res_conv merge (WM <= 2) costs.
#define SF_FCONV_PIPE_VARIANTS(X) \
  X(1, 1, 4) \
  X(1, 2, 4) \
  X(1, 1, 6) \
  X(1, 2, 6) \
  X(2, 1, 6) \
  X(2, 2, 6) \
  X(2, 1, 8) \
  X(2, 2, 8) \
  X(2, 1, 12) \
  X(2, 2, 12) \
  X(4, 1, 16) \
  X(4, 2, 16)

// Pairs (conv1 || res_conv in one launch, k_conv_fused_pair): (WM, WN, D, NORM of the first conv, LAZY of both)
#define SF_FCONV_PAIR_VARIANTS(X) \
  X(1, 1, 12, FNORM_GN_SELF, 0) \
  X(1, 1, 12, FNORM_GN_SELF, 1) \
  X(1, 1, 12, FNORM_GN_SELF, 2) \
  X(1, 1, 12, FNORM_GN_SLOTS, 0) \
  X(1, 2, 8, FNORM_GN_SLOTS, 0) \
  X(2, 2, 8, FNORM_GN_SLOTS, 0) \
  X(4, 2, 8, FNORM_GN_SLOTS, 0)

static inline int fconv_pix_stride(int Cs) {
  const int raw = Cs * 2;
  return raw + ((32 - raw % 256) + 256) % 256;      // stride = 32 (mod 256): conflict-free 16-byte fragment reads
}

// Returns 0 on success; on failure writes a message to err.
static inline int fconv_setup(const sf_op& op, FConvArgs& a, int& WM, int& WN, uint32_t& grid, uint32_t& lds_bytes, char* err,
                              size_t errn) {
#define FC_FAIL(...) do { snprintf(err, errn, __VA_ARGS__); return 1; } while (0)
  a.s1.p = (float*)op.p[0]; a.s1.a = (const float*)op.p[1]; a.s1.b = (const float*)op.p[2]; a.s1.r = (const float*)op.p[3];
  a.s1.slots = (const float*)op.p[4];
  a.s2.p = (float*)op.p[5]; a.s2.a = a.s2.b = a.s2.r = nullptr; a.s2.slots = (const float*)op.p[6];
  a.w = (const bf16x8*)op.p[7]; a.bias = (const float*)op.p[8]; a.out = (float*)op.p[9]; a.resid = (const float*)op.p[10];
  a.ws = (float*)op.p[11]; a.slots_out = (float*)op.p[12];
  a.dbg = (long long*)op.p[16];
  a.wk = (const float*)op.p[17]; a.logit_part = (float*)op.p[18];
  a.weff = nullptr; a.pool_part = nullptr; a.weff_off = 0;
  a.attn = FAttn{}; a.attn_off = 0;
  a.rc_w = nullptr; a.rc_bias = nullptr; a.rc_out = nullptr; a.rc_off = 0; a.rc_buf_bytes = 0;
  if (op.flags & 64) {                 // epilogue pooling (k_conv_fused_pipe<.., POOL>): p 17 = w_eff bf16 [KS * 32], p 18 = pooled fragments
    a.weff = (const sf_opnd*)op.p[17]; a.pool_part = (float*)op.p[18];
    a.wk = nullptr; a.logit_part = nullptr;
    if (!(op.flags & 32) || !a.weff || !a.pool_part) FC_FAIL("fconv: epilogue pooling needs the pipelined kernel, w_eff and a pooled-fragment buffer");
  }
  if ((a.wk == nullptr) != (a.logit_part == nullptr)) FC_FAIL("fconv: context logits need both to_k weight and the partial buffer");
  a.gamma = (const float*)op.p[13]; a.beta = (const float*)op.p[14]; a.ss = (const float*)op.p[15];
  a.B = op.i[0]; a.H = op.i[1]; a.W = op.i[2];
  a.s1.C = op.i[3]; a.s2.C = op.i[4]; a.C = a.s1.C + a.s2.C;
  a.Cout = op.i[5]; a.ldc = op.i[6]; a.co_off = op.i[7]; a.k = op.i[8];
  a.s1.mode = op.i[9]; a.s1.groups = op.i[10]; a.s1.npad = op.i[11];
  a.s2.mode = 0; a.s2.groups = 0; a.s2.npad = 0;
  a.norm = op.i[12]; a.G = op.i[13] > 0 ? op.i[13] : 8; a.TR = op.i[14];
  WM = op.i[15]; WN = op.i[16];
  a.S = op.i[17] > 0 ? op.i[17] : 1;
  a.ss_stride = op.i[18];
  a.silu = (op.flags & 1) ? 1 : 0; a.pre_gelu = (op.flags & 2) ? 1 : 0; a.accum = (op.flags & 4) ? 1 : 0; a.out_gelu = (op.flags & 8) ? 1 : 0;
  a.eps = op.f[0]; a.s1.scale = op.f[1]; a.s2.scale = op.f[2];
  if ((!a.s1.p && a.s1.mode == 0) || !a.w || a.B < 1 || a.H < 1 || a.W < 1) FC_FAIL("fconv: missing operand");
  if ((long)a.B * a.H * a.W * (a.s1.C > a.s2.C ? a.s1.C : a.s2.C) >= (1L << 29)) FC_FAIL("fconv: source too large for 32-bit element offsets");
  if (a.k != 1 && a.k != 3) FC_FAIL("fconv: k must be 1 or 3 (stride 1, same padding)");
  if (a.W & (a.W - 1)) FC_FAIL("fconv: W must be a power of two");
  if (a.C % 32 || a.s1.C % 32 || a.s1.C <= 0 || a.s2.C < 0) FC_FAIL("fconv: channel counts must be multiples of 32");
  if (a.s2.C && !a.s2.p) FC_FAIL("fconv: second source missing");
  if (a.TR < 1 || a.H % a.TR || a.TR * a.W != 16 * WM) FC_FAIL("fconv: tile of %d rows x %d != 16*WM (WM=%d)", a.TR, a.W, WM);
  if (!((WM == 1 || WM == 2 || WM == 4) && (WN == 1 || WN == 2))) FC_FAIL("fconv: unsupported wave tile %dx%d", WM, WN);
  a.cchunks = a.C / 32;
  if (a.cchunks % a.S) FC_FAIL("fconv: %d chunks do not split into %d slices", a.cchunks, a.S);
  a.cps = a.cchunks / a.S;
  a.KS = a.k * a.k * a.cchunks;
  a.M = a.B * a.H * a.W;
  a.mt_per_img = a.H / a.TR;
  a.n_frags = (a.Cout + 15) / 16;
  a.n_tiles = (a.n_frags + WN - 1) / WN;
  a.npad = a.n_frags * 16;
  if (a.s1.mode < 0 || a.s1.mode > 2) FC_FAIL("fconv: unknown lazy mode %d", a.s1.mode);
  if (a.s1.mode == 1 && (!a.s1.a || a.s1.groups < 1 || a.s1.groups > 8 || a.s1.npad % 4)) FC_FAIL("fconv: bad split-K source (1..8 slabs)");
  if (a.s1.mode == 2 && (!a.s1.a || !a.s1.b || !a.s1.r)) FC_FAIL("fconv: gated source needs h, gate and res");
  if (a.s1.mode && a.s1.scale != 1.0f) FC_FAIL("fconv: lazy sources are unscaled");
  if (a.s1.mode && a.s1.p && (a.s1.p == a.s1.a || a.s1.p == a.s1.r)) FC_FAIL("fconv: a lazy source must materialise into its own buffer");
  if (a.s1.mode && !a.s1.p && a.norm != FNORM_NONE) FC_FAIL("fconv: only the plain (res_conv) half of a pair may leave a lazy source unmaterialised");
  const int Cs = a.cps * 32;
  if (a.norm == FNORM_GN_SELF || a.norm == FNORM_GN_SLOTS) {
    if (!a.gamma || !a.beta || a.C % a.G) FC_FAIL("fconv: GroupNorm parameters missing");
    const int Cg = a.C / a.G;
    if (Cs % Cg || Cs / Cg > 8 || Cg % 4) FC_FAIL("fconv: a slice must hold 1..8 whole groups (Cs=%d Cg=%d)", Cs, Cg);
    if (a.norm == FNORM_GN_SELF && a.TR != a.H) FC_FAIL("fconv: GN_SELF needs the whole image in the tile");
    if (a.norm == FNORM_GN_SLOTS) {
      if (!a.s1.slots || (a.s2.C && !a.s2.slots)) FC_FAIL("fconv: GN_SLOTS without slots");
      if (Cg % 16 || (a.H * a.W) % 16 || a.s1.mode == 1) FC_FAIL("fconv: GN_SLOTS geometry");
    }
  } else if (a.norm == FNORM_LN) {
    if (a.S != 1 || a.k != 1 || !a.gamma || a.s2.C) FC_FAIL("fconv: LayerNorm prologue needs S=1, k=1, one source, a gain");
    if ((a.C / 4) * 16 * WM > 16 * SF_FCONV_WAVES * 64) FC_FAIL("fconv: LayerNorm row too long for the register-resident prologue (C <= 2048)");
    if (a.s1.scale != 1.0f) FC_FAIL("fconv: LayerNorm source is unscaled");
  } else if (a.norm == FNORM_ATTN) {
    if (a.S != 1 || a.k != 1 || a.s2.C || a.s1.mode || a.C != 64 * SF_FCONV_WAVES || a.H * a.W != 16 || WM != 1 || (op.flags & (1 | 2 | 32)))
      FC_FAIL("fconv: the attention prologue needs k=1, one plain source of 8 x 64 channels, a 16-token map, no activation");
    FAttn& at = a.attn;
    at.q = (const float*)op.p[0]; at.ldq = op.i[19]; at.scale = op.f[6]; at.J = 0; at.per_head = 0;
    for (int sgi = 0; sgi < 3; ++sgi) {
      FAttnSeg& sg = at.seg[sgi];
      sg.k = (const float*)op.p[19 + sgi]; sg.v_off = (int)op.f[3 + sgi];
      sg.rows = op.i[20 + 4 * sgi]; sg.row_stride = op.i[21 + 4 * sgi]; sg.batch_stride = op.i[22 + 4 * sgi]; sg.head_stride = op.i[23 + 4 * sgi];
      if (sg.rows < 0 || (sg.rows && !sg.k)) FC_FAIL("fconv: attention segment %d without keys", sgi);
      if (!sg.rows) sg.k = at.q;                                   // never dereferenced; keeps the struct free of null arithmetic
      at.J += sg.rows;
      if (sg.rows && sg.head_stride) at.per_head = 1;
    }
    if (at.ldq < a.C || at.J < 1 || at.J > (at.per_head ? 4 : SF_ATTN_MAX_KEYS)) FC_FAIL("fconv: attention wants 1..%d keys (%d given)", at.per_head ? 4 : SF_ATTN_MAX_KEYS, at.J);
  } else if (a.norm != FNORM_NONE) {
    FC_FAIL("fconv: unknown norm %d", a.norm);
  }
  if (a.S > 1) {
    if (!a.ws || a.accum || a.slots_out || a.out_gelu) FC_FAIL("fconv: split-K slices write slabs only");
  } else {
    if (!a.out) FC_FAIL("fconv: output missing");
    if (a.slots_out && (a.Cout % 16 || a.ldc % 16 || a.co_off % 16)) FC_FAIL("fconv: slots need 16-aligned channels");
  }
  if (a.norm == FNORM_GN_SELF && (a.H * a.W * (Cs / 4) > 4096)) FC_FAIL("fconv: GN_SELF tile exceeds the register-resident prologue");
  if (a.norm == FNORM_GN_SELF && Cs / (a.C / a.G) > 16) FC_FAIL("fconv: GN_SELF slice holds more than 16 groups");
  a.logW = 0;
  while ((1 << a.logW) < a.W) ++a.logW;
  auto mk = [](uint32_t d) { FDiv f; f.d = d ? d : 1; f.magic = (uint32_t)(0x100000000ull / f.d) + 1u; return f; };
  a.d_cs4 = mk(Cs / 4);
  a.d_cg = mk((a.norm == FNORM_GN_SELF || a.norm == FNORM_GN_SLOTS) ? a.C / a.G : 1);
  a.d_cps = mk(a.cps);
  a.d_tc = mk((Cs / 4) < SF_FCONV_WAVES * 64 ? (Cs / 4) : SF_FCONV_WAVES * 64);
  a.d_ncf = mk((a.norm == FNORM_GN_SLOTS) ? (a.C / a.G) / 16 : 1);
  a.det_w = 0;
  if (a.norm == FNORM_GN_SELF) {
    const int NT = SF_FCONV_WAVES * 64, cs4 = Cs / 4, cg4 = (a.C / a.G) / 4;
    const int w = cg4 >= 64 ? (cg4 % 64 == 0 ? 64 : 0) : ((cg4 & (cg4 - 1)) == 0 ? cg4 : 0);
    if (NT % cs4 == 0 && w >= 1 && (cs4 >= 64 || 64 % cs4 == 0)) a.det_w = w;
  }
  a.pix_stride = fconv_pix_stride(Cs);
  const int h = a.k >> 1;
  const uint32_t frame = ((uint32_t)(a.TR + 2 * h) * (a.W + 2 * h) + 1) * a.pix_stride;      // + 1 spare pixel (dead staging stores)
  a.red_off = (int)frame;
  a.tab_off = a.red_off + 1024 * SF_FCONV_WAVES * WM * WN;
  a.misc_off = a.tab_off + 2 * Cs * 4;
  lds_bytes = a.misc_off + 640 + 2048;      // misc: 160 floats of statistics + 512 floats of reduction partials
  a.attn_off = 0;
  if (a.norm == FNORM_ATTN) { a.attn_off = (int)((lds_bytes + 15) & ~15u); lds_bytes = a.attn_off + SF_ATTN_LDS_BYTES; }
  if (lds_bytes > SF_LDS_MAX && !(op.flags & 32)) FC_FAIL("fconv: tile needs %u bytes of LDS", lds_bytes);   // pipe: its own (chunked) frame below
  const int MT = a.B * a.mt_per_img;
  // XCD-aware tile map (fconv_tile_of): R row groups x 8 / R channel groups.  R = 1 (every XCD owns n-tiles == x mod 8 of ALL
  // rows: each weight byte crosses the fabric once, the activation map is fetched by all 8 L2s) is the measured best on every
  // layer: sharing rows instead (R = 2 / 4, each XCD then pulls 1 / R of the activations and R x the weights) was slower even
  // on the 32x32 layers whose activations outweigh their weights (r03: eval 1.311 / 1.335 / 1.402 ms for R = 1 / 2 / 4;
  // choosing R per layer by fabric bytes: 1.314; the SF_XCD_R switch of that A/B was retired in r04, fconv_tile_of keeps the general map).
  a.xcd_map = 0;
  if (MT > 1 && a.S == 1) {
    const int R = 1;
    if (MT % R == 0 && a.n_tiles % (8 / R) == 0) a.xcd_map = R;
    else if (a.n_tiles % 8 == 0) a.xcd_map = 1;
  }
  grid = (uint32_t)a.S * MT * a.n_tiles;
  a.buf_bytes = 0;
  a.inv_n = (a.norm == FNORM_GN_SELF || a.norm == FNORM_GN_SLOTS) ? 1.0 / ((double)a.H * a.W * (a.C / a.G)) : 0.0;
  if (op.flags & 32) {       // k_conv_fused_pipe: 128-channel chunks, two frame buffers, 4 matrix + 4 staging waves
    if (a.norm != FNORM_GN_SLOTS || a.k != 3 || a.S != 1 || a.s1.mode != 0 || a.C % 128 || a.C > 4 * SF_FCONV_WAVES * 64 || a.G != 8 ||
        ((a.TR + 2) * a.W) % 8)
      FC_FAIL("fconv pipe: needs slot GroupNorm (8 groups), k = 3, one slice, a plain source, C %% 128 == 0");
    a.pix_stride = fconv_pix_stride(128);
    a.buf_bytes = (int)((((uint32_t)(a.TR + 2) * (a.W + 2) + 1) * a.pix_stride + 15) & ~15u);
    a.red_off = 2 * a.buf_bytes;
    a.tab_off = a.red_off + 1024 * (SF_FCONV_WAVES / 2) * (WM * WN + (a.weff ? WM : 0));
    a.misc_off = a.tab_off + 2 * a.C * 4;
    lds_bytes = a.misc_off + 640 + 2048;
    if (a.weff) {
      if (a.accum || a.resid || a.out_gelu || a.co_off || a.ldc != a.Cout) FC_FAIL("fconv: epilogue pooling wants a plain conv output");
      a.weff_off = (int)lds_bytes;
      lds_bytes += (uint32_t)a.KS * 64;
    }
    if (lds_bytes > SF_LDS_MAX) FC_FAIL("fconv pipe: tile needs %u bytes of LDS", lds_bytes);
  }
  return 0;
#undef FC_FAIL
}

static inline int fconv_pipe_ept(const FConvArgs& a) { return (a.TR + 2) * a.W / 8; }

// k_lin4_ln (fused_conv4.h, r05): LayerNorm -> Linear on the 16-token map, plain source of 1024 | 2048 channels; returns C4T (8 | 16) or 0
static inline int lin4_c4t(const sf_op& op, const FConvArgs& a, int WM, int WN) {
  if (op.flags & (16 | 32 | 64 | 128)) return 0;
  if (a.norm != FNORM_LN || a.H != 4 || a.W != 4 || a.k != 1 || a.TR != 4 || WM != 1 || (WN != 1 && WN != 2)) return 0;
  if (a.S != 1 || a.s1.mode != 0 || a.s2.C || a.dbg || a.logit_part || (a.C != 1024 && a.C != 2048) || a.s1.scale != 1.0f) return 0;
  if (((uintptr_t)a.gamma | (uintptr_t)a.beta | (uintptr_t)a.s1.p) & 15) return 0;
  return a.C / 128;
}

// k_lin4_attn (fused_conv4.h, r06): the attention-prologue output projection of the 16-token map (8 heads x 64 inner channels) on k_lin4_ln's
// skeleton; returns WN (1 | 2) or 0 = the general kernel.  Op flag 128 (planner attribute Unet.conv4 = False) keeps the general kernel.
static inline int lin4_attn_wn(const sf_op& op, const FConvArgs& a, int WM, int WN) {
  if (op.flags & (1 | 2 | 16 | 32 | 64 | 128)) return 0;
  if (a.norm != FNORM_ATTN || a.H != 4 || a.W != 4 || a.k != 1 || a.TR != 4 || WM != 1 || (WN != 1 && WN != 2)) return 0;
  if (a.S != 1 || a.C != 512 || a.s2.C || a.s1.mode || a.dbg || a.logit_part) return 0;
  return WN;
}

// k_conv4_gn (fused_conv4.h, r05) takes the op when it is the 4x4 level's GroupNorm-self 3x3 conv in the geometry the kernel is written
// for; returns CS4 (64 | 128) or 0 = the general kernel.  Op flag 128 (planner attribute Unet.conv4 = False) keeps the general kernel.
// (r06: also the same geometry WITHOUT a norm -- k_conv4_gn<CS4, 0, false>: one plain source, slices of 256 channels, B = 1)
static inline bool conv4_nonorm(const FConvArgs& a) {
  return a.norm == FNORM_NONE && a.B == 1 && a.s1.mode == 0 && a.s2.C == 0 && a.cps == 8 && !a.wk && !a.pre_gelu && ((uintptr_t)a.s1.p & 15) == 0;
}
static inline int conv4_cs4(const sf_op& op, const FConvArgs& a, int WM, int WN) {
  if (op.flags & (16 | 32 | 64 | 128)) return 0;
  if ((a.norm != FNORM_GN_SELF && !conv4_nonorm(a)) || a.H != 4 || a.W != 4 || a.k != 3 || a.TR != 4 || WM != 1 || WN != 1) return 0;
  if (a.S < 2 || !a.ws || a.dbg || a.G != 8 || (a.cps != 8 && a.cps != 16)) return 0;
  if (a.norm == FNORM_NONE) return a.cps * 8;
  if (((a.C / a.G) / 4) * 2 != a.cps * 8) return 0;                        // a slice = two whole groups
  if (a.s1.mode == 1 && a.s1.groups > 4) return 0;
  if (((uintptr_t)a.gamma | (uintptr_t)a.beta | (uintptr_t)a.ss) & 15 || (a.ss && a.ss_stride % 4)) return 0;      // float4 affine operands
  return a.cps * 8;
}

// k_conv3s (fused_conv3s.h, r06): the recurring single-source slot-GroupNorm 3x3 convs on a kernel with compile-time geometry.
// (HL = log2 of the map side, C = Cout, TWL = log2 of the tile width, WM, WN): the first block = the B = 1 plan's layers as full-width
// strips and as 2-D tiles, the second = the 64-pixel tiles of B >= 2 at 32x32, the third = the 32-pixel tiles of B >= 2 at 16x16 (8 x 4 pixels) and 8x8 (4 rows).
#define SF_CONV3S_VARIANTS(X) \
  X(5, 256, 5, 2, 2) X(5, 256, 3, 2, 2) \
  X(4, 256, 4, 1, 1) X(4, 256, 2, 1, 1) \
  X(4, 512, 4, 1, 2) X(4, 512, 2, 1, 2) \
  X(3, 512, 3, 1, 1) X(3, 1024, 3, 1, 1) \
  X(5, 256, 5, 4, 2) X(5, 256, 3, 4, 2) \
  X(4, 256, 2, 2, 1) X(4, 256, 2, 2, 2) X(4, 512, 2, 2, 2) \
  X(3, 512, 3, 2, 1) X(3, 1024, 3, 2, 1) X(3, 1024, 3, 2, 2)

// Does k_conv3s take this (pipelined, op flag 32) conv?  Returns the tile width's log2, or -1 = the general kernel.  Op field i[19]: bit 1 = keep
// the general kernel (planner attribute Unet.conv3s = False), bits 2.. = tile width in pixels (0 = full-width strips of TR rows).
static inline int conv3s_twl(const sf_op& op, const FConvArgs& a, int WM, int WN) {
  if (!(op.flags & 32) || (op.flags & (2 | 4 | 8 | 16)) || (op.i[19] & 2)) return -1;
  if (a.norm != FNORM_GN_SLOTS || a.k != 3 || a.S != 1 || a.s2.C || a.s1.mode || a.s1.scale != 1.0f || a.H != a.W || a.G != 8 || !a.silu) return -1;
  if (a.Cout != a.C || a.ldc != a.Cout || a.co_off || a.accum || a.out_gelu || a.logit_part || a.dbg || !a.bias || !a.out) return -1;
  if (((uintptr_t)a.gamma | (uintptr_t)a.beta | (uintptr_t)a.ss | (uintptr_t)a.s1.p) & 15 || (a.ss && a.ss_stride % 4)) return -1;
  const int tw = (op.i[19] >> 2) ? (op.i[19] >> 2) : a.W;
  if (tw < 4 || tw > a.W || (tw & (tw - 1)) || (16 * WM) % tw || (16 * WM) / tw > a.H || a.H % ((16 * WM) / tw)) return -1;
  int twl = 0;
  while ((1 << twl) < tw) ++twl;
  return twl;
}

// k_conv3s_rc: conv1 of a ResnetBlock on the concat of two sources with the block's res_conv in the same workgroups (the pipelined pairs of the
// B = 1 plan).  (HL, C1, C2, COUT, TWL, WM, WN): the 2-D tiles at 32x32 / 16x16 (their full-width strips need more registers than a wave has: 35 / 4
// spilled VGPRs, not instantiated), the 2-row strip at 8x8.  The 32- / 64-pixel tiles of B >= 2 do not fit either (two sets of accumulators + the staging
// batches: 38-91 spilled VGPRs): those pairs stay on k_conv_fused_pipe_rc / _pipe_pair.
#define SF_CONV3S_RC_VARIANTS(X) \
  X(5, 256, 256, 256, 3, 2, 2) \
  X(4, 512, 256, 512, 2, 1, 2) \
  X(3, 1024, 512, 1024, 3, 1, 1)

// The res_conv `b` can ride in conv1 `a`'s workgroups: a 1x1 un-normalised conv of the same raw sources onto plain rows of the same width
static inline bool fconv_rc_compatible(const FConvArgs& a, const FConvArgs& b) {
  return !(a.weff || a.logit_part || b.k != 1 || b.norm != FNORM_NONE || b.S != 1 || b.Cout != a.Cout || b.ldc != b.Cout || b.co_off ||
           b.resid || b.accum || b.slots_out || b.out_gelu || b.silu || b.logit_part || !b.out || b.s1.mode || a.s1.mode ||
           b.s1.p != a.s1.p || b.s2.p != a.s2.p || b.s1.C != a.s1.C || b.s2.C != a.s2.C || b.s1.scale != a.s1.scale || b.s2.scale != a.s2.scale);
}

// Does k_conv3s_rc take this pipelined pair (op1 = conv1 with flag 16, b = its res_conv as set up by fconv_setup)?  Tile width's log2 or -1.
static inline int conv3s_rc_twl(const sf_op& op1, const FConvArgs& a, const FConvArgs& b, int WM, int WN) {
  if (!(op1.flags & 32) || !(op1.flags & 16) || (op1.flags & (2 | 4 | 8 | 64)) || (op1.i[19] & 2)) return -1;
  if (a.norm != FNORM_GN_SLOTS || a.k != 3 || a.S != 1 || !a.s2.C || !a.s2.p || !a.s2.slots || a.s1.mode || a.s1.scale != 1.0f || a.H != a.W || a.G != 8 || !a.silu) return -1;
  if (a.ldc != a.Cout || a.co_off || a.accum || a.out_gelu || a.logit_part || a.weff || a.dbg || !a.bias || !a.out || !fconv_rc_compatible(a, b)) return -1;
  if (((uintptr_t)a.gamma | (uintptr_t)a.beta | (uintptr_t)a.ss | (uintptr_t)a.s1.p | (uintptr_t)a.s2.p) & 15 || (a.ss && a.ss_stride % 4)) return -1;
  const int tw = (op1.i[19] >> 2) ? (op1.i[19] >> 2) : a.W;
  if (tw < 4 || tw > a.W || (tw & (tw - 1)) || (16 * WM) % tw || (16 * WM) / tw > a.H || a.H % ((16 * WM) / tw)) return -1;
  int twl = 0;
  while ((1 << twl) < tw) ++twl;
  return twl;
}

// (CS4, LAZY, NB) of k_conv4_gn_mb
#define SF_CONV4_MB_VARIANTS(X) \
  X(64, 0, 2) X(64, 0, 4) X(64, 1, 2) X(64, 1, 4) X(64, 2, 2) X(64, 2, 4) X(128, 0, 2) X(128, 2, 2)

// k_conv4_gn_mb (fused_conv4.h, r05): NB = 2 | 4 images per workgroup for an op that fits k_conv4_gn at B >= 2.  Returns NB and rewrites the
// LDS layout (NB frames, a reduction buffer per image) and the grid, or 0 = one image per workgroup.  Op field i[19] bit 0 (planner
// attribute Unet.conv4_mb = False; FNORM_ATTN ops use i[19] otherwise) keeps k_conv4_gn.
static inline int conv4_mb_setup(const sf_op& op, FConvArgs& a, int cs4, uint32_t& grid, uint32_t& lds_bytes) {
  if ((op.i[19] & 1) || a.B < 2 || a.norm != FNORM_GN_SELF) return 0;
  if (a.s1.C % (cs4 * 4)) return 0;                                         // a slice lies in ONE source: the kernel selects its base pointers per workgroup
  if ((long)a.M * (a.s1.mode == 1 ? a.s1.npad : a.s1.C) >= (1L << 30)) return 0;     // 32-bit element offsets
  int nb = 0;
  if (cs4 == 64) nb = a.B % 4 == 0 ? 4 : (a.B % 2 == 0 ? 2 : 0);
  else if (cs4 == 128 && a.s1.mode != 1) nb = a.B % 2 == 0 ? 2 : 0;       // two 36-pixel x 512-channel frames: 76 KB; a split-K source at Cs = 512
                                                                            // stays on k_conv4_gn (18 weight fragments + one image's 80 gather registers: 256 VGPRs + 8 spilled; measured neutral at B = 4, not kept)
  if (!nb) return 0;
  const uint32_t frame = (36u * (uint32_t)a.pix_stride + 15u) & ~15u;
  const uint32_t red = (uint32_t)nb * frame, misc = red + (uint32_t)nb * 8192u, total = misc + 640 + 2048;
  if (total > SF_LDS_MAX) return 0;
  a.buf_bytes = (int)frame;
  a.red_off = (int)red;
  a.tab_off = a.misc_off = (int)misc;                                       // no affine table: the affine lives in registers
  lds_bytes = total;
  grid = (uint32_t)a.S * (uint32_t)(a.B / nb) * (uint32_t)a.n_tiles;
  return nb;
}

// (WM, WN, EPT) of k_conv_fused_pipe_rc: the pipelined tiles with registers to spare for the res_conv's accumulators and ring slot
#define SF_FCONV_PIPE_RC_VARIANTS(X) \
  X(1, 1, 4) \
  X(1, 2, 4) \
  X(1, 1, 6) \
  X(1, 2, 6) \
  X(2, 1, 6) \
  X(2, 2, 6) \
  X(2, 1, 8) \
  X(2, 2, 8) \
  X(2, 1, 12) \
  X(2, 2, 12)

// A pipelined pair whose res_conv can ride in conv1's workgroups (k_conv_fused_pipe_rc): fills a.rc_* / the LDS layout and returns true.
// `a` = conv1 as set up by fconv_setup, `b` = the res_conv, WM / WN the common tile, lds_bytes conv1's.
// Measured (profiles/r04_pipe_rc_merge_ab.log): B = 1 eval 1.233 -> 1.190 ms, B = 2 1.417 -> 1.401, B = 4 1.908 -> 1.876 (its 32x32 pairs
// run the WM = 4 tile, which keeps the two-kernel launch); a pair launch 22-23 -> 17-18 us.
static inline bool fconv_pipe_rc_merge(FConvArgs& a, const FConvArgs& b, int WM, int WN, uint32_t& lds_bytes) {
  if (WM > 2 || !fconv_rc_compatible(a, b)) return false;
  const uint32_t raw = (((uint32_t)(16 * WM + 1) * a.pix_stride) + 15) & ~15u;
  // LDS: [frames][red: + WM * WN fragments per matrix wave][table][misc][raw operand x 2]
  const int red_old = 1024 * (SF_FCONV_WAVES / 2) * (WM * WN), red_new = 1024 * (SF_FCONV_WAVES / 2) * (2 * WM * WN);
  const uint32_t total = lds_bytes + (uint32_t)(red_new - red_old) + 2 * raw;
  if (total > SF_LDS_MAX) return false;
  a.tab_off += red_new - red_old;
  a.misc_off += red_new - red_old;
  a.rc_off = (int)((lds_bytes + (uint32_t)(red_new - red_old) + 15) & ~15u);
  a.rc_buf_bytes = (int)raw;
  lds_bytes = (uint32_t)a.rc_off + 2 * raw;
  a.rc_w = b.w; a.rc_bias = b.bias; a.rc_out = b.out;
  return lds_bytes <= SF_LDS_MAX;
}

// Pair = op1 (flags & 16) + the op after it: same tile shape, op2 un-normalised, same lazy mode (op2 with s1.p == null when lazy).
static inline int fconv_pair_setup(const sf_op& op1, const sf_op& op2, FConvPairArgs& p, int& WM, int& WN, uint32_t& grid, uint32_t& lds_bytes,
                                   char* err, size_t errn) {
  int WM2, WN2;
  uint32_t g1, g2, l1, l2;
  if (fconv_setup(op1, p.a, WM, WN, g1, l1, err, errn) || fconv_setup(op2, p.b, WM2, WN2, g2, l2, err, errn)) return 1;
  if (op2.type != SF_OP_FCONV || WM2 != WM || WN2 != WN || p.b.norm != FNORM_NONE || p.b.s1.mode != p.a.s1.mode || (op2.flags & 16)) {
    snprintf(err, errn, "fconv pair: the second op must be an un-normalised fconv of the same tile shape and lazy mode");
    return 1;
  }
  if (p.a.dbg || p.b.dbg) { snprintf(err, errn, "fconv pair: no phase stamps"); return 1; }
  p.grid_b = (int)g2;
  grid = g1 + g2;
  lds_bytes = l1 > l2 ? l1 : l2;
  return 0;
}

// SF_OP_GCA operands (flags = stage)
//   1 POOL  p: 0 h2  1 split-K slabs or null  2 conv bias or null  3 logit_part  4 part_pool  5 part_ms
//           i: 0 M  1 C  2 HW  3 CH (pixels per chunk)  4 chunks per image  5 nparts  6 groups  7 npad
//   2 NET0  p: 0 part_pool  1 part_ms  2 W0 bf16 [HID][Kp]  3 b0  4 hid ;  i: 0 B  1 C  2 Kp  3 HID  4 chunks  5 bit 0: keep k_gca_net0 (the canonical (C, chunks) run k_gca_net0_t otherwise)
//   3 GATE  p: 0 h2  1 res  2 hid  3 W2 bf16 [C][Kp2]  4 b2  5 out  6 slots or null ;  i: 0 M  1 C  2 HW  3 HID  4 Kp2  5 bit 0: keep k_gca_gate (HID = 128 | 256 | 512 run k_gca_gate_t otherwise)
static inline int gca_setup(const sf_op& op, GcaPoolArgs& pa, GcaNetArgs& na, GcaGateArgs& ga, uint32_t& grid, char* err, size_t errn) {
#define GC_FAIL(...) do { snprintf(err, errn, __VA_ARGS__); return 1; } while (0)
  if (op.flags == 1) {
    pa.h2 = (float*)op.p[0]; pa.ws = (const float*)op.p[1]; pa.bias = (const float*)op.p[2];
    pa.logit_part = (const float*)op.p[3]; pa.part_pool = (float*)op.p[4]; pa.part_ms = (float*)op.p[5];
    pa.M = op.i[0]; pa.C = op.i[1]; pa.HW = op.i[2]; pa.CH = op.i[3]; pa.chunks = op.i[4]; pa.nparts = op.i[5];
    pa.groups = op.i[6]; pa.npad = op.i[7];
    if (!pa.h2 || !pa.logit_part || !pa.part_pool || !pa.part_ms) GC_FAIL("gca pool: missing operand");
    if (pa.C % 64 || pa.CH < 16 || pa.CH > 128 || (pa.CH & (pa.CH - 1)) || pa.CH * pa.chunks != pa.HW || pa.M % pa.HW || pa.nparts < 1)
      GC_FAIL("gca pool: C %% 64, CH a power of two in 16..128, CH * chunks == HW required");
    if (pa.ws && (pa.groups < 1 || pa.groups > 8 || pa.npad % 4)) GC_FAIL("gca pool: bad split-K source (1..8 slabs)");
    grid = (uint32_t)(pa.M / pa.HW) * pa.chunks * (pa.C / 64);
    return 0;
  }
  if (op.flags == 2) {
    na.part_pool = (const float*)op.p[0]; na.part_ms = (const float*)op.p[1]; na.W0 = (const sf_opnd*)op.p[2];
    na.b0 = (const float*)op.p[3]; na.hid = (float*)op.p[4];
    na.B = op.i[0]; na.C = op.i[1]; na.Kp = op.i[2]; na.HID = op.i[3]; na.chunks = op.i[4];
    if (!na.part_pool || !na.part_ms || !na.W0 || !na.b0 || !na.hid) GC_FAIL("gca net0: missing operand");
    if (na.C > 2048 || na.C % 8 || na.Kp < na.C || na.chunks < 1 || na.chunks > 64) GC_FAIL("gca net0: C <= 2048, 1..64 chunks");
    grid = (uint32_t)na.B * ((na.HID + 15) / 16);
    return 0;
  }
  if (op.flags == 3) {
    ga.h2 = (const float*)op.p[0]; ga.res = (const float*)op.p[1]; ga.hid = (const float*)op.p[2];
    ga.W2 = (const sf_opnd*)op.p[3]; ga.b2 = (const float*)op.p[4]; ga.out = (float*)op.p[5]; ga.slots = (float*)op.p[6];
    ga.M = op.i[0]; ga.C = op.i[1]; ga.HW = op.i[2]; ga.HID = op.i[3]; ga.Kp2 = op.i[4];
    if (!ga.h2 || !ga.res || !ga.hid || !ga.W2 || !ga.b2 || !ga.out) GC_FAIL("gca gate: missing operand");
    if (ga.M % 16 || ga.C % 16 || ga.HW % 16 || ga.Kp2 % 8 || ga.Kp2 < ga.HID || ga.HID > 1024 || ga.HID < 8) GC_FAIL("gca gate: 16-aligned M, C, HW and 8 <= HID <= 1024 required");
    grid = ((uint32_t)(ga.M / 16) * (ga.C / 16) + 3) / 4;
    return 0;
  }
  GC_FAIL("gca: unknown stage %d", op.flags);
#undef GC_FAIL
}

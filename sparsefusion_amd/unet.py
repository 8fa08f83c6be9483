"""View-conditioned latent UNet on the HIP op set (libsparsefusion_hip.so, csrc/unet_ops.hip).

Drop-in for the reference's `Unet` (external/imagen_pytorch.py:1078-1671) in the configuration
SparseFusion builds (utils/load_model.py:58-69): same constructor keywords, the same 477
state-dict keys/shapes (checkpoints load unchanged through `load_state_dict`), the same call
surface `forward(x, time, cond_images=...)` / `forward_with_cond_scale(..., cond_scale=1.)`
with NCHW fp32 tensors at the boundary.

Inside, nothing of the torch module graph survives: `nn.Parameter`s only hold the weights; the
forward is a static list of ~330 kernel launches (an "op plan") built once per batch size over
pre-packed bf16 weights and two device arenas, replayed by ONE C call (`sf_plan_run`).  See
DESIGN.md section 4 for the dataflow and the fusion map."""
import math
import os

import numpy as np
import torch
import torch.nn as nn

from . import _lib

OP_CONV, OP_GN_ACT, OP_LN, OP_GEMV, OP_ATTN, OP_GCA_POOL, OP_ELTWISE, OP_MEMSET, OP_TIME_EMB, OP_SPLITK_REDUCE = range(1, 11)
OP_FCONV, OP_SLOTS, OP_GCA, OP_INITX, OP_GN_FINALIZE = 14, 15, 16, 17, 18
# (WM, WN, norm of conv1) for which k_conv_fused_pair is instantiated (csrc/fused_host.h SF_FCONV_PAIR_VARIANTS); FNORM_GN_SELF = 1, _SLOTS = 2
PAIR_TILES = {(1, 1, 1), (1, 1, 2), (1, 2, 2), (2, 2, 2), (4, 2, 2)}
# (WM, WN, (TR + 2) * W / 8) for which k_conv_fused_pipe is instantiated (SF_FCONV_PIPE_VARIANTS)
# (log2 H, C = Cout, log2 tile width, WM, WN) of k_conv3s (csrc/fused_host.h SF_CONV3S_VARIANTS; tests/test_plans_cpu.py compares the two tables)
CONV3S_VARIANTS = {(5, 256, 5, 2, 2), (5, 256, 3, 2, 2), (4, 256, 4, 1, 1), (4, 256, 2, 1, 1), (4, 512, 4, 1, 2), (4, 512, 2, 1, 2),
                   (3, 512, 3, 1, 1), (3, 1024, 3, 1, 1), (5, 256, 5, 4, 2), (5, 256, 3, 4, 2),
                   (4, 256, 2, 2, 1), (4, 256, 2, 2, 2), (4, 512, 2, 2, 2), (3, 512, 3, 2, 1), (3, 1024, 3, 2, 1), (3, 1024, 3, 2, 2)}
# (log2 H, C1, C2, Cout, log2 tile width, WM, WN) of k_conv3s_rc (SF_CONV3S_RC_VARIANTS): conv1 on a concat + the block's res_conv in one set of workgroups
CONV3S_RC_VARIANTS = {(5, 256, 256, 256, 3, 2, 2), (4, 512, 256, 512, 2, 1, 2), (3, 1024, 512, 1024, 3, 1, 1)}
PIPE_TILES = {(1, 1, 4), (1, 2, 4), (1, 1, 6), (1, 2, 6), (2, 1, 6), (2, 2, 6), (2, 1, 8), (2, 2, 8), (2, 1, 12), (2, 2, 12), (4, 1, 16), (4, 2, 16)}
FNORM_NONE, FNORM_GN_SELF, FNORM_GN_SLOTS, FNORM_LN, FNORM_ATTN = range(5)      # csrc/fused_kernels.h
ATTN_LDS_BYTES = 8 * 16 * 36 * 4 + 2 * 8 * 4 * 68 * 4      # SF_ATTN_LDS_BYTES: scratch of the attention prologue (FNORM_ATTN)
# Measured (WM, WN, split-K groups) of implicit-GEMM launches where the cost model of Unet.conv_tiling picks a slower tile
# (tools/tile_sweep.py on MI355X, whole-eval time, r03: B = 1 eval 1.3246 -> 1.3004 ms): key = (m_frags, n_frags, KS, pixshuf).
# The up-sampling 1x1 convs (PixelShuffle epilogue, no split-K) want ONE 16-row fragment per wave: a (4, 1) tile left 128
# workgroups pulling 256 KB of fp32 activations each.
TILE_PICKS = {(4, 32, 128, False): (1, 2, 8),      # Downsample 16x16 -> 8x8, 256 -> 512, k 4 s 2
              (4, 128, 32, True): (1, 1, 1),       # Upsample 8x8 -> 16x16: 1x1 conv 1024 -> 2048 + PixelShuffle
              (16, 64, 16, True): (1, 2, 1),       # Upsample 16x16 -> 32x32: 1x1 conv 512 -> 1024 + PixelShuffle
              (64, 1, 72, False): (1, 1, 4),       # final 3x3 conv 256 -> 4 at 32x32
              # B = 4 (BASELINE configs[3], 4 novel views per GPU; tools/tile_sweep.py 4, r04: eval 1.953 -> 1.904 ms)
              (64, 16, 128, False): (2, 4, 4), (16, 32, 128, False): (1, 4, 4), (4, 64, 256, False): (1, 2, 4),
              (4, 64, 288, False): (1, 4, 4), (4, 256, 32, True): (1, 2, 1), (16, 128, 32, True): (1, 4, 1),
              (64, 64, 16, True): (1, 2, 1)}
LDS_MAX = 163840
SKIP_SCALE = 2 ** -0.5            # scale_skip_connection (imagen_pytorch.py:1283)


def init_x_weight_table(ws, dtype=torch.bfloat16):
    """bf16 MFMA B-fragments of SF_OP_INITX (csrc/initx.hip) for the three CrossEmbed convs (k = 3 / 7 / 15), latent channels only:
    [k-step][n-frag][lane 64][8] with lane = (n = lane & 15, g = lane >> 4) and element j = (tap kx0 + j // 4, channel j % 4):
    k = 15: step s = (ky = s // 2, kx0 = 8 * (s % 2) + 2 g); k = 7: (ky = s, kx0 = 2 g); k = 3: (ky = 2 s + g // 2, kx0 = 2 * (g % 2)).
    Returns (int16 tensor of bf16 bits, fragment offsets of the three convs)."""
    parts, offs, acc = [], [], 0
    for w in ws:
        cw, ci, k, _ = w.shape
        steps = {15: 30, 7: 7, 3: 2}[k]
        nfr = cw // 16
        wp = torch.zeros(cw, 4, k + 2, 16 + 2)                              # zero-padded taps / channels
        wp[:, :ci, :k, :k] = w
        t = torch.zeros(steps, nfr, 64, 8)
        for s in range(steps):
            for g in range(4):
                if k == 15:
                    ky, kx0 = s // 2, 8 * (s % 2) + 2 * g
                elif k == 7:
                    ky, kx0 = s, 2 * g
                else:
                    ky, kx0 = 2 * s + g // 2, 2 * (g % 2)
                blk = wp[:, :, ky, kx0:kx0 + 2]                             # [cw, 4 ch, 2 taps]
                t[s, :, g * 16:(g + 1) * 16, :] = blk.permute(0, 2, 1).reshape(nfr, 16, 8)
        parts.append(t.reshape(-1))
        offs.append(acc)
        acc += steps * nfr
    return torch.cat(parts).to(dtype).view(torch.int16).contiguous(), offs       # dtype = the library's MFMA operand type


def _cast_tuple(v, n):
    return tuple(v) if isinstance(v, (tuple, list)) else (v,) * n


# ------------------------------------------------------------------------------------------------
# parameter specification (names + shapes of the reference state dict)
# ------------------------------------------------------------------------------------------------
def unet_param_spec(dim=256, dim_mults=(1, 2, 4, 4), num_resnet_blocks=(2, 2, 2, 2),
                    layer_attns=(False, False, False, True), cond_images_channels=256, channels=4, cond_dim=None,
                    attn_heads=8, attn_dim_head=64, ff_mult=2, learned_sinu_pos_emb_dim=16, max_conditional_len=256):
    """Ordered (name, shape) list of every tensor `Unet(...).state_dict()` holds in the reference for
    this family of configurations (cond_on_z=False, attn_pool_text=False, init_cross_embed, pixel
    shuffle upsampling, gca, final resnet block)."""
    cond_dim = cond_dim or dim
    tdim = dim * 4
    inner = attn_heads * attn_dim_head
    dims = [dim] + [dim * m for m in dim_mults]
    in_out = list(zip(dims[:-1], dims[1:]))
    n_lv = len(in_out)
    nres = _cast_tuple(num_resnet_blocks, n_lv)
    attns = _cast_tuple(layer_attns, n_lv)
    spec = [("null_conditional_embed", (1, max_conditional_len, cond_dim)), ("null_conditional_hidden", (1, tdim))]
    cin0 = channels + cond_images_channels
    scales = [dim // 2, dim // 4]
    scales.append(dim - sum(scales))
    for i, (k, co) in enumerate(zip((3, 7, 15), scales)):
        spec += [(f"init_conv.convs.{i}.weight", (co, cin0, k, k)), (f"init_conv.convs.{i}.bias", (co,))]
    spec += [("to_time_hiddens.0.weights", (learned_sinu_pos_emb_dim // 2,)),
             ("to_time_hiddens.1.weight", (tdim, learned_sinu_pos_emb_dim + 1)), ("to_time_hiddens.1.bias", (tdim,)),
             ("to_time_cond.0.weight", (tdim, tdim)), ("to_time_cond.0.bias", (tdim,)),
             ("to_time_tokens.0.weight", (cond_dim * 2, tdim)), ("to_time_tokens.0.bias", (cond_dim * 2,)),
             ("norm_cond.weight", (cond_dim,)), ("norm_cond.bias", (cond_dim,))]

    def resnet(p, cin, cout, gca, cross):
        s = [(f"{p}.time_mlp.1.weight", (cout * 2, tdim)), (f"{p}.time_mlp.1.bias", (cout * 2,))]
        if cross:
            s += [(f"{p}.cross_attn.fn.null_kv", (2, attn_dim_head)), (f"{p}.cross_attn.fn.norm.g", (cout,)),
                  (f"{p}.cross_attn.fn.to_q.weight", (inner, cout)), (f"{p}.cross_attn.fn.to_kv.weight", (inner * 2, cond_dim)),
                  (f"{p}.cross_attn.fn.to_out.0.weight", (cout, inner)), (f"{p}.cross_attn.fn.to_out.1.g", (cout,))]
        for b, ci in (("block1", cin), ("block2", cout)):
            s += [(f"{p}.{b}.groupnorm.weight", (ci,)), (f"{p}.{b}.groupnorm.bias", (ci,)),
                  (f"{p}.{b}.project.weight", (cout, ci, 3, 3)), (f"{p}.{b}.project.bias", (cout,))]
        if gca:
            s += [(f"{p}.gca.to_k.weight", (1, cout, 1, 1)), (f"{p}.gca.to_k.bias", (1,)),
                  (f"{p}.gca.net.0.weight", (max(3, cout // 2), cout, 1, 1)), (f"{p}.gca.net.0.bias", (max(3, cout // 2),)),
                  (f"{p}.gca.net.2.weight", (cout, max(3, cout // 2), 1, 1)), (f"{p}.gca.net.2.bias", (cout,))]
        if cin != cout:
            s += [(f"{p}.res_conv.weight", (cout, cin, 1, 1)), (f"{p}.res_conv.bias", (cout,))]
        return s

    def attention(p, d, context):
        s = [(f"{p}.null_kv", (2, attn_dim_head)), (f"{p}.norm.g", (d,)), (f"{p}.to_q.weight", (inner, d)),
             (f"{p}.to_kv.weight", (attn_dim_head * 2, d))]
        if context:
            s += [(f"{p}.to_context.0.weight", (cond_dim,)), (f"{p}.to_context.0.bias", (cond_dim,)),
                  (f"{p}.to_context.1.weight", (attn_dim_head * 2, cond_dim)), (f"{p}.to_context.1.bias", (attn_dim_head * 2,))]
        return s + [(f"{p}.to_out.0.weight", (d, inner)), (f"{p}.to_out.1.g", (d,))]

    def transformer(p, d):
        hid = int(d * ff_mult)
        return attention(f"{p}.layers.0.0.fn", d, True) + [
            (f"{p}.layers.0.1.0.g", (1, d, 1, 1)), (f"{p}.layers.0.1.1.weight", (hid, d, 1, 1)),
            (f"{p}.layers.0.1.3.g", (1, hid, 1, 1)), (f"{p}.layers.0.1.4.weight", (d, hid, 1, 1))]

    for lv, (di, do) in enumerate(in_out):
        last = lv == n_lv - 1
        spec += resnet(f"downs.{lv}.1", di, di, False, False)
        for r in range(nres[lv]):
            spec += resnet(f"downs.{lv}.2.{r}", di, di, True, False)
        if attns[lv]:
            spec += transformer(f"downs.{lv}.3", di)
        if not last:
            spec += [(f"downs.{lv}.4.weight", (do, di, 4, 4)), (f"downs.{lv}.4.bias", (do,))]
        else:
            spec += [(f"downs.{lv}.4.fns.0.weight", (do, di, 3, 3)), (f"downs.{lv}.4.fns.0.bias", (do,)),
                     (f"downs.{lv}.4.fns.1.weight", (do, di, 1, 1)), (f"downs.{lv}.4.fns.1.bias", (do,))]
    for ui, lv in enumerate(reversed(range(n_lv))):
        di, do = in_out[lv]
        last = ui == n_lv - 1
        spec += resnet(f"ups.{ui}.0", do + di, do, False, False)
        for r in range(nres[lv]):
            spec += resnet(f"ups.{ui}.1.{r}", do + di, do, True, False)
        if attns[lv]:
            spec += transformer(f"ups.{ui}.2", do)
        if not last:
            spec += [(f"ups.{ui}.3.net.0.weight", (di * 4, do, 1, 1)), (f"ups.{ui}.3.net.0.bias", (di * 4,))]
    mid = dims[-1]
    spec += resnet("mid_block1", mid, mid, False, True)
    spec += attention("mid_attn.fn.fn", mid, False)
    spec += resnet("mid_block2", mid, mid, False, True)
    spec += resnet("final_res_block", dim, dim, True, False)
    spec += [("final_conv.weight", (channels, dim, 3, 3)), ("final_conv.bias", (channels,))]
    return spec


class _Node(nn.Module):
    """Anonymous container used to reproduce the reference's dotted parameter names."""


def _register(root, dotted, tensor):
    parts = dotted.split(".")
    node = root
    for p in parts[:-1]:
        if p not in node._modules:
            node.add_module(p, _Node())
        node = node._modules[p]
    node.register_parameter(parts[-1], nn.Parameter(tensor))


# ------------------------------------------------------------------------------------------------
# plan building
# ------------------------------------------------------------------------------------------------
class _Arena:
    """Bump allocator over one device tensor.  `base=None` = sizing pass (offsets only)."""

    def __init__(self, nbytes=None, device=None):
        self.off = 0
        self.buf = torch.empty(nbytes, dtype=torch.uint8, device=device) if nbytes else None

    def alloc(self, nbytes):
        nbytes = (nbytes + 255) // 256 * 256
        off = self.off
        self.off += nbytes
        return (self.buf.data_ptr() + off) if self.buf is not None else (1 << 20) + off


class _T:
    """A planned activation: device pointer + logical shape [B, HW, C] (NHWC) or [rows, C]."""
    __slots__ = ("ptr", "rows", "C", "HW", "lazy", "slots", "writer", "twin")

    def __init__(self, ptr, rows, C, HW=None):
        self.ptr, self.rows, self.C, self.HW = ptr, rows, C, HW
        # lazy: None, or how the first consumer must materialise the tensor (csrc/unet_ops.hip LazySrc):
        #   ("splitk", ws, bias, resid, groups, npad, ws index)   or   ("gate", h, gate, res)
        self.lazy = None
        # slots: device pointer of the [rows/16][C/16][2] (sum, sum of squares) table of the materialised values that a
        # GroupNorm-fused conv reads its statistics from (csrc/fused_kernels.h), or None
        self.slots = None
        self.twin = None            # operand-type copy [rows][C] written by the producing conv's epilogue (_Plan.conv twin=), or None
        # writer: the OP_CONV op (k_conv_lds) that wrote the whole tensor last, or None (experimental GroupNorm-partials epilogue)
        self.writer = None


class _Plan:
    def __init__(self, unet, B, device, sizing=None):
        self.u, self.B, self.dev = unet, B, device
        self.ops = []
        self.graph = None
        if sizing is None:
            self.zero, self.misc = _Arena(), _Arena()
        else:
            self.zero, self.misc = _Arena(sizing[0], device), _Arena(sizing[1], device)
        self.w = unet._packed(device)
        self.written = set()                 # (ptr, channel offset) of conv outputs that already hold data
        # two split-K workspaces (max demand over the ops that use each; ops run serially): a fused conv reads its input's
        # slabs from one while it writes its own partial tiles to the other
        self.ws_need = [0, 0]
        self.ws_ptrs = [self.misc.alloc(sizing[2]) if sizing is not None else 0,
                        self.misc.alloc(sizing[3]) if sizing is not None and len(sizing) > 3 and sizing[3] else 0]
        self.ws_owners = [None, None]        # tensors whose un-reduced split-K partials currently live in each workspace

    @property
    def ws_bytes(self):
        return self.ws_need[0]

    @property
    def ws2_bytes(self):
        return self.ws_need[1]

    @property
    def ws_owner(self):
        return next((o for o in self.ws_owners if o is not None and o.lazy is not None), None)

    def acquire_ws(self, nbytes, avoid=()):
        """Index of a workspace the op being emitted may overwrite: a free one if possible, never one that holds the
        partials of a tensor in `avoid` (an input of that op); a workspace owned by another tensor is reduced first."""
        for i in sorted(range(2), key=lambda j: (self.ws_owners[j] is not None and self.ws_owners[j].lazy is not None, j)):
            o = self.ws_owners[i]
            if o is not None and o.lazy is not None:
                if any(o is t for t in avoid):
                    continue
                self.need(o)
            self.ws_need[i] = max(self.ws_need[i], nbytes)
            self.ws_owners[i] = None
            return i
        raise AssertionError("no split-K workspace available")

    # -------- allocation helpers
    def zf32(self, rows, C, HW=None):        # conv outputs: first writer stores, later writers accumulate
        return self.f32(rows, C, HW)

    def f32(self, rows, C, HW=None):
        return _T(self.misc.alloc(rows * C * 4), rows, C, HW)

    def bf16(self, rows, C, HW=None):
        return _T(self.misc.alloc(rows * C * 2), rows, C, HW)

    def op(self, type_, flags=0, p=(), i=(), f=()):
        o = _lib.SfOp()
        o.type, o.flags = type_, flags
        for k, v in enumerate(p):
            o.p[k] = v if v else None
        for k, v in enumerate(i):
            o.i[k] = int(v)
        for k, v in enumerate(f):
            o.f[k] = float(v)
        self.ops.append(o)

    def wptr(self, name):
        return self.w[name].data_ptr()

    # -------- lazy tensors: the first consumer materialises them (saves one dependent launch each)
    def need(self, t):
        """Emit the stand-alone materialisation of a lazy tensor for consumers that cannot do it themselves."""
        if t is None or t.lazy is None:
            return t
        lz, t.lazy = t.lazy, None
        if lz[0] == "splitk":
            _, ws, bias, resid, groups, npad, wi = lz
            self.op(OP_SPLITK_REDUCE, 0, p=(ws, bias, resid, t.ptr), i=(t.rows, t.C, npad, groups))
            self.ws_owners[wi] = None
        else:
            _, h, gate, res = lz
            self.op(OP_ELTWISE, 1, p=(h, gate, res, t.ptr), i=(self.B, t.HW, t.C))
        return t

    def take_lazy(self, t, allow):
        """(p[8..10], (mode, groups, npad)) for a consumer that materialises `t` itself; clears the lazy state."""
        if t.lazy is None or t.lazy[0] not in allow:
            self.need(t)
            return (0, 0, 0), (0, 0, 0)
        lz, t.lazy = t.lazy, None
        if lz[0] == "splitk":
            self.ws_owners[lz[6]] = None
            return (lz[1], lz[2], lz[3]), (1, lz[4], lz[5])
        return (lz[1], lz[2], lz[3]), (2, 0, 0)

    # -------- op emitters
    def conv(self, x, x_f32, H, W, wname, bname, out, ldc, co_off, Cout, k, stride=1, pad=0, resid=None, pixshuf=False,
             defer=False, w_ptr=None, batch=None, out_hw=None, upsampled=False, relu=False, gelu=False, twin=None, want_slots=False,
             nchw=False, defer_max_groups=8):
        """One implicit-GEMM launch.  `w_ptr` replaces the named weight by a device-packed B operand (attention),
        `batch` overrides the plan batch (per-sample GEMMs), `out_hw` the output size (asymmetric padding),
        `upsampled` makes (H, W) the dims of a nearest-x2 view of the stored [H/2, W/2] input.  `twin`: a dense operand-type
        [M][Cout] buffer the epilogue of an LDS-tiled kernel also fills (the A operand of a following conv: no fp32 round trip);
        out.twin is set when the chosen kernel writes it."""
        B = self.B if batch is None else batch
        self.need(x)
        self.need(resid)
        Ho, Wo = out_hw or ((H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1)
        M = B * Ho * Wo
        m_frags, n_frags = (M + 15) // 16, (Cout + 15) // 16
        KS = k * k * (x.C // 32)
        WM, WN, groups = self.u.conv_tiling(m_frags, n_frags, KS, pixshuf)
        accum = (out.ptr, co_off) in self.written
        self.written.add((out.ptr, co_off))
        tile = WM * 16 + WN
        twin_ws = 0
        # large-M layers (VAE, VGG, B >= 4): the LDS-tiled kernel, 128 pixels x 128 (or 64) channels per workgroup
        lds_min = getattr(self.u, "lds_conv_min_blocks", 0)
        mid = getattr(self.u, "lds_mid_min_rows", 0) if B >= getattr(self.u, "lds_mid_min_batch", 8) else 0
        if lds_min and n_frags >= 4 and (not pixshuf or (mid and M >= mid and Cout % 4 == 0)):
            bnf = 8 if n_frags > 4 else 4
            blocks = ((m_frags + 7) // 8) * ((n_frags + bnf - 1) // bnf)
            if bnf == 8 and blocks < 256:                     # fewer tiles than CUs: halve the channel tile instead of idling CUs
                bnf, blocks = 4, ((m_frags + 7) // 8) * ((n_frags + 3) // 4)
            # 3x3 layers k_conv3_halo takes (csrc/conv_halo.h) beat the weight-streaming kernel from 64 tiles on (measured 21 vs 29 us
            # on the 32x32 512->512 layer); everything else needs lds_min tiles
            halo = (k == 3 and stride == 1 and pad == 1 and not x_f32 and x.C % 64 == 0 and W % 16 == 0 and H % 8 == 0 and M % 128 == 0
                    and Cout % 4 == 0 and ldc % 4 == 0 and co_off % 4 == 0 and (Ho, Wo) == (H, W) and not pixshuf)
            if blocks >= (min(lds_min, 64) if halo else lds_min) and not pixshuf:
                tile, groups = 256 + bnf, 1
                if twin is not None and ldc == Cout and co_off == 0:
                    twin_ws = twin.ptr
            elif mid and M >= mid:
                # r06: M of a few hundred rows (the 4x4 level of B >= 8, the Upsample 1x1 convs): still 128-row MFMA tiles out of LDS, the
                # workgroups that are missing come from split-K groups over the stage range (k_conv_lds / k_conv_glds; not under the
                # pixel shuffle, whose epilogue writes the output itself) -- every weight byte is still fetched by ONE workgroup per pixel tile
                glds_ok = not x_f32 and x.C % 64 == 0 and Cout % 4 == 0 and ldc % 4 == 0 and co_off % 4 == 0
                stages = KS // 2 if glds_ok else (KS + 1) // 2
                g = 1 if pixshuf else max(1, min(256 // blocks, stages // 4, 16))
                if glds_ok and k == 3 and stride == 1 and pad == 1 and H == W and H in (4, 8) and not pixshuf:
                    # whole 4x4 / 8x8 maps: k_conv3_halo_sm (csrc/conv_halo_small.h) splits K by 64-channel chunk, two chunks per group at least
                    # (the frames of the second are staged under the taps of the first)
                    g = max(1, min(256 // blocks, x.C // 128, 16))
                tile, groups = 256 + bnf, g
        ws, wi = 0, 0
        if groups > 1:
            wi = self.acquire_ws(groups * M * n_frags * 16 * 4)   # a workspace about to be overwritten is reduced first
            ws = self.ws_ptrs[wi]
        elif twin_ws:
            ws = twin_ws
        defer = bool(defer and not relu and not gelu and 1 < groups <= defer_max_groups and not accum and not pixshuf and co_off == 0 and ldc == Cout == out.C and M == out.rows
                     and (self.u.lazy_consumers & 1))
        bias, res = self.wptr(bname) if bname else 0, resid.ptr if resid else 0
        # r04: the SiLU + PixelShuffle epilogue of an Upsample also leaves the (sum, sum of squares) slots of its output for the next
        # GroupNorm-fused conv (was a k_slots launch): one slot per MFMA fragment, filed under the right (image, 16-channel column)
        slots = 0
        if want_slots and pixshuf and tile < 256 and (Cout // 4) % 16 == 0 and ldc % 16 == 0 and co_off % 16 == 0 and (Ho * Wo) % 16 == 0 and not accum:
            slots = self.misc.alloc(4 * M // 16 * (ldc // 16) * 2 * 4)
        # nchw: the (non-deferred) split-K reduction of this conv writes the plan's NCHW output directly (was k_unpack_out)
        nchw = bool(nchw and groups > 1 and not defer and not accum and not resid and tile < 256 and co_off == 0 and ldc == Cout)
        self.op(OP_CONV, (1 if x_f32 else 0) | (2 if pixshuf else 0) | (4 if accum else 0) | (8 if defer else 0) |
                (16 if upsampled else 0) | (32 if relu else 0) | (64 if gelu else 0) | (256 if nchw else 0),
                p=(x.ptr, w_ptr if w_ptr is not None else self.wptr(wname), bias, out.ptr, res, ws, 0, slots),
                i=(B, H, W, x.C, Ho, Wo, Cout, ldc, co_off, k, k, stride, pad, groups, tile))
        if defer:
            out.lazy = ("splitk", ws, bias, res, groups, n_frags * 16, wi)
            self.ws_owners[wi] = out
        out.slots = slots or None
        self.last_conv_nchw = nchw
        out.writer = self.ops[-1] if (tile >= 256 and co_off == 0 and ldc == Cout == out.C and M == out.rows) else None
        out.twin = twin if (twin is not None and tile >= 256 and groups == 1 and ws == twin.ptr) else None
        return Ho, Wo

    def gn_act(self, x, skip, gname, ss_ptr, out, raw=None, silu=True, groups=8, eps=1e-5):
        C1, C2 = x.C, (skip.C if skip else 0)
        self.need(skip)
        lp, li = self.take_lazy(x, ("splitk", "gate"))
        stats = self.zero.alloc(self.B * groups * 2 * 8)     # f64 (sum, sum of squares) per (b, group), zeroed per eval
        ready = 0
        wr = x.writer
        cg = (C1 // groups) if not skip else 0
        if (getattr(self.u, "gn_epilogue", False) and wr is not None and not (wr.flags & 128) and li[0] == 0 and cg in (4, 8, 16)
                and x.HW and x.HW % 128 == 0 and x.rows == self.B * x.HW):
            # (VAE plans; SF_VAE_GN_EPI=0 disables) the producing k_conv_lds leaves per-tile partial sums, k_gn_finalize adds them up,
            # and the statistics pass over the tensor (k_gn_stats_px) is skipped
            part = self.misc.alloc(x.rows // 128 * groups * 2 * 8)
            wr.flags |= 128
            wr.p[6] = part or None
            wr.i[15] = cg
            self.op(OP_GN_FINALIZE, 0, p=(part, stats), i=(self.B, x.HW // 128, groups))
            ready = 2
        self.op(OP_GN_ACT, (0 if silu else 1) | ready | (0 if getattr(self.u, "gn_one", True) else 4),      # (flag 4: k_gn_stats + k_gn_apply even where k_gn_one fits)
                p=(x.ptr, skip.ptr if skip else 0, self.wptr(gname + ".weight"), self.wptr(gname + ".bias"), ss_ptr, out.ptr,
                   raw.ptr if raw else 0, stats) + lp,
                i=(self.B, x.HW, C1, C2, getattr(self.u, "tb_stride", 0)) + li + (groups,), f=(eps, SKIP_SCALE))

    def ln(self, x, gname, bname, out, C, rows, eps=1e-5, gelu=False, out_f32=False, resid=None, twin=None):
        """`twin`: an operand-type [rows][C] buffer that also receives the (fp32-output) result -- the A operand of the next linear
        (k_layernorm_w256: C = 256, rows >= 1024 only)."""
        self.need(x)
        self.need(resid)
        assert twin is None or (out_f32 and C == 256 and rows >= 1024)
        self.op(OP_LN, (1 if gelu else 0) | (2 if out_f32 else 0) | (0 if getattr(self.u, "ln_wave", True) else 4),
                p=(x.ptr, self.wptr(gname), self.wptr(bname) if bname else 0, out.ptr, resid.ptr if resid else 0, twin.ptr if twin else 0),
                i=(rows, C), f=(eps,))
        out.twin = twin

    def gemv(self, x_ptr, M, ldx, wname, bname, y_ptr, ldy, N, K, in_silu=False, out_act=0):
        Kp = (K + 7) // 8 * 8
        # > 8 rows (a sampler's time table): k_gemm_rows, up to 64 rows per launch 
        step = 8 if M <= 8 else 64
        for m0 in range(0, M, step):
            mm = min(step, M - m0)
            self.op(OP_GEMV, (1 if in_silu else 0) | (out_act << 1),
                    p=(x_ptr + m0 * ldx * 4, self.wptr(wname), self.wptr(bname) if bname else 0, y_ptr + m0 * ldy * 4),
                    i=(mm, N, K, Kp, ldx, ldy))

    def attn(self, q, out, segs, heads, ldq, scale, out_f32=False):
        p = [q.ptr, out.ptr]
        i = [self.B, heads, ldq, 0]
        for s in (segs + [None] * 3)[:3]:
            if s is None:
                p += [0, 0]
                i += [0, 0, 0, 0]
            else:
                p += [s[0], s[1]]
                i += list(s[2:])
        self.op(OP_ATTN, 1 if out_f32 else 0, p=p, i=i, f=(scale,))


    # -------- fused GroupNorm / LayerNorm -> conv ops (csrc/fused_kernels.h)
    def fused_geometry(self, H, C, Cout, norm, k):
        """(TR, WM, WN, S) of a k_conv_fused launch on an H x H map, or None when the layer does not fit the kernel."""
        B = self.B
        if not getattr(self.u, "fused", False) or H not in (4, 8, 16, 32) or C % 32 or Cout % 16:
            return None
        TR, WM = {4: (4, 1), 8: (2, 1), 16: (1, 1), 32: (1, 2)}[H]
        if B >= getattr(self.u, "big_tile_min_batch", 2) and H in (8, 16, 32):
            # twice the pixels per workgroup (32 at 8x8 / 16x16, 64 at 32x32) once the batch fills the chip: half the weight
            # re-reads per pixel, a smaller halo share.  conv1 || res_conv of a
            # ResnetBlock share a launch and must agree on the tile: the 1x1 half decides as its 3x3 partner (same C) does.
            fit3 = self._fused_geometry(H, C, Cout, norm if k == 3 else FNORM_GN_SLOTS, 3, 2 * TR, 2 * WM)
            g = self._fused_geometry(H, C, Cout, norm, k, 2 * TR, 2 * WM)
            if fit3 is not None and g is not None:
                return g
        return self._fused_geometry(H, C, Cout, norm, k, TR, WM)

    def _fused_geometry(self, H, C, Cout, norm, k, TR, WM):
        B = self.B
        n_frags = Cout // 16
        MT = B * (H // TR)
        h = k // 2

        pipe_ok = (getattr(self.u, "fconv_pipe", False) and norm == FNORM_GN_SLOTS and k == 3 and C % 128 == 0
                   and ((TR + 2) * H) % 8 == 0)

        def lds_bytes(S_, WN_):
            if pipe_ok and S_ == 1 and (WM, WN_, (TR + 2) * H // 8) in PIPE_TILES:      # k_conv_fused_pipe: two 128-channel frames
                return self.pipe_lds_bytes(H, C, TR, WM, WN_)
            Cs = C // S_
            stride = Cs * 2 + ((32 - (Cs * 2) % 256) + 256) % 256
            return ((TR + 2 * h) * (H + 2 * h) + 1) * stride + 8192 * WM * WN_ + 2 * Cs * 4 + 2688 + (ATTN_LDS_BYTES + 16 if norm == FNORM_ATTN else 0)

        cand = [1]
        if norm in (FNORM_GN_SELF, FNORM_GN_SLOTS):
            if C % 8:
                return None
            Cg = C // 8
            if Cg % 4 or (norm == FNORM_GN_SLOTS and Cg % 16):
                return None
            if norm == FNORM_GN_SELF:
                if H != 4:
                    return None
                # input-channel slices of whole groups: enough of them to give every CU a workgroup and to fit the LDS
                cand = [d for d in (1, 2, 4, 8) if (C // 32) % d == 0 and (C // d) % Cg == 0]
        cand = [d for d in cand if lds_bytes(d, 1) <= LDS_MAX]
        if not cand:
            return None
        S = next((d for d in cand if MT * n_frags * d >= 256), cand[-1])
        if ((B <= getattr(self.u, "conv4_slices_max_batch", 4) or (getattr(self.u, "conv4_mb", True) and B % 2 == 0)) and getattr(self.u, "conv4", True)
                and norm == FNORM_GN_SELF and H == 4 and k == 3 and C in (1024, 2048) and 4 in cand):
            S = 4               # r05: keep k_conv4_gn's 4-slice geometry where the workgroup count alone would pick fewer slices (B = 2: S = 2, B = 4:
                                # S = 1, both on the general kernel): B = 2 eval 1.287 -> 1.240 ms, B = 4 1.734 -> 1.730 (profiles/r05_conv4_batch_ab.log);
                                # even B of any size: k_conv4_gn_mb, 2 | 4 images per workgroup on one weight slice (csrc/fused_conv4.h)
        WN = 2 if (n_frags % 2 == 0 and MT * (n_frags // 2) * S >= 256 and lds_bytes(S, 2) <= LDS_MAX and norm != FNORM_GN_SELF) else 1
        return TR, WM, WN, S

    @staticmethod
    def pipe_lds_bytes(H, C, TR, WM, WN, pool=False):
        """LDS bytes of a k_conv_fused_pipe launch (csrc/fused_host.h fconv_setup, pipe branch): two 128-channel frames, the K-slice
        reduction buffer of the 4 matrix waves, the affine table, statistics; POOL adds one logit fragment per m-fragment to the
        reduction buffer and the w_eff table (9 * C / 32 k-steps x 64 bytes)."""
        buf = (((TR + 2) * (H + 2) + 1) * 288 + 15) // 16 * 16
        return 2 * buf + 4096 * (WM * WN + (WM if pool else 0)) + 2 * C * 4 + 2688 + (9 * (C // 32) * 64 if pool else 0)

    def ensure_slots(self, t):
        """Make `t` a materialised tensor with a (sum, sum of squares) slot table (one launch when it has none)."""
        if t.slots is not None and t.lazy is None:
            return t
        assert t.rows % 16 == 0 and t.C % 16 == 0
        slots = self.misc.alloc(t.rows // 16 * (t.C // 16) * 2 * 4)
        if t.lazy is not None and t.lazy[0] == "gate" and t.lazy[3] and t.lazy[3] != t.ptr:
            _, h, gate, res = t.lazy
            t.lazy = None
            self.op(OP_SLOTS, 0, p=(h, gate, res, t.ptr, slots), i=(t.rows, t.C, t.HW))
        elif t.lazy is not None and t.lazy[0] == "splitk" and not t.lazy[3] and t.lazy[5] % 4 == 0:
            _, ws, bias, _, groups, npad, wi = t.lazy          # deferred split-K conv: reduce + slots in one launch
            t.lazy = None
            self.ws_owners[wi] = None
            self.op(OP_SLOTS, 0, p=(0, 0, 0, t.ptr, slots, ws, bias), i=(t.rows, t.C, t.HW or t.rows, groups, npad))
        else:
            self.need(t)
            self.op(OP_SLOTS, 0, p=(t.ptr, 0, 0, 0, slots), i=(t.rows, t.C, t.HW or t.rows))
        t.slots = slots
        return t

    def fconv(self, x, skip, H, wname, bname, out, Cout, k, norm, geom, gname=None, ss_ptr=0, silu=True, resid=None,
              want_slots=False, pre_gelu=False, beta_name=None, ldc=None, co_off=0, logit=None, out_gelu=False,
              pair_first=False, pair_lazy=None, pool=None, attn=None):
        """One k_conv_fused launch: out = conv_k(act(norm(concat(x, skip * 2^-1/2)))).  With S > 1 input-channel slices the
        output stays a lazy split-K tensor (slabs + bias + resid) that the next fused conv / GroupNorm / gca pass reduces."""
        TR, WM, WN, S = geom
        B = self.B
        C1, C2 = x.C, (skip.C if skip else 0)
        self.need(skip)
        self.need(resid)
        if norm == FNORM_GN_SLOTS:
            self.ensure_slots(x)
            if skip:
                self.ensure_slots(skip)
        ldc = ldc or Cout
        accum = (out.ptr, co_off) in self.written
        n_frags = (Cout + 15) // 16
        ws, wi = 0, 0
        if S > 1:                                               # before x's slabs are released: never write the workspace we read
            assert not accum and ldc == Cout and co_off == 0
            wi = self.acquire_ws(S * x.rows * n_frags * 16 * 4, avoid=(x,))
            ws = self.ws_ptrs[wi]
        else:
            self.written.add((out.ptr, co_off))
        lp, li = (0, 0, 0), (0, 0, 0)
        x_ptr = x.ptr
        if pair_lazy is not None:                               # second half of a pair: the first half's (lazy) view of x, read-only
            lp, li = pair_lazy
            if li[0]:
                x_ptr = 0
        elif x.lazy is not None:
            if x.lazy[0] == "gate" and (not x.lazy[3] or x.lazy[3] == x.ptr):
                self.need(x)                                    # res lives in the target buffer: only a one-reader kernel may do that
            else:
                lp, li = self.take_lazy(x, ("splitk", "gate"))
        slots_out = 0
        if want_slots and S == 1:
            if out.slots is None:
                out.slots = self.misc.alloc(out.rows // 16 * (ldc // 16) * 2 * 4)
            slots_out = out.slots
        bias, res = (self.wptr(bname) if bname else 0), (resid.ptr if resid else 0)
        gam = self.wptr(gname + ".weight") if norm in (FNORM_GN_SELF, FNORM_GN_SLOTS) else (self.wptr(gname) if gname else 0)
        bet = self.wptr(gname + ".bias") if norm in (FNORM_GN_SELF, FNORM_GN_SLOTS) else (self.wptr(beta_name) if beta_name else 0)
        assert not (out_gelu and S > 1)
        # staging and matrix work overlapped inside the workgroup (k_conv_fused_pipe) where the layer fits that kernel
        pipe = self.pipe_ok(C1, C2, H, geom, norm, k, silu) and li[0] == 0 and pair_lazy is None
        if pool is not None:                                    # GlobalContext pooling in the epilogue: (w_eff ptr, pooled-fragment buffer)
            assert pipe and logit is None and not accum and resid is None and ldc == Cout and co_off == 0
            logit = pool
        ap, ai, af = (), (), ()
        if attn is not None:                                    # FNORM_ATTN: x.ptr = the q rows; <= 3 key / value segments (csrc/fused_host.h)
            segs, ldq, scale = attn
            assert norm == FNORM_ATTN and len(segs) <= 3 and not silu and x.lazy is None and lp == (0, 0, 0)
            segs = list(segs) + [(0, 0, 0, 0, 0, 0)] * (3 - len(segs))
            assert all((sv - sk) % 4 == 0 and 0 <= (sv - sk) // 4 < (1 << 24) for sk, sv, *_ in segs)
            ap = tuple(sk for sk, *_ in segs)
            ai = (ldq,) + tuple(v for _, _, rows, rs, bs, hs in segs for v in (rows, rs, bs, hs))
            af = tuple((sv - sk) // 4 for sk, sv, *_ in segs) + (scale,)
        elif norm == FNORM_GN_SELF and not getattr(self.u, "conv4_mb", True):
            ai = (1,)                                           # i[19] bit 0: one image per workgroup (k_conv4_gn) where k_conv4_gn_mb would take the op
        elif pipe:
            # r06: the recurring single-source geometries run on k_conv3s (csrc/fused_conv3s.h; the host takes the op when a variant of its
            # shape exists).  i[19] bit 1 keeps the general pipelined kernel, bits 2.. = tile width of a 2-D pixel tile (0 = full-width strips)
            code = 0 if getattr(self.u, "conv3s", True) else 2
            tw = getattr(self.u, f"conv3s_tw{H}", 0)
            if (code == 0 and tw and tw != H and C2 == 0 and Cout == C1 and ldc == Cout and co_off == 0 and not accum and not out_gelu and not pre_gelu
                    and not pair_first and pair_lazy is None and (logit is None or pool is not None) and bname
                    and (H.bit_length() - 1, C1, tw.bit_length() - 1, WM, WN) in CONV3S_VARIANTS):
                code |= tw << 2
            elif (code == 0 and pair_first and C2 and ldc == Cout and co_off == 0 and not accum and not out_gelu and not pre_gelu and logit is None and bname
                    and (H.bit_length() - 1, C1, C2, Cout, (tw or H).bit_length() - 1, WM, WN) in CONV3S_RC_VARIANTS):
                code |= (tw if tw != H else 0) << 2                # the pair runs on k_conv3s_rc (the host checks that the res_conv fits)
            ai = (code,)
        self.op(OP_FCONV, (1 if silu else 0) | (2 if pre_gelu else 0) | (4 if accum else 0) | (8 if out_gelu else 0) | (16 if pair_first else 0)
                | (32 if pipe else 0) | (64 if pool is not None else 0) | (0 if getattr(self.u, "conv4", True) else 128),
                p=(x_ptr, lp[0], lp[1], lp[2], x.slots or 0, skip.ptr if skip else 0, (skip.slots or 0) if skip else 0,
                   self.wptr(wname), 0 if S > 1 else bias, out.ptr, 0 if S > 1 else res, ws, slots_out, gam, bet, ss_ptr, 0,
                   logit[0] if logit else 0, logit[1] if logit else 0) + ap,
                i=(B, H, H, C1, C2, Cout, ldc, co_off, k, li[0], li[1], li[2], norm, 8, TR, WM, WN, S, self.u.tb_stride) + ai,
                f=(1e-5, 1.0, SKIP_SCALE) + af)
        if S > 1:
            out.lazy = ("splitk", ws, bias, res, S, n_frags * 16, wi)
            self.ws_owners[wi] = out
            out.slots = None
            thr = getattr(self.u, "conv4_reduce_min_batch", 0)
            if thr and B >= thr and H == 4 and norm == FNORM_GN_SELF and logit is None and not pair_first:
                self.need(out)      # (a paired conv1 is followed by its res_conv in the SAME launch: no op may stand between them -- resnet_fused reduces after the pair)
                                    # r05: reduce the slabs in their own launch -- the consuming conv's 64 n-tile workgroups then gather one float4 per
                                    # element instead of five (~100 MB of L2 reads per consuming launch at B = 4): B = 4 eval 1.693 -> 1.681 ms,
                                    # B = 32 7.24 -> 6.86 ms (profiles/r05_conv4_mb_ab.log); at B = 1 the extra launch costs more than the gather
        return lp, li

    def pipe_ok(self, C1, C2, H, geom, norm, k, silu=True):
        """A slot-GroupNorm 3x3 conv with a plain source runs on k_conv_fused_pipe (staging || matrix work) when its tile exists."""
        TR, WM, WN, S = geom
        return bool(getattr(self.u, "fconv_pipe", False) and norm == FNORM_GN_SLOTS and k == 3 and S == 1 and silu
                    and (C1 + C2) % 128 == 0 and C1 % 4 == 0 and ((TR + 2) * H) % 8 == 0 and (WM, WN, (TR + 2) * H // 8) in PIPE_TILES)

    def resnet_fused(self, name, x, skip, cout, H, gca=False, cross=False):
        """ResnetBlock (imagen_pytorch.py:665-729) with both GroupNorms inside their convs: 2 launches (+ res_conv, + gca)
        instead of 6-10.  Returns None when a layer of the block does not fit the fused kernel."""
        B, HW = self.B, H * H
        cin = x.C + (skip.C if skip else 0)
        rows = B * HW
        norm = FNORM_GN_SELF if H == 4 else FNORM_GN_SLOTS
        g1 = self.fused_geometry(H, cin, cout, norm, 3)
        g2 = self.fused_geometry(H, cout, cout, norm, 3)
        gr = self.fused_geometry(H, cin, cout, FNORM_NONE, 1) if cin != cout else g1
        if g1 is None or g2 is None or gr is None or x.C % 32 or rows % 16:
            return None
        if (B >= 2 and (getattr(self.u, "rc_small_tiles", 0) & H) and cin != cout and skip is not None and getattr(self.u, "conv3s", True)
                and getattr(self.u, "pair_res_conv", True)):
            # r06: a (conv1 on the concat || res_conv) pair of B >= 2 keeps the B = 1 tile where k_conv3s_rc has it: the 32- / 64-pixel tiles of
            # B >= 2 do not fit that kernel's registers, and the general pair kernel they fall back to is slower than more workgroups of the small one
            TR0, WM0 = {8: (2, 1), 16: (1, 1), 32: (1, 2)}[H]
            s1, sr = self._fused_geometry(H, cin, cout, norm, 3, TR0, WM0), self._fused_geometry(H, cin, cout, FNORM_NONE, 1, TR0, WM0)
            tw = getattr(self.u, f"conv3s_tw{H}", 0) or H
            if (s1 is not None and sr is not None and s1[1:3] == sr[1:3] and sr[3] == 1
                    and (H.bit_length() - 1, x.C, skip.C, cout, tw.bit_length() - 1, s1[1], s1[2]) in CONV3S_RC_VARIANTS):
                g1, gr = s1, sr
        slots = norm == FNORM_GN_SLOTS
        h = self.zf32(rows, cout, HW)
        # conv1 and res_conv read the same input and are independent: one launch (k_conv_fused_pair) when their tiles match
        pair = (cin != cout and getattr(self.u, "pair_res_conv", True) and g1[1:3] == gr[1:3] and gr[3] == 1
                and (g1[1], g1[2], norm) in PAIR_TILES)
        # r04, 4x4 level: a gca block's res_conv is only needed by the gate, so it runs beside the 16-workgroup pooling launch
        # (k_gca_pool_rc) on the materialised input instead of as a second round of workgroups re-reducing conv1's lazy source
        late_rc = (cin != cout and gca and H == 4 and getattr(self.u, "res_conv_beside_pool", True) and gr[1:4] == (1, 1, 1)
                   and cout % 64 == 0 and cout <= 2048 and cin <= 4096)    # (LDS: 16 pixels x cin operands beside the pooling's 9 KB)
        # r05: conv1 of the 4x4 level runs on k_conv4_gn where its geometry fits (csrc/fused_conv4.h); the pair kernel is the general one,
        # so such a block emits conv1 and its res_conv as two launches (26.7 us as a pair against ~10 + 9.3 us)
        conv4_geom = (H == 4 and getattr(self.u, "conv4", True) and g1[1:3] == (1, 1) and g1[3] > 1 and cin // g1[3] in (256, 512)
                      and (cin // 8) * 2 == cin // g1[3])
        pair = pair and not late_rc and not conv4_geom
        lz = self.fconv(x, skip, H, f"{name}.block1.project.weight", f"{name}.block1.project.bias", h, cout, 3, norm, g1,
                        gname=f"{name}.block1.groupnorm", want_slots=slots, pair_first=pair)
        rc = None
        emit_rc = None
        if cin != cout:                                             # res_conv reads the raw concat (x is materialised by conv1)
            rc = self.zf32(rows, cout, HW)
            emit_rc = lambda first=False: self.fconv(x, skip, H, f"{name}.res_conv.weight", f"{name}.res_conv.bias", rc, cout, 1,
                                                     FNORM_NONE, gr, silu=False, pair_lazy=lz if pair else None, pair_first=first)
            if not late_rc:
                emit_rc()
                thr = getattr(self.u, "conv4_reduce_min_batch", 0)
                if pair and thr and B >= thr and H == 4 and norm == FNORM_GN_SELF and h.lazy is not None and h.lazy[0] == "splitk":
                    self.need(h)                                        # the own-launch reduction of fconv(), behind the pair
        if cross:
            h = self.cross_attention(f"{name}.cross_attn.fn", h)
        ss_ptr = self.ss.ptr + self.u.ss_offset[name] * 4
        w2, b2, gn2 = f"{name}.block2.project.weight", f"{name}.block2.project.bias", f"{name}.block2.groupnorm"
        res = rc if rc is not None else x
        out = self.zf32(rows, cout, HW)
        if not gca:
            self.fconv(h, None, H, w2, b2, out, cout, 3, norm, g2, gname=gn2, ss_ptr=ss_ptr, resid=res, want_slots=slots)
            return out
        h2 = self.zf32(rows, cout, HW)
        if (getattr(self.u, "gca_epilogue_pool", True) and cout % 64 == 0 and cout <= 2048 and HW % 16 == 0 and HW // 16 <= 64
                and self.pipe_ok(cout, 0, H, g2, norm, 3) and h.lazy is None and f"{name}.__weff__" in self.w
                and self.pipe_lds_bytes(H, cout, g2[0], g2[1], g2[2], pool=True) <= LDS_MAX):      # else: the k_gca_pool plan below
            # r03: the softmax pooling of the GlobalContext rides in conv2's epilogue (k_conv_fused_pipe<.., POOL>: context logits
            # from the conv's own staged input through w_eff, one pooled fragment per 16 pixels) -> net0 -> gate: 2 launches
            if late_rc:                                           # (unreachable today: late_rc needs H == 4 = GN_SELF, pipe_ok needs GN_SLOTS;
                emit_rc()                                         #  a planner change must not leave `res = rc` unwritten)
            pbuf = self.f32(rows // 16, cout + 2)                 # [M/16][cout] pooled fragments, then [M/16][2] (max, sum of exp)
            self.fconv(h, None, H, w2, b2, h2, cout, 3, norm, g2, gname=gn2, ss_ptr=ss_ptr, pool=(self.wptr(f"{name}.__weff__"), pbuf.ptr))
            self.gca_fused(name, h2, cout, res, out, None, 0, slots, pooled=(pbuf.ptr, pbuf.ptr + rows // 16 * cout * 4, HW // 16))
            return out
        if cout % 64 == 0 and cout <= 2048 and HW % 16 == 0:
            # fused GlobalContext: conv2's epilogue leaves partial context logits, then pool -> net0 -> gate (+ residual, + slots)
            nparts = g2[3] * (cout // 16)
            lpart = self.f32(nparts, rows)
            self.fconv(h, None, H, w2, b2, h2, cout, 3, norm, g2, gname=gn2, ss_ptr=ss_ptr,
                       logit=(self.wptr(f"{name}.gca.to_k.weight"), lpart.ptr))
            self.gca_fused(name, h2, cout, res, out, lpart, nparts, slots, before_pool=emit_rc if late_rc else None)
            return out
        if late_rc:
            emit_rc()
        self.fconv(h, None, H, w2, b2, h2, cout, 3, norm, g2, gname=gn2, ss_ptr=ss_ptr)
        gate = self.gca_gate(name, h2, cout)
        out.lazy = ("gate", h2.ptr, gate.ptr, res.ptr)
        return out

    def gca_fused(self, name, h2, cout, res, out, lpart, nparts, want_slots, pooled=None, before_pool=None):
        """GlobalContext + gated residual in three launches (csrc/fused_gca.h): out = h2 * gca(h2) + res, materialised, with
        its statistics slots when the consumer is a slot-GroupNorm conv."""
        B, HW, rows = self.B, h2.HW, h2.rows
        chunks = min(8, HW // 16)
        CH = HW // chunks
        hidc = max(3, cout // 2)
        part_pool, part_ms, hid = self.f32(B * chunks, cout), self.f32(B * chunks, 2), self.f32(B, hidc)
        ws = bias = groups = npad = 0
        if h2.lazy is not None:
            assert h2.lazy[0] == "splitk" and not h2.lazy[3]
            _, ws, bias, _, groups, npad, wi = h2.lazy
            h2.lazy = None
            self.ws_owners[wi] = None
        if before_pool is None:
            self.need(res)
        if pooled is not None:                                  # the producing conv's epilogue already pooled 16-pixel fragments
            pp, pm, chunks = pooled
        else:
            pp, pm = part_pool.ptr, part_ms.ptr
            if before_pool is not None:                         # the block's res_conv shares this launch (flag 16 on the fconv)
                before_pool(True)
            self.op(OP_GCA, 1, p=(h2.ptr, ws, bias, lpart.ptr, pp, pm), i=(rows, cout, HW, CH, chunks, nparts, groups, npad))
        if want_slots:
            out.slots = self.misc.alloc(rows // 16 * (cout // 16) * 2 * 4)
        self.op(OP_GCA, 2, p=(pp, pm, self.wptr(f"{name}.gca.net.0.weight"), self.wptr(f"{name}.gca.net.0.bias"),
                              hid.ptr), i=(B, cout, (cout + 7) // 8 * 8, hidc, chunks, 0 if getattr(self.u, "gate_t", True) else 1))
        self.op(OP_GCA, 3, p=(h2.ptr, res.ptr, hid.ptr, self.wptr(f"{name}.gca.net.2.weight"), self.wptr(f"{name}.gca.net.2.bias"),
                              out.ptr, out.slots or 0), i=(rows, cout, HW, hidc, (hidc + 7) // 8 * 8, 0 if getattr(self.u, "gate_t", True) else 1))

    def gca_gate(self, name, h2, cout):
        """GlobalContext gate of h2 (imagen_pytorch.py:916-941): softmax-pooled context -> two 1x1 convs -> sigmoid."""
        B, HW = self.B, h2.HW
        pooled = _T(self.zero.alloc(B * cout * 4), B, cout)        # accumulated with atomics: zeroed per eval
        hid = self.f32(B, max(3, cout // 2))
        gate = self.f32(B, cout)
        logits = self.f32(B, HW)
        lp, li = self.take_lazy(h2, ("splitk",))
        self.op(OP_GCA_POOL, 0, p=(h2.ptr, self.wptr(f"{name}.gca.to_k.weight"), self.wptr(f"{name}.gca.to_k.bias"), pooled.ptr,
                                   logits.ptr, 0, 0, 0) + lp, i=(B, HW, cout) + li)
        self.gemv(pooled.ptr, B, cout, f"{name}.gca.net.0.weight", f"{name}.gca.net.0.bias", hid.ptr, hid.C, hid.C, cout,
                  out_act=1)
        self.gemv(hid.ptr, B, hid.C, f"{name}.gca.net.2.weight", f"{name}.gca.net.2.bias", gate.ptr, cout, cout, hid.C,
                  out_act=2)
        return gate

    # -------- blocks
    def resnet(self, name, x, skip, cout, H, gca=False, cross=False):
        B, HW = self.B, H * H
        cin = x.C + (skip.C if skip else 0)
        rows = B * HW
        # large batches at the 32x32 / 16x16 levels: GroupNorm as its own pass + the 3x3 convs on k_conv3_halo (csrc/conv_halo.h, 700-950
        # TFLOP/s from 128 tiles on) beat the GroupNorm-fused weight-streaming kernels, which are built for M of a few tiles
        big = (rows >= (getattr(self.u, f"unfused_min_rows_{H}", 0) or getattr(self.u, "unfused_min_rows", 1 << 30)) and H % 16 == 0 and cin % 64 == 0
               and cout % 64 == 0)                  # (unfused_min_rows_16 / _32: a per-level threshold, 0 = unfused_min_rows)
        # r06: the 4x4 / 8x8 levels of large batches likewise (their 3x3 convs then run on k_conv_glds with split-K groups, conv(): lds_mid_min_rows)
        low = getattr(self.u, f"unfused_min_rows_{H}", 0) if H in (4, 8) else 0
        big = big or bool(low and rows >= low and getattr(self.u, "lds_mid_min_rows", 0) and cin % 64 == 0 and cout % 64 == 0)
        if getattr(self.u, "fused", False) and not big:
            y = self.resnet_fused(name, x, skip, cout, H, gca, cross)
            if y is not None:
                return y
        a1 = self.bf16(rows, cin, HW)
        raw = self.bf16(rows, cin, HW) if cin != cout else None
        self.gn_act(x, skip, f"{name}.block1.groupnorm", 0, a1, raw)
        h = self.zf32(rows, cout, HW)
        self.conv(a1, False, H, H, f"{name}.block1.project.weight", f"{name}.block1.project.bias", h, cout, 0, cout, 3, 1, 1,
                  defer=True)                                  # block2's GroupNorm statistics pass reduces the partials
        if cross:
            h = self.cross_attention(f"{name}.cross_attn.fn", h)
        a2 = self.bf16(rows, cout, HW)
        ss_ptr = self.ss.ptr + self.u.ss_offset[name] * 4
        self.gn_act(h, None, f"{name}.block2.groupnorm", ss_ptr, a2)
        out = self.zf32(rows, cout, HW)
        w2, b2 = f"{name}.block2.project.weight", f"{name}.block2.project.bias"
        if gca:
            h2 = self.zf32(rows, cout, HW)
            self.conv(a2, False, H, H, w2, b2, h2, cout, 0, cout, 3, 1, 1, defer=True)   # reduced by the gca logits pass
            pooled = _T(self.zero.alloc(B * cout * 4), B, cout)        # accumulated with atomics: zeroed per eval
            hid = self.f32(B, max(3, cout // 2))
            gate = self.f32(B, cout)
            logits = self.f32(B, HW)
            lp, li = self.take_lazy(h2, ("splitk",))
            self.op(OP_GCA_POOL, 0, p=(h2.ptr, self.wptr(f"{name}.gca.to_k.weight"), self.wptr(f"{name}.gca.to_k.bias"), pooled.ptr,
                                       logits.ptr, 0, 0, 0) + lp, i=(B, HW, cout) + li)
            self.gemv(pooled.ptr, B, cout, f"{name}.gca.net.0.weight", f"{name}.gca.net.0.bias", hid.ptr, hid.C, hid.C, cout,
                      out_act=1)
            self.gemv(hid.ptr, B, hid.C, f"{name}.gca.net.2.weight", f"{name}.gca.net.2.bias", gate.ptr, cout, cout, hid.C,
                      out_act=2)
            if raw is not None:
                self.conv(raw, False, H, H, f"{name}.res_conv.weight", f"{name}.res_conv.bias", out, cout, 0, cout, 1)
                out.lazy = ("gate", h2.ptr, gate.ptr, 0)               # out already holds res_conv(x)
            else:
                out.lazy = ("gate", h2.ptr, gate.ptr, x.ptr)
            if not (self.u.lazy_consumers & 2):
                self.need(out)
        else:
            if raw is not None:
                self.conv(raw, False, H, H, f"{name}.res_conv.weight", f"{name}.res_conv.bias", out, cout, 0, cout, 1)
                self.conv(a2, False, H, H, w2, b2, out, cout, 0, cout, 3, 1, 1)
            else:
                self.conv(a2, False, H, H, w2, b2, out, cout, 0, cout, 3, 1, 1, resid=x)
        return out

    def _attn_out(self, name, att, x, d, rows):
        o = self.zf32(rows, d)
        self.conv(att, False, 1, rows // self.B, f"{name}.to_out.0.weight", None, o, d, 0, d, 1)
        y = self.f32(rows, d, x.HW)
        self.ln(o, f"{name}.to_out.1.g", None, y, d, rows, out_f32=True, resid=x)
        return y

    # ---- attention on the fused linear (k_conv_fused k = 1: LayerNorm in the prologue, no split-K reduce launches)
    def lin_geometry(self, x, N, norm):
        """Geometry of a fused linear on a 4x4 token map, or None (the first-round ops then run)."""
        if x.HW != 16 or x.rows != self.B * 16:
            return None
        if x.rows >= getattr(self.u, "unfused_lin_min_rows", 1 << 30) and getattr(self.u, "lds_mid_min_rows", 0):
            return None     # r06: enough token rows for 128-row MFMA tiles -- LayerNorm pass + LDS-tiled 1x1 convs (conv(): lds_mid_min_rows)
        return self.fused_geometry(4, x.C, N, norm, 1)

    def attention_fused(self, name, x, context, cross):
        """x + Attention(x) with 4 launches: [LayerNorm + q (| k | v) projection], the 16-query attention core, the output
        projection, [LayerNorm + residual]  (imagen_pytorch.py:480-566, :731-805).  None when the shapes do not fit."""
        B, d, rows = self.B, x.C, x.rows
        heads, dh = self.u.attn_heads, self.u.attn_dim_head
        inner = heads * dh
        nq = inner if cross else inner + 2 * dh
        g1, g2 = self.lin_geometry(x, nq, FNORM_LN), None
        if g1 is None or d % 32:
            return None
        att = _T(self.misc.alloc(rows * inner * 4), rows, inner, 16)
        g2 = self.lin_geometry(att, d, FNORM_ATTN)            # the larger LDS frame of the two forms of the output projection
        if g2 is None:
            return None
        self.need(x)       # split-K slabs: one reduce launch beats re-reducing them in each of the projection's 40 workgroups
        qkv = self.zf32(rows, nq, 16)
        self.fconv(x, None, 4, f"{name}.to_q.weight" if cross else f"{name}.__qkv__", None, qkv, nq, 1, FNORM_LN, g1,
                   gname=f"{name}.norm.g", silu=False)
        nk = self.wptr(f"{name}.null_kv")
        segs = []
        if cross:                                              # keys = [null, the 2 time tokens] (time block)
            kvp = self.tb.ptr + self.u.tb_off[name] * 4
            segs = [(nk, nk + dh * 4, 1, 0, 0, 0), (kvp, kvp + inner * 4, 2, inner * 2, self.u.tb_stride, dh)]
        else:
            if context:
                ckvp = self.tb.ptr + self.u.tb_off[name] * 4
                segs.append((ckvp, ckvp + dh * 4, 2, 2 * dh, self.u.tb_stride, 0))
            kp = qkv.ptr + inner * 4                           # one shared k/v head right of the 8 query heads
            segs += [(nk, nk + dh * 4, 1, 0, 0, 0), (kp, kp + dh * 4, 16, nq, 16 * nq, 0)]
        o = self.zf32(rows, d, 16)
        # (the guard mirrors fused_host.h: the TOTAL number of keys, <= 4 if any segment keeps its own k / v per head, else <= 24)
        nkeys, per_head = sum(sg[2] for sg in segs), any(sg[5] for sg in segs if sg[2])
        if getattr(self.u, "attn_in_out_proj", True) and heads == 8 and dh == 64 and len(segs) <= 3 and 1 <= nkeys <= (4 if per_head else 24):
            # r04: the attention core runs in the prologue of its output projection (k_conv_fused<.., FNORM_ATTN>: wave = head, the 64
            # workgroups each redo the 16-token core -- 0.3 MFLOP -- instead of one more dependent launch)
            qv = _T(qkv.ptr, rows, inner, 16)
            self.fconv(qv, None, 4, f"{name}.to_out.0.weight", None, o, d, 1, FNORM_ATTN, g2, silu=False, attn=(segs, nq, dh ** -0.5))
        else:
            self.attn(qkv, att, segs, heads, nq, dh ** -0.5, out_f32=True)
            self.fconv(att, None, 4, f"{name}.to_out.0.weight", None, o, d, 1, FNORM_NONE, g2, silu=False)
        y = self.f32(rows, d, x.HW)
        self.ln(o, f"{name}.to_out.1.g", None, y, d, rows, out_f32=True, resid=x)
        return y

    def cross_attention(self, name, h):
        """h + CrossAttention(h, context=c)  (imagen_pytorch.py:731-805; keys = [null, 2 time tokens])."""
        if getattr(self.u, "fused", False):
            y = self.attention_fused(name, h, False, True)
            if y is not None:
                return y
        B, d, rows = self.B, h.C, h.rows
        heads, dh = self.u.attn_heads, self.u.attn_dim_head
        inner = heads * dh
        xn = self.bf16(rows, d)
        self.ln(h, f"{name}.norm.g", None, xn, d, rows)
        q = self.zf32(rows, inner)
        self.conv(xn, False, 1, rows // B, f"{name}.to_q.weight", None, q, inner, 0, inner, 1)
        att = self.bf16(rows, inner)
        nk = self.wptr(f"{name}.null_kv")
        kvp = self.tb.ptr + self.u.tb_off[name] * 4            # k/v of the 2 time tokens: computed by emit_time (time-only)
        segs = [(nk, nk + dh * 4, 1, 0, 0, 0), (kvp, kvp + inner * 4, 2, inner * 2, self.u.tb_stride, dh)]
        self.attn(q, att, segs, heads, inner, dh ** -0.5)
        return self._attn_out(name, att, h, d, rows)

    def self_attention(self, name, x, context):
        """x + Attention(x[, context=c])  (imagen_pytorch.py:480-566; one shared k/v head)."""
        if getattr(self.u, "fused", False):
            y = self.attention_fused(name, x, context, False)
            if y is not None:
                return y
        B, d, rows = self.B, x.C, x.rows
        heads, dh = self.u.attn_heads, self.u.attn_dim_head
        inner = heads * dh
        xn = self.bf16(rows, d)
        self.ln(x, f"{name}.norm.g", None, xn, d, rows)
        q, kv = self.zf32(rows, inner), self.zf32(rows, 2 * dh)
        self.conv(xn, False, 1, rows // B, f"{name}.to_q.weight", None, q, inner, 0, inner, 1)
        self.conv(xn, False, 1, rows // B, f"{name}.to_kv.weight", None, kv, 2 * dh, 0, 2 * dh, 1)
        segs = []
        if context:
            ckvp = self.tb.ptr + self.u.tb_off[name] * 4
            segs.append((ckvp, ckvp + dh * 4, 2, 2 * dh, self.u.tb_stride, 0))
        nk = self.wptr(f"{name}.null_kv")
        tok = rows // B
        segs += [(nk, nk + dh * 4, 1, 0, 0, 0), (kv.ptr, kv.ptr + dh * 4, tok, 2 * dh, tok * 2 * dh, 0)]
        att = self.bf16(rows, inner)
        self.attn(q, att, segs, heads, inner, dh ** -0.5)
        return self._attn_out(name, att, x, d, rows)

    def transformer(self, name, x):
        B, d, rows = self.B, x.C, x.rows
        x1 = self.self_attention(f"{name}.layers.0.0.fn", x, True)
        hid = int(d * self.u.ff_mult)
        if getattr(self.u, "fused", False):                      # ChanFeedForward with both ChanLayerNorms inside the 1x1 convs
            ga, gb = self.lin_geometry(x1, hid, FNORM_LN), None
            if ga is not None and hid % 32 == 0:
                f1 = self.zf32(rows, hid, 16)
                gb = self.lin_geometry(f1, d, FNORM_LN)
            if ga is not None and gb is not None:
                # GELU once, in ff1's epilogue (the 64 workgroups of ff2 would each redo it on the whole 16 x 2048 input)
                self.fconv(x1, None, 4, f"{name}.layers.0.1.1.weight", None, f1, hid, 1, FNORM_LN, ga,
                           gname=f"{name}.layers.0.1.0.g", silu=False, out_gelu=True)
                x2 = self.zf32(rows, d, x.HW)
                self.fconv(f1, None, 4, f"{name}.layers.0.1.4.weight", None, x2, d, 1, FNORM_LN, gb,
                           gname=f"{name}.layers.0.1.3.g", silu=False, resid=x1)
                return x2
        xn = self.bf16(rows, d)
        self.ln(x1, f"{name}.layers.0.1.0.g", None, xn, d, rows)
        f1 = self.zf32(rows, hid)
        self.conv(xn, False, 1, rows // B, f"{name}.layers.0.1.1.weight", None, f1, hid, 0, hid, 1)
        xn3 = self.bf16(rows, hid)
        self.ln(f1, f"{name}.layers.0.1.3.g", None, xn3, hid, rows, gelu=True)
        x2 = self.zf32(rows, d, x.HW)
        self.conv(xn3, False, 1, rows // B, f"{name}.layers.0.1.4.weight", None, x2, d, 0, d, 1, resid=x1)
        return x2

    # -------- time path (imagen_pytorch.py:1175-1190, :1514-1604) + the context projections that depend on it alone
    def emit_time(self, rows, tb_ptr, t_ptr):
        """Ops that fill `rows` time-block rows (u.tb_stride floats each, layout u.tb_off) from `rows` log-snr values."""
        u = self.u
        st = u.tb_stride
        half = u.learned_sinu_pos_emb_dim // 2
        heads, dh = u.attn_heads, u.attn_dim_head
        four, hid, t = self.f32(rows, 2 * half + 1), self.f32(rows, u.tdim), self.f32(rows, u.tdim)
        tok = self.f32(rows, 2 * u.cond_dim)
        self.op(OP_TIME_EMB, 0, p=(t_ptr, self.wptr("to_time_hiddens.0.weights"), 0, four.ptr), i=(rows, half))
        self.gemv(four.ptr, rows, four.C, "to_time_hiddens.1.weight", "to_time_hiddens.1.bias", hid.ptr, u.tdim, u.tdim, four.C, out_act=1)
        self.gemv(hid.ptr, rows, u.tdim, "to_time_cond.0.weight", "to_time_cond.0.bias", t.ptr, u.tdim, u.tdim, u.tdim)
        self.gemv(hid.ptr, rows, u.tdim, "to_time_tokens.0.weight", "to_time_tokens.0.bias", tok.ptr, 2 * u.cond_dim, 2 * u.cond_dim, u.tdim)
        c = self.f32(2 * rows, u.cond_dim)                 # the 2 time tokens of every row, norm_cond applied
        self.ln(tok, "norm_cond.weight", "norm_cond.bias", c, u.cond_dim, 2 * rows, out_f32=True)
        # all 27 time_mlp outputs in one GEMV (SiLU -> Linear)
        self.gemv(t.ptr, rows, u.tdim, "__time_mlps__.weight", "__time_mlps__.bias", tb_ptr, st, u.ss_total, u.tdim, in_silu=True)
        for name, off in u.tb_off.items():
            if name.endswith(".cross_attn.fn"):            # CrossAttention.to_kv(context) (:731-805)
                n_out, src, wname, bname = 2 * heads * dh, c, f"{name}.to_kv.weight", None
            else:                                          # Attention.to_context = LayerNorm + Linear (:480-566)
                src = self.f32(2 * rows, u.cond_dim)
                self.ln(c, f"{name}.to_context.0.weight", f"{name}.to_context.0.bias", src, u.cond_dim, 2 * rows, out_f32=True)
                n_out, wname, bname = 2 * dh, f"{name}.to_context.1.weight", f"{name}.to_context.1.bias"
            for tk in range(2):                            # token tk of every row -> its k|v slot of that row's block
                self.gemv(src.ptr + tk * u.cond_dim * 4, rows, 2 * u.cond_dim, wname, bname, tb_ptr + (off + tk * n_out) * 4, st,
                          n_out, u.cond_dim)

    # -------- whole network
    def build(self):
        u, B = self.u, self.B
        R = u.image_size
        HW = R * R
        cin0 = u.channels + u.cond_images_channels
        cpad = (cin0 + 31) // 32 * 32
        self.x_in, self.t_in = self.f32(B, u.channels * HW), self.f32(B, 1)
        self.cond_in = self.f32(B, u.cond_images_channels * HW)
        self.tb = self.f32(B, u.tb_stride)
        self.ss = self.tb
        self.emit_time(B, self.tb.ptr, self.t_in.ptr)
        self.n_time_ops = len(self.ops)                   # a sampler replays ops[n_time_ops:] after gathering its table row
        self.op(OP_MEMSET, 0, p=(self.zero.buf.data_ptr() if self.zero.buf is not None else 1,), i=(0,))   # size patched below
        memset_op = self.ops[-1]
        xin = self.f32(B * HW, cpad, HW)
        self.op(OP_ELTWISE, 2, p=(self.cond_in.ptr, self.x_in.ptr, 0, xin.ptr), i=(B, HW, u.cond_images_channels, u.channels, cpad))
        # init conv: CrossEmbedLayer k = 3 / 7 / 15 into channel slices (:1017-1042)
        x = self.zf32(B * HW, u.dim, HW)
        co = 0
        for i, k in enumerate((3, 7, 15)):
            cw = u.spec_shapes[f"init_conv.convs.{i}.weight"][0]
            self.conv(xin, True, R, R, f"init_conv.convs.{i}.weight", f"init_conv.convs.{i}.bias", x, u.dim, co, cw, k, 1, k // 2)
            co += cw
        self.n_init_ops = len(self.ops)
        # Sampler variant of the init conv.  The conv is linear in its input channels and the conditioning image (256 of the
        # 260 channels) is the same for every eval of a trajectory: `begin_sampling` runs the ops above once with x = 0 and
        # keeps the result (bias included) in `base`; an eval then only convolves the 4 latent channels and adds `base`.
        self.x0, self.base = x, self.f32(B * HW, u.dim, HW)
        full_ops, full_written, self.ops, self.written = self.ops, self.written, [], set()
        cws = [u.spec_shapes[f"init_conv.convs.{i}.weight"][0] for i in range(3)]
        if (getattr(u, "initx_direct", True) and u.fused and R % 8 == 0 and u.channels <= 4 and cws[0] in (64, 128)
                and cws[1] in (32, 64) and cws[2] in (32, 64) and "__init_xw__" in self.w):
            # ONE launch (csrc/initx.hip: MFMA straight out of a [row][column][4 channels] LDS patch) instead of pack + 3 implicit
            # GEMMs + reduce
            offs, woffs = [0, cws[0], cws[0] + cws[1]], u.initx_woffs
            self.op(OP_INITX, 0, p=(self.x_in.ptr, self.base.ptr, self.wptr("__init_xw__"), x.ptr),
                    i=(B, R, R, u.channels, u.dim, cws[0], cws[1], cws[2], offs[0], offs[1], offs[2], woffs[0], woffs[1], woffs[2]))
        else:
            xin_x = self.f32(B * HW, 32, HW)
            self.op(OP_ELTWISE, 2, p=(self.x_in.ptr, self.x_in.ptr, 0, xin_x.ptr), i=(B, HW, 0, u.channels, 32))
            co = 0
            for i, k in enumerate((3, 7, 15)):
                cw = cws[i]
                self.conv(xin_x, True, R, R, f"__init_x__.{i}", None, x, u.dim, co, cw, k, 1, k // 2, resid=self.base)
                co += cw
        self.init_x_ops, self.ops, self.written = self.ops, full_ops, full_written
        hiddens = []
        H = R
        n_lv = len(u.in_out)
        for lv, (di, do) in enumerate(u.in_out):
            x = self.resnet(f"downs.{lv}.1", x, None, di, H)
            for r in range(u.nres[lv]):
                x = self.resnet(f"downs.{lv}.2.{r}", x, None, di, H, gca=True)
                hiddens.append(x)
            if u.attns[lv]:
                x = self.transformer(f"downs.{lv}.3", x)
            hiddens.append(x)
            if lv < n_lv - 1:
                y = self.zf32(B * (H // 2) ** 2, do, (H // 2) ** 2)
                # split-K partials stay in the workspace: the consumer (slot pass / 4x4 GroupNorm prologue) reduces them
                # (into the 4x4 level: k_conv4_gn gathers at most 4 slabs -- more would send its consumer to the general kernel, silently: ADVICE r05)
                self.conv(x, True, H, H, f"downs.{lv}.4.weight", f"downs.{lv}.4.bias", y, do, 0, do, 4, 2, 1, defer=bool(u.fused),
                          defer_max_groups=4 if (H // 2 == 4 and getattr(u, "conv4", True)) else 8)
                H //= 2
            else:
                y = self.zf32(B * H * H, do, H * H)
                # Parallel(conv3x3, conv1x1) of the last level (imagen_pytorch.py:1294-1297: the two outputs are summed) = ONE 3x3
                # conv whose centre tap carries the 1x1 weights too (merged at pack time, Unet._packed): one launch and one
                # 18.9 MB weight stream instead of two launches + a split-K reduction; its partials stay lazy for mid_block1
                if (B == 1 and u.fused and getattr(u, "conv4", True) and getattr(u, "merged_down_conv4", True) and H == 4 and x.C == do == 1024
                        and x.rows == 16 and (x.lazy is None or x.lazy[0] != "splitk")):
                    # r06: ... on k_conv4_gn's un-normalised form (csrc/fused_conv4.h, NORM = false): 4 slices of 256 channels, slabs out
                    # (13.5 -> ~7 us at B = 1; k_conv_igemm keeps it for B >= 2, where k_conv4_gn_mb has no such form)
                    self.need(x)
                    self.fconv(x, None, 4, f"downs.{lv}.4.__merged__.weight", f"downs.{lv}.4.__merged__.bias", y, do, 3, FNORM_NONE, (4, 1, 1, 4),
                               silu=False)
                else:
                    self.conv(x, True, H, H, f"downs.{lv}.4.__merged__.weight", f"downs.{lv}.4.__merged__.bias", y, do, 0, do, 3, 1, 1,
                              defer=bool(u.fused), defer_max_groups=4 if (H == 4 and getattr(u, "conv4", True)) else 8)
            x = y
        mid = x.C
        x = self.resnet("mid_block1", x, None, mid, H, cross=True)
        x = self.self_attention("mid_attn.fn.fn", x, False)
        x = self.resnet("mid_block2", x, None, mid, H, cross=True)
        for ui, lv in enumerate(reversed(range(n_lv))):
            di, do = u.in_out[lv]
            x = self.resnet(f"ups.{ui}.0", x, hiddens.pop(), do, H)
            for r in range(u.nres[lv]):
                x = self.resnet(f"ups.{ui}.1.{r}", x, hiddens.pop(), do, H, gca=True)
            if u.attns[lv]:
                x = self.transformer(f"ups.{ui}.2", x)
            if ui < n_lv - 1:
                y = self.f32(B * 4 * H * H, di, 4 * H * H)
                self.conv(x, True, H, H, f"ups.{ui}.3.net.0.weight", f"ups.{ui}.3.net.0.bias", y, di, 0, di * 4, 1, pixshuf=True,
                          want_slots=bool(u.fused and getattr(u, "producer_slots", True)))
                H *= 2
                x = y
        x = self.resnet("final_res_block", x, None, u.dim, H, gca=True)
        self.out = self.f32(B, u.channels * HW)
        # the split-K reduction of the final conv writes the NCHW output itself when the conv IS split-K; decided BEFORE emitting (r04
        # emitted, looked at the result and popped the op again, which left the first attempt's workspace reservation, `written`
        # entry and a possible reduce op of the workspace's previous owner in the plan)
        _, _, fgroups = u.conv_tiling((B * HW + 15) // 16, (u.channels + 15) // 16, 9 * (x.C // 32), False)
        if getattr(u, "producer_slots", True) and fgroups > 1 and (u.channels + 15) // 16 < 4:      # (n_frags < 4: never the LDS-tiled kernel)
            o = _T(self.out.ptr, B * HW, u.channels, HW)
            self.conv(x, True, R, R, "final_conv.weight", "final_conv.bias", o, u.channels, 0, u.channels, 3, 1, 1, nchw=True)
            assert self.last_conv_nchw, "final conv: the split-K reduction was expected to write NCHW"
        else:                                                        # rows, then unpack
            o = self.zf32(B * HW, u.channels, HW)
            self.conv(x, True, R, R, "final_conv.weight", "final_conv.bias", o, u.channels, 0, u.channels, 3, 1, 1)
            self.op(OP_ELTWISE, 3, p=(o.ptr, 0, 0, self.out.ptr), i=(B, HW, u.channels, u.channels))
        # r04: the sampler's init conv (k_init_x) leaves the statistics slots of x0 itself.  The full plan's k_slots pass over x0 --
        # emitted by the first GroupNorm-fused conv, i.e. the first op behind the init convs -- then belongs to the init part only.
        first = self.ops[self.n_init_ops] if self.n_init_ops < len(self.ops) else None
        if (getattr(u, "producer_slots", True) and first is not None and first.type == OP_SLOTS and first.p[0] == self.x0.ptr and not first.p[1]
                and not first.p[5] and self.x0.slots and len(self.init_x_ops) == 1 and self.init_x_ops[0].type == OP_INITX):
            self.init_x_ops[0].p[4] = self.x0.slots
            self.n_init_ops += 1
        memset_op.i[0] = (self.zero.off + 3) // 4
        if self.zero.off == 0:                          # nothing accumulates with atomics in this plan: no memset launch
            self.ops.remove(memset_op)
        self.op_array = (_lib.SfOp * len(self.ops))(*self.ops)
        n_init = self.n_init_ops - (1 if self.zero.off == 0 else 0)     # the memset op may just have been dropped
        # sampler: [memset] + x-only init conv + everything after the init conv; begin_sampling: [memset] + full init conv
        pre = [o for o in self.ops[self.n_time_ops:n_init] if o.type == OP_MEMSET]
        body = pre + self.init_x_ops + self.ops[n_init:]
        self.body_array, self.n_body_ops = (_lib.SfOp * len(body))(*body), len(body)
        init = self.ops[self.n_time_ops:n_init]
        self.init_array, self.n_init_run = (_lib.SfOp * len(init))(*init), len(init)
        if self.misc.buf is not None:
            self.tb_view = self.tview(self.tb)
            self.x_view, self.t_view = self.tview(self.x_in), self.tview(self.t_in)
            self.cond_view, self.out_view = self.tview(self.cond_in), self.tview(self.out)
            self.x0_view, self.base_view = self.tview(self.x0), self.tview(self.base)
        return self

    def tview(self, t):
        """torch view of a planned fp32 buffer of the misc arena (static input / output staging)."""
        off = t.ptr - self.misc.buf.data_ptr()
        return self.misc.buf[off:off + t.rows * t.C * 4].view(torch.float32).view(t.rows, t.C)


class _TimePlan(_Plan):
    """The time path alone, for T time steps at once: one table row per step (Unet.time_table)."""

    def build(self):
        T = self.B
        self.t_in = self.f32(T, 1)
        self.table = self.f32(T, self.u.tb_stride)
        self.emit_time(T, self.table.ptr, self.t_in.ptr)
        self.op_array = (_lib.SfOp * len(self.ops))(*self.ops)
        if self.misc.buf is not None:
            self.t_view, self.table_view = self.tview(self.t_in), self.tview(self.table)
        return self


class Unet(nn.Module):
    def __init__(self, *, dim, dim_mults=(1, 2, 4, 8), num_resnet_blocks=1, layer_attns=True, layer_cross_attns=True,
                 cond_images_channels=0, channels=3, channels_out=None, attn_pool_text=True, cond_dim=None,
                 attn_dim_head=64, attn_heads=8, ff_mult=2., learned_sinu_pos_emb_dim=16, max_conditional_len=256,
                 cond_on_z=False, lowres_cond=False, image_size=32, **unsupported):
        super().__init__()
        n_lv = len(dim_mults)
        if any(_cast_tuple(layer_cross_attns, n_lv)) or lowres_cond or cond_on_z or cond_images_channels <= 0:
            raise NotImplementedError("sparsefusion_amd.Unet covers the SparseFusion configuration only "
                                      "(utils/load_model.py:58-69: image-conditioned, no text, no lowres, no layer cross-attn)")
        for k, v in unsupported.items():
            if k not in ("image_embed_dim", "conditional_embed_dim", "num_image_tokens", "num_time_tokens", "out_dim"):
                raise NotImplementedError(f"Unet keyword '{k}' is not supported by the HIP plan")
        if channels_out not in (None, channels):
            raise NotImplementedError("channels_out != channels")
        self.dim, self.channels, self.cond_images_channels = dim, channels, cond_images_channels
        self.channels_out = channels
        self.cond_dim = cond_dim or dim
        self.tdim = dim * 4
        self.attn_heads, self.attn_dim_head, self.ff_mult = attn_heads, attn_dim_head, ff_mult
        self.learned_sinu_pos_emb_dim = learned_sinu_pos_emb_dim
        self.image_size = image_size
        self.lowres_cond, self.cond_on_z, self.has_cond_image = False, False, True
        dims = [dim] + [dim * m for m in dim_mults]
        self.in_out = list(zip(dims[:-1], dims[1:]))
        self.nres = _cast_tuple(num_resnet_blocks, n_lv)
        self.attns = _cast_tuple(layer_attns, n_lv)
        self._locals = dict(dim=dim, dim_mults=tuple(dim_mults), num_resnet_blocks=num_resnet_blocks, layer_attns=layer_attns,
                            layer_cross_attns=layer_cross_attns, cond_images_channels=cond_images_channels, channels=channels,
                            attn_pool_text=attn_pool_text, cond_dim=cond_dim, attn_dim_head=attn_dim_head, attn_heads=attn_heads,
                            ff_mult=ff_mult, learned_sinu_pos_emb_dim=learned_sinu_pos_emb_dim,
                            max_conditional_len=max_conditional_len, image_size=image_size)
        if attn_heads * attn_dim_head != 512 or attn_dim_head != 64:
            raise NotImplementedError("attention kernel is built for 8 heads x 64")
        spec = unet_param_spec(dim, tuple(dim_mults), self.nres, self.attns, cond_images_channels, channels, self.cond_dim,
                               attn_heads, attn_dim_head, ff_mult, learned_sinu_pos_emb_dim, max_conditional_len)
        self.spec_shapes = dict(spec)
        g = torch.Generator().manual_seed(0)
        for name, shape in spec:
            _register(self, name, self._default_init(name, shape, g))
        # offsets of each ResnetBlock's (scale, shift) inside the batched time-MLP output
        self.ss_offset, off = {}, 0
        for name, shape in spec:
            if name.endswith(".time_mlp.1.weight"):
                self.ss_offset[name[:-len(".time_mlp.1.weight")]] = off
                off += shape[0]
        self.ss_total = off
        # "time block": everything of one eval that depends on the time ONLY, one row per batch element --
        # [all (scale, shift) pairs | per attention-with-context: k/v of the 2 time tokens | per cross-attention: k/v of the 2 time
        # tokens].  A sampler evaluates it once per trajectory for all its time steps (`time_table`), the eval gathers one row.
        self.tb_off = {}
        for name, shape in spec:
            if name.endswith(".to_context.1.weight"):
                self.tb_off[name[:-len(".to_context.1.weight")]] = off
                off += 2 * shape[0]
            elif name.endswith(".cross_attn.fn.to_kv.weight"):
                self.tb_off[name[:-len(".to_kv.weight")]] = off
                off += 2 * shape[0]
        self.tb_stride = (off + 63) // 64 * 64
        self.conv_waves_target = 1024       # waves wanted per conv launch (4 per CU) before split-K stops
        self.lds_conv_min_blocks = 96       # use k_conv_lds when a layer has at least this many 128 x 128 output tiles
        self.lds_mid_min_rows = 128      # r06: convs of >= this many rows that have too few 128-row tiles for lds_conv_min_blocks run on the LDS-tiled kernels with split-K groups (_Plan.conv); 0 = off
        self.merged_down_conv4 = True     # r06: the merged 3x3 + 1x1 conv of the last Downsample on k_conv4_gn<64, 0, false> at B = 1; False: k_conv_igemm
        self.rc_small_tiles = 32         # r06: bit mask of map sides (32 | 16 | 8) whose B >= 2 (conv1 || res_conv) pairs keep the B = 1 tile of k_conv3s_rc (32: B = 2 eval 1.155 -> 1.139 ms, B = 4 1.562 -> 1.520; 32 | 16: B = 4 1.580, B = 8 2.451 -> 2.550)
        self.gn_one = True               # r06: GroupNorm passes of B >= 32 plans (B * 8 >= 256 workgroups) in one launch (k_gn_one); False: k_gn_stats + k_gn_apply everywhere
        self.lds_mid_min_batch = 8       # ... in plans of at least this many images (the B = 1 .. 4 plans keep their measured kernels)
        self.unfused_min_rows_4 = 256    # r06: 4x4-level ResnetBlocks with B*16 >= this leave the fused kernels (as unfused_min_rows at 32x32 / 16x16; measured: B = 8 loses 0.12 ms, B = 16 gains 0.22, B = 32 0.49); 0 = never
        self.unfused_min_rows_8 = 1024   # ... 8x8 level (k_conv3_halo_sm; measured: B = 8 loses 0.14 ms, B = 16 gains 0.04, B = 32 0.56)
        self.unfused_lin_min_rows = 256      # r06: attention / feed-forward linears of the 4x4 level with >= this many token rows leave the fused kernels (measured: B = 16 3.683 -> 3.670 ms, B = 32 4.664 -> 4.472)
        self.unfused_min_rows_16 = 4096  # r06: the 16x16 level's own threshold (0 = unfused_min_rows; measured with k_gemm_rows_ks / the r06 plans: B = 16 3.641 -> 3.313 ms; 2048 rows lose: B = 8 2.454 -> 2.512)
        self.unfused_min_rows_32 = 0     # ... 32x32 level (4096 rows lose: B = 4 1.575 -> 1.686 ms)
        self.unfused_min_rows = 8192     # ResnetBlocks with B*H*W >= this at the 32x32 / 16x16 levels leave the fused kernels (_Plan.resnet; measured r03: B = 8 eval 3.33 -> 2.96 ms, B = 32 11.5 -> 7.8 ms, B = 4 unchanged)
        self.lazy_consumers = 3             # bit 0: split-K reductions, bit 1: gated residuals are materialised by their first consumer
        # Planner attributes (plain Python attributes since r04 -- the SF_* environment switches of the r01-r03 A/B runs are retired;
        # tools/unet_time.py sets them through SF_UNET_ATTRS, tests/test_gpu_unet.py::test_plan_switches_match_oracle covers them).
        # GroupNorm inside the conv launches (k_conv_fused) wherever the layer fits; False = the first-round plan
        self.fused = True
        self.pair_res_conv = True           # conv1 || res_conv of a ResnetBlock in one launch
        self.res_conv_beside_pool = True    # 4x4 gca blocks: the res_conv shares the GlobalContext pooling launch instead
        self.initx_direct = True            # latent half of the init conv as one direct-convolution launch
        self.big_tile_min_batch = 2         # batch from which the 8x8 / 16x16 / 32x32 maps use 32- / 32- / 64-pixel tiles (r03: B = 2 eval 1.70 -> 1.50 ms, B = 4 2.62 -> 2.02, B = 32 16.4 -> 11.5; at B = 1 they would leave half the CUs idle; 999 = never)
        self.gca_epilogue_pool = True       # GlobalContext pooling in conv2's epilogue (False = k_gca_pool launch)
        self.fconv_pipe = True              # slot-GroupNorm 3x3 convs on k_conv_fused_pipe (staging || matrix work)
        self.producer_slots = True          # r04: k_init_x / the Upsample epilogue leave their consumers' statistics slots, the final conv's split-K reduction writes NCHW (False: the r03 k_slots / k_unpack_out launches; parity tests)
        self.attn_in_out_proj = True        # the 16-token attention core in the prologue of its output projection (False: k_attn16 launch; parity tests)
        self.ln_wave = True                 # r05: LayerNorm of <= 256 rows of 512 | 1024 | 2048 channels on k_layernorm_wave (False: op flag 4 = k_layernorm)
        self.gate_t = True                  # r05: GlobalContext gate with a compile-time hidden width (k_gca_gate_t; False: k_gca_gate)
        self.conv4_reduce_min_batch = 4     # r05: from this batch on a 4x4 split-K conv1 is reduced by its own launch instead of by conv2's gather (0 = never)
        self.conv4_mb = True                # r05: even B: the 4x4 GroupNorm-self convs run 2 | 4 images per workgroup (k_conv4_gn_mb; False = op field i[19] bit 0)
        self.conv4_slices_max_batch = 4     # r05: up to this batch the 4x4 GroupNorm-self convs keep 4 input-channel slices (= k_conv4_gn's geometry); 0 = the workgroup-count rule alone
        self.conv3s = True                  # r06: the recurring single-source slot-GroupNorm 3x3 convs on k_conv3s (compile-time geometry, csrc/fused_conv3s.h); False: op field i[19] bit 1 = k_conv_fused_pipe
        self.conv3s_tw32, self.conv3s_tw16, self.conv3s_tw8 = 8, 4, 0      # r06: tile width of k_conv3s's 2-D pixel tiles per map size (0 = full-width strips)
        self.conv4 = True                   # r05: the 4x4 level's GroupNorm-self 3x3 convs on k_conv4_gn (csrc/fused_conv4.h); False: op flag 128 = k_conv_fused (parity tests)
        self.use_hip_graph = True           # replay one captured hipGraph per eval instead of ~370 host launches
        self._pack_cache = None
        self._plans = {}
        self.tile_override = dict(TILE_PICKS)

    @staticmethod
    def _default_init(name, shape, g):
        """Plain deterministic init (checkpoints overwrite it).  final_conv is zero like the reference (:1388)."""
        if name.startswith("final_conv"):
            return torch.zeros(shape)
        if name.endswith(".g") or name.endswith("groupnorm.weight") or name in ("norm_cond.weight",) or name.endswith("to_context.0.weight"):
            return torch.ones(shape)
        if name.endswith("bias"):
            return torch.zeros(shape)
        if name.endswith("null_kv") or name.startswith("null_conditional") or name.endswith("weights"):
            return torch.randn(shape, generator=g)
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
        return (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(fan_in)

    def conv_tiling(self, m_frags, n_frags, KS, pixshuf=False):
        """(WM, WN, split-K groups) of one implicit-GEMM launch, by a small cost model.  A wave owns a
        (16*WM x 16*WN) tile; the 4 waves of a workgroup split K four ways (reduced in LDS); `groups` further K
        slices go through the workspace + k_splitk_reduce.  Terms: weight streaming from HBM (needs ~4 waves/CU
        to saturate), fragment traffic from L2 (1 KiB per fragment, amortised over the tile), MFMA issue, and the
        partial-tile round trip of split-K."""
        ov = getattr(self, "tile_override", None)               # {(m_frags, n_frags, KS, pixshuf): (WM, WN, groups)}: measured picks
        if ov and (m_frags, n_frags, KS, bool(pixshuf)) in ov:  # (TILE_PICKS below) and tools/tile_sweep.py
            return ov[(m_frags, n_frags, KS, bool(pixshuf))]
        best = None
        mfma = m_frags * n_frags * KS
        w_bytes = n_frags * KS * 1024
        for WM in (1, 2, 4):
            if m_frags % WM or (m_frags <= 4 and WM != m_frags and m_frags in (1, 2, 4)):
                continue                                   # small maps: all rows in one tile -> weights fetched once
            for WN in (1, 2, 4):
                if n_frags % WN:
                    continue
                tiles = (m_frags // WM) * (n_frags // WN)
                for groups in ((1,) if pixshuf else (1, 2, 4, 8, 16, 32)):
                    if groups > 1 and KS // (4 * groups) < 2:
                        continue
                    waves = tiles * groups * 4
                    util = min(1.0, waves / self.conv_waves_target)
                    t_hbm = w_bytes / 4.0e12 / util
                    t_l2 = mfma * 1024 * (1.0 / WN + 1.0 / WM) / 12.0e12 / util
                    t_mfma = mfma * 20 / (1024 * 2.1e9) / util
                    t_part = (2.0 * groups * m_frags * n_frags * 1024 / 3.0e12 + 2.0e-6) if groups > 1 else 0.0
                    cost = max(t_hbm, t_l2, t_mfma) + t_part + 1e-9 * WM * WN
                    if best is None or cost < best[0]:
                        best = (cost, WM, WN, groups)
        return best[1], best[2], best[3]

    # ---- reference API odds and ends
    def cast_model_parameters(self, *, lowres_cond, conditional_embed_dim, channels, channels_out, cond_on_z):
        if lowres_cond or cond_on_z or channels != self.channels or channels_out != self.channels_out:
            raise NotImplementedError("only the first, unconditional-on-text unet of a cascade is supported")
        return self

    def to_config_and_state_dict(self):
        return self._locals, self.state_dict()

    @classmethod
    def from_config_and_state_dict(klass, config, state_dict):
        unet = klass(**config)
        unet.load_state_dict(state_dict)
        return unet

    def load_state_dict(self, *args, **kwargs):
        r = super().load_state_dict(*args, **kwargs)
        self.invalidate()
        return r

    def drop_plans(self):
        """Retire the launch plans but keep the packed weights (plan-level tuning: tools/tile_sweep.py)."""
        for plan in self._plans.values():
            plan.generation = -1
        self._plans = {}

    MAX_TIME_PLANS = 4                   # ("time", T) plans kept per module: one per distinct trajectory length, LRU

    def invalidate(self):
        """Forget packed weights / plans (call after mutating parameters in place).  Contexts of `begin_sampling` that still
        point at a dropped plan are refused by `eval_prepared` (their plan's generation is retired)."""
        for plan in self._plans.values():
            plan.generation = -1
        self._pack_cache, self._plans = None, {}
        self.__dict__["_table_cache"] = {}

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self.invalidate()
        return r

    # ---- weight packing
    def _packed(self, device):
        if self._pack_cache is not None and self._pack_cache[0] == str(device):
            return self._pack_cache[1]
        lib = self.clib
        odt = _lib.operand_dtype(lib)
        packed = {}
        sd = {k: v.detach() for k, v in self.named_parameters()}
        tm_w, tm_b = [], []
        for name, w in sd.items():
            wc = w.float().cpu().contiguous()
            is_conv = name.endswith(".weight") and (
                w.dim() == 4 or any(s in name for s in (".to_q.", ".to_out.0.")) or
                (".to_kv." in name and ".cross_attn." not in name))
            if name.endswith(".time_mlp.1.weight"):
                tm_w.append(wc)
                tm_b.append(sd[name[:-6] + "bias"].float().cpu())
            elif name.endswith(".time_mlp.1.bias"):
                pass
            elif ".gca.to_k.weight" in name:
                packed[name] = wc.reshape(-1).to(device)
            elif ".gca.net." in name and name.endswith(".weight"):
                packed[name] = self._gemv_pack(wc.reshape(wc.shape[0], -1), device)
            elif is_conv:
                w4 = wc if wc.dim() == 4 else wc.reshape(wc.shape[0], wc.shape[1], 1, 1)
                co, ci, kh, kw = w4.shape
                cpad = (ci + 31) // 32 * 32
                n = lib.sf_conv_packed_elems(co, cpad, kh, kw)
                buf = torch.empty(n, dtype=torch.int16)
                _lib.check(lib.sf_conv_pack_weights(w4.data_ptr(), co, ci, cpad, kh, kw, buf.data_ptr()), "pack " + name)
                packed[name] = buf.to(device)
            elif name.endswith(".weight") and w.dim() == 2:
                packed[name] = self._gemv_pack(wc, device)
            else:
                packed[name] = wc.reshape(-1).to(device)      # biases, gains, null_kv, sinusoid weights: fp32
        for name in [k[:-len(".to_kv.weight")] for k in sd if k.endswith(".to_kv.weight") and ".cross_attn." not in k]:
            wq, wkv = sd[name + ".to_q.weight"].float().cpu(), sd[name + ".to_kv.weight"].float().cpu()
            w4 = torch.cat([wq, wkv], 0).contiguous()               # [8 heads * 64 | k 64 | v 64, d]: one fused projection
            co, ci = w4.shape
            if ci % 32 == 0:
                buf = torch.empty(lib.sf_conv_packed_elems(co, ci, 1, 1), dtype=torch.int16)
                _lib.check(lib.sf_conv_pack_weights(w4.data_ptr(), co, ci, ci, 1, 1, buf.data_ptr()), "pack qkv " + name)
                packed[name + ".__qkv__"] = buf.to(device)
        for name in [k[:-len(".fns.0.weight")] for k in sd if k.endswith(".fns.0.weight")]:      # Parallel(3x3, 1x1) -> one 3x3 conv
            w3, w1 = sd[name + ".fns.0.weight"].float().cpu().clone(), sd[name + ".fns.1.weight"].float().cpu()
            w3[:, :, 1, 1] += w1[:, :, 0, 0]
            co, ci, kh, kw = w3.shape
            cpad = (ci + 31) // 32 * 32
            buf = torch.empty(lib.sf_conv_packed_elems(co, cpad, kh, kw), dtype=torch.int16)
            _lib.check(lib.sf_conv_pack_weights(w3.contiguous().data_ptr(), co, ci, cpad, kh, kw, buf.data_ptr()), "pack merged " + name)
            packed[name + ".__merged__.weight"] = buf.to(device)
            packed[name + ".__merged__.bias"] = (sd[name + ".fns.0.bias"].float().cpu() + sd[name + ".fns.1.bias"].float().cpu()).to(device)
        for name in [k[:-len(".gca.to_k.weight")] for k in sd if k.endswith(".gca.to_k.weight")]:
            # w_eff[tap][channel] = sum_n wk[n] * W2[n][channel][tap]: the context logit as a 1-output-channel conv of conv2's input
            w2c = sd.get(name + ".block2.project.weight")
            if w2c is not None and w2c.shape[1] % 32 == 0 and tuple(w2c.shape[2:]) == (3, 3):
                wk = sd[name + ".gca.to_k.weight"].float().cpu().reshape(-1)
                weff = torch.einsum("n,ncyx->yxc", wk, w2c.float().cpu()).reshape(-1).to(odt).contiguous()
                packed[name + ".__weff__"] = weff.to(device)
        for i in range(3):                                      # latent-channel slices of the init conv (sampler path)
            w4 = sd[f"init_conv.convs.{i}.weight"].float().cpu()[:, self.cond_images_channels:].contiguous()
            co, ci, kh, kw = w4.shape
            buf = torch.empty(lib.sf_conv_packed_elems(co, 32, kh, kw), dtype=torch.int16)
            _lib.check(lib.sf_conv_pack_weights(w4.data_ptr(), co, ci, 32, kh, kw, buf.data_ptr()), "pack init x")
            packed[f"__init_x__.{i}"] = buf.to(device)
        # the same slices as fp32 [tap = (ci, ky, kx)][channel] tables for the direct init-x conv (csrc/initx.hip)
        if self.channels <= 4:
            tab, self.initx_woffs = init_x_weight_table([sd[f"init_conv.convs.{i}.weight"].float().cpu()[:, self.cond_images_channels:]
                                                         for i in range(3)], dtype=odt)
            packed["__init_xw__"] = tab.to(device)
        packed["__time_mlps__.weight"] = self._gemv_pack(torch.cat(tm_w, 0), device)
        packed["__time_mlps__.bias"] = torch.cat(tm_b, 0).to(device)
        self._pack_cache = (str(device), packed)
        return packed

    def _gemv_pack(self, w2, device):
        K = w2.shape[1]
        Kp = (K + 7) // 8 * 8
        if Kp != K:
            w2 = torch.nn.functional.pad(w2, (0, Kp - K))
        return w2.to(_lib.operand_dtype(self.clib)).contiguous().to(device)

    # ---- MFMA operand type: bf16 (default) or IEEE half.  `unet.half()` -- what a reference user does for fp16 inference --
    # selects the IEEE-half build of the library for THIS module (libsparsefusion_hip_f16.so, csrc/sf_operand.h): weights are
    # packed as fp16, activations are rounded to fp16 in front of every MFMA, accumulation / residual stream / norms / softmax
    # stay fp32.  `set_operand("f16" | "bf16" | None)` overrides the parameter-dtype rule (None = follow the parameters, or
    # SF_OPERAND for the whole process).
    def set_operand(self, operand):
        if operand not in (None, "bf16", "f16"):
            raise ValueError("operand must be None, 'bf16' or 'f16'")
        self._operand_override = operand
        self.invalidate()
        return self

    @property
    def operand(self):
        ov = getattr(self, "_operand_override", None)
        if ov is not None:
            return ov
        p = next(self.parameters(), None)
        return "f16" if (p is not None and p.dtype == torch.float16) else None

    @property
    def clib(self):
        return _lib.lib(self.operand)

    def _plan(self, B, device):
        key = (B, str(device))
        if key not in self._plans:
            sizing = _Plan(self, B, device).build()
            plan = _Plan(self, B, device, (sizing.zero.off, sizing.misc.off + sizing.ws_bytes + sizing.ws2_bytes + 512,
                                           sizing.ws_bytes, sizing.ws2_bytes)).build()
            plan.generation = 0
            self._plans[key] = plan
        return self._plans[key]

    # ---- forward
    @torch.no_grad()
    def forward(self, x, time, *, cond_images=None, cond_drop_prob=0., **unsupported):
        if cond_images is None:
            raise AssertionError("you requested to condition on an image on the unet, but the conditioning image is not supplied")
        for k, v in unsupported.items():
            if v is not None:
                raise NotImplementedError(f"Unet.forward argument '{k}' is not supported")
        _lib.require_cuda(x, time, cond_images)
        B = x.shape[0]
        assert x.shape[1:] == (self.channels, self.image_size, self.image_size), x.shape
        assert cond_images.shape[1] == self.cond_images_channels, \
            'the number of channels on the conditioning image you are passing in does not match'
        if cond_images.shape[-1] != self.image_size:        # resize_image_to: nearest (imagen_pytorch.py:150-165)
            cond_images = torch.nn.functional.interpolate(cond_images, self.image_size, mode='nearest')
        if cond_drop_prob != 0.:                            # conditioning dropout: the image condition is zeroed per sample
            if cond_drop_prob >= 1.:                        # (prob_mask_like + `cond_images * keep_mask`, :1499-1503)
                keep = torch.zeros(B, dtype=torch.bool, device=x.device)
            else:
                keep = torch.zeros(B, device=x.device).float().uniform_(0, 1) < (1 - cond_drop_prob)
            cond_images = cond_images * keep.view(B, 1, 1, 1)
        plan = self._plan(B, x.device)
        plan.generation += 1                                # the plan's arena is about to be overwritten: any sampler context
        plan.x_view.copy_(x.reshape(B, -1))                 # of this batch size is dead from here on (eval_prepared checks)
        plan.t_view.copy_(time.reshape(-1).expand(B).reshape(B, 1))
        plan.cond_view.copy_(cond_images.reshape(B, -1))
        if self.use_hip_graph and not torch.cuda.is_current_stream_capturing():
            if plan.graph is None:                     # first call: eager warm-up, then capture the ~370 launches once
                self._run_plan(plan)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):   # other threads (RCCL watchdog) may touch HIP meanwhile
                    self._run_plan(plan)
                plan.graph = g
            plan.graph.replay()
        else:
            self._run_plan(plan)
        y = plan.out_view.clone().view(B, self.channels, self.image_size, self.image_size)
        return y if x.dtype == torch.float32 else y.to(x.dtype)          # a .half() model answers in half, as the reference's would

    @staticmethod
    def _run_plan(plan, body_only=False):
        if body_only:
            _lib.check(plan.u.clib.sf_plan_run(plan.body_array, plan.n_body_ops, _lib.stream_ptr()), "unet plan", plan.u.clib)
        else:
            _lib.check(plan.u.clib.sf_plan_run(plan.op_array, len(plan.ops), _lib.stream_ptr()), "unet plan", plan.u.clib)

    # ---- sampler fast path: the time path is evaluated once per trajectory, the eval replays the body only
    def _time_plan(self, T, device):
        key = ("time", T, str(device))
        if key not in self._plans:
            old = [k for k in self._plans if k[0] == "time"]
            for k in old[:max(0, len(old) + 1 - self.MAX_TIME_PLANS)]:          # dicts keep insertion order: oldest first
                del self._plans[k]
            sizing = _TimePlan(self, T, device).build()
            self._plans[key] = _TimePlan(self, T, device, (sizing.zero.off, sizing.misc.off + 512, 0, 0)).build()
        else:
            self._plans[key] = self._plans.pop(key)                             # most recently used last
        return self._plans[key]

    @torch.no_grad()
    def time_table(self, log_snrs, key=None):
        """[T, tb_stride] time-block rows for T log-snr values: time MLPs, every ResnetBlock's (scale, shift), and the k/v of the
        time tokens of every attention that sees them (external/imagen_pytorch.py:1514-1604) -- everything that depends on
        the time alone.  One call per sampler trajectory replaces ~14 launches and a 72 MB weight read in each eval.
        `key` (hashable, e.g. the tuple of the trajectory's times): the table is a function of the weights and the times only, so a
        caller that runs the SAME schedule again gets the same tensor back -- read-only by contract -- until the parameters change
        (invalidate(): load_state_dict / .to() / .half()).  Who profits: samplers driven with a fixed `max_thres` (evaluation
        sweeps, the r01-r04 bench lines: -0.35 ms per call).  The reference's distillation loop does NOT: it draws `max_thres`
        anew on every step (sparsefusion/distillation.py:303) and the times are linspace(max_thres, 0, n + 1), so its key differs
        on essentially every step and every table is built (r05: bench.py draws per step for that reason).  The cache therefore
        holds the TWO most recent schedules only (T x tb_stride floats each), not eight dead tables."""
        _lib.require_cuda(log_snrs)
        ck = None
        if key is not None:
            ck = (key, str(log_snrs.device), self.operand)
            hit = self.__dict__.get("_table_cache", {}).get(ck)
            if hit is not None:
                cache = self.__dict__["_table_cache"]
                cache[ck] = cache.pop(ck)                                   # most recently used last
                return hit
        T = log_snrs.numel()
        plan = self._time_plan(T, log_snrs.device)
        plan.t_view.copy_(log_snrs.reshape(T, 1))
        _lib.check(self.clib.sf_plan_run(plan.op_array, len(plan.ops), _lib.stream_ptr()), "unet time plan", self.clib)
        table = plan.table_view.clone()
        if ck is not None:
            cache = self.__dict__.setdefault("_table_cache", {})
            while len(cache) >= 2:
                cache.pop(next(iter(cache)))
            cache[ck] = table
        return table

    @torch.no_grad()
    def begin_sampling(self, cond_images, log_snrs, table_key=None):
        """Prepare a trajectory: time table for `log_snrs` [T] + the (fixed) conditioning image.  Returns the context for
        `eval_prepared`.  Same numerics as `forward` (same kernels, same operands), fewer launches per eval."""
        _lib.require_cuda(cond_images)
        B = cond_images.shape[0]
        if cond_images.shape[-1] != self.image_size:
            cond_images = torch.nn.functional.interpolate(cond_images, self.image_size, mode='nearest')
        plan = self._plan(B, cond_images.device)
        plan.generation += 1                                   # ONE trajectory per (batch size, device) at a time: the latents,
        plan.cond_view.copy_(cond_images.reshape(B, -1))       # `base` and the time rows of a trajectory live in the plan's arena
        plan.x_view.zero_()                                    # init conv of the conditioning image alone (+ bias) -> base
        _lib.check(self.clib.sf_plan_run(plan.init_array, plan.n_init_run, _lib.stream_ptr()), "unet init conv", self.clib)
        plan.base_view.copy_(plan.x0_view)
        return {"plan": plan, "table": self.time_table(log_snrs, key=table_key), "B": B, "generation": plan.generation}

    @torch.no_grad()
    def eval_prepared(self, ctx, x, row, row_ready=False):
        """eps = unet(x, time = log_snrs[row]) for a context of `begin_sampling`.  `x` may be ctx['plan'].x_view itself (a
        sampler that writes its latents there saves the copy); `row_ready`: ctx['table'][row] already sits in ctx['plan'].tb_view.  Returns a VIEW of the plan's output buffer: consume or clone
        it before the next eval."""
        plan, B = ctx["plan"], ctx["B"]
        if ctx.get("generation") != plan.generation:
            raise RuntimeError("Unet.eval_prepared: this sampling context is stale -- a later forward() / begin_sampling() with the same "
                               "batch size, or invalidate() / .to(), took over the plan that held its latents and time rows; "
                               "call begin_sampling() again")
        if x.data_ptr() != plan.x_view.data_ptr():
            plan.x_view.copy_(x.reshape(B, -1))
        if not row_ready:                                       # (a sampler whose step kernel already moved the row there skips the copy launch)
            plan.tb_view.copy_(ctx["table"][row].expand(B, -1))
        if self.use_hip_graph and not torch.cuda.is_current_stream_capturing():
            if getattr(plan, "body_graph", None) is None:
                self._run_plan(plan, True)
                torch.cuda.synchronize()
                try:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, capture_error_mode="thread_local"):
                        self._run_plan(plan, True)
                    plan.body_graph = g
                except Exception as e:                      # e.g. another library's thread touched HIP mid-capture
                    import warnings
                    warnings.warn(f"sparsefusion_amd.Unet: hipGraph capture failed ({e}); replaying the plan launch by launch")
                    self.use_hip_graph = False
                    torch.cuda.synchronize()
                    self._run_plan(plan, True)
                    return plan.out_view.view(B, self.channels, self.image_size, self.image_size)
            plan.body_graph.replay()
        else:
            self._run_plan(plan, True)
        return plan.out_view.view(B, self.channels, self.image_size, self.image_size)

    def forward_with_cond_scale(self, *args, cond_scale=1., **kwargs):
        """imagen_pytorch.py:1456-1468: classifier-free guidance = a second eval with the condition dropped."""
        logits = self.forward(*args, **kwargs)
        if cond_scale == 1:
            return logits
        null_logits = self.forward(*args, cond_drop_prob=1., **kwargs)
        return null_logits + (logits - null_logits) * cond_scale

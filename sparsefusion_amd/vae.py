"""SD-VAE encode / decode on the HIP op plan -- the two no-grad calls inside every distillation step.

Mirrors the call surface of the reference's AutoencoderKL (external/ldm/models/autoencoder.py:285-333) that
sparsefusion/distillation.py uses: `vae.encode(x).mode()` (:299) and `vae.decode(z)` (:309), and its state-dict
keys (`encoder.*`, `decoder.*`, `quant_conv.*`, `post_quant_conv.*`; config external/ldm/configs/sd-vae.yaml:6-20).
The network (external/ldm/modules/diffusionmodules/model.py:368-569) is compiled once per (direction, batch)
into a static launch plan of the same ops the UNet uses (csrc/unet_ops.hip): implicit-GEMM convs on bf16 MFMA
with fp32 accumulation, GroupNorm(32, eps 1e-6)+swish producing the bf16 conv operand, the stride-2
right/bottom-padded Downsample and the nearest-x2 Upsample folded into the conv's address generation, and the
1024-token single-head AttnBlock as two more implicit GEMMs whose B operands are packed on the device.

Forward only (the reference calls both under torch.no_grad()).  No CPU fallback: the plan needs the HIP library."""
import math

import numpy as np
import os

import torch
import torch.nn as nn

from . import _lib
from .unet import (OP_ELTWISE, OP_MEMSET, Unet, _Plan, _register, _T)

SD_VAE_DDCONFIG = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128, ch_mult=(1, 2, 4, 4),
                       num_res_blocks=2, attn_resolutions=(), dropout=0.0)          # sd-vae.yaml:6-20


def _resnet_spec(out, p, cin, cout):
    out += [(f"{p}.norm1.weight", (cin,)), (f"{p}.norm1.bias", (cin,)), (f"{p}.conv1.weight", (cout, cin, 3, 3)),
            (f"{p}.conv1.bias", (cout,)), (f"{p}.norm2.weight", (cout,)), (f"{p}.norm2.bias", (cout,)),
            (f"{p}.conv2.weight", (cout, cout, 3, 3)), (f"{p}.conv2.bias", (cout,))]
    if cin != cout:
        out += [(f"{p}.nin_shortcut.weight", (cout, cin, 1, 1)), (f"{p}.nin_shortcut.bias", (cout,))]


def _attn_spec(out, p, c):
    out += [(f"{p}.norm.weight", (c,)), (f"{p}.norm.bias", (c,))]
    for n in ("q", "k", "v", "proj_out"):
        out += [(f"{p}.{n}.weight", (c, c, 1, 1)), (f"{p}.{n}.bias", (c,))]


def vae_param_spec(*, ch, ch_mult, num_res_blocks, in_channels, out_ch, z_channels, embed_dim, double_z=True, **_):
    """(name, shape) of every parameter, in the registration order of the reference modules
    (Encoder model.py:383-432, Decoder :479-531 -- `up` levels are inserted at the front --, autoencoder.py:302-303)."""
    mult = tuple(ch_mult)
    out = [("encoder.conv_in.weight", (ch, in_channels, 3, 3)), ("encoder.conv_in.bias", (ch,))]
    in_mult = (1,) + mult
    block_in = ch
    for lv in range(len(mult)):
        block_in, block_out = ch * in_mult[lv], ch * mult[lv]
        for b in range(num_res_blocks):
            _resnet_spec(out, f"encoder.down.{lv}.block.{b}", block_in, block_out)
            block_in = block_out
        if lv != len(mult) - 1:
            out += [(f"encoder.down.{lv}.downsample.conv.weight", (block_in, block_in, 3, 3)),
                    (f"encoder.down.{lv}.downsample.conv.bias", (block_in,))]
    _resnet_spec(out, "encoder.mid.block_1", block_in, block_in)
    _attn_spec(out, "encoder.mid.attn_1", block_in)
    _resnet_spec(out, "encoder.mid.block_2", block_in, block_in)
    zo = 2 * z_channels if double_z else z_channels
    out += [("encoder.norm_out.weight", (block_in,)), ("encoder.norm_out.bias", (block_in,)),
            ("encoder.conv_out.weight", (zo, block_in, 3, 3)), ("encoder.conv_out.bias", (zo,))]
    block_in = ch * mult[-1]
    out += [("decoder.conv_in.weight", (block_in, z_channels, 3, 3)), ("decoder.conv_in.bias", (block_in,))]
    _resnet_spec(out, "decoder.mid.block_1", block_in, block_in)
    _attn_spec(out, "decoder.mid.attn_1", block_in)
    _resnet_spec(out, "decoder.mid.block_2", block_in, block_in)
    levels = {}
    for lv in reversed(range(len(mult))):
        cur = []
        for b in range(num_res_blocks + 1):
            _resnet_spec(cur, f"decoder.up.{lv}.block.{b}", block_in, ch * mult[lv])
            block_in = ch * mult[lv]
        if lv != 0:
            cur += [(f"decoder.up.{lv}.upsample.conv.weight", (block_in, block_in, 3, 3)),
                    (f"decoder.up.{lv}.upsample.conv.bias", (block_in,))]
        levels[lv] = cur
    for lv in range(len(mult)):
        out += levels[lv]
    out += [("decoder.norm_out.weight", (block_in,)), ("decoder.norm_out.bias", (block_in,)),
            ("decoder.conv_out.weight", (out_ch, block_in, 3, 3)), ("decoder.conv_out.bias", (out_ch,)),
            ("quant_conv.weight", (2 * embed_dim, 2 * z_channels, 1, 1)), ("quant_conv.bias", (2 * embed_dim,)),
            ("post_quant_conv.weight", (z_channels, embed_dim, 1, 1)), ("post_quant_conv.bias", (z_channels,))]
    return out


class _VaePlan(_Plan):
    """Static launch plan of one direction ('enc' or 'dec') at batch B."""

    def __init__(self, vae, kind, B, device, sizing=None):
        super().__init__(vae, B, device, sizing)
        self.kind = kind

    def gn(self, x, name, out, silu=True):
        self.gn_act(x, None, name, 0, out, None, silu=silu, groups=32, eps=1e-6)

    def conv3(self, x, x_f32, H, name, out, cout, resid=None, twin=None):
        self.conv(x, x_f32, H, H, name + ".weight", name + ".bias", out, cout, 0, cout, 3, 1, 1, resid=resid, twin=twin)

    def resnet_block(self, p, x, cout, H, twin=False):
        """model.py:119-141 (temb None, dropout 0).  twin: conv2 also leaves its output in operand type (out.twin) for a conv that
        reads the block output directly (Upsample)."""
        rows, HW, cin = x.rows, H * H, x.C
        a1 = self.bf16(rows, cin, HW)
        self.gn(x, p + ".norm1", a1)
        h = self.zf32(rows, cout, HW)
        self.conv3(a1, False, H, p + ".conv1", h, cout)
        a2 = self.bf16(rows, cout, HW)
        self.gn(h, p + ".norm2", a2)
        out = self.zf32(rows, cout, HW)
        tw = self.bf16(rows, cout, HW) if twin else None
        if cin != cout:                                   # out = nin_shortcut(x), then conv2 accumulates into it
            xs = x.twin if self.u.conv_twin else None    # (the producing conv left an operand-type copy: half the bytes, LDS-DMA kernel)
            self.conv(xs if xs is not None else x, xs is None, H, H, p + ".nin_shortcut.weight", p + ".nin_shortcut.bias", out, cout, 0, cout, 1)
            self.conv3(a2, False, H, p + ".conv2", out, cout, twin=tw)
        else:
            self.conv3(a2, False, H, p + ".conv2", out, cout, resid=x, twin=tw)
        return out

    def attn_block(self, p, x, H):
        """model.py:178-203: tokens are the H*W pixels (= rows of the NHWC map), one head of width C."""
        B, HW, C = self.B, H * H, x.C
        rows = x.rows
        hn = self.bf16(rows, C, HW)
        self.gn(x, p + ".norm", hn, silu=False)
        q, k, v = (self.zf32(rows, C, HW) for _ in range(3))
        for t, n in ((q, "q"), (k, "k"), (v, "v")):
            self.conv(hn, False, H, H, f"{p}.{n}.weight", f"{p}.{n}.bias", t, C, 0, C, 1)
        att = self.zf32(rows, C, HW)
        lib = _lib.lib()
        kp_elems = lib.sf_conv_packed_elems(HW, C, 1, 1)          # B operand of q.k^T: N = HW keys, K = C
        vp_elems = lib.sf_conv_packed_elems(C, (HW + 31) // 32 * 32, 1, 1)   # B operand of P.v: N = C, K = HW
        hwp = (HW + 31) // 32 * 32
        for b in range(B):
            kp = self.misc.alloc(kp_elems * 2)
            vp = self.misc.alloc(vp_elems * 2)
            s = self.f32(HW, HW)
            pr = _T(self.misc.alloc(HW * hwp * 2), HW, hwp)
            qb, kb, vb = (_T(t.ptr + b * HW * C * 4, HW, C, HW) for t in (q, k, v))
            ob = _T(att.ptr + b * HW * C * 4, HW, C, HW)
            self.op(OP_ELTWISE, 5, p=(kb.ptr, 0, 0, kp), i=(HW, C, C, 0))
            self.op(OP_ELTWISE, 5, p=(vb.ptr, 0, 0, vp), i=(C, HW, C, 1))
            self.conv(qb, True, 1, HW, None, None, s, HW, 0, HW, 1, w_ptr=kp, batch=1)
            if hwp != HW:
                raise NotImplementedError("AttnBlock token count must be a multiple of 32")
            self.op(OP_ELTWISE, 6, p=(s.ptr, 0, 0, pr.ptr), i=(HW, HW), f=(float(int(C) ** (-0.5)),))
            self.conv(pr, False, 1, HW, None, None, ob, C, 0, C, 1, w_ptr=vp, batch=1)
        out = self.zf32(rows, C, HW)
        self.conv(att, True, H, H, p + ".proj_out.weight", p + ".proj_out.bias", out, C, 0, C, 1, resid=x)
        return out

    def build(self):
        v, B = self.u, self.B
        mult, nres, ch = v.ch_mult, v.num_res_blocks, v.ch
        n_lv = len(mult)
        self.op(OP_MEMSET, 0, p=(self.zero.buf.data_ptr() if self.zero.buf is not None else 1,), i=(0,))
        memset_op = self.ops[-1]
        if self.kind == "enc":
            R = v.resolution
            HW = R * R
            self.x_in = self.f32(B, v.in_channels * HW)
            xin = self.f32(B * HW, 32, HW)
            self.op(OP_ELTWISE, 2, p=(0, self.x_in.ptr, 0, xin.ptr), i=(B, HW, 0, v.in_channels, 32))
            h = self.zf32(B * HW, ch, HW)
            self.conv3(xin, True, R, "encoder.conv_in", h, ch)
            H = R
            for lv in range(n_lv):
                for b in range(nres):
                    h = self.resnet_block(f"encoder.down.{lv}.block.{b}", h, ch * mult[lv], H,
                                          twin=(b == nres - 1 and lv != n_lv - 1 and v.conv_twin))
                if lv != n_lv - 1:                        # Downsample: zero pad right/bottom, conv3x3 stride 2 (model.py:72-76)
                    y = self.zf32(B * (H // 2) ** 2, h.C, (H // 2) ** 2)
                    src = h.twin                          # operand-type copy from the block's conv2 epilogue, when it runs an LDS-tiled kernel
                    ytw = self.bf16(y.rows, h.C, y.HW) if (v.conv_twin and ch * mult[lv + 1] != h.C) else None      # ... and one for the next nin_shortcut
                    self.conv(src if src is not None else h, src is None, H, H, f"encoder.down.{lv}.downsample.conv.weight",
                              f"encoder.down.{lv}.downsample.conv.bias", y, h.C, 0, h.C, 3, 2, 0, out_hw=(H // 2, H // 2), twin=ytw)
                    h, H = y, H // 2
            h = self.resnet_block("encoder.mid.block_1", h, h.C, H)
            h = self.attn_block("encoder.mid.attn_1", h, H)
            h = self.resnet_block("encoder.mid.block_2", h, h.C, H)
            a = self.bf16(h.rows, h.C, H * H)
            self.gn(h, "encoder.norm_out", a)
            zo = 2 * v.z_channels
            # 8 moment channels in a 32-wide row (quant_conv reads it as its A operand): the pad lanes must be
            # finite, so the row lives in the arena that is zeroed at the top of every run
            mo = _T(self.zero.alloc(h.rows * 32 * 4), h.rows, 32, H * H)
            self.conv(a, False, H, H, "encoder.conv_out.weight", "encoder.conv_out.bias", mo, 32, 0, zo, 3, 1, 1)
            qm = self.zf32(h.rows, 2 * v.embed_dim, H * H)
            self.conv(mo, True, H, H, "quant_conv.weight", "quant_conv.bias", qm, 2 * v.embed_dim, 0, 2 * v.embed_dim, 1)
            self.out = self.f32(B, 2 * v.embed_dim * H * H)
            self.op(OP_ELTWISE, 3, p=(qm.ptr, 0, 0, self.out.ptr), i=(B, H * H, 2 * v.embed_dim, 2 * v.embed_dim))
            self.out_shape = (2 * v.embed_dim, H, H)
        else:
            H = v.resolution // 2 ** (n_lv - 1)
            HW = H * H
            self.x_in = self.f32(B, v.embed_dim * HW)
            zin = self.f32(B * HW, 32, HW)
            self.op(OP_ELTWISE, 2, p=(0, self.x_in.ptr, 0, zin.ptr), i=(B, HW, 0, v.embed_dim, 32))
            z = _T(self.zero.alloc(B * HW * 32 * 4), B * HW, 32, HW)     # post_quant_conv output, again a zero-padded 32-wide row
            self.conv(zin, True, H, H, "post_quant_conv.weight", "post_quant_conv.bias", z, 32, 0, v.z_channels, 1)
            block_in = ch * mult[-1]
            h = self.zf32(B * HW, block_in, HW)
            self.conv3(z, True, H, "decoder.conv_in", h, block_in)
            h = self.resnet_block("decoder.mid.block_1", h, block_in, H)
            h = self.attn_block("decoder.mid.attn_1", h, H)
            h = self.resnet_block("decoder.mid.block_2", h, block_in, H)
            for lv in reversed(range(n_lv)):
                for b in range(nres + 1):
                    h = self.resnet_block(f"decoder.up.{lv}.block.{b}", h, ch * mult[lv], H, twin=(b == nres and lv != 0 and v.conv_twin))
                if lv != 0:                               # Upsample: nearest x2 folded into the conv's input addressing (:53-57)
                    y = self.zf32(B * 4 * H * H, h.C, 4 * H * H)
                    src = h.twin                          # the block's conv2 left an operand-type copy: the 3x3 runs on k_conv3_halo
                    ytw = self.bf16(y.rows, h.C, y.HW) if (v.conv_twin and ch * mult[lv - 1] != h.C) else None      # for the next nin_shortcut
                    self.conv(src if src is not None else h, src is None, 2 * H, 2 * H, f"decoder.up.{lv}.upsample.conv.weight",
                              f"decoder.up.{lv}.upsample.conv.bias", y, h.C, 0, h.C, 3, 1, 1, upsampled=True, twin=ytw)
                    h, H = y, 2 * H
            a = self.bf16(h.rows, h.C, H * H)
            self.gn(h, "decoder.norm_out", a)
            o = self.zf32(h.rows, v.out_ch, H * H)
            self.conv3(a, False, H, "decoder.conv_out", o, v.out_ch)
            self.out = self.f32(B, v.out_ch * H * H)
            self.op(OP_ELTWISE, 3, p=(o.ptr, 0, 0, self.out.ptr), i=(B, H * H, v.out_ch, v.out_ch))
            self.out_shape = (v.out_ch, H, H)
        memset_op.i[0] = (self.zero.off + 3) // 4
        self.op_array = (_lib.SfOp * len(self.ops))(*self.ops)
        if self.misc.buf is not None:
            self.x_view, self.out_view = self.tview(self.x_in), self.tview(self.out)
        return self


class DiagonalGaussianDistribution:
    """Posterior returned by encode(): external/ldm/modules/distributions/distributions.py:24-64."""

    def __init__(self, parameters, deterministic=False):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.deterministic = deterministic
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)
        if self.deterministic:
            self.var = self.std = torch.zeros_like(self.mean)

    def sample(self):
        return self.mean + self.std * torch.randn(self.mean.shape, device=self.parameters.device)

    def mode(self):
        return self.mean

    def kl(self, other=None):
        if self.deterministic:
            return torch.Tensor([0.])
        if other is None:
            return 0.5 * torch.sum(torch.pow(self.mean, 2) + self.var - 1.0 - self.logvar, dim=[1, 2, 3])
        return 0.5 * torch.sum(torch.pow(self.mean - other.mean, 2) / other.var + self.var / other.var - 1.0 - self.logvar
                               + other.logvar, dim=[1, 2, 3])


class AutoencoderKL(nn.Module):
    """Drop-in for the inference surface of external/ldm/models/autoencoder.py:285-343."""

    def __init__(self, ddconfig=None, lossconfig=None, embed_dim=4, ckpt_path=None, ignore_keys=(), **unsupported):
        super().__init__()
        dd = dict(SD_VAE_DDCONFIG if ddconfig is None else ddconfig)
        if not dd.get("double_z", True):
            raise NotImplementedError("AutoencoderKL asserts double_z (autoencoder.py:300)")
        if list(dd.get("attn_resolutions", ())):
            raise NotImplementedError("attn_resolutions is empty in sd-vae.yaml; per-level attention is not planned")
        if dd.get("dropout", 0.0) != 0.0:
            raise NotImplementedError("dropout is a training feature")
        self.ch, self.ch_mult, self.num_res_blocks = dd["ch"], tuple(dd["ch_mult"]), dd["num_res_blocks"]
        self.in_channels, self.out_ch, self.z_channels = dd["in_channels"], dd["out_ch"], dd["z_channels"]
        self.resolution, self.embed_dim = dd["resolution"], embed_dim
        if self.ch % 128 or self.in_channels > 32 or 2 * self.z_channels > 32:
            raise NotImplementedError("HIP plan: ch must be a multiple of 128 (GroupNorm(32) with >= 4 channels per group)")
        g = torch.Generator().manual_seed(0)
        self.spec_shapes = {}
        for name, shape in vae_param_spec(embed_dim=embed_dim, **dd):
            self.spec_shapes[name] = tuple(shape)
            _register(self, name, nn.Parameter(self._default_init(name, shape, g), requires_grad=False))
        self.conv_waves_target = 1024
        self.lazy_consumers = 0                # VAE convs are large-M: no split-K partials worth deferring
        self.ss_total = 0
        self.lds_conv_min_blocks = 96
        # EXPERIMENTAL, not yet measured: GroupNorm statistics from the producing conv's epilogue (csrc/conv_lds.h) instead of a pass
        # over the tensor; parity-checked on CPU threads (tests/test_hostemu_conv_lds.py)
        self.conv_twin = True             # Upsample convs read an operand-type twin of the block output (r03) instead of the fp32 tensor
        self.gn_epilogue = True           # GroupNorm statistics out of the producing conv's epilogue (r03: encode 1.80 -> 1.72 ms, decode 2.94 -> 2.76 ms); False = the statistics pass
        self._pack_cache, self._plans = None, {}
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path, ignore_keys=ignore_keys)

    conv_tiling = Unet.conv_tiling

    @staticmethod
    def _default_init(name, shape, g):
        if ".norm" in name:                    # GroupNorm affine: ones / zeros
            return torch.ones(shape) if name.endswith("weight") else torch.zeros(shape)
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
        bound = 1.0 / math.sqrt(fan_in)        # torch Conv2d default (kaiming_uniform a=sqrt(5)) has this range
        if name.endswith("bias"):
            return torch.zeros(shape)
        return (torch.rand(shape, generator=g) * 2 - 1) * bound

    def init_from_ckpt(self, path, ignore_keys=()):
        """autoencoder.py:312-322."""
        sd = torch.load(path, map_location="cpu")["state_dict"]
        for k in list(sd.keys()):
            if any(k.startswith(ik) for ik in ignore_keys):
                del sd[k]
        self.load_state_dict(sd, strict=False)

    def load_state_dict(self, *args, **kwargs):
        r = super().load_state_dict(*args, **kwargs)
        self.invalidate()
        return r

    def invalidate(self):
        self._pack_cache, self._plans = None, {}

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self.invalidate()
        return r

    def _packed(self, device):
        if self._pack_cache is not None and self._pack_cache[0] == str(device):
            return self._pack_cache[1]
        lib = _lib.lib()
        packed = {}
        for name, w in self.named_parameters():
            wc = w.detach().float().cpu().contiguous()
            if wc.dim() == 4:
                co, ci, kh, kw = wc.shape
                cpad = (ci + 31) // 32 * 32
                buf = torch.empty(lib.sf_conv_packed_elems(co, cpad, kh, kw), dtype=torch.int16)
                _lib.check(lib.sf_conv_pack_weights(wc.data_ptr(), co, ci, cpad, kh, kw, buf.data_ptr()), "pack " + name)
                packed[name] = buf.to(device)
            else:
                packed[name] = wc.reshape(-1).to(device)
        self._pack_cache = (str(device), packed)
        return packed

    def _plan(self, kind, B, device):
        key = (kind, B, str(device))
        if key not in self._plans:
            sizing = _VaePlan(self, kind, B, device).build()
            self._plans[key] = _VaePlan(self, kind, B, device,
                                        (sizing.zero.off, sizing.misc.off + sizing.ws_bytes + 256, sizing.ws_bytes)).build()
        return self._plans[key]

    def _run(self, kind, x):
        _lib.require_cuda(x)
        B = x.shape[0]
        plan = self._plan(kind, B, x.device)
        plan.x_view.copy_(x.reshape(B, -1))
        _lib.check(_lib.lib().sf_plan_run(plan.op_array, len(plan.ops), _lib.stream_ptr()), f"vae {kind} plan")
        return plan.out_view.clone().view(B, *plan.out_shape)

    @torch.no_grad()
    def encode(self, x):
        """autoencoder.py:324-328.  x [B, in_channels, R, R] fp32 -> posterior over [B, embed_dim, R/f, R/f]."""
        assert x.shape[1:] == (self.in_channels, self.resolution, self.resolution), x.shape
        return DiagonalGaussianDistribution(self._run("enc", x.float()))

    @torch.no_grad()
    def decode(self, z):
        """autoencoder.py:330-333."""
        f = 2 ** (len(self.ch_mult) - 1)
        assert z.shape[1:] == (self.embed_dim, self.resolution // f, self.resolution // f), z.shape
        return self._run("dec", z.float())

    def forward(self, input, sample_posterior=True):
        """autoencoder.py:335-342."""
        posterior = self.encode(input)
        z = posterior.sample() if sample_posterior else posterior.mode()
        return self.decode(z), posterior

"""Kernel-logic tests of k_layernorm and k_attn16 (sparsefusion_amd/csrc/attn_ln.h) on CPU threads against torch: LayerNorm with
optional GELU in front, bias, residual, bf16 / fp32 output (external/imagen_pytorch.py:92-107, 944-961); the 16-query attention
core over up to 3 key / value segments -- time tokens, null key, the 16 pixels (Attention.forward, :480-566)."""
import ctypes as C
import os
import subprocess

import pytest
import torch
import torch.nn.functional as F

from hostemu import fused

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostemu")
SO = os.path.join(HERE, "_build", "libattn_ln_emu.so")
pytestmark = pytest.mark.skipif(not fused.available(), reason="host clang not found")
ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _lib():
    srcs = [os.path.join(HERE, "attn_ln_emu.cpp"), os.path.join(HERE, "hip_emu.h")] + \
           [os.path.join(HERE, "..", "..", "sparsefusion_amd", "csrc", f) for f in ("attn_ln.h", "sf_dev.h")]
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(s) for s in srcs):
        os.makedirs(os.path.dirname(SO), exist_ok=True)
        subprocess.check_call([fused.CLANG, "-std=c++17", "-O2", "-fPIC", "-shared", "-I" + HERE, "-Wall", "-Wno-unused-function",
                               "-ffp-contract=off", srcs[0], "-o", SO, "-lpthread"])
    return C.CDLL(SO)


@pytest.mark.parametrize("R,Cc,gelu,bias,resid,out_f32", [(16, 1024, False, True, True, True), (5, 192, True, False, False, False),
                                                         (3, 2048, False, True, False, False),      # (512 | 1024 | 2048 channels, <= 256 rows: k_layernorm_wave, r05)
                                                         (6, 512, True, False, True, True), (16, 1024, 4, True, True, True),      # gelu = 4: flag 4, k_layernorm on the same shape
                                                         (10, 256, False, True, True, True), (7, 256, True, False, False, False)])      # k_layernorm_w256
def test_layernorm_kernel(R, Cc, gelu, bias, resid, out_f32):
    lib = _lib()
    g = torch.Generator().manual_seed(R + Cc)
    x, gain = torch.randn(R, Cc, generator=g) * 2 + 0.5, torch.randn(Cc, generator=g)
    b = torch.randn(Cc, generator=g) if bias else None
    r = torch.randn(R, Cc, generator=g) if resid else None
    xin = F.gelu(x) if (int(gelu) & 1) else x
    want = F.layer_norm(xin, (Cc,), gain, b, 1e-5)
    if resid:
        want = want + r
    out = torch.full((R, Cc), float("nan")) if out_f32 else torch.zeros(R, Cc, dtype=torch.bfloat16)
    lib.emu_layernorm(ptr(x), ptr(gain), ptr(b), ptr(out), ptr(r), R, Cc, C.c_float(1e-5), int(gelu), int(out_f32))
    if out_f32:
        assert torch.allclose(out, want, rtol=1e-5, atol=2e-5), float((out - want).abs().max())
    else:
        assert torch.allclose(out.float(), want, rtol=8e-3, atol=8e-3)          # bf16 output


@pytest.mark.parametrize("out_f32", [True, False])
def test_attention_core_kernel(out_f32):
    lib = _lib()
    g = torch.Generator().manual_seed(9)
    B, heads, dh = 2, 8, 64
    q = torch.randn(B * 16, 512, generator=g)
    kv = torch.randn(B * 16, 128, generator=g)                  # per pixel: k | v (one shared head: imagen's single-head k / v)
    ckv = torch.randn(B * 2, 128, generator=g)                  # 2 time tokens per image
    null = torch.randn(2, 64, generator=g)                      # null key / value
    out = torch.full((B * 16, 512), float("nan")) if out_f32 else torch.zeros(B * 16, 512, dtype=torch.bfloat16)
    ks = (C.c_void_p * 3)(ckv.data_ptr(), null.data_ptr(), kv.data_ptr())
    vs = (C.c_void_p * 3)(ckv.data_ptr() + 256, null.data_ptr() + 256, kv.data_ptr() + 256)
    geo = (C.c_int * 12)(2, 128, 256, 0, 1, 0, 0, 0, 16, 128, 2048, 0)
    lib.emu_attn16(ptr(q), ptr(out), ks, vs, geo, B, heads, 512, C.c_float(dh ** -0.5), int(out_f32))
    qh = q.view(B, 16, heads, dh).permute(0, 2, 1, 3) * dh ** -0.5
    k = torch.cat([ckv[:, :64].view(B, 2, 64), null[0].view(1, 1, 64).expand(B, 1, 64), kv[:, :64].view(B, 16, 64)], 1)
    v = torch.cat([ckv[:, 64:].view(B, 2, 64), null[1].view(1, 1, 64).expand(B, 1, 64), kv[:, 64:].view(B, 16, 64)], 1)
    att = torch.softmax(torch.einsum("bhid,bjd->bhij", qh, k), -1)
    want = torch.einsum("bhij,bjd->bhid", att, v).permute(0, 2, 1, 3).reshape(B * 16, 512)
    if out_f32:
        assert torch.allclose(out, want, rtol=1e-5, atol=1e-5), float((out - want).abs().max())
    else:
        assert torch.allclose(out.float(), want, rtol=8e-3, atol=8e-3)

"""Kernel-logic test of k_init_x (sparsefusion_amd/csrc/initx.h: the latent half of the UNet's init CrossEmbed conv, k = 3 / 7 /
15 into channel slices, external/imagen_pytorch.py:1017-1042) on CPU threads: the kernel source with the weight table of
`unet.init_x_weight_table` against torch conv2d on the same bf16-rounded operands, added to `base`."""
import ctypes as C
import os
import subprocess

import pytest
import torch
import torch.nn.functional as F

from hostemu import fused
from sparsefusion_amd.unet import init_x_weight_table

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostemu")
SO = os.path.join(HERE, "_build", "libinitx_emu.so")
pytestmark = pytest.mark.skipif(not fused.available(), reason="host clang not found")


def _lib():
    srcs = [os.path.join(HERE, "initx_emu.cpp"), os.path.join(HERE, "hip_emu.h")] + \
           [os.path.join(HERE, "..", "..", "sparsefusion_amd", "csrc", f) for f in ("initx.h", "sf_dev.h")]
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(s) for s in srcs):
        os.makedirs(os.path.dirname(SO), exist_ok=True)
        subprocess.check_call([fused.CLANG, "-std=c++17", "-O2", "-fPIC", "-shared", "-I" + HERE, "-Wall", "-Wno-unused-function",
                               "-ffp-contract=off", srcs[0], "-o", SO, "-lpthread"])
    return C.CDLL(SO)


@pytest.mark.parametrize("B,R,Cx,cws", [(1, 16, 4, (128, 64, 64)), (2, 8, 3, (64, 32, 32))])
def test_init_x_kernel_matches_conv2d(B, R, Cx, cws):
    lib = _lib()
    bf = lambda t: t.to(torch.bfloat16).float()
    g = torch.Generator().manual_seed(5 + B)
    ks = (3, 7, 15)
    dim = sum(cws)
    x = torch.randn(B, Cx, R, R, generator=g)
    ws = [torch.randn(cw, Cx, k, k, generator=g) / (Cx * k * k) ** 0.5 for cw, k in zip(cws, ks)]
    base = torch.randn(B * R * R, dim, generator=g)
    want = torch.cat([F.conv2d(bf(x), bf(w), padding=k // 2) for w, k in zip(ws, ks)], 1).permute(0, 2, 3, 1).reshape(B * R * R, dim) + base
    tab, woffs = init_x_weight_table(ws)
    out = torch.full((B * R * R, dim), float("nan"))
    arr = lambda v: (C.c_int * 3)(*v)
    ptr = lambda t: C.c_void_p(t.data_ptr())
    lib.emu_init_x(ptr(x), ptr(base), ptr(tab), ptr(out), B, R, R, Cx, dim, arr(cws), arr([0, cws[0], cws[0] + cws[1]]), arr(woffs), None)
    assert torch.allclose(out, want, rtol=1e-4, atol=2e-4), float((out - want).abs().max())
    # with the statistics slots of the output (what the next GroupNorm-fused conv sums per (image, 16-channel column)): one slot
    # per MFMA fragment = 16 pixels (two rows of an 8 x 8 tile) x 16 channels; every slot written, column sums = the tensor's
    out2 = torch.full_like(out, float("nan"))
    slots = torch.full((B * R * R // 16, dim // 16, 2), float("nan"))
    lib.emu_init_x(ptr(x), ptr(base), ptr(tab), ptr(out2), B, R, R, Cx, dim, arr(cws), arr([0, cws[0], cws[0] + cws[1]]), arr(woffs), ptr(slots))
    assert torch.equal(out2, out) and torch.isfinite(slots).all()
    v = out.view(B, R * R, dim // 16, 16).double()
    direct = torch.stack([v.sum((1, 3)), (v * v).sum((1, 3))], -1)                      # [B][column][2]
    got = slots.view(B, R * R // 16, dim // 16, 2).double().sum(1)
    assert torch.allclose(got, direct, rtol=1e-5, atol=1e-3), float((got - direct).abs().max())

"""Kernel-logic test of k_conv_igemm (sparsefusion_amd/csrc/conv_igemm.h: the weight-streaming implicit GEMM of the UNet's
non-fused layers, the small VAE / VGG maps and the EFT linears) on CPU threads against torch conv2d on the same bf16-rounded
operands: tile shapes, in-workgroup and workspace split-K (summed as k_splitk_reduce does), 4x4 stride-2 Downsample, the
SiLU + PixelShuffle(2) epilogue of the Upsample, residual / accumulate / GELU epilogues, ragged M and Cout."""
import ctypes as C
import os
import subprocess

import pytest
import torch
import torch.nn.functional as F

from hostemu import fused

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostemu")
SO = os.path.join(HERE, "_build", "libconv_igemm_emu" + "".join("_" + d.replace("=", "") for d in fused._DEFS) + ".so")      # SF_EMU_DEFINES: an A/B harness build
pytestmark = pytest.mark.skipif(not fused.available(), reason="host clang not found")


def _lib():
    srcs = [os.path.join(HERE, "conv_igemm_emu.cpp"), os.path.join(HERE, "hip_emu.h")] + \
           [os.path.join(HERE, "..", "..", "sparsefusion_amd", "csrc", f) for f in ("conv_igemm.h", "conv_lds.h", "conv_lds_body.inc", "sf_dev.h")]
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(s) for s in srcs):
        os.makedirs(os.path.dirname(SO), exist_ok=True)
        subprocess.check_call([fused.CLANG, "-std=c++17", "-O2", "-fPIC", "-shared", "-I" + HERE, "-Wall", "-Wno-unused-function",
                               "-ffp-contract=off"] + ["-D" + d for d in fused._DEFS] + [srcs[0], "-o", SO, "-lpthread"])
    return C.CDLL(SO)


def _pack(w):
    from sparsefusion_amd import _lib as L
    lib = L.lib()
    co, ci, kh, kw = w.shape
    cpad = (ci + 31) // 32 * 32
    buf = torch.empty(lib.sf_conv_packed_elems(co, cpad, kh, kw), dtype=torch.int16)
    L.check(lib.sf_conv_pack_weights(w.contiguous().data_ptr(), co, ci, cpad, kh, kw, buf.data_ptr()))
    return buf, cpad


bf = lambda t: t.to(torch.bfloat16).float()

CASES = [
    # B, H, Cin, Cout, k, stride, pad, groups, (WM, WN), a_f32, resid, accum, relu, pixshuf
    (1, 4, 256, 64, 3, 1, 1, 1, (1, 1), False, True, False, 0, False),      # 4x4 map, one tile, residual
    (1, 4, 128, 72, 3, 1, 1, 3, (1, 2), True, False, False, 0, False),      # workspace split-K (3 groups), ragged Cout
    (2, 8, 64, 64, 1, 1, 0, 1, (2, 2), False, False, True, 2, False),       # 1x1, accumulate, GELU
    (1, 8, 32, 64, 4, 2, 1, 1, (1, 2), True, False, False, 0, False),       # Downsample conv 4x4 stride 2 pad 1 (imagen)
    (1, 4, 64, 128, 3, 1, 1, 1, (1, 2), True, False, False, 0, True),       # Upsample: conv -> SiLU -> PixelShuffle(2)
    (2, 4, 64, 256, 1, 1, 0, 1, (1, 1), True, False, False, 0, True),       # the UNet's Upsample: 1x1 conv, two images, + output slots
    (1, 10, 32, 40, 3, 1, 1, 1, (4, 2), False, False, False, 1, False),     # ragged M = 100 (6.25 fragments), ReLU
]


@pytest.mark.parametrize("B,H,Cin,Cout,k,stride,pad,groups,tile,a_f32,resid,accum,relu,pixshuf", CASES)
def test_conv_igemm_matches_conv2d(B, H, Cin, Cout, k, stride, pad, groups, tile, a_f32, resid, accum, relu, pixshuf):
    lib = _lib()
    g = torch.Generator().manual_seed(Cin + Cout + H + k)
    x = torch.randn(B, Cin, H, H, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    b = torch.randn(Cout, generator=g)
    conv = F.conv2d(bf(x), bf(w), None if groups > 1 else b, stride=stride, padding=pad)
    Ho = conv.shape[-1]
    M = B * Ho * Ho
    wp, cpad = _pack(w)
    assert cpad == Cin
    xn = x.permute(0, 2, 3, 1).contiguous()
    xa = xn if a_f32 else xn.to(torch.bfloat16)
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
    WM, WN = tile
    npad = (Cout + 15) // 16 * 16
    if pixshuf:                                                 # [B, 2Ho, 2Ho, Cout/4] = PixelShuffle(2)(SiLU(conv))
        want = F.pixel_shuffle(F.silu(conv), 2).permute(0, 2, 3, 1).reshape(B * 4 * Ho * Ho, Cout // 4)
        out = torch.full((B * 4 * Ho * Ho, Cout // 4), float("nan"))
        Co, Mo = Cout // 4, B * 4 * Ho * Ho
        slots = torch.full((Mo // 16, Co // 16, 2), float("nan")) if Co % 16 == 0 else None
        rc = lib.emu_conv_igemm(ptr(xa), ptr(wp), ptr(b), ptr(out), None, None, B, H, H, Cin, Ho, Ho, Cout, Cout // 4, 0, k, stride, pad, 1,
                                WM, WN, int(a_f32), 0, 0, 0, 1, ptr(slots))
        assert rc == 0 and torch.allclose(out, want, rtol=1e-4, atol=2e-4), float((out - want).abs().max())
        if slots is not None:          # (sum, sum of squares) per MFMA fragment, filed under the right (image, 16-channel column)
            assert torch.isfinite(slots).all()
            v = out.view(B, 4 * Ho * Ho, Co // 16, 16).double()
            direct = torch.stack([v.sum((1, 3)), (v * v).sum((1, 3))], -1)
            got = slots.view(B, 4 * Ho * Ho // 16, Co // 16, 2).double().sum(1)
            assert torch.allclose(got, direct, rtol=1e-5, atol=1e-3), float((got - direct).abs().max())
        return
    want = conv.permute(0, 2, 3, 1).reshape(M, Cout)
    ldc = Cout + 4
    res = torch.randn(M, ldc, generator=g) if resid else None
    out = torch.randn(M, ldc, generator=g) if accum else torch.full((M, ldc), float("nan"))
    if groups > 1:                                              # partial tiles only: sum the slabs + bias as k_splitk_reduce does
        ws = torch.full((groups, M, npad), float("nan"))
        rc = lib.emu_conv_igemm(ptr(xa), ptr(wp), ptr(b), ptr(out), None, ptr(ws), B, H, H, Cin, Ho, Ho, Cout, ldc, 0, k, stride, pad,
                                groups, WM, WN, int(a_f32), 0, 0, 0, 0, None)
        assert rc == 0 and bool(torch.isnan(out).all())         # the kernel itself writes no output in this mode
        got = ws[:, :, :Cout].sum(0) + b
        assert torch.allclose(got, want + b, rtol=1e-4, atol=2e-4), float((got - want - b).abs().max())
        return
    if resid:
        want = want + res[:, :Cout]
    if accum:
        want = want + out[:, :Cout]
    want = want.relu() if relu == 1 else F.gelu(want) if relu == 2 else want
    rc = lib.emu_conv_igemm(ptr(xa), ptr(wp), ptr(b), ptr(out), ptr(res), None, B, H, H, Cin, Ho, Ho, Cout, ldc, 0, k, stride, pad, 1,
                            WM, WN, int(a_f32), int(accum), 0, relu, 0, None)
    assert rc == 0
    got = out[:, :Cout]
    assert torch.allclose(got, want, rtol=1e-4, atol=2e-4), float((got - want).abs().max())

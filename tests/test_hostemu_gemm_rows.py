"""Kernel-logic test of k_gemm_rows (sparsefusion_amd/csrc/gemm_rows.h; the sampler's time table: 9..64 rows x bf16 weights on
the MFMA M side) on CPU threads against torch: fp32 x times bf16-rounded W, the arithmetic of k_gemv, for the shapes of
Unet.emit_time (K = 17 first layer, SiLU on the input, SiLU on the output, ragged N, row strides wider than K / N)."""
import ctypes as C
import os
import subprocess

import pytest
import torch
import torch.nn.functional as F

from hostemu import fused

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostemu")
SO = os.path.join(HERE, "_build", "libgemm_rows_emu.so")
pytestmark = pytest.mark.skipif(not fused.available(), reason="host clang not found")


def _lib():
    srcs = [os.path.join(HERE, "gemm_rows_emu.cpp"), os.path.join(HERE, "hip_emu.h")] + \
           [os.path.join(HERE, "..", "..", "sparsefusion_amd", "csrc", f) for f in ("gemm_rows.h", "sf_dev.h")]
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(s) for s in srcs):
        os.makedirs(os.path.dirname(SO), exist_ok=True)
        subprocess.check_call([fused.CLANG, "-std=c++17", "-O2", "-fPIC", "-shared", "-I" + HERE, "-Wall", "-Wno-unused-function",
                               "-ffp-contract=off", srcs[0], "-o", SO, "-lpthread"])
    return C.CDLL(SO)


@pytest.mark.parametrize("M,N,K,ldx,ldy,in_silu,out_act,bias", [
    (51, 96, 17, 17, 96, 0, 1, True),          # to_time_hiddens.1: K = 17 (padded to 24), SiLU out
    (51, 200, 320, 320, 264, 1, 0, True),      # batched time_mlps: SiLU in, ragged N (200 = 3 x 64 + 8), ldy > N
    (20, 64, 300, 640, 64, 0, 2, False),       # token k/v: row stride 2 x cond_dim, K crosses the 128-column chunks raggedly, no bias
    (64, 48, 64, 64, 48, 0, 0, True),
    (32, 512, 1024, 1024, 512, 0, 1, True),    # GlobalContext net.0 of a 1024-channel block at B = 32: 8 chunks, two per wave (float4 staging)
    (9, 1024, 512, 512, 1024, 0, 2, True),     # net.2 at B = 9: one m-fragment, sigmoid
    (16, 40, 648, 648, 40, 1, 0, False),       # 6 chunks over 8 waves (two idle, the last ragged: 8 columns), ragged N
    (40, 64, 1024, 1024, 64, 0, 0, True),      # 40 rows: the 4-wave form, two chunks per wave
])
@pytest.mark.parametrize("ks", [False, True])
def test_gemm_rows_matches_fp32_x_times_bf16_w(M, N, K, ldx, ldy, in_silu, out_act, bias, ks):
    """ks: k_gemm_rows_ks (r06: N / 16 workgroups, the waves split K by 128-column chunk; the GlobalContext MLPs of the large-batch plans)."""
    lib = _lib()
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, ldx, generator=g)
    Kp = (K + 7) // 8 * 8
    w = torch.zeros(N, Kp)
    w[:, :K] = torch.randn(N, K, generator=g) / K ** 0.5
    wb = w.to(torch.bfloat16)
    b = torch.randn(N, generator=g) if bias else None
    y = torch.full((M, ldy), float("nan"))
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
    (lib.emu_gemm_rows_ks if ks else lib.emu_gemm_rows)(ptr(x), ptr(wb.view(torch.int16)), ptr(b), ptr(y), M, N, K, Kp, ldx, ldy, in_silu, out_act)
    xin = F.silu(x[:, :K]) if in_silu else x[:, :K]
    want = xin.double() @ wb[:, :K].double().t()
    if bias:
        want = want + b.double()
    want = F.silu(want) if out_act == 1 else torch.sigmoid(want) if out_act == 2 else want
    got = y[:, :N].double()
    assert bool(torch.isnan(y[:, N:]).all())                 # nothing written beyond N
    err = float((got - want).abs().max()) / float(want.abs().max())
    assert err < 3e-5, err                                    # hi + lo split of x: 16 mantissa bits

"""Kernel-logic test of the (experimental, SF_NGP_FWD_MFMA=1) MFMA field-forward kernel (sparsefusion_amd/csrc/ngp_fwd_mfma.h) on
CPU threads against the per-point loop of k_ngp_field (ngp_device.h math): depths bit-exact, sigma / albedo to fp32 summation
order, both depth modes, a ragged last trip, one and several workgroups."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import ngp_ref
from hostemu import fused

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostemu")
SO = os.path.join(HERE, "_build", "libngp_fwd_emu.so")
pytestmark = pytest.mark.skipif(not fused.available(), reason="host clang not found")


def _lib():
    srcs = [os.path.join(HERE, "ngp_fwd_emu.cpp"), os.path.join(HERE, "hip_emu.h")] + \
           [os.path.join(HERE, "..", "..", "sparsefusion_amd", "csrc", f) for f in ("ngp_fwd_mfma.h", "ngp_bwd_mfma.h", "ngp_device.h", "sf_dev.h")]
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(s) for s in srcs):
        os.makedirs(os.path.dirname(SO), exist_ok=True)
        subprocess.check_call([fused.CLANG, "-std=c++17", "-O2", "-fPIC", "-shared", "-I" + HERE, "-Wall", "-Wno-unused-function",
                               "-ffp-contract=off", srcs[0], "-o", SO, "-lpthread"])
    return C.CDLL(SO)


@pytest.mark.parametrize("mode", [0, 1])
def test_field_forward_mfma_matches_per_point_math(mode):
    lib = _lib()
    p = ngp_ref.init_params(bound=4, seed=3, table_std=0.5, sigma_bias=-1.0)
    g = torch.Generator().manual_seed(mode)
    N, T = 9, 22                                              # P = 198: six full trips of 32 points + a ragged one
    P = N * T
    o, d = ngp_ref.circle_rays(3, view=2)
    o, d = o[:N].contiguous(), d[:N].contiguous()
    nears = torch.full((N,), 2.0) + torch.rand(N, generator=g)
    fars = nears + 6.0                                        # some samples leave the box
    lin = torch.linspace(0.0, 1.0, T)
    u = torch.rand(N, T, generator=g)
    z_in = (torch.rand(N, T, generator=g) * 9.0 + 1.0).sort(1).values.contiguous()
    offs = p["encoder.offsets"].to(torch.int32).contiguous()
    L = offs.numel() - 1
    S = float(np.log2(ngp_ref.per_level_scale(4)))
    ws = [p[f"sigma_net.net.{i}.{w}"].contiguous() for i in range(3) for w in ("weight", "bias")]
    aabb = p["aabb_train"].contiguous()
    ptr = lambda t: C.c_void_p(t.data_ptr())

    def run(use_ref, grid):
        out = [torch.full((P,), -7.0), torch.full((P,), float("nan")), torch.full((P, 3), float("nan"))]
        lib.emu_field_fwd(ptr(p["encoder.embeddings"]), ptr(offs), C.c_uint32(L), C.c_float(S), C.c_uint32(16), C.c_uint32(1),
                          *[ptr(w) for w in ws], C.c_float(4.0), ptr(o), ptr(d), ptr(aabb), ptr(nears), ptr(fars), ptr(lin), ptr(u),
                          ptr(z_in), C.c_uint32(P), C.c_uint32(T), C.c_int(mode), C.c_uint32(grid), C.c_int(use_ref),
                          *[ptr(t) for t in out])
        return out

    ref = run(1, 1)
    assert float(ref[1].max()) > 10 * float(ref[1].min()) > 0           # a non-trivial field: densities over a range
    for grid in (1, 3):
        got = run(0, grid)
        if mode == 0:
            assert torch.equal(got[0], ref[0])                           # sample depths: bit-exact
        assert torch.allclose(got[1], ref[1], rtol=2e-5, atol=1e-7), float(((got[1] - ref[1]).abs() / ref[1]).max())
        assert torch.allclose(got[2], ref[2], rtol=2e-5, atol=1e-6), float((got[2] - ref[2]).abs().max())

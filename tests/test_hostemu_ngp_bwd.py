"""Kernel-logic test of the MFMA field-backward kernel (sparsefusion_amd/csrc/ngp_bwd_mfma.h) on CPU threads against the
per-point reference math of ngp_device.h (the functions the oracle-pinned host emulation of the render uses): MLP weight /
bias gradients and the per-level feature gradients for random rays, and a ragged last trip."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import ngp_ref
from hostemu import fused

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostemu")
SO = os.path.join(HERE, "_build", "libngp_bwd_emu.so")
pytestmark = pytest.mark.skipif(not fused.available(), reason="host clang not found")


def _lib():
    srcs = [os.path.join(HERE, "ngp_bwd_emu.cpp"), os.path.join(HERE, "hip_emu.h")] + \
           [os.path.join(HERE, "..", "..", "sparsefusion_amd", "csrc", f) for f in ("ngp_bwd_mfma.h", "ngp_scatter_bin.h", "ngp_device.h", "sf_dev.h")]
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(s) for s in srcs):
        os.makedirs(os.path.dirname(SO), exist_ok=True)
        subprocess.check_call([fused.CLANG, "-std=c++17", "-O2", "-fPIC", "-shared", "-I" + HERE, "-Wall", "-Wno-unused-function",
                               "-ffp-contract=off", srcs[0], "-o", SO, "-lpthread"])
    return C.CDLL(SO)


def test_field_backward_mfma_matches_per_point_math():
    lib = _lib()
    p = ngp_ref.init_params(bound=4, seed=3, table_std=0.5, sigma_bias=-1.0)
    g = torch.Generator().manual_seed(0)
    N, T2 = 9, 22                                             # P = 198: six full trips of 32 points + a ragged one
    P = N * T2
    o, d = ngp_ref.circle_rays(3, view=2)
    o, d = o[:N].contiguous(), d[:N].contiguous()
    z = (torch.rand(N, T2, generator=g) * 9.0 + 1.0).sort(1).values.contiguous()      # some samples leave the box
    dsig = torch.randn(P, generator=g)
    drgb = torch.randn(P, 3, generator=g)
    offs = p["encoder.offsets"].to(torch.int32).contiguous()
    L = offs.numel() - 1
    S = float(np.log2(ngp_ref.per_level_scale(4)))
    ws = [p[f"sigma_net.net.{i}.{w}"].contiguous() for i in range(3) for w in ("weight", "bias")]
    aabb = p["aabb_train"].contiguous()
    ptr = lambda t: C.c_void_p(t.data_ptr())

    def run(use_ref, grid):
        out = [torch.zeros(64, 32), torch.zeros(64), torch.zeros(64, 64), torch.zeros(64), torch.zeros(4, 64), torch.zeros(4),
               torch.zeros(L, P, 2)]
        lib.emu_field_bwd(ptr(p["encoder.embeddings"]), ptr(offs), C.c_uint32(L), C.c_float(S), C.c_uint32(16), C.c_uint32(1),
                          *[ptr(w) for w in ws], C.c_float(4.0), ptr(o), ptr(d), ptr(aabb), ptr(z), ptr(dsig), ptr(drgb),
                          C.c_uint32(P), C.c_uint32(T2), C.c_uint32(grid), C.c_int(use_ref), *[ptr(t) for t in out])
        return out

    ref = run(1, 1)
    for grid in (1, 3):                                       # one workgroup (4 waves share the trips) and several
        got = run(0, grid)
        for name, a, b in zip(("g_w0", "g_b0", "g_w1", "g_b1", "g_w2", "g_b2", "dfeat"), got, ref):
            err = float((a - b).abs().max()) / max(float(b.abs().max()), 1e-20)
            assert err < 2e-5, (grid, name, err)
    assert float(ref[6].abs().max()) > 0 and float(ref[2].abs().max()) > 0
    # r03: the same kernel reading the forward's field cache (features of every sample through the sort permutation) instead of
    # re-gathering them
    got = run(2, 3)
    for name, a, b in zip(("g_w0", "g_b0", "g_w1", "g_b1", "g_w2", "g_b2", "dfeat"), got, ref):
        err = float((a - b).abs().max()) / max(float(b.abs().max()), 1e-20)
        assert err < 2e-5, ("cache", name, err)


@pytest.mark.parametrize("case", ["roomy", "overflow", "tiled"])
def test_binned_scatter_matches_per_corner_adds(case):
    """k_ngp_bin + k_ngp_bin_reduce (csrc/ngp_scatter_bin.h, r04: the hashed levels' table gradient without per-corner device
    atomics) against ngp_scatter of ngp_device.h on the same feature gradients: roomy buckets, buckets so small that most
    contributions take the overflow path (direct adds), and a `tiled` grid (z-dropped levels, skewed buckets); two chunks share the
    entries buffer, several tiles per workgroup, a ragged last tile."""
    lib = _lib()
    g = torch.Generator().manual_seed(1)
    N, T2 = 150, 22                                           # P = 3300: tiles of 1024 samples, the last one ragged
    P = N * T2
    o, d = ngp_ref.circle_rays(13, view=1)
    o, d = o[:N].contiguous(), d[:N].contiguous()
    z = (torch.rand(N, T2, generator=g) * 9.0 + 1.0).sort(1).values.contiguous()
    L, S = 16, float(np.log2(ngp_ref.per_level_scale(4)))
    sizes = [min(1 << 12, (int(np.ceil(16 * 2.0 ** (l * S))) + 1) ** 3) for l in range(L)]      # a 2^12-row hash map: 4 buckets per level
    offs = torch.tensor(np.concatenate([[0], np.cumsum([(n + 7) // 8 * 8 for n in sizes])]), dtype=torch.int32)
    dfeat = torch.randn(L, P, 2, generator=g)
    dfeat[:, ::7] = 0.0                                       # dead samples
    aabb = torch.tensor([-4.0] * 3 + [4.0] * 3)
    ptr = lambda t: C.c_void_p(t.data_ptr())
    first, cap, gridtype = {"roomy": (6, 16384, 0), "overflow": (10, 8, 0), "tiled": (9, 1024, 1)}[case]

    def run(use_ref, chunks=1, grid=1):
        tab = torch.zeros(int(offs[-1]), 2)
        rc = lib.emu_bin_scatter(ptr(offs), C.c_uint32(L), C.c_float(S), C.c_uint32(16), C.c_uint32(gridtype), C.c_float(4.0), ptr(o), ptr(d),
                                 ptr(aabb), ptr(z), ptr(dfeat), C.c_uint32(N), C.c_uint32(T2), C.c_uint32(first), C.c_uint32(cap),
                                 C.c_uint32(chunks), C.c_uint32(grid), C.c_int(use_ref), ptr(tab))
        assert rc == 0, rc
        return tab

    ref = run(1)
    lo = int(offs[first])
    assert float(ref[lo:].abs().max()) > 0 and float(ref[:lo].abs().max()) == 0
    for chunks, grid in ((1, 1), (2, 3)) if case == "roomy" else ((2, 2),):
        got = run(0, chunks, grid)
        err = float((got - ref).abs().max()) / float(ref.abs().max())
        assert err < 1e-5, (case, chunks, grid, err)

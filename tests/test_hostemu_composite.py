"""Kernel-logic test of the wave-per-ray merge + composite (sparsefusion_amd/csrc/ngp_composite_wave.h) on CPU threads
against the per-ray loop of ngp_device.h (`ngp_merge_composite`, external/nerf/renderer_df.py:300-345: sort of
cat([coarse, fine]) depths, alpha compositing): unsorted fine depths, ties between coarse and fine samples, a miss ray
(near == far -> NaN depth), T below the wave width and a ragged last workgroup."""
import ctypes as C
import os
import subprocess

import pytest
import torch

from hostemu import fused

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostemu")
SO = os.path.join(HERE, "_build", "libngp_composite_emu.so")
pytestmark = pytest.mark.skipif(not fused.available(), reason="host clang not found")


def _lib():
    srcs = [os.path.join(HERE, "ngp_composite_emu.cpp"), os.path.join(HERE, "hip_emu.h")] + \
           [os.path.join(HERE, "..", "..", "sparsefusion_amd", "csrc", f) for f in ("ngp_composite_wave.h", "ngp_device.h", "sf_dev.h")]
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(s) for s in srcs):
        os.makedirs(os.path.dirname(SO), exist_ok=True)
        subprocess.check_call([fused.CLANG, "-std=c++17", "-O2", "-fPIC", "-shared", "-I" + HERE, "-Wall", "-Wno-unused-function",
                               "-ffp-contract=off", srcs[0], "-o", SO, "-lpthread"])
    return C.CDLL(SO)


@pytest.mark.parametrize("N,T", [(6, 64), (5, 16), (3, 4)])
def test_wave_composite_matches_per_ray_loop(N, T):
    lib = _lib()
    g = torch.Generator().manual_seed(N * 100 + T)
    near = torch.rand(N, generator=g) + 0.5
    far = near + 2.0 + torch.rand(N, generator=g)
    z_c = (near[:, None] + (far - near)[:, None] * torch.rand(N, T, generator=g)).sort(1).values.contiguous()
    z_f = (near[:, None] + (far - near)[:, None] * torch.rand(N, T, generator=g)).contiguous()       # NOT sorted
    z_f[0, : T // 2] = z_c[0, : T // 2]                       # ties: the coarse sample goes first
    z_f[1, 1] = z_f[1, 0]                                     # tie inside the fine samples: source order
    far[2] = near[2]                                          # miss ray: (z - near) / 0
    z_c[2], z_f[2] = near[2], near[2]
    sig_c, sig_f = torch.rand(N, T, generator=g) * 3, torch.rand(N, T, generator=g) * 30
    rgb_c, rgb_f = torch.rand(N, T, 3, generator=g), torch.rand(N, T, 3, generator=g)
    ptr = lambda t: C.c_void_p(t.data_ptr())

    def run(use_ref):
        out = [torch.full((N, 2 * T), -1.0), torch.full((N, 2 * T), -1.0), torch.full((N, 2 * T, 3), -1.0), torch.zeros(N, 3),
               torch.zeros(N), torch.zeros(N)]
        lib.emu_composite(ptr(z_c), ptr(sig_c), ptr(rgb_c), ptr(z_f), ptr(sig_f), ptr(rgb_f), ptr(near), ptr(far), C.c_uint32(N),
                          C.c_uint32(T), C.c_float(0.25), C.c_int(use_ref), *[ptr(t) for t in out])
        return out

    ref, got = run(1), run(0)
    for name, a, b in zip(("z_sorted", "sigma_sorted", "rgb_sorted"), got[:3], ref[:3]):
        assert torch.equal(a, b), name                        # the merge is bit-exact: same stable order
    for name, a, b in zip(("image", "depth", "weights_sum"), got[3:], ref[3:]):
        assert torch.allclose(a, b, rtol=0, atol=5e-7, equal_nan=True), (name, float((a - b).abs().nan_to_num().max()))
    assert bool(torch.isnan(got[4][2])) and bool(torch.isnan(ref[4][2]))          # miss ray: NaN depth, as the reference


@pytest.mark.parametrize("N,T,with_ws", [(6, 64, True), (5, 16, False), (3, 4, True)])
def test_wave_composite_backward_matches_per_ray_loop(N, T, with_ws):
    """k_ngp_composite_bwd_wave against ngp_composite_backward (the adjoint of the alpha compositing, renderer_df.py:318-345):
    d(loss)/d(sigma), d(loss)/d(rgb) of sorted rays, with and without a weights_sum gradient."""
    lib = _lib()
    g = torch.Generator().manual_seed(7 * N + T)
    near = torch.rand(N, generator=g) + 0.5
    far = near + 2.0 + torch.rand(N, generator=g)
    z = (near[:, None] + (far - near)[:, None] * torch.rand(N, 2 * T, generator=g)).sort(1).values.contiguous()
    z[0, 3] = z[0, 2]                                         # a zero-length interval
    sig = (torch.rand(N, 2 * T, generator=g) * 8).contiguous()
    rgb = torch.rand(N, 2 * T, 3, generator=g)
    gi = torch.randn(N, 3, generator=g)
    gw = torch.randn(N, generator=g) if with_ws else None
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)

    def run(use_ref):
        out = [torch.full((N, 2 * T), float("nan")), torch.full((N, 2 * T, 3), float("nan"))]
        lib.emu_composite_bwd(ptr(z), ptr(sig), ptr(rgb), ptr(near), ptr(far), C.c_uint32(N), C.c_uint32(T), C.c_float(0.25), ptr(gi),
                              ptr(gw), C.c_int(use_ref), *[ptr(t) for t in out])
        return out

    ref, got = run(1), run(0)
    for name, a, b in zip(("dsigma", "drgb"), got, ref):
        scale = float(b.abs().max())
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6 * scale), (name, float((a - b).abs().max()), scale)

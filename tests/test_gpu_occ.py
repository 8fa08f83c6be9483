"""Occupancy-grid ray marching / compositing (`_raymarching` cuda_ray=True entry points) on the HIP library vs the
C oracle.  Index / position work (sample positions, deltas, (ray, offset, count) bookkeeping, counters, alive
flags) must be BIT-EXACT; the compositing sums use the hardware fast exp and are compared at 2e-6 absolute."""
import pytest
import torch

from oracle import ngp_native
from occ_common import BOUND, CASCADE, H, MAX_STEPS, ball_bitfield, camera_rays, near_far, oracle_march_train

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def scene():
    grid, bits, center = ball_bitfield()
    o, d = camera_rays(n_side=40)                      # 1600 rays: spans two scan groups
    nears, fars = near_far(o, d)
    return grid, bits, o, d, nears, fars


def _dev(*ts):
    return [t.to(DEV) for t in ts]


@pytest.mark.parametrize("dt_gamma,perturb", [(0.0, False), (0.0, True), (1.0 / 128, True)])
def test_march_rays_train_bit_exact(scene, dt_gamma, perturb):
    from sparsefusion_amd import raymarching
    grid, bits, o, d, nears, fars = scene
    N = o.shape[0]
    noises = torch.rand(N, generator=torch.Generator().manual_seed(5)) if perturb else torch.zeros(N)
    xyzs, dirs, deltas, rays, counter = oracle_march_train(o, d, bits, nears, fars, noises, dt_gamma)
    total = int(counter[0])
    cnt = torch.zeros(2, dtype=torch.int32, device=DEV)
    gx, gd, gl, gr = raymarching.march_rays_train(*_dev(o, d), BOUND, bits.to(DEV), CASCADE, H, *_dev(nears, fars), cnt, -1,
                                                  perturb, 128, True, dt_gamma, MAX_STEPS, noises=noises.to(DEV))
    assert cnt.cpu().tolist() == [total, N]
    assert torch.equal(gr.cpu(), rays)
    m = total + (128 - total % 128)                                    # align=128 padding, as the reference wrapper
    assert gx.shape[0] == m
    assert torch.equal(gx.cpu()[:total], xyzs[:total]) and torch.equal(gd.cpu()[:total], dirs[:total])
    assert torch.equal(gl.cpu()[:total], deltas[:total])
    assert (gx.cpu()[total:] == 0).all()


def test_march_rays_train_overflow_counter_base_and_empty(scene):
    from sparsefusion_amd.raymarching import backend
    grid, bits, o, d, nears, fars = scene
    N = o.shape[0]
    noises = torch.zeros(N)
    for M, base in ((5000, 0), (None, 777)):
        counter = torch.tensor([base, 0], dtype=torch.int32)
        rx, rd, rl, rr, rc = oracle_march_train(o, d, bits, nears, fars, noises, M=M, counter=counter.clone())
        Mv = rx.shape[0]
        gx, gd, gl = (torch.zeros(Mv, k, device=DEV) for k in (3, 3, 2))
        gr = torch.full((N, 3), -7, dtype=torch.int32, device=DEV)
        gc = counter.to(DEV)
        backend.march_rays_train(*_dev(o, d, bits), BOUND, 0.0, MAX_STEPS, N, CASCADE, H, Mv, *_dev(nears, fars), gx, gd, gl, gr, gc,
                                 noises.to(DEV))
        assert gc.cpu().tolist() == rc.tolist() and torch.equal(gr.cpu(), rr)
        assert torch.equal(gx.cpu(), rx) and torch.equal(gl.cpu(), rl) and torch.equal(gd.cpu(), rd)
    # N = 0 is a no-op; a CPU tensor or wrong dtype raises (no silent .cuda() copies, no CPU path)
    e = torch.zeros(0, 3, device=DEV)
    backend.march_rays_train(e, e, bits.to(DEV), BOUND, 0.0, MAX_STEPS, 0, CASCADE, H, 0, e[:, 0], e[:, 0], e, e, e[:, :2],
                             torch.zeros(0, 3, dtype=torch.int32, device=DEV), torch.zeros(2, dtype=torch.int32, device=DEV), e[:, 0])
    with pytest.raises(RuntimeError):
        backend.march_rays_train(o, d, bits, BOUND, 0.0, MAX_STEPS, N, CASCADE, H, 10, nears, fars, torch.zeros(10, 3),
                                 torch.zeros(10, 3), torch.zeros(10, 2), torch.zeros(N, 3, dtype=torch.int32),
                                 torch.zeros(2, dtype=torch.int32), noises)


def test_composite_rays_train_forward_backward(scene):
    from sparsefusion_amd import raymarching
    grid, bits, o, d, nears, fars = scene
    N = o.shape[0]
    xyzs, dirs, deltas, rays, counter = oracle_march_train(o, d, bits, nears, fars, torch.zeros(N), M=60000)   # some rays overflow
    M = 60000
    g = torch.Generator().manual_seed(9)
    sigmas, rgbs = torch.rand(M, generator=g) * 4, torch.rand(M, 3, generator=g)
    gws, gim = torch.randn(N, generator=g), torch.randn(N, 3, generator=g)
    for T_thresh in (1e-4, 0.0):
        ws, depth, image = torch.empty(N), torch.empty(N), torch.empty(N, 3)
        ngp_native.composite_rays_train_forward(sigmas, rgbs, deltas, rays, M, N, T_thresh, ws, depth, image)
        gs, gc = torch.zeros(M), torch.zeros(M, 3)
        ngp_native.composite_rays_train_backward(gws, gim, sigmas, rgbs, deltas, rays, ws, image, M, N, T_thresh, gs, gc)
        s_d, c_d = sigmas.to(DEV).requires_grad_(True), rgbs.to(DEV).requires_grad_(True)
        w_d, z_d, i_d = raymarching.composite_rays_train(s_d, c_d, deltas.to(DEV), rays.to(DEV), T_thresh)
        assert torch.allclose(w_d.cpu(), ws, atol=2e-6) and torch.allclose(i_d.detach().cpu(), image, atol=2e-6)
        assert torch.allclose(z_d.detach().cpu(), depth, rtol=1e-5, atol=1e-5)
        ((w_d * gws.to(DEV)).sum() + (i_d * gim.to(DEV)).sum() + z_d.sum() * 0.0).backward()
        assert torch.allclose(c_d.grad.cpu(), gc, atol=2e-6)
        assert torch.allclose(s_d.grad.cpu(), gs, rtol=1e-4, atol=2e-5)
        assert (s_d.grad.cpu()[gs == 0] == 0).all()                    # terminated / overflowed samples stay exactly zero


def test_inference_march_and_composite_loop(scene):
    """The reference's eval loop (renderer_df.py:523-556) on both sides, round by round."""
    from sparsefusion_amd import raymarching
    grid, bits, o, d, nears, fars = scene
    N = o.shape[0]

    def field(x):
        return (2.0 + torch.sin(3 * x).sum(-1)).clamp(min=0).contiguous(), torch.sigmoid(x * 2).contiguous()

    ws, depth, image = torch.zeros(N), torch.zeros(N), torch.zeros(N, 3)
    gw, gz, gi = (t.clone().to(DEV) for t in (ws, depth, image))
    alive, rays_t = torch.arange(N, dtype=torch.int32), nears.clone()
    g_alive, g_t = alive.to(DEV), rays_t.to(DEV)
    od, dd, bd, nd, fd = _dev(o, d, bits, nears, fars)
    step = 0
    while step < MAX_STEPS and alive.numel() > 0:
        n_alive = alive.numel()
        n_step = max(min(N // n_alive, 8), 1)
        noises = torch.rand(n_alive, generator=torch.Generator().manual_seed(step)) if step == 0 else torch.zeros(n_alive)
        x, dr, dl = torch.zeros(n_alive * n_step, 3), torch.zeros(n_alive * n_step, 3), torch.zeros(n_alive * n_step, 2)
        ngp_native.march_rays(n_alive, n_step, alive, rays_t, o, d, BOUND, 0.0, MAX_STEPS, CASCADE, H, bits, nears, fars, x, dr, dl,
                              noises)
        gx, gd, gl = raymarching.march_rays(n_alive, n_step, g_alive, g_t, od, dd, BOUND, bd, CASCADE, H, nd, fd, 128,
                                            step == 0, 0.0, MAX_STEPS, noises=noises.to(DEV))
        m = n_alive * n_step
        assert torch.equal(gx.cpu()[:m], x) and torch.equal(gl.cpu()[:m], dl) and torch.equal(gd.cpu()[:m], dr)
        s, c = field(x)
        ngp_native.composite_rays(n_alive, n_step, 1e-2, alive, rays_t, s, c, dl, ws, depth, image)
        raymarching.composite_rays(n_alive, n_step, g_alive, g_t, s.to(DEV), c.to(DEV), gl[:m].contiguous(), gw, gz, gi, 1e-2)
        assert torch.equal(g_alive.cpu(), alive) and torch.equal(g_t.cpu(), rays_t)     # termination flags and ray times
        alive = alive[alive >= 0].contiguous()
        g_alive = g_alive[g_alive >= 0].contiguous()
        step += n_step
    assert alive.numel() == 0
    assert torch.allclose(gw.cpu(), ws, atol=2e-6) and torch.allclose(gi.cpu(), image, atol=2e-6)
    assert torch.allclose(gz.cpu(), depth, rtol=1e-5, atol=1e-5)


def test_sph_from_ray_matches_oracle():
    from sparsefusion_amd import raymarching
    g = torch.Generator().manual_seed(0)
    o, d = torch.randn(1000, 3, generator=g) * 0.5, torch.randn(1000, 3, generator=g)
    ref = torch.empty(1000, 2)
    ngp_native.sph_from_ray(o, d, 5.0, 1000, ref)
    out = raymarching.sph_from_ray(o.to(DEV), d.to(DEV), 5.0).cpu()
    assert torch.allclose(out, ref, atol=2e-6)


def test_shim_exports_every_reference_entry_point():
    """raymarching/src/bindings.cpp:7-18 registers exactly these ten names."""
    import importlib
    import os
    import sys
    import sparsefusion_amd
    shim_dir = os.path.join(os.path.dirname(sparsefusion_amd.__file__), "shims")
    sys.path.insert(0, shim_dir)
    try:
        m = importlib.import_module("_raymarching")
    finally:
        sys.path.remove(shim_dir)
    for name in ("near_far_from_aabb", "sph_from_ray", "morton3D", "morton3D_invert", "packbits", "march_rays_train",
                 "composite_rays_train_forward", "composite_rays_train_backward", "march_rays", "composite_rays"):
        assert callable(getattr(m, name))

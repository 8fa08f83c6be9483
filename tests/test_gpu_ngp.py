"""GPU parity tests of the NGP side of the hot path: HIP kernels (through the C ABI) vs the CPU oracle and the
golden fixtures generated from the real reference.  Bit-exact for index / ray bookkeeping, stated
tolerances for floating point."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import ngp_native, ngp_ref
from ngp_common import BOUND, grad_leaf, log2_scale, params_from_cfg, psnr

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def golden(golden_dir):
    return torch.load(f"{golden_dir}/ngp_render.pt")


def _net(p):
    from sparsefusion_amd.nerf import NeRFNetwork, get_default_torch_ngp_opt
    net = NeRFNetwork(get_default_torch_ngp_opt())
    net.load_state_dict({k: p[k] for k in net.state_dict().keys()})
    return net.to(DEV)


# ----------------------------------------------------------------------------- grid encode (G1/G2/G3)
@pytest.mark.parametrize("D,Cc,gridtype,L,B", [(3, 2, 1, 16, 10007), (3, 2, 0, 16, 4096), (2, 4, 0, 8, 777),
                                                 (3, 8, 1, 4, 300), (1, 1, 0, 4, 65)])
def test_grid_forward_bit_exact_and_backward(D, Cc, gridtype, L, B):
    from sparsefusion_amd.gridencoder import backend, encoder
    g = torch.Generator().manual_seed(B)
    scale = 1.5
    offs = torch.from_numpy(encoder.level_offsets(D, L, scale, 16, 14, False))
    table = torch.randn(int(offs[-1]), Cc, generator=g)
    x = torch.rand(B, D, generator=g)
    x[0] = 0.0; x[1] = 1.0; x[2, 0] = 1.5; x[3, D - 1] = -0.25       # interval ends + out-of-range rows
    S = float(np.log2(scale))
    want, want_dx = torch.empty(L, B, Cc), torch.empty(B, L * D * Cc)
    ngp_native.grid_encode_forward(x, table, offs, want, B, D, Cc, L, S, 16, want_dx, gridtype, False)
    got, got_dx = torch.empty(L, B, Cc, device=DEV), torch.empty(B, L * D * Cc, device=DEV)
    xd, td, od = x.to(DEV), table.to(DEV), offs.to(DEV)
    backend.grid_encode_forward(xd, td, od, got, B, D, Cc, L, S, 16, got_dx, gridtype, False)
    assert torch.equal(got.cpu(), want), "features (and therefore cell indices) must be bit-exact"
    assert torch.equal(got_dx.cpu(), want_dx)
    assert got[:, 2].abs().max() == 0 and got[:, 3].abs().max() == 0
    gy = torch.randn(L, B, Cc, generator=g)
    want_gt, want_gx = torch.zeros_like(table), torch.zeros_like(x)
    ngp_native.grid_encode_backward(gy, x, table, offs, want_gt, B, D, Cc, L, S, 16, want_dx, want_gx, gridtype, False)
    got_gt, got_gx = torch.zeros_like(td), torch.zeros_like(xd)
    backend.grid_encode_backward(gy.to(DEV), xd, td, od, got_gt, B, D, Cc, L, S, 16, got_dx, got_gx, gridtype, False)
    assert torch.allclose(got_gt.cpu(), want_gt, rtol=1e-4, atol=1e-5)      # fp32 atomics: order-dependent
    assert torch.equal(got_gx.cpu(), want_gx)


def test_grid_encoder_module_autograd():
    from sparsefusion_amd.gridencoder import GridEncoder
    torch.manual_seed(0)
    enc = GridEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=16,
                      desired_resolution=2048 * BOUND, gridtype='tiled')
    enc.embeddings.data.uniform_(-0.5, 0.5)
    x = (torch.rand(5000, 3) * 2 - 1) * BOUND
    w = torch.randn(5000, 32)
    ref_table = enc.embeddings.detach().clone().requires_grad_(True)
    y_ref = ngp_native.GridEncodeCPU.apply(((x + BOUND) / (2 * BOUND)), ref_table, enc.offsets.clone(),
                                           enc.per_level_scale, 16, False, 1, False)
    (y_ref * w).sum().backward()
    enc = enc.to(DEV)
    y = enc(x.to(DEV), bound=BOUND)
    (y * w.to(DEV)).sum().backward()
    assert torch.equal(y.detach().cpu(), y_ref.detach())
    assert torch.allclose(enc.embeddings.grad.cpu(), ref_table.grad, rtol=1e-4, atol=1e-5)


def test_grid_errors_and_empty():
    from sparsefusion_amd.gridencoder import backend
    x = torch.rand(8, 3, device=DEV)
    offs = torch.tensor([0, 64], dtype=torch.int32, device=DEV)
    with pytest.raises(RuntimeError, match="C must be 1, 2, 4, or 8"):
        backend.grid_encode_forward(x, torch.zeros(64, 3, device=DEV), offs, torch.empty(1, 8, 3, device=DEV), 8, 3, 3, 1,
                                    1.0, 16, None, 0, False)
    with pytest.raises(RuntimeError, match="D must be"):
        backend.grid_encode_forward(torch.rand(8, 6, device=DEV), torch.zeros(64, 2, device=DEV), offs,
                                    torch.empty(1, 8, 2, device=DEV), 8, 6, 2, 1, 1.0, 16, None, 0, False)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        backend.grid_encode_forward(x.cpu(), torch.zeros(64, 2, device=DEV), offs, torch.empty(1, 8, 2, device=DEV), 8, 3,
                                    2, 1, 1.0, 16, None, 0, False)
    backend.grid_encode_forward(x[:0].contiguous(), torch.zeros(64, 2, device=DEV), offs, torch.empty(1, 0, 2, device=DEV),
                                0, 3, 2, 1, 1.0, 16, None, 0, False)      # B = 0 is a no-op


# ----------------------------------------------------------------------------- ray utilities (R1)
def test_near_far_bit_exact():
    from sparsefusion_amd import raymarching
    g = torch.Generator().manual_seed(5)
    N = 100003
    o = torch.randn(N, 3, generator=g) * 6
    d = torch.randn(N, 3, generator=g)
    d[0, 1] = 0.0; d[1] = torch.tensor([0.0, 0.0, 1.0]); o[2] = torch.tensor([50.0, 50.0, 50.0])
    aabb = torch.tensor([-4.0, -4, -4, 4, 4, 4])
    wn, wf = torch.empty(N), torch.empty(N)
    ngp_native.near_far_from_aabb(o, d, aabb, N, 0.1, wn, wf)
    n, f = raymarching.near_far_from_aabb(o.to(DEV), d.to(DEV), aabb.to(DEV), 0.1)
    assert torch.equal(n.cpu(), wn) and torch.equal(f.cpu(), wf)
    assert float((wn == torch.finfo(torch.float32).max).float().mean()) > 0.05      # misses are exercised


def test_morton_and_packbits_bit_exact():
    from sparsefusion_amd import raymarching
    g = torch.Generator().manual_seed(6)
    coords = torch.randint(0, 128, (128 * 128 * 4, 3), generator=g, dtype=torch.int32)
    want = torch.empty(coords.shape[0], dtype=torch.int32)
    ngp_native.morton3D(coords, coords.shape[0], want)
    got = raymarching.morton3D(coords.to(DEV))
    assert torch.equal(got.cpu(), want)
    assert torch.equal(raymarching.morton3D_invert(got).cpu(), coords)
    grid = torch.rand(3, 128 ** 3 // 64, generator=g) * 20
    wb = torch.empty(grid.numel() // 8, dtype=torch.uint8)
    ngp_native.packbits(grid.view(-1), grid.numel() // 8, 10.0, wb)
    assert torch.equal(raymarching.packbits(grid.to(DEV), 10.0).cpu(), wb)


# ----------------------------------------------------------------------------- field (N1)
def test_density_matches_oracle(golden):
    p = params_from_cfg(golden["teacher"]["cfg"])
    net = _net(p).eval()
    x = (torch.rand(20000, 3) * 2 - 1) * BOUND
    x[0] = torch.tensor([BOUND, -BOUND, BOUND]); x[1] = 0.0
    sig_ref, alb_ref = ngp_ref.common_forward(p, x, BOUND)
    with torch.no_grad():
        out = net.density(x.to(DEV))
    assert torch.allclose(out["sigma"].cpu(), sig_ref, rtol=2e-5, atol=1e-7)
    assert torch.allclose(out["albedo"].cpu(), alb_ref, rtol=2e-5, atol=1e-6)
    # differentiable route (HIP encode op + torch MLP) agrees with the fused one
    net.train()
    s2, a2 = net.common_forward(x.to(DEV))
    assert torch.allclose(s2.detach(), out["sigma"], rtol=2e-5, atol=1e-7)
    assert torch.allclose(a2.detach(), out["albedo"], rtol=2e-5, atol=1e-6)


# ----------------------------------------------------------------------------- fused render (R2)
@pytest.mark.parametrize("name", ["teacher", "default_init"])
def test_render_matches_reference_golden(golden, name):
    g = golden[name]
    p = params_from_cfg(g["cfg"])
    net = _net(p).train()
    noise = dict(u_coarse=g["u_coarse"].to(DEV), u_fine=g["u_fine"].to(DEV))
    opt = vars(net.opt)
    r = net.render(g["rays_o"][None].to(DEV), g["rays_d"][None].to(DEV), staged=False, perturb=True, bg_color=0,
                   ambient_ratio=1.0, shading='albedo', force_all_rays=True, noise=noise, **opt)
    assert r["image"].shape == (1, 256, 3) and r["weights_sum"].shape == (256,)
    live = g["mask"]
    assert torch.equal(r["mask"][0].cpu(), live)                               # ray bookkeeping: bit-exact
    assert torch.allclose(r["image"][0].cpu(), g["image"], atol=1e-5)
    assert torch.allclose(r["weights_sum"].cpu(), g["weights_sum"], atol=1e-5)
    assert torch.allclose(r["depth"][0].cpu()[live], g["depth"][live], atol=1e-5)
    assert bool(torch.isnan(r["depth"][0, 5]))                                  # miss ray: 0 * NaN as the reference
    loss = (r["image"][0] * g["g_image"].to(DEV)).sum() + (r["weights_sum"] * g["g_ws"].to(DEV)).sum()
    loss.backward()
    # end to end the fine sample positions differ by ~1e-5 (cdf rounding) and the finest table levels turn that into
    # ~1 % weight changes, so the end-to-end gradient check is loose; the isolated check below is tight.
    for k, ref in g["grad_mlp"].items():
        got = dict(net.sigma_net.named_parameters())[k].grad.cpu()
        assert (got - ref).norm() <= 2e-2 * ref.norm() + 1e-7, k
    ge = net.encoder.embeddings.grad.cpu()
    assert abs(ge.norm() - g["grad_table_norm"]) <= 2e-2 * g["grad_table_norm"]
    lvl = torch.stack([ge[p["encoder.offsets"][l]:p["encoder.offsets"][l + 1]].abs().sum() for l in range(16)])
    assert torch.allclose(lvl, g["grad_table_level_abs"], rtol=2e-2)
    # eval: deterministic sampling, white background (render_batched route)
    net.eval()
    opt_small = dict(opt, max_ray_batch=100)                                    # 3 chunks: 100 + 100 + 56 rays
    re = net.render_batched(g["rays_o"][None].to(DEV), g["rays_d"][None].to(DEV), batched=True, perturb=False, bg_color=1,
                            ambient_ratio=1.0, shading='albedo', force_all_rays=True, **opt_small)
    assert torch.allclose(re["image"][0].cpu(), g["eval_image"], atol=1e-5)
    assert torch.allclose(re["weights_sum"][0].cpu(), g["eval_weights_sum"], atol=1e-5)


def test_render_backward_isolated_tight(golden):
    """C-ABI backward fed with the ORACLE's sorted samples: isolates the gradient kernels from the
    sample-position sensitivity.  Upstream d(sigma), d(rgb) and the last layer agree to 1e-5; deeper
    gradients differ by ReLU-mask flips of pre-activations within fp32 rounding of zero (a flip rate eps gives a
    relative L2 difference ~sqrt(eps); measured 4e-4 .. 4e-3 on the MI355X), so the bound is 1e-2
    in norm together with a cosine bound."""
    from sparsefusion_amd import _lib
    from sparsefusion_amd.nerf.renderer import _FieldHandle
    g = golden["teacher"]
    p = params_from_cfg(g["cfg"])
    pl = grad_leaf(p)
    ref = ngp_ref.render_run(pl, g["rays_o"], g["rays_d"], u_coarse=g["u_coarse"], u_fine=g["u_fine"], bg_color=0.0,
                             training=True, return_aux=True)
    ref["sigma_sorted"].retain_grad()
    ((ref["image"] * g["g_image"]).sum() + (ref["weights_sum"] * g["g_ws"]).sum()).backward()
    net = _net(p)
    h = _FieldHandle(net)
    params = [t.detach().contiguous() for t in net._field_params()]
    grads = [torch.zeros_like(t) for t in params]
    f = h.struct(params)
    gs = _lib.SfNgpFieldGrad()
    (gs.g_embeddings, gs.g_w0, gs.g_b0, gs.g_w1, gs.g_b1, gs.g_w2, gs.g_b2) = (t.data_ptr() for t in grads)
    N, T = 256, 64
    d = lambda t: t.detach().contiguous().to(DEV)
    o, dd, aabb = d(g["rays_o"]), d(g["rays_d"]), d(p["aabb_train"])
    nears, fars, zs, ss, rs = d(ref["nears"]), d(ref["fars"]), d(ref["z_sorted"]), d(ref["sigma_sorted"]), d(ref["rgb_sorted"])
    gi, gw = d(g["g_image"]), d(g["g_ws"])
    lib = _lib.lib()
    wb = lib.sf_ngp_render_workspace_bytes(N, T)
    work = torch.empty(wb // 4, device=DEV)
    rc = lib.sf_ngp_render_backward(C.byref(f), C.byref(gs), _lib.ptr(o), _lib.ptr(dd), _lib.ptr(aabb), N, T, _lib.ptr(nears),
                                    _lib.ptr(fars), _lib.ptr(zs), _lib.ptr(ss), _lib.ptr(rs), 0.0, _lib.ptr(gi), _lib.ptr(gw),
                                    0, None, _lib.ptr(work), wb, _lib.stream_ptr())      # no field cache: the re-gather path
    _lib.check(rc)
    torch.cuda.synchronize()
    names = ["encoder.embeddings"] + [f"sigma_net.net.{i}.{w}" for i in range(3) for w in ("weight", "bias")]
    for n, got in zip(names, grads):
        want = pl[n].grad
        rel = ((got.cpu() - want).norm() / want.norm()).item()
        cos = torch.nn.functional.cosine_similarity(got.cpu().flatten(), want.flatten(), dim=0).item()
        tight = n.startswith("sigma_net.net.2")
        assert rel < (2e-5 if tight else 1e-2) and cos > 0.99995, (n, rel, cos)
    dsig = work[:N * 2 * T].view(N, 2 * T).cpu()
    assert ((dsig - ref["sigma_sorted"].grad).norm() / ref["sigma_sorted"].grad.norm()).item() < 1e-5


def test_render_full_size_properties(golden):
    """BASELINE size (128x128 rays): size-independent properties + determinism + linearity of backward."""
    p = params_from_cfg(golden["teacher"]["cfg"])
    net = _net(p).train()
    o, d = ngp_ref.circle_rays(128, view=7)
    o, d = o.to(DEV), d.to(DEV)
    N = o.shape[0]
    assert N == 16384
    g = torch.Generator().manual_seed(11)
    noise = dict(u_coarse=torch.rand(N, 64, generator=g).to(DEV), u_fine=torch.rand(N, 64, generator=g).to(DEV))
    kw = dict(staged=False, perturb=True, bg_color=0, shading='albedo', noise=noise, **vars(net.opt))

    def fwd_bwd(gi, gw):
        net.zero_grad()
        r = net.render(o[None], d[None], **kw)
        ((r["image"][0] * gi).sum() + (r["weights_sum"] * gw).sum()).backward()
        return r, [q.grad.clone() for q in net.parameters()]

    g1, w1 = torch.randn(N, 3, generator=g).to(DEV), torch.randn(N, generator=g).to(DEV)
    g2, w2 = torch.randn(N, 3, generator=g).to(DEV), torch.randn(N, generator=g).to(DEV)
    r1, gr1 = fwd_bwd(g1, w1)
    r2, gr2 = fwd_bwd(g2, w2)
    r3, gr3 = fwd_bwd(g1 + g2, w1 + w2)
    assert torch.equal(r1["image"], r2["image"]) and torch.equal(r1["weights_sum"], r3["weights_sum"])   # deterministic fwd
    ws = r1["weights_sum"]
    assert bool(torch.isfinite(r1["image"]).all()) and float(ws.min()) >= 0 and float(ws.max()) <= 1 + 1e-5
    assert 0.05 < float(ws.mean()) < 0.95
    for a, b, c in zip(gr1, gr2, gr3):                                           # backward is linear in the upstream grads
        assert ((a + b) - c).norm() <= 1e-3 * c.norm() + 1e-6
    # PSNR against the oracle render of a 64x64 view (>= 50 dB)
    o2, d2 = ngp_ref.circle_rays(64, view=20)
    n2 = o2.shape[0]
    uc, uf = torch.rand(n2, 64, generator=g), torch.rand(n2, 64, generator=g)
    with torch.no_grad():
        ref = ngp_ref.render_run(p, o2, d2, u_coarse=uc, u_fine=uf, bg_color=0.0, training=True)
        got = net.render(o2[None].to(DEV), d2[None].to(DEV), staged=False, perturb=True, bg_color=0, shading='albedo',
                         noise=dict(u_coarse=uc.to(DEV), u_fine=uf.to(DEV)), **vars(net.opt))
    assert psnr(got["image"][0].cpu(), ref["image"]) > 50.0


def _table_gradient_vs_grid_encoder(net, n_side, view, seed, tol):
    """The table gradient of the fused render backward against an independent scatter of the SAME per-level feature gradients: the
    reference-ABI grid-encoder backward (csrc/gridencoder.hip, one device atomic per corner and channel =
    external/gridencoder/src/gridencoder.cu:203-262), level by level."""
    from sparsefusion_amd import _lib
    o, d = ngp_ref.circle_rays(n_side, view=view)
    o, d = o.to(DEV), d.to(DEV)
    N, T = o.shape[0], 64
    g = torch.Generator().manual_seed(seed)
    noise = dict(u_coarse=torch.rand(N, T, generator=g).to(DEV), u_fine=torch.rand(N, T, generator=g).to(DEV))
    r = net.render(o[None], d[None], staged=False, perturb=True, bg_color=0, shading='albedo', noise=noise, **vars(net.opt))
    fn = r["image"].grad_fn
    while type(fn).__name__ != "_RenderFnBackward":
        fn = fn.next_functions[0][0]
    rays_o, rays_d, aabb, nears, fars, z_s, sig_s, rgb_s, *rest = fn.saved_tensors
    cache = rest.pop(0) if fn.has_cache else None
    params = rest
    grads = [torch.zeros_like(t) for t in params]
    gs = _lib.SfNgpFieldGrad()
    (gs.g_embeddings, gs.g_w0, gs.g_b0, gs.g_w1, gs.g_b1, gs.g_w2, gs.g_b2) = (t.data_ptr() for t in grads)
    f = fn.handle.struct(params)
    lib = _lib.lib()
    wb = lib.sf_ngp_render_workspace_bytes(N, T)
    work = torch.empty(wb // 4, device=DEV)
    gi, gw = torch.randn(N, 3, generator=g).to(DEV), torch.randn(N, generator=g).to(DEV)
    rc = lib.sf_ngp_render_backward(C.byref(f), C.byref(gs), _lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(aabb), N, T, _lib.ptr(nears),
                                    _lib.ptr(fars), _lib.ptr(z_s), _lib.ptr(sig_s), _lib.ptr(rgb_s), 0.0, _lib.ptr(gi), _lib.ptr(gw),
                                    n_side, _lib.ptr(cache), _lib.ptr(work), wb, _lib.stream_ptr())
    _lib.check(rc)
    torch.cuda.synchronize()
    M = N * 2 * T
    dfeat = work[4 * M:4 * M + 32 * M].view(16, M, 2)                          # level-major d(features) the field backward left
    x = rays_o[:, None, :] + rays_d[:, None, :] * z_s[:, :, None]
    x = torch.minimum(torch.maximum(x, aabb[:3]), aabb[3:]).reshape(M, 3)     # ngp_point: clipped to the box
    enc = net.encoder
    enc.embeddings.grad = None
    enc(x, bound=net.bound).backward(dfeat.permute(1, 0, 2).reshape(M, 32))
    want, got = enc.embeddings.grad, grads[0]
    offs = enc.offsets.tolist()
    assert float(want.abs().max()) > 0
    for l in range(16):                                                       # every level on its own: dense, wrapped, z-dropped
        a, b = got[offs[l]:offs[l + 1]], want[offs[l]:offs[l + 1]]
        assert float(b.norm()) > 0 and float((a - b).norm() / b.norm()) < tol, (l, float((a - b).norm() / b.norm()))


def test_binned_table_gradient_equals_per_corner_atomics_at_full_size(golden):
    """r04: the table gradient of the fused backward (csrc/ngp_scatter_bin.h: wrapped levels binned by table slice and reduced in
    LDS doubles, two ray chunks sharing the bins; dense levels through the LDS cache) at the BASELINE size against the per-corner
    atomics of the grid-encoder backward.  Equal up to fp32 summation order (measured <= 1.1e-4: the per-corner path adds ~2000 fp32
    contributions per row in arrival order, the binned one in doubles)."""
    net = _net(params_from_cfg(golden["teacher"]["cfg"])).train()
    _table_gradient_vs_grid_encoder(net, 128, 5, 3, 5e-4)


def test_table_gradient_with_unequal_ray_chunks(golden):
    """r05: 100 x 100 = 10 000 rays have no equal split into chunks of a multiple of 256 rays: the backward runs a chunk of 8192 and one
    of 1808 rays (ngp_bwd_plan), bins sized for 8192 -- same comparison as above."""
    net = _net(params_from_cfg(golden["teacher"]["cfg"])).train()
    _table_gradient_vs_grid_encoder(net, 100, 4, 5, 5e-4)


def test_table_gradient_of_a_large_hash_map_keeps_the_atomic_scatter(golden):
    """A field whose levels exceed 2^19 rows (log2_hashmap_size = 20, `hash` grid type) is outside the binned scatter's bucket table:
    sf_ngp_render_backward must take k_ngp_scatter + k_ngp_scatter_fine (the r02 / r03 path) for it -- same comparison."""
    from sparsefusion_amd.gridencoder import GridEncoder
    net = _net(params_from_cfg(golden["teacher"]["cfg"])).train()
    enc = GridEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=20,
                      desired_resolution=2048 * net.bound, gridtype='hash', align_corners=False).to(DEV)
    with torch.no_grad():
        enc.embeddings.uniform_(-0.5, 0.5, generator=None)
    net.encoder, net._handle = enc, None
    _table_gradient_vs_grid_encoder(net, 48, 9, 7, 5e-4)


def test_field_cache_equals_regather(golden, monkeypatch):
    """r03: the forward's field cache (features of every sample + sort permutation, sf_ngp_render_forward's field_cache) against
    the backward that re-gathers the features: same render, gradients equal to the last bits (the cached features ARE the values
    the forward multiplied; the re-gather recomputes them with the same arithmetic)."""
    from sparsefusion_amd.nerf import renderer as R
    p = params_from_cfg(golden["teacher"]["cfg"])
    net = _net(p).train()
    o, d = ngp_ref.circle_rays(48, view=3)
    o, d = o.to(DEV), d.to(DEV)
    N = o.shape[0]
    g = torch.Generator().manual_seed(5)
    noise = dict(u_coarse=torch.rand(N, 64, generator=g).to(DEV), u_fine=torch.rand(N, 64, generator=g).to(DEV))
    gi, gw = torch.randn(N, 3, generator=g).to(DEV), torch.randn(N, generator=g).to(DEV)
    kw = dict(staged=False, perturb=True, bg_color=0, shading='albedo', noise=noise, **vars(net.opt))
    out = []
    for cache in (True, False, True):
        monkeypatch.setattr(R, "_FEAT_CACHE", cache)
        net.zero_grad()
        r = net.render(o[None], d[None], **kw)
        ((r["image"][0] * gi).sum() + (r["weights_sum"] * gw).sum()).backward()
        out.append((r["image"].clone(), [q.grad.clone() for q in net.parameters()]))
    assert torch.equal(out[0][0], out[1][0])
    for a, b, c in zip(out[0][1], out[1][1], out[2][1]):
        assert a.abs().max() > 0
        assert (a - b).norm() <= 1e-5 * b.norm() + 1e-9, float((a - b).norm() / b.norm())
        assert (a - c).norm() <= 1e-5 * c.norm() + 1e-9                             # (atomics in the scatters: equal to summation order)


def test_render_batched_eval_sizes(golden):
    """BASELINE config 4 renders 512^2 at evaluation time through render_batched (renderer_df.py:681-717): chunks of
    max_ray_batch rays must reproduce the single-launch result bit for bit (rays are independent; eval sampling is
    deterministic), also for a ragged last chunk, and stay finite at 262 144 rays."""
    p = params_from_cfg(golden["teacher"]["cfg"])
    net = _net(p).eval()
    kw = dict(perturb=False, bg_color=1, shading='albedo', **{k: v for k, v in vars(net.opt).items() if k != 'max_ray_batch'})
    o, d = ngp_ref.circle_rays(256, view=5)
    o, d = o[None].to(DEV), d[None].to(DEV)
    whole = net.render_batched(o, d, batched=False, **kw)
    chunks = net.render_batched(o, d, batched=True, max_ray_batch=128 * 128, **kw)
    ragged = net.render_batched(o, d, batched=True, max_ray_batch=10007, **kw)
    for k in ("image", "depth", "weights_sum"):
        a = whole[k].reshape(chunks[k].shape)
        assert torch.equal(torch.nan_to_num(a), torch.nan_to_num(chunks[k])), k
        assert torch.equal(torch.nan_to_num(a), torch.nan_to_num(ragged[k])), k
    assert not whole["image"].requires_grad
    o5, d5 = ngp_ref.circle_rays(512, view=9)
    big = net.render_batched(o5[None].to(DEV), d5[None].to(DEV), batched=True, max_ray_batch=128 * 128, **kw)
    assert big["image"].shape == (1, 512 * 512, 3) and bool(torch.isfinite(big["image"]).all())
    assert 0.0 <= float(big["weights_sum"].min()) and float(big["weights_sum"].max()) <= 1 + 1e-5


def test_render_rng_stream_matches_reference_order():
    """With no injected noise `run` draws randn(3), rand(N,T), rand(N,T) in the reference's order."""
    from sparsefusion_amd.nerf import NeRFNetwork, get_default_torch_ngp_opt
    torch.manual_seed(0)
    net = NeRFNetwork(get_default_torch_ngp_opt()).to(DEV).train()
    net.encoder.embeddings.data.uniform_(-0.5, 0.5)
    o, d = ngp_ref.circle_rays(16, view=1)
    o, d = o.to(DEV), d.to(DEV)
    torch.manual_seed(42)
    torch.randn(3, device=DEV); uc = torch.rand(256, 64, device=DEV); uf = torch.rand(256, 64, device=DEV)
    with torch.no_grad():
        a = net.render(o[None], d[None], perturb=True, bg_color=0, shading='albedo', noise=dict(u_coarse=uc, u_fine=uf),
                       **vars(net.opt))
        torch.manual_seed(42)
        b = net.render(o[None], d[None], perturb=True, bg_color=0, shading='albedo', **vars(net.opt))
    assert torch.equal(a["image"], b["image"])


# ----------------------------------------------------------------------------- sample bookkeeping of the default path (R2, r05)
def _forward_with_bookkeeping(net, p, o, d, uc, uf):
    """sf_ngp_render_forward through the C ABI with its workspace and field cache read back: coarse z (workspace [0, NT)), fine z
    (workspace [5 NT, 6 NT)), z_sorted, the sort permutation (behind the two feature blocks of the cache), near / far."""
    from sparsefusion_amd import _lib
    from sparsefusion_amd.nerf.renderer import _FieldHandle
    N, T = o.shape[0], 64
    h = _FieldHandle(net)
    params = [t.detach().contiguous() for t in net._field_params()]
    f = h.struct(params)
    lib = _lib.lib()
    f32 = dict(dtype=torch.float32, device=DEV)
    od, dd, aabb = o.to(DEV).contiguous(), d.to(DEV).contiguous(), p["aabb_train"].to(DEV).contiguous()
    lin = net._table(T, torch.device(DEV))[0]                      # the table the renderer itself passes (host linspace, copied)
    assert torch.equal(lin.cpu(), torch.linspace(0.0, 1.0, T))
    ucd, ufd = uc.to(DEV).contiguous(), uf.to(DEV).contiguous()
    nears, fars = torch.empty(N, **f32), torch.empty(N, **f32)
    z_s, sig_s, rgb_s = torch.empty(N, 2 * T, **f32), torch.empty(N, 2 * T, **f32), torch.empty(N, 2 * T, 3, **f32)
    image, depth, ws = torch.empty(N, 3, **f32), torch.empty(N, **f32), torch.empty(N, **f32)
    wb = lib.sf_ngp_render_forward_workspace_bytes(N, T)
    work = torch.empty(wb // 4, **f32)
    cache = torch.empty(lib.sf_ngp_render_cache_bytes(N, T) // 4, **f32)
    rc = lib.sf_ngp_render_forward(C.byref(f), _lib.ptr(od), _lib.ptr(dd), _lib.ptr(aabb), N, T, 0.1, _lib.ptr(lin), _lib.ptr(ucd),
                                   _lib.ptr(ufd), T, 0.0, _lib.ptr(nears), _lib.ptr(fars), _lib.ptr(z_s), _lib.ptr(sig_s), _lib.ptr(rgb_s),
                                   _lib.ptr(image), _lib.ptr(depth), _lib.ptr(ws), _lib.ptr(cache), _lib.ptr(work), wb, _lib.stream_ptr())
    _lib.check(rc)
    torch.cuda.synchronize()
    NT = N * T
    perm = cache[2 * NT * 32:2 * NT * 32 + 2 * NT].view(torch.int32).view(N, 2 * T).cpu().long()
    return dict(nears=nears.cpu(), fars=fars.cpu(), z_coarse=work[:NT].view(N, T).cpu(), z_fine=work[5 * NT:6 * NT].view(N, T).cpu(),
                z_sorted=z_s.cpu(), perm=perm)


def _finest_cells(p, o, d, z):
    """Cell index triple of every sample on the FINEST level (resolution 2048 * bound per unit box: the level a position error
    reaches first), computed in double from (ray, z) the way ngp_point + the encoder do: clip to the box, map to [0, 1], scale."""
    x = o[:, None, :].double() + d[:, None, :].double() * z[:, :, None].double()
    x = torch.minimum(torch.maximum(x, p["aabb_train"][:3].double()), p["aabb_train"][3:].double())
    u = (x + BOUND) / (2 * BOUND)
    scale = 16.0 * ngp_ref.per_level_scale(BOUND) ** 15 - 1.0
    return torch.floor(u * scale + 0.5).long()


@pytest.mark.parametrize("case", ["teacher_256_rays", "full_16384_rays"])
def test_render_sample_bookkeeping_vs_oracle(golden, case):
    """north_star: "bit-exact for ray-index bookkeeping"; SURVEY 8(d): "indices / per-ray (near, far) / sample ordering: bit-exact".
    The reference's bookkeeping on the cuda_ray=False path is z_vals -> sample_pdf -> torch.sort -> z_index gathers
    (external/nerf/renderer_df.py:356-412, :15-49).  What the HIP forward leaves in device buffers is compared with the oracle's
    `render_run(return_aux=True)` (itself bit-identical to the reference's own `run` on CPU, tests/golden/ngp_render.pt):
      * per-ray (near, far), the mask and the T coarse sample depths: BIT-EXACT;
      * the sort: z_sorted is bit-exactly the ascending order of the HIP path's own cat([coarse, fine]) and the permutation the field
        cache keeps is the stable argsort of it (coarse before fine on ties = torch.sort of the concatenation where values are
        distinct); on every ray whose fine depths are bit-equal to the oracle's the permutation EQUALS the oracle's z_index;
      * the T fine depths are NOT bit-exact and cannot be: they are an inverse-CDF draw whose pdf is weights / torch.sum(weights), and
        torch's CPU float reduction order (a vectorised cascade, ISA dependent) differs from the reference's CUDA order and from ours
        (sequential, in double, rounded once); a last-place difference of the total moves every cdf entry, and the interpolation
        divides by cdf_hi - cdf_lo (>= 1e-5).  Measured on MI355X (r05): 21-24 % of the fine depths bit-equal, the rest a few units in
        the last place away: max |dz| = 1.5e-6 / 7.6e-6 of (far - near) at 256 / 16 384 rays; 0.15 % / 0.13 % of the fine samples
        land in a different cell of the FINEST grid level (24 of 16 320 / 1 318 of 1 048 576).  Stated bounds: |dz| <= 5e-5 * (far - near)
        on every sample, fewer than 0.5 % finest-level cell changes.
    (r05 found the coarse depths 1 ulp off on 19 % of the samples -- hipcc had fused the multiply-adds of the position arithmetic,
    csrc/ngp_device.h SF_MUL / SF_ADD -- and the renderer's linspace table built by the device kernel; both fixed with this test.)"""
    g = golden["teacher"]
    p = params_from_cfg(g["cfg"])
    net = _net(p).train()
    if case == "teacher_256_rays":
        o, d, uc, uf = g["rays_o"], g["rays_d"], g["u_coarse"], g["u_fine"]
    else:
        o, d = ngp_ref.circle_rays(128, view=7)
        gen = torch.Generator().manual_seed(23)
        uc, uf = torch.rand(o.shape[0], 64, generator=gen), torch.rand(o.shape[0], 64, generator=gen)
    N, T = o.shape[0], 64
    with torch.no_grad():
        ref = ngp_ref.render_run(p, o, d, u_coarse=uc, u_fine=uf, bg_color=0.0, training=True, return_aux=True)
    got = _forward_with_bookkeeping(net, p, o, d, uc, uf)
    live = ref["mask"]
    # (1) per-ray bookkeeping and the coarse samples: bit-exact
    assert torch.equal(got["nears"], ref["nears"]) and torch.equal(got["fars"], ref["fars"])
    assert torch.equal(got["nears"] < got["fars"], live)
    zc_bad = got["z_coarse"][live] != ref["z_coarse"][live]
    assert not bool(zc_bad.any()), ("coarse sample depths must be bit-exact", int(zc_bad.sum()),
                                    float((got["z_coarse"][live] - ref["z_coarse"][live]).abs().max()))
    # (2) the sort of the HIP path's own samples: exact, stable
    cat = torch.cat([got["z_coarse"], got["z_fine"]], dim=1)
    want_sorted, want_perm = torch.sort(cat, dim=1, stable=True)
    assert torch.equal(got["z_sorted"][live], want_sorted[live]), "z_sorted must be the ascending order of cat([coarse, fine])"
    assert torch.equal(got["perm"][live], want_perm[live]), "the cached permutation must be the stable argsort (coarse first on ties)"
    assert torch.equal(torch.gather(cat, 1, got["perm"])[live], got["z_sorted"][live])
    # (3) fine samples against the oracle
    zf, zr = got["z_fine"][live], ref["z_fine"][live]
    span = (ref["fars"] - ref["nears"])[live][:, None]
    same = zf == zr
    frac_equal = same.float().mean().item()
    worst = ((zf - zr).abs() / span).max().item()
    cells_hip = _finest_cells(p, o[live], d[live], zf)
    cells_ref = _finest_cells(p, o[live], d[live], zr)
    flips = (cells_hip != cells_ref).any(dim=-1)
    flip_frac = flips.float().mean().item()
    print(f"\n[bookkeeping {case}] rays {N} live {int(live.sum())}: fine z bit-equal {100 * frac_equal:.2f} %, max |dz| / (far - near) "
          f"{worst:.2e}, finest-level cell flips {int(flips.sum())} of {flips.numel()} fine samples ({100 * flip_frac:.3f} %)")
    assert frac_equal >= 0.05            # (informational: a collapse to ~0 would mean a systematic difference, not rounding)
    assert worst <= 5e-5
    assert flip_frac < 0.005
    # (4) rays whose fine depths are all bit-equal: the permutation IS the oracle's z_index (torch.sort of the concatenation),
    # checked where the oracle's sorted depths are distinct (torch.sort is not stable by default)
    ray_same = same.all(dim=1)
    ref_sorted, ref_index = torch.sort(torch.cat([ref["z_coarse"], ref["z_fine"]], dim=1), dim=1)
    assert torch.equal(ref_sorted[live], ref["z_sorted"][live])
    distinct = torch.ones_like(ref_sorted[live], dtype=torch.bool)
    distinct[:, 1:] &= ref_sorted[live][:, 1:] != ref_sorted[live][:, :-1]
    distinct[:, :-1] &= ref_sorted[live][:, 1:] != ref_sorted[live][:, :-1]
    sel = ray_same[:, None] & distinct
    # (few rays qualify -- all 64 fine depths bit-equal --; the comparison is exact on those, and (2) above pins the sort itself on all)
    print(f"[bookkeeping {case}] rays with all fine depths bit-equal: {int(ray_same.sum())}")
    assert torch.equal(got["perm"][live][sel], ref_index[live][sel])
    assert torch.equal(got["z_sorted"][live][ray_same], ref["z_sorted"][live][ray_same])
    # every ray: the permutation restricted to the COARSE samples is the oracle's (their depths are bit-equal and ascending), and the
    # rank of a fine sample among all samples differs from the oracle's only where two depths are closer than the fine-depth bound
    rank_hip = torch.argsort(got["perm"][live], dim=1)           # rank_hip[n, j] = sorted position of sample j of cat([coarse, fine])
    rank_ref = torch.argsort(ref_index[live], dim=1)
    moved = rank_hip != rank_ref
    zcat_ref = torch.cat([ref["z_coarse"], ref["z_fine"]], dim=1)[live]
    gap = torch.full_like(zcat_ref, float("inf"))
    srt = ref_sorted[live]
    d = srt[:, 1:] - srt[:, :-1]
    near_gap = torch.minimum(torch.cat([d, torch.full_like(d[:, :1], float("inf"))], 1), torch.cat([torch.full_like(d[:, :1], float("inf")), d], 1))
    gap.scatter_(1, ref_index[live], near_gap)                   # distance of every sample to its nearest neighbour in depth (oracle)
    print(f"[bookkeeping {case}] samples whose sorted position differs from the oracle's: {int(moved.sum())} of {moved.numel()}")
    assert bool((gap[moved] <= 2 * 5e-5 * span.expand_as(gap)[moved]).all()), "a sample changed its sorted position without a near-tie in depth"
    assert moved.float().mean().item() < 0.005

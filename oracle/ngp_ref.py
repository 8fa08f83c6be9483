"""torch-CPU restatement of the NGP field and the hierarchical volume renderer -- TEST
INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows the reference literally, including its redundancies (the field is evaluated 256x per
ray: coarse, fine, then all 128 again for colour), so that autograd reproduces the
reference's gradients:
  MLP / common_forward / gaussian   external/nerf/network_grid.py:14-33, :69-88
  trunc_exp                         external/ngp_activation.py:10-23
  sample_pdf                        external/nerf/renderer_df.py:15-49
  run                               external/nerf/renderer_df.py:310-468
All randomness is injected (`light noise` is irrelevant for shading='albedo'):
  u_coarse [N,T] in [0,1)  <- torch.rand(z_vals.shape)            renderer_df.py:363
  u_fine   [N,t] in [0,1)  <- torch.rand(cdf.shape[:-1]+[n])      renderer_df.py:31
Pinned against the reference's own NeRFNetwork.render (run verbatim on CPU through
oracle/ref_loader.py) by tests/golden/ngp_render_*.pt.
"""
import numpy as np
import torch
import torch.nn.functional as F

from .ngp_native import GridEncodeCPU, near_far_from_aabb as _near_far_native

# Geometry of get_encoder('tiledgrid', input_dim=3, log2_hashmap_size=16, desired_resolution=2048*bound)
# (network_grid.py:50, ngp_encoder.py:50-79 defaults num_levels=16, level_dim=2, base_resolution=16)
NUM_LEVELS, LEVEL_DIM, BASE_RES, LOG2_HASHMAP = 16, 2, 16, 16


def per_level_scale(bound):
    return float(np.exp2(np.log2(2048 * bound / BASE_RES) / (NUM_LEVELS - 1)))


def level_offsets(bound, input_dim=3, align_corners=False):
    s = per_level_scale(bound)
    offs, off = [], 0
    for i in range(NUM_LEVELS):
        res = int(np.ceil(BASE_RES * s ** i))
        n = min(2 ** LOG2_HASHMAP, (res if align_corners else res + 1) ** input_dim)
        n = int(np.ceil(n / 8) * 8)
        offs.append(off)
        off += n
    offs.append(off)
    return torch.tensor(offs, dtype=torch.int32)


def init_params(bound=4, seed=0, table_std=1e-4, sigma_bias=None):
    """Parameters under the reference's state-dict names.  Default init = the reference's
    (U(-1e-4,1e-4) table, nn.Linear default MLP); `table_std` / `sigma_bias` give the
    'trained-like' and 'teacher' variants of SURVEY.md section 8(d)."""
    g = torch.Generator().manual_seed(seed)
    offs = level_offsets(bound)
    p = {"encoder.offsets": offs,
         "encoder.embeddings": (torch.rand(int(offs[-1]), LEVEL_DIM, generator=g) * 2 - 1) * table_std}
    dims = [(64, 32), (64, 64), (4, 64)]
    for i, (o, k) in enumerate(dims):
        lim = 1.0 / np.sqrt(k)
        p[f"sigma_net.net.{i}.weight"] = (torch.rand(o, k, generator=g) * 2 - 1) * lim
        p[f"sigma_net.net.{i}.bias"] = (torch.rand(o, generator=g) * 2 - 1) * lim
    if sigma_bias is not None:
        p["sigma_net.net.2.bias"][0] = sigma_bias
    p["aabb_train"] = torch.tensor([-bound] * 3 + [bound] * 3, dtype=torch.float32)
    p["aabb_infer"] = p["aabb_train"].clone()
    return p


class _TruncExp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * torch.exp(x.clamp(-15, 15))


def encode(params, x, bound):
    """GridEncoder.forward (grid.py:138-154): map to [0,1], encode, [B, 32]."""
    inputs = (x + bound) / (2 * bound)
    return GridEncodeCPU.apply(inputs.view(-1, 3), params["encoder.embeddings"], params["encoder.offsets"],
                               per_level_scale(bound), BASE_RES, False, 1, False)


def common_forward(params, x, bound):
    h = encode(params, x, bound)
    for i in range(3):
        h = F.linear(h, params[f"sigma_net.net.{i}.weight"], params[f"sigma_net.net.{i}.bias"])
        if i != 2:
            h = F.relu(h)
    d = (x ** 2).sum(-1)
    g = 5 * torch.exp(-d / (2 * 0.2 ** 2))
    sigma = _TruncExp.apply(h[..., 0] + g)
    albedo = torch.sigmoid(h[..., 1:])
    return sigma, albedo


def near_far(rays_o, rays_d, aabb, min_near):
    n = rays_o.shape[0]
    nears, fars = torch.empty(n), torch.empty(n)
    _near_far_native(rays_o.contiguous(), rays_d.contiguous(), aabb.contiguous(), n, min_near, nears, fars)
    return nears, fars


def sample_pdf(bins, weights, n_samples, u):
    """renderer_df.py:15-49 with the uniform draws `u` [B, n_samples] injected (det=False) or
    u=None for the deterministic mid-point grid (det=True)."""
    weights = weights + 1e-5
    pdf = weights / torch.sum(weights, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    if u is None:
        u = torch.linspace(0. + 0.5 / n_samples, 1. - 0.5 / n_samples, steps=n_samples)
        u = u.expand(list(cdf.shape[:-1]) + [n_samples])
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.max(torch.zeros_like(inds - 1), inds - 1)
    above = torch.min((cdf.shape[-1] - 1) * torch.ones_like(inds), inds)
    cdf_lo, cdf_hi = torch.gather(cdf, 1, below), torch.gather(cdf, 1, above)
    bin_lo, bin_hi = torch.gather(bins, 1, below), torch.gather(bins, 1, above)
    denom = cdf_hi - cdf_lo
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - cdf_lo) / denom
    return bin_lo + t * (bin_hi - bin_lo)


def render_run(params, rays_o, rays_d, *, bound=4, min_near=0.1, num_steps=64, upsample_steps=64,
               u_coarse=None, u_fine=None, bg_color=0.0, training=True, return_aux=False):
    """NeRFRenderer.run (renderer_df.py:310-468) for shading='albedo', bg_radius=0.
    rays_o/rays_d [N,3].  u_coarse=None <=> perturb=False; u_fine=None <=> det sampling."""
    aabb = params["aabb_train"] if training else params["aabb_infer"]
    N = rays_o.shape[0]
    nears, fars = near_far(rays_o, rays_d, aabb, min_near)
    nears = nears.unsqueeze(-1)
    fars = fars.unsqueeze(-1)

    z_vals = torch.linspace(0.0, 1.0, num_steps).unsqueeze(0).expand((N, num_steps))
    z_vals = nears + (fars - nears) * z_vals
    sample_dist = (fars - nears) / num_steps
    if u_coarse is not None:
        z_vals = z_vals + (u_coarse - 0.5) * sample_dist

    xyzs = rays_o.unsqueeze(-2) + rays_d.unsqueeze(-2) * z_vals.unsqueeze(-1)
    xyzs = torch.min(torch.max(xyzs, aabb[:3]), aabb[3:])
    sigma_c, _ = common_forward(params, xyzs.reshape(-1, 3), bound)
    sigma_c = sigma_c.view(N, num_steps)

    with torch.no_grad():
        deltas = z_vals[..., 1:] - z_vals[..., :-1]
        deltas = torch.cat([deltas, sample_dist * torch.ones_like(deltas[..., :1])], dim=-1)
        alphas = 1 - torch.exp(-deltas * sigma_c)
        alphas_shifted = torch.cat([torch.ones_like(alphas[..., :1]), 1 - alphas + 1e-15], dim=-1)
        weights = alphas * torch.cumprod(alphas_shifted, dim=-1)[..., :-1]
        z_vals_mid = (z_vals[..., :-1] + 0.5 * deltas[..., :-1])
        new_z_vals = sample_pdf(z_vals_mid, weights[:, 1:-1], upsample_steps, u_fine).detach()
        new_xyzs = rays_o.unsqueeze(-2) + rays_d.unsqueeze(-2) * new_z_vals.unsqueeze(-1)
        new_xyzs = torch.min(torch.max(new_xyzs, aabb[:3]), aabb[3:])

    sigma_f, _ = common_forward(params, new_xyzs.reshape(-1, 3), bound)
    sigma_f = sigma_f.view(N, upsample_steps)

    z_all = torch.cat([z_vals, new_z_vals], dim=1)
    z_all, z_index = torch.sort(z_all, dim=1)
    xyz_all = torch.cat([xyzs, new_xyzs], dim=1)
    xyz_all = torch.gather(xyz_all, dim=1, index=z_index.unsqueeze(-1).expand_as(xyz_all))
    sigma_all = torch.gather(torch.cat([sigma_c, sigma_f], dim=1), dim=1, index=z_index)

    deltas = z_all[..., 1:] - z_all[..., :-1]
    deltas = torch.cat([deltas, sample_dist * torch.ones_like(deltas[..., :1])], dim=-1)
    alphas = 1 - torch.exp(-deltas * sigma_all)
    alphas_shifted = torch.cat([torch.ones_like(alphas[..., :1]), 1 - alphas + 1e-15], dim=-1)
    weights = alphas * torch.cumprod(alphas_shifted, dim=-1)[..., :-1]

    _, rgbs = common_forward(params, xyz_all.reshape(-1, 3), bound)      # sigma of this pass is discarded
    rgbs = rgbs.view(N, -1, 3)

    weights_sum = weights.sum(dim=-1)
    ori_z_vals = ((z_all - nears) / (fars - nears)).clamp(0, 1)
    depth = torch.sum(weights * ori_z_vals, dim=-1)
    image = torch.sum(weights.unsqueeze(-1) * rgbs, dim=-2)
    image = image + (1 - weights_sum).unsqueeze(-1) * bg_color
    out = {"image": image, "depth": depth, "weights_sum": weights_sum, "mask": (nears < fars).view(-1)}
    if return_aux:
        out.update(nears=nears.view(-1), fars=fars.view(-1), z_sorted=z_all, sigma_sorted=sigma_all, rgb_sorted=rgbs,
                   weights=weights, z_coarse=z_vals, z_fine=new_z_vals)
    return out


def circle_rays(n_side, view, n_views=34, radius=6.0, focal=2.0, unit_dir=False, elevation=0.3):
    """Synthetic pinhole rays: camera on a circle of `radius` around the origin
    looking at it; pytorch3d-style non-unit directions (unit z-step) unless unit_dir."""
    ang = 2 * np.pi * view / n_views
    eye = torch.tensor([radius * np.cos(ang), radius * elevation, radius * np.sin(ang)], dtype=torch.float32)
    fwd = -eye / eye.norm()
    up = torch.tensor([0.0, 1.0, 0.0])
    right = torch.linalg.cross(fwd, up); right = right / right.norm()
    upv = torch.linalg.cross(right, fwd)
    ax = torch.linspace(1 - 1 / n_side, -1 + 1 / n_side, n_side)
    yy, xx = torch.meshgrid(ax, ax, indexing="ij")
    d = fwd[None, None] + (xx[..., None] * right[None, None] + yy[..., None] * upv[None, None]) / focal
    d = d.reshape(-1, 3)
    if unit_dir:
        d = d / d.norm(dim=-1, keepdim=True)
    o = eye[None].expand_as(d).contiguous()
    return o, d.contiguous()

"""Ray utilities with the reference's Python call surface (raymarching/raymarching.py:19-155):
`near_far_from_aabb(rays_o, rays_d, aabb, min_near=0.2) -> (nears, fars)`, `morton3D`,
`morton3D_invert`, `packbits`, and the occupancy-grid path `sph_from_ray`, `march_rays_train`,
`composite_rays_train` (differentiable, :238-290), `march_rays`, `composite_rays` (:160-380)."""
import torch

from . import backend


@torch.no_grad()
def near_far_from_aabb(rays_o, rays_d, aabb, min_near=0.2):
    """Slab test of N rays against aabb=(xmin,ymin,zmin,xmax,ymax,zmax); misses get FLT_MAX twice."""
    o = rays_o.float().contiguous().view(-1, 3)
    d = rays_d.float().contiguous().view(-1, 3)
    n = o.shape[0]
    nears = torch.empty(n, dtype=torch.float32, device=o.device)
    fars = torch.empty(n, dtype=torch.float32, device=o.device)
    backend.near_far_from_aabb(o, d, aabb.float().contiguous(), n, min_near, nears, fars)
    return nears, fars


@torch.no_grad()
def morton3D(coords):
    c = coords.int().contiguous()
    out = torch.empty(c.shape[0], dtype=torch.int32, device=c.device)
    backend.morton3D(c, c.shape[0], out)
    return out


@torch.no_grad()
def morton3D_invert(indices):
    i = indices.int().contiguous()
    out = torch.empty(i.shape[0], 3, dtype=torch.int32, device=i.device)
    backend.morton3D_invert(i, i.shape[0], out)
    return out


@torch.no_grad()
def packbits(grid, thresh, bitfield=None):
    g = grid.float().contiguous()
    n = g.shape[0] * g.shape[1] // 8
    if bitfield is None:
        bitfield = torch.empty(n, dtype=torch.uint8, device=g.device)
    backend.packbits(g, n, thresh, bitfield)
    return bitfield


# ---------------------------------------------------------------------------------------------
# occupancy-grid path (cuda_ray=True): raymarching/raymarching.py:50-77, 160-380
# ---------------------------------------------------------------------------------------------
@torch.no_grad()
def sph_from_ray(rays_o, rays_d, radius):
    """Spherical coordinates (theta, phi in [-1, 1]) of the far intersection with the sphere of `radius`."""
    o = rays_o.float().contiguous().view(-1, 3)
    d = rays_d.float().contiguous().view(-1, 3)
    coords = torch.empty(o.shape[0], 2, dtype=torch.float32, device=o.device)
    backend.sph_from_ray(o, d, radius, o.shape[0], coords)
    return coords


@torch.no_grad()
def march_rays_train(rays_o, rays_d, bound, density_bitfield, C, H, nears, fars, step_counter=None, mean_count=-1,
                     perturb=False, align=-1, force_all_rays=False, dt_gamma=0, max_steps=1024, noises=None):
    """Samples of N rays inside occupied cells (raymarching.py:160-235).  Returns (xyzs [M,3], dirs [M,3],
    deltas [M,2], rays [N,3] = (ray id, point offset, point count)).  `noises` (extra, optional) injects the
    per-ray jitter that the reference draws with torch.rand when `perturb` is set."""
    o = rays_o.float().contiguous().view(-1, 3)
    d = rays_d.float().contiguous().view(-1, 3)
    bitfield = density_bitfield.contiguous()
    N = o.shape[0]
    M = N * max_steps
    if not force_all_rays and mean_count > 0:
        if align > 0:
            mean_count += align - mean_count % align
        M = mean_count
    dev = o.device
    xyzs = torch.zeros(M, 3, dtype=torch.float32, device=dev)
    dirs = torch.zeros(M, 3, dtype=torch.float32, device=dev)
    deltas = torch.zeros(M, 2, dtype=torch.float32, device=dev)
    rays = torch.empty(N, 3, dtype=torch.int32, device=dev)
    if step_counter is None:
        step_counter = torch.zeros(2, dtype=torch.int32, device=dev)
    if noises is None:
        noises = torch.rand(N, dtype=torch.float32, device=dev) if perturb else torch.zeros(N, dtype=torch.float32, device=dev)
    backend.march_rays_train(o, d, bitfield, bound, dt_gamma, max_steps, N, C, H, M, nears.float().contiguous(),
                             fars.float().contiguous(), xyzs, dirs, deltas, rays, step_counter, noises.float().contiguous())
    if force_all_rays or mean_count <= 0:
        m = int(step_counter[0].item())
        if align > 0:
            m += align - m % align
        xyzs, dirs, deltas = xyzs[:m], dirs[:m], deltas[:m]
    return xyzs, dirs, deltas, rays


class _CompositeRaysTrain(torch.autograd.Function):
    """raymarching.py:238-290: differentiable w.r.t. sigmas and rgbs (grad_depth is ignored, as in the reference)."""

    @staticmethod
    def forward(ctx, sigmas, rgbs, deltas, rays, T_thresh=1e-4):
        sigmas, rgbs = sigmas.float().contiguous(), rgbs.float().contiguous()
        deltas = deltas.float().contiguous()
        M, N = sigmas.shape[0], rays.shape[0]
        weights_sum = torch.empty(N, dtype=torch.float32, device=sigmas.device)
        depth = torch.empty(N, dtype=torch.float32, device=sigmas.device)
        image = torch.empty(N, 3, dtype=torch.float32, device=sigmas.device)
        backend.composite_rays_train_forward(sigmas, rgbs, deltas, rays, M, N, T_thresh, weights_sum, depth, image)
        ctx.save_for_backward(sigmas, rgbs, deltas, rays, weights_sum, depth, image)
        ctx.dims = [M, N, T_thresh]
        return weights_sum, depth, image

    @staticmethod
    def backward(ctx, grad_weights_sum, grad_depth, grad_image):
        sigmas, rgbs, deltas, rays, weights_sum, depth, image = ctx.saved_tensors
        M, N, T_thresh = ctx.dims
        grad_sigmas, grad_rgbs = torch.zeros_like(sigmas), torch.zeros_like(rgbs)
        backend.composite_rays_train_backward(grad_weights_sum.float().contiguous(), grad_image.float().contiguous(), sigmas, rgbs,
                                              deltas, rays, weights_sum, image, M, N, T_thresh, grad_sigmas, grad_rgbs)
        return grad_sigmas, grad_rgbs, None, None, None


composite_rays_train = _CompositeRaysTrain.apply


@torch.no_grad()
def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, density_bitfield, C, H, near, far, align=-1,
               perturb=False, dt_gamma=0, max_steps=1024, noises=None):
    """n_step samples for each of the first n_alive rays of `rays_alive` (raymarching.py:296-346)."""
    o = rays_o.float().contiguous().view(-1, 3)
    d = rays_d.float().contiguous().view(-1, 3)
    M = n_alive * n_step
    if align > 0:
        M += align - (M % align)
    dev = o.device
    xyzs = torch.zeros(M, 3, dtype=torch.float32, device=dev)
    dirs = torch.zeros(M, 3, dtype=torch.float32, device=dev)
    deltas = torch.zeros(M, 2, dtype=torch.float32, device=dev)
    if noises is None:
        noises = (torch.rand(n_alive, dtype=torch.float32, device=dev) if perturb
                  else torch.zeros(n_alive, dtype=torch.float32, device=dev))
    backend.march_rays(n_alive, n_step, rays_alive, rays_t, o, d, bound, dt_gamma, max_steps, C, H, density_bitfield.contiguous(),
                       near, far, xyzs, dirs, deltas, noises.float().contiguous())
    return xyzs, dirs, deltas


@torch.no_grad()
def composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image, T_thresh=1e-2):
    """In-place accumulation into weights_sum / depth / image; terminated rays get rays_alive = -1 (raymarching.py:349-373)."""
    backend.composite_rays(n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas.float().contiguous(), rgbs.float().contiguous(),
                           deltas, weights_sum, depth, image)
    return tuple()

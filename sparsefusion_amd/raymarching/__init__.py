"""Ray utilities with the reference's Python call surface (raymarching/raymarching.py:19-155):
`near_far_from_aabb(rays_o, rays_d, aabb, min_near=0.2) -> (nears, fars)`, `morton3D`,
`morton3D_invert`, `packbits`.  None of them is differentiable in the reference either."""
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

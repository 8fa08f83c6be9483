"""Shared scene for the occupancy-grid (cuda_ray=True) tests: a solid ball inside bound 4 (3 cascades, 128^3 cells)."""
import math

import torch

from oracle import ngp_native

BOUND, H, MAX_STEPS = 4.0, 128, 1024
CASCADE = 1 + math.ceil(math.log2(BOUND))
RADIUS = 1.3


def ball_bitfield(radius=RADIUS, center=(0.2, -0.1, 0.3)):
    """density grid [C, H^3] in Morton order (1 inside the ball, 0 outside) and its packed bitfield."""
    idx = torch.arange(H ** 3, dtype=torch.int32)
    coords = torch.empty(H ** 3, 3, dtype=torch.int32)
    ngp_native.morton3D_invert(idx, H ** 3, coords)
    grid = torch.zeros(CASCADE, H ** 3)
    c = torch.tensor(center)
    for cas in range(CASCADE):
        b = min(2.0 ** cas, BOUND)
        xyz = ((coords.float() + 0.5) / H * 2 - 1) * b              # cell centres of this cascade
        grid[cas] = ((xyz - c).norm(dim=-1) < radius).float()
    bits = torch.empty(CASCADE * H ** 3 // 8, dtype=torch.uint8)
    ngp_native.packbits(grid.reshape(-1).contiguous(), bits.numel(), 0.5, bits)
    return grid, bits, c


def camera_rays(n_side=16, eye=(0.5, 0.8, -3.2), spread=0.9, seed=0):
    g = torch.Generator().manual_seed(seed)
    ax = torch.linspace(-spread, spread, n_side)
    yy, xx = torch.meshgrid(ax, ax, indexing="ij")
    o = torch.tensor(eye).expand(n_side * n_side, 3).contiguous()
    d = torch.stack([xx.reshape(-1), yy.reshape(-1), torch.full((n_side * n_side,), 2.0)], -1)
    d = d + 0.01 * torch.randn(d.shape, generator=g)
    d = (d / d.norm(dim=-1, keepdim=True)) * 1.3                    # non-unit directions, as the pytorch3d rays
    o = o - d * 0.2
    return o.contiguous(), d.contiguous()


def near_far(o, d):
    aabb = torch.tensor([-BOUND, -BOUND, -BOUND, BOUND, BOUND, BOUND])
    nears, fars = torch.empty(o.shape[0]), torch.empty(o.shape[0])
    ngp_native.near_far_from_aabb(o, d, aabb, o.shape[0], 0.05, nears, fars)
    return nears, fars


def oracle_march_train(o, d, bits, nears, fars, noises, dt_gamma=0.0, M=None, counter=None, max_steps=MAX_STEPS):
    N = o.shape[0]
    M = N * max_steps if M is None else M
    xyzs, dirs, deltas = torch.zeros(M, 3), torch.zeros(M, 3), torch.zeros(M, 2)
    rays = torch.full((N, 3), -7, dtype=torch.int32)
    counter = torch.zeros(2, dtype=torch.int32) if counter is None else counter
    ngp_native.march_rays_train(o, d, bits, BOUND, dt_gamma, max_steps, N, CASCADE, H, M, nears, fars, xyzs, dirs, deltas, rays,
                                counter, noises)
    return xyzs, dirs, deltas, rays, counter

"""Pinhole camera batch + grid / Monte-Carlo ray samplers with the conventions of the pytorch3d classes the reference uses
(`PerspectiveCameras`, `GridRaysampler`, `MonteCarloRaysampler`, `RayBundle`; call sites utils/render_utils.py:43-92,
sparsefusion/distillation.py:201,274, sparsefusion/eft.py:239,316).  pytorch3d itself is an unpinned third-party
dependency that is absent here: the published conventions are restated --
  * row-vector world -> view transform X_cam = X_world R + T, NDC = focal * (x, y) / z + principal point, +X left / +Y up;
  * a ray is the un-projection of its pixel's NDC xy at depths 1 and 2: directions = p2 - p1 (NOT unit length),
    origins = p1 - directions (the camera centre), lengths = linspace(min_depth, max_depth, n_pts_per_ray);
  * the grid sampler walks y from min_y to max_y (rows) and x from min_x to max_x (columns), both INCLUSIVE.
Any object with `transform_points_ndc`, `get_camera_center`, `unproject_points` and `__len__` can stand in for these
cameras (the EFT module and the samplers are duck-typed).  Everything here is O(rays) torch glue on the caller's device."""
import collections

import torch

RayBundle = collections.namedtuple("RayBundle", ["origins", "directions", "lengths", "xys"])


def ray_bundle_to_ray_points(rb):
    """pytorch3d.renderer.ray_bundle_to_ray_points: [..., n_pts, 3] world points of a bundle."""
    return rb.origins[..., None, :] + rb.lengths[..., :, None] * rb.directions[..., None, :]


class PinholeCameras(torch.nn.Module):
    """N cameras: R [N,3,3], T [N,3] (row-vector convention), focal_length [N,2] and principal_point [N,2] in NDC units."""

    def __init__(self, R, T, focal_length, principal_point=None):
        super().__init__()
        self.register_buffer("R", R.float())
        self.register_buffer("T", T.float())
        focal_length = focal_length.float()
        if focal_length.dim() == 1:
            focal_length = focal_length[:, None].expand(-1, 2)
        self.register_buffer("focal_length", focal_length.contiguous())
        pp = torch.zeros_like(self.focal_length) if principal_point is None else principal_point.float()
        self.register_buffer("principal_point", pp)

    def __len__(self):
        return self.R.shape[0]

    def __getitem__(self, idx):
        idx = [idx] if isinstance(idx, int) else idx
        return PinholeCameras(self.R[idx], self.T[idx], self.focal_length[idx], self.principal_point[idx])

    @property
    def device(self):
        return self.R.device

    def get_camera_center(self):
        return -torch.bmm(self.T[:, None], self.R.transpose(1, 2))[:, 0]          # C = -T R^T

    def transform_points_ndc(self, pts):
        p = pts.expand(len(self), -1, -1) if pts.shape[0] == 1 else pts
        cam = torch.bmm(p, self.R) + self.T[:, None]
        z = cam[..., 2:3]
        xy = cam[..., :2] / z * self.focal_length[:, None] + self.principal_point[:, None]
        return torch.cat([xy, 1.0 / z], -1)

    def unproject_points(self, xy_depth, world_coordinates=True):
        """[N, P, 3] (ndc x, ndc y, depth) -> world (or view) points: the inverse of transform_points_ndc at that depth."""
        xy, depth = xy_depth[..., :2], xy_depth[..., 2:3]
        cam_xy = (xy - self.principal_point[:, None]) / self.focal_length[:, None] * depth
        cam = torch.cat([cam_xy, depth], -1)
        if not world_coordinates:
            return cam
        return torch.bmm(cam - self.T[:, None], self.R.transpose(1, 2))


def _xy_to_ray_bundle(cameras, xy_grid, min_depth, max_depth, n_pts_per_ray):
    """pytorch3d `_xy_to_ray_bundle`: xy_grid [N, ..., 2] -> RayBundle with spatial dims `...`."""
    N = xy_grid.shape[0]
    spatial = xy_grid.shape[1:-1]
    P = 1
    for s in spatial:
        P *= s
    dev = xy_grid.device
    depths = torch.linspace(min_depth, max_depth, n_pts_per_ray, dtype=xy_grid.dtype, device=dev)
    lengths = depths[None, None].expand(N, P, n_pts_per_ray)
    flat = xy_grid.reshape(N, P, 2)
    to_unproject = torch.cat((flat.repeat(1, 2, 1),
                              torch.cat((flat.new_ones(N, P, 1), 2.0 * flat.new_ones(N, P, 1)), 1)), -1)
    world = cameras.unproject_points(to_unproject)
    p1, p2 = world[:, :P], world[:, P:]
    directions = p2 - p1
    origins = p1 - directions
    return RayBundle(origins.view(N, *spatial, 3), directions.view(N, *spatial, 3), lengths.view(N, *spatial, n_pts_per_ray),
                     xy_grid)


class GridRaysampler(torch.nn.Module):
    """pytorch3d.renderer.GridRaysampler (NDC): one ray per node of an image_height x image_width lattice."""

    def __init__(self, min_x, max_x, min_y, max_y, image_width, image_height, n_pts_per_ray, min_depth, max_depth):
        super().__init__()
        self._n_pts_per_ray, self._min_depth, self._max_depth = n_pts_per_ray, min_depth, max_depth
        ys = torch.linspace(min_y, max_y, image_height, dtype=torch.float32)
        xs = torch.linspace(min_x, max_x, image_width, dtype=torch.float32)
        Y, X = torch.meshgrid(ys, xs, indexing="ij")
        self.register_buffer("_xy_grid", torch.stack((X, Y), -1), persistent=False)        # [H, W, 2], (x, y) order

    def forward(self, cameras, **kwargs):
        n = len(cameras)
        grid = self._xy_grid.to(cameras.get_camera_center().device)[None].expand(n, -1, -1, -1)
        return _xy_to_ray_bundle(cameras, grid, self._min_depth, self._max_depth, self._n_pts_per_ray)


class MonteCarloRaysampler(torch.nn.Module):
    """pytorch3d.renderer.MonteCarloRaysampler: n_rays_per_image uniform samples of the NDC rectangle per camera."""

    def __init__(self, min_x, max_x, min_y, max_y, n_rays_per_image, n_pts_per_ray, min_depth, max_depth):
        super().__init__()
        self._box = (min_x, max_x, min_y, max_y)
        self._n_rays, self._n_pts_per_ray, self._min_depth, self._max_depth = n_rays_per_image, n_pts_per_ray, min_depth, max_depth

    def forward(self, cameras, generator=None, **kwargs):
        n = len(cameras)
        dev = cameras.get_camera_center().device
        min_x, max_x, min_y, max_y = self._box
        u = torch.rand(n, self._n_rays, 2, generator=generator, device=dev if generator is None else generator.device).to(dev)
        xy = torch.stack((u[..., 0] * (max_x - min_x) + min_x, u[..., 1] * (max_y - min_y) + min_y), -1)
        return _xy_to_ray_bundle(cameras, xy, self._min_depth, self._max_depth, self._n_pts_per_ray)

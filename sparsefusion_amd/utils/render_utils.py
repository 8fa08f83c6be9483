"""`init_ray_sampler` / `init_light_field_renderer` of the reference (utils/render_utils.py:16-92, :95-187): the grid sampler at
full resolution, the Monte-Carlo sampler, and the feature-resolution grid sampler (20 depths per ray) the EFT pre-pass
renders through.  Same arguments and return tuples; `gpu` selects the device the renderers' buffers move to."""
import torch

from .cameras import GridRaysampler, MonteCarloRaysampler
from .eft_raymarcher import LightFieldRaymarcher
from .eft_renderer import CustomImplicitRenderer


def _samplers(img_h, img_w, min, max, bbox, n_pts_per_ray, n_rays, scale_factor):
    half_w, half_h = 1.0 / img_w, 1.0 / img_h
    grid = GridRaysampler(min_x=1.0 - half_w, max_x=-1.0 + half_w, min_y=1.0 - half_h, max_y=-1.0 + half_h, image_height=img_h,
                          image_width=img_w, n_pts_per_ray=n_pts_per_ray, min_depth=min, max_depth=max)
    feat = None
    if scale_factor is not None:       # the reference keeps the FULL-resolution half-pixel margin for the coarse lattice
        feat = GridRaysampler(min_x=1.0 - half_w, max_x=-1.0 + half_w, min_y=1.0 - half_h, max_y=-1.0 + half_h,
                              image_height=int(img_h // scale_factor), image_width=int(img_w // scale_factor), n_pts_per_ray=20,
                              min_depth=min, max_depth=max)
    if bbox is None:
        mc = MonteCarloRaysampler(min_x=-1.0, max_x=1.0, min_y=-1.0, max_y=1.0, n_rays_per_image=n_rays,
                                  n_pts_per_ray=n_pts_per_ray, min_depth=min, max_depth=max)
    else:
        mc = MonteCarloRaysampler(min_x=-float(bbox[0, 1]), max_x=-float(bbox[0, 3]), min_y=-float(bbox[0, 0]), max_y=-float(bbox[0, 2]),
                                  n_rays_per_image=n_rays, n_pts_per_ray=n_pts_per_ray, min_depth=min, max_depth=max)
    return grid, mc, feat


def init_ray_sampler(gpu, img_h, img_w, min=0.1, max=4.0, bbox=None, n_pts_per_ray=128, n_rays=750, scale_factor=None):
    grid, mc, feat = _samplers(img_h, img_w, min, max, bbox, n_pts_per_ray, n_rays, scale_factor)
    return (grid, mc, feat) if scale_factor is not None else (grid, mc)


def init_light_field_renderer(gpu, img_h, img_w, min=0.1, max=4.0, bbox=None, n_pts_per_ray=128, n_rays=750, scale_factor=None):
    grid, mc, feat = _samplers(img_h, img_w, min, max, bbox, n_pts_per_ray, n_rays, scale_factor)
    marcher = LightFieldRaymarcher()
    dev = torch.device("cuda", gpu) if isinstance(gpu, int) else torch.device(gpu)
    renderer_grid = CustomImplicitRenderer(raysampler=grid, raymarcher=marcher, reg=True).to(dev)
    renderer_mc = CustomImplicitRenderer(raysampler=mc, raymarcher=marcher, reg=True).to(dev)
    if scale_factor is None:
        return renderer_grid, renderer_mc
    renderer_feat = CustomImplicitRenderer(raysampler=feat, raymarcher=marcher, reg=True).to(dev)
    return renderer_grid, renderer_mc, renderer_feat

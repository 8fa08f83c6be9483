"""`CustomImplicitRenderer` of the reference (utils/eft_renderer.py:18-167): ray sampler -> volumetric function -> ray
marcher.  With `reg` set the volumetric function's third return value travels along, which is how the distillation pre-pass
calls it: `renderer_feat(cameras=, volumetric_function=eft.batched_forward, n_batches=16, input_cameras=, input_rgb=)`
-> (features [1, 32, 32, 3 + 256], ray_bundle, reg) (sparsefusion/distillation.py:103-109)."""
import torch


class CustomImplicitRenderer(torch.nn.Module):
    def __init__(self, raysampler, raymarcher, reg=None):
        super().__init__()
        if not callable(raysampler):
            raise ValueError('"raysampler" has to be a "Callable" object.')
        if not callable(raymarcher):
            raise ValueError('"raymarcher" has to be a "Callable" object.')
        self.raysampler, self.raymarcher, self.reg = raysampler, raymarcher, reg

    def forward(self, cameras, volumetric_function, **kwargs):
        if not callable(volumetric_function):
            raise ValueError('"volumetric_function" has to be a "Callable" object.')
        ray_bundle = self.raysampler(cameras=cameras, volumetric_function=volumetric_function, **kwargs)
        rays_densities, rays_features, reg_term = volumetric_function(ray_bundle=ray_bundle, cameras=cameras, **kwargs)
        images = self.raymarcher(rays_densities=rays_densities, rays_features=rays_features, ray_bundle=ray_bundle, **kwargs)
        if self.reg is not None:
            return images, ray_bundle, reg_term
        return images, ray_bundle, 0                    # utils/eft_renderer.py:164-167: always a 3-tuple

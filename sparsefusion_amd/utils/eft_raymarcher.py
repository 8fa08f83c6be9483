"""`LightFieldRaymarcher` of the reference (utils/eft_raymarcher.py:16-31): a nominal ray marcher -- the light-field features
come back as they are, opacity first."""
import torch


class LightFieldRaymarcher(torch.nn.Module):
    def forward(self, rays_densities, rays_features, **kwargs):
        return torch.cat((rays_densities, rays_features), dim=-1)

"""Diffusion container + continuous-time noise schedule for the PLMS path.

Host-side counterpart of the parts of the reference that `PLMSSampler` touches:
  GaussianDiffusionContinuousTimes   external/imagen_pytorch.py:201-297 (cosine log-SNR :194-199)
  DDPM                               sparsefusion/vldm.py:53-285 -- only the attributes the sampler reads
                                     (`unets`, `noise_schedulers`, `pred_objectives`, `dynamic_thresholding`,
                                     `sample_channels`, `image_sizes`, `device`, `clip_output`, `clip_value`,
                                     `unnormalize_img`) and `load_state_dict` with the `unets.0.*` keys
                                     (utils/load_model.py:76-94).
Training (`p_losses`, `forward`) and the ancestral sampler are out of scope."""
import math

import torch
import torch.nn as nn

from .unet import Unet


def alpha_cosine_log_snr(t, s: float = 0.008):
    return -torch.log(((torch.cos((t + s) / (1 + s) * math.pi * 0.5) ** -2) - 1).clamp(min=1e-5))


def log_snr_to_alpha_sigma(log_snr):
    return torch.sqrt(torch.sigmoid(log_snr)), torch.sqrt(torch.sigmoid(-log_snr))


def _pad_dims(x, t):
    return t.view(*t.shape, *((1,) * (x.ndim - t.ndim)))


class GaussianDiffusionContinuousTimes(nn.Module):
    def __init__(self, *, noise_schedule, timesteps=1000):
        super().__init__()
        if noise_schedule != "cosine":
            raise NotImplementedError("SparseFusion uses the cosine schedule (sparsefusion/vldm.py:66, plms.py:81)")
        self.log_snr = alpha_cosine_log_snr
        self.num_timesteps = timesteps

    def get_times(self, batch_size, noise_level, *, device):
        return torch.full((batch_size,), noise_level, device=device, dtype=torch.float32)

    def get_condition(self, times):
        return None if times is None else self.log_snr(times)

    def _pairs(self, times, batch):
        times = times.unsqueeze(0).expand(batch, -1)
        return list(zip(times[:, :-1].unbind(dim=-1), times[:, 1:].unbind(dim=-1)))

    def get_sampling_timesteps(self, batch, *, device):
        return self._pairs(torch.linspace(1., 0., self.num_timesteps + 1, device=device), batch)

    def get_sampling_timesteps_custom(self, batch, min_thres=0.0, max_thres=0.999, *, device, n_steps=5):
        return self._pairs(torch.linspace(max_thres, min_thres, n_steps + 1, device=device), batch)

    def q_sample(self, x_start, t, noise=None):
        if isinstance(t, float):
            t = torch.full((x_start.shape[0],), t, device=x_start.device, dtype=x_start.dtype)
        noise = torch.randn_like(x_start) if noise is None else noise
        log_snr = self.log_snr(t)
        alpha, sigma = log_snr_to_alpha_sigma(_pad_dims(x_start, log_snr))
        return alpha * x_start + sigma * noise, log_snr

    def q_posterior(self, x_start, x_t, t, *, t_next=None):
        if t_next is None:
            t_next = (t - 1. / self.num_timesteps).clamp(min=0.)
        log_snr, log_snr_next = _pad_dims(x_t, self.log_snr(t)), _pad_dims(x_t, self.log_snr(t_next))
        alpha, _ = log_snr_to_alpha_sigma(log_snr)
        alpha_next, sigma_next = log_snr_to_alpha_sigma(log_snr_next)
        c = -torch.expm1(log_snr - log_snr_next)
        mean = alpha_next * (x_t * (1 - c) / alpha + c * x_start)
        var = (sigma_next ** 2) * c
        return mean, var, torch.log(var.clamp(min=1e-20))

    def predict_start_from_noise(self, x_t, t, noise):
        alpha, sigma = log_snr_to_alpha_sigma(_pad_dims(x_t, self.log_snr(t)))
        return (x_t - sigma * noise) / alpha.clamp(min=1e-8)


def _tup(v, n):
    return tuple(v) if isinstance(v, (tuple, list)) else (v,) * n


class DDPM(nn.Module):
    def __init__(self, unets, *, image_sizes, conditional_encoder=None, conditional_embed_dim=None, channels=3,
                 timesteps=1000, cond_drop_prob=0.1, noise_schedules='cosine', pred_objectives='noise', conditional=True,
                 auto_normalize_img=False, dynamic_thresholding=True, dynamic_thresholding_percentile=0.95,
                 clip_output=True, clip_value=1.0, **unused):
        super().__init__()
        unets = _tup(unets, 1)
        if len(unets) != 1 or not isinstance(unets[0], Unet):
            raise NotImplementedError("one sparsefusion_amd.Unet (SparseFusion has no cascade: utils/load_model.py:76-91)")
        if conditional or auto_normalize_img:
            raise NotImplementedError("SparseFusion builds DDPM(conditional=False, auto_normalize_img=False)")
        self.timesteps, self.channels, self.conditional = timesteps, channels, conditional
        self.unconditional = not conditional
        self.noise_schedulers = nn.ModuleList([GaussianDiffusionContinuousTimes(noise_schedule=_tup(noise_schedules, 1)[0],
                                                                                timesteps=_tup(timesteps, 1)[0])])
        self.pred_objectives = _tup(pred_objectives, 1)
        self.unets = nn.ModuleList([unets[0].cast_model_parameters(lowres_cond=False, cond_on_z=False, conditional_embed_dim=None,
                                                                   channels=channels, channels_out=channels)])
        self.image_sizes = _tup(image_sizes, 1)
        self.sample_channels = (channels,)
        self.cond_drop_prob = cond_drop_prob
        self.can_classifier_guidance = cond_drop_prob > 0.
        self.normalize_img = self.unnormalize_img = lambda t: t
        self.dynamic_thresholding = _tup(dynamic_thresholding, 1)
        self.dynamic_thresholding_percentile = dynamic_thresholding_percentile
        self.clip_output, self.clip_value = clip_output, clip_value
        self.register_buffer('_temp', torch.tensor([0.]), persistent=False)

    @property
    def device(self):
        return self._temp.device

    def get_unet(self, unet_number):
        assert unet_number == 1
        return self.unets[0]

    def load_state_dict(self, *args, **kwargs):
        r = super().load_state_dict(*args, **kwargs)
        self.unets[0].invalidate()
        return r

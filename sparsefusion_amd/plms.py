"""PLMS / PNDM sampler on the HIP UNet -- call surface of external/plms.py:13-214.

    PLMSSampler(vldm, 50).sample(latents, cond_images=feat, use_tqdm=False, return_noise=True, max_thres=m)
        -> (pred_x0, x_noisy, noise, alpha_cumprod)                (sparsefusion/distillation.py:304)

Same step count (n = min(int(max_thres*2*steps), steps), n+1 UNet evaluations, 0 when n == 0), same
Adams-Bashforth coefficients, same clamp(+-clip_value), and the same order of `torch.randn_like`
draws (one per get_model_output call, used or not) so a seeded run consumes the generator like the
reference.  Restructured for the GPU: the schedule scalars of a step are computed once on the host in
fp32 (they are identical for every batch element), the latent updates are two fused kernels
(sf_plms_update / sf_plms_combine), and nothing synchronises with the host inside the loop."""
import numpy as np
import torch

from . import _lib
from .vldm import DDPM, alpha_cosine_log_snr, log_snr_to_alpha_sigma


def _f(v):
    return torch.tensor(float(v), dtype=torch.float32)


def step_coefficients(t, t_next, clip):
    """[alpha, sigma, alpha_next, c, noise_scale, clip] of one get_model_output call (plms.py:176-212)."""
    ls, lsn = alpha_cosine_log_snr(_f(t)), alpha_cosine_log_snr(_f(t_next))
    alpha, sigma = log_snr_to_alpha_sigma(ls)
    alpha_next, sigma_next = log_snr_to_alpha_sigma(lsn)
    c = -torch.expm1(ls - lsn)
    logvar = torch.log(((sigma_next ** 2) * c).clamp(min=1e-20))
    noise_scale = 0.0 if float(t_next) == 0.0 else float((0.5 * logvar).exp())
    return np.array([float(alpha), float(sigma), float(alpha_next), float(c), noise_scale, float(clip)], dtype=np.float32)


class PLMSSampler():
    def __init__(self, diffusion: DDPM, plms_steps=100):
        self.diffusion = diffusion
        self.plms_steps = plms_steps

    @torch.no_grad()
    def sample(self, image=None, max_thres=.999, cond_images=None, cond_scale=1.0, use_tqdm=True, return_noise=False,
               noises=None, **kwargs):
        d = self.diffusion
        batch = cond_images.shape[0]
        shape = (batch, d.sample_channels[0], d.image_sizes[0], d.image_sizes[0])
        if d.pred_objectives[0] != 'noise':
            raise NotImplementedError("pred_objective must be 'noise' (plms.py:170)")
        if d.dynamic_thresholding[0] and d.clip_output:
            raise NotImplementedError("dynamic thresholding is off in SparseFusion (utils/load_model.py:88)")
        out = self._loop(d.unets[0], image, shape, cond_images, cond_scale, max_thres, noises)
        return out if return_noise else out[0]

    def _loop(self, unet, image, shape, cond_images, cond_scale, max_thres, noises):
        d = self.diffusion
        dev = cond_images.device
        lib = _lib.lib()
        draw = (lambda: torch.randn(shape, device=dev)) if noises is None else iter(noises).__next__
        if image is None:
            image = torch.randn(shape, device=dev)
        else:
            assert max_thres is not None
        image = image.float().contiguous()
        n = image.numel()
        B = shape[0]
        clip = d.clip_value if d.clip_output else 3.0e38
        if max_thres >= .99:
            times = torch.linspace(1., 0., self.plms_steps + 1)
        else:
            n_steps = min(int(max_thres * self.plms_steps * 2), self.plms_steps)
            times = torch.linspace(max_thres, 0.0, n_steps + 1)
        noise = draw()
        ls0 = alpha_cosine_log_snr(_f(max_thres))
        a0, s0 = log_snr_to_alpha_sigma(ls0)
        x_noisy = float(a0) * image + float(s0) * noise
        img = image if max_thres >= .99 else x_noisy
        tl = times.tolist()

        # the UNet's time path depends on the step only: one table for the whole trajectory (Unet.time_table), every
        # eval then replays the body of the plan (classifier-free guidance, cond_scale != 1, keeps the generic path)
        fast = cond_scale == 1 and hasattr(unet, "begin_sampling") and len(tl) > 1
        if fast:
            eval_times = list(dict.fromkeys(tl[:-1] + [tl[1]]))     # every step's t, plus t_next of the first (improved Euler)
            row_of = {t: k for k, t in enumerate(eval_times)}
            ctx = unet.begin_sampling(cond_images, torch.stack([alpha_cosine_log_snr(_f(t)) for t in eval_times]).to(dev))

        def eps_model(x, t):
            if fast:
                return unet.eval_prepared(ctx, x, row_of[t]).clone()
            ls = torch.full((B,), float(alpha_cosine_log_snr(_f(t))), dtype=torch.float32, device=dev)
            return unet.forward_with_cond_scale(x, ls, cond_images=cond_images, cond_scale=cond_scale)

        def update(x, e, t, t_next):
            coef = step_coefficients(t, t_next, clip)
            nz = draw()
            x_prev = torch.empty_like(x)
            _lib.check(lib.sf_plms_update(_lib.ptr(x), _lib.ptr(e), _lib.ptr(nz), coef.ctypes.data, n, _lib.ptr(x_prev),
                                          None, _lib.stream_ptr()), "plms_update")
            return x_prev

        def combine(es, cs):
            c4 = np.zeros(4, dtype=np.float32)
            c4[:len(cs)] = cs
            ptrs = [_lib.ptr(e) for e in es] + [None] * (4 - len(es))
            out = torch.empty_like(es[0])
            _lib.check(lib.sf_plms_combine(*ptrs, c4.ctypes.data, n, _lib.ptr(out), _lib.stream_ptr()), "plms_combine")
            return out

        old = []
        for t, t_next in zip(tl[:-1], tl[1:]):
            e_t = eps_model(img, t)
            draw()                                    # get_model_output draws a noise tensor it does not use here
            if len(old) == 0:                         # pseudo improved Euler (plms.py:137-143)
                x_prev = update(img, e_t, t, t_next)
                e_next = eps_model(x_prev, t_next)
                draw()
                e_prime = combine([e_t, e_next], [0.5, 0.5])
            elif len(old) == 1:
                e_prime = combine([e_t, old[-1]], [3 / 2, -1 / 2])
            elif len(old) == 2:
                e_prime = combine([e_t, old[-1], old[-2]], [23 / 12, -16 / 12, 5 / 12])
            else:
                e_prime = combine([e_t, old[-1], old[-2], old[-3]], [55 / 24, -59 / 24, 37 / 24, -9 / 24])
            img = update(img, e_prime, t, t_next)
            old.append(e_t)
            if len(old) >= 4:
                old.pop(0)
        if d.clip_output:
            img = img.clamp(-d.clip_value, d.clip_value)
        alpha_cumprod = torch.sigmoid(ls0).to(dev).expand(B).clone()
        return d.unnormalize_img(img), x_noisy, noise, alpha_cumprod

"""PLMS / PNDM sampler on the HIP UNet -- call surface of external/plms.py:13-214.

    PLMSSampler(vldm, 50).sample(latents, cond_images=feat, use_tqdm=False, return_noise=True, max_thres=m)
        -> (pred_x0, x_noisy, noise, alpha_cumprod)                (sparsefusion/distillation.py:304)

Same step count (n = min(int(max_thres*2*steps), steps), n+1 UNet evaluations, 0 when n == 0), same
Adams-Bashforth coefficients, same clamp(+-clip_value), and the same NUMBER and ROLE of gaussian draws (one
per get_model_output call, used or not) -- but they come out of ONE `torch.randn` of the whole trajectory on
the device generator, so a seeded run is reproducible here without reproducing the reference's CPU-generator
stream (pass `noises=` to inject the reference's own draws, as the golden tests do).
Restructured for the GPU: the schedule scalars of a step are computed once on the host in fp32 (they are
identical for every batch element); the UNet's time path is evaluated once per trajectory (Unet.time_table) and
each eval replays the plan body; the latents live in the plan's input buffer and are updated in place by two
fused kernels (sf_plms_update / sf_plms_combine, the latter also files the eps history entry); nothing
synchronises with the host inside the loop."""
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
        B = shape[0]
        clip = d.clip_value if d.clip_output else 3.0e38
        if max_thres >= .99:
            times = torch.linspace(1., 0., self.plms_steps + 1)
        else:
            n_steps = min(int(max_thres * self.plms_steps * 2), self.plms_steps)
            times = torch.linspace(max_thres, 0.0, n_steps + 1)
        tl = times.tolist()
        n_sampler_steps = len(tl) - 1
        if noises is None:
            # every draw of the trajectory in ONE launch (reference order and count: 1 for q_sample, 1 per model call whether
            # it is used or not, plms.py:158-214); `image is None` draws the start image first, as the reference does
            n_draws = 1 + 2 * n_sampler_steps + (2 if n_sampler_steps > 0 else 0) + (1 if image is None else 0)
            bulk = torch.randn((n_draws,) + tuple(shape), device=dev)
            draw = iter(bulk.unbind(0)).__next__
        else:
            draw = iter(noises).__next__
        if image is None:
            image = draw()
        else:
            assert max_thres is not None
        image = image.float().contiguous()
        n = image.numel()
        noise = draw()
        ls0 = alpha_cosine_log_snr(_f(max_thres))
        a0, s0 = log_snr_to_alpha_sigma(ls0)
        x_noisy = float(a0) * image + float(s0) * noise
        img = image if max_thres >= .99 else x_noisy

        # the UNet's time path depends on the step only: one table for the whole trajectory (Unet.time_table), every
        # eval then replays the body of the plan (classifier-free guidance, cond_scale != 1, keeps the generic path)
        fast = cond_scale == 1 and hasattr(unet, "begin_sampling") and n_sampler_steps > 0
        if fast:
            eval_times = list(dict.fromkeys(tl[:-1] + [tl[1]]))     # every step's t, plus t_next of the first (improved Euler)
            row_of = {t: k for k, t in enumerate(eval_times)}
            ctx = unet.begin_sampling(cond_images, torch.stack([alpha_cosine_log_snr(_f(t)) for t in eval_times]).to(dev),
                                      table_key=tuple(eval_times))        # same schedule as the last call: the cached time table
            x_slot = ctx["plan"].x_view.view(shape)                 # the plan's input buffer: latents live there between steps

        row_ready = [None]          # the table row the last sf_plms_step already copied into the plan (one launch less per eval)

        def eps_model(x, t):
            """eps for latents x at time t.  Fast path: a VIEW of the plan's output buffer (valid until the next eval)."""
            if fast:
                ready, row_ready[0] = row_ready[0] == row_of[t], None
                return unet.eval_prepared(ctx, x, row_of[t], row_ready=ready)
            ls = torch.full((B,), float(alpha_cosine_log_snr(_f(t))), dtype=torch.float32, device=dev)
            return unet.forward_with_cond_scale(x, ls, cond_images=cond_images, cond_scale=cond_scale)

        def update(x, e, t, t_next, out=None):
            coef = step_coefficients(t, t_next, clip)
            nz = draw()
            x_prev = torch.empty_like(x) if out is None else out    # out may alias x: the kernel is elementwise
            _lib.check(lib.sf_plms_update(_lib.ptr(x), _lib.ptr(e), _lib.ptr(nz), coef.ctypes.data, n, _lib.ptr(x_prev),
                                          None, _lib.stream_ptr()), "plms_update")
            return x_prev

        def combine(es, cs, keep=None):
            """sum_k cs[k] * es[k]; `keep` also receives a copy of es[0] (the history entry of this step)."""
            c4 = np.zeros(4, dtype=np.float32)
            c4[:len(cs)] = cs
            ptrs = [_lib.ptr(e) for e in es] + [None] * (4 - len(es))
            out = torch.empty(shape, device=dev)
            _lib.check(lib.sf_plms_combine(*ptrs, c4.ctypes.data, n, _lib.ptr(out), _lib.ptr(keep), _lib.stream_ptr()), "plms_combine")
            return out

        def step(es, cs, keep, x, t, t_next, out):
            """combine + update in ONE launch: x_prev = update(x, sum_k cs[k] * es[k]); `keep` receives es[0].  On the fast path the
            same launch also moves the NEXT eval's time-block row into the plan (its eps inputs are read before anything else runs)."""
            c4 = np.zeros(4, dtype=np.float32)
            c4[:len(cs)] = cs
            ptrs = [_lib.ptr(e) for e in es] + [None] * (4 - len(es))
            coef = step_coefficients(t, t_next, clip)
            nz = draw()
            x_prev = torch.empty_like(x) if out is None else out
            rsrc = rdst = None
            rn = 0
            if fast and t_next in row_of and B == 1 and ctx["table"].shape[1] % 4 == 0:
                rsrc, rdst, rn = ctx["table"][row_of[t_next]], ctx["plan"].tb_view, ctx["table"].shape[1]
                row_ready[0] = row_of[t_next]
            _lib.check(lib.sf_plms_step(*ptrs, c4.ctypes.data, _lib.ptr(keep), _lib.ptr(x), _lib.ptr(nz), coef.ctypes.data, n,
                                        _lib.ptr(x_prev), _lib.ptr(rsrc), _lib.ptr(rdst), rn, _lib.stream_ptr()), "plms_step")
            return x_prev

        ring = [torch.empty(shape, device=dev) for _ in range(4)]   # eps history (the eval's output buffer is reused)
        old = []
        for k, (t, t_next) in enumerate(zip(tl[:-1], tl[1:])):
            e_view = eps_model(img, t)
            draw()                                    # get_model_output draws a noise tensor it does not use here
            e_t = ring[k % 4]
            if len(old) == 0:                         # pseudo improved Euler (plms.py:137-143)
                e_t.copy_(e_view)
                x_prev = update(img, e_t, t, t_next)
                e_next = eps_model(x_prev, t_next)
                draw()
                e_prime = combine([e_t, e_next], [0.5, 0.5])
            if len(old) == 0:
                img = update(img, e_prime, t, t_next, out=x_slot if fast else None)
            elif len(old) == 1:
                img = step([e_view, old[-1]], [3 / 2, -1 / 2], e_t, img, t, t_next, x_slot if fast else None)
            elif len(old) == 2:
                img = step([e_view, old[-1], old[-2]], [23 / 12, -16 / 12, 5 / 12], e_t, img, t, t_next, x_slot if fast else None)
            else:
                img = step([e_view, old[-1], old[-2], old[-3]], [55 / 24, -59 / 24, 37 / 24, -9 / 24], e_t, img, t, t_next,
                           x_slot if fast else None)
            old.append(e_t)
            if len(old) >= 4:
                old.pop(0)
        if d.clip_output:
            img = img.clamp(-d.clip_value, d.clip_value)
        alpha_cumprod = torch.sigmoid(ls0).to(dev).expand(B).clone()
        return d.unnormalize_img(img), x_noisy, noise, alpha_cumprod

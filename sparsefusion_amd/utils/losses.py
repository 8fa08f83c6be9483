"""Loss glue of the distillation step on HIP kernels (csrc/loss_ops.hip), differentiable through torch autograd:

    upsample2x(x)                         == F.interpolate(x, scale_factor=2, mode='bilinear')   (distillation.py:287-288)
    render_loss(img, sil, rgb, mask, ..)  == lambda_color * huber(img, rgb).abs().mean() + lambda_sil * huber(sil, mask).abs().mean()
                                             + lambda_opacity * sqrt(sil^2 + .01).mean() + lambda_entropy * H2(sil).mean()   (:217-241)
    fusion_loss(img, sil, pred, w, ..)    == (w[view] * (img - pred).abs()).mean() + lambda_opacity * ... + lambda_entropy * ...   (:310-343)

Each is one forward launch (+ one tiny row sum) and one backward launch; torch ran 6-15 elementwise kernels per term plus
their autograd twins.  The perceptual term stays with sparsefusion_amd.lpips.PerceptualLoss and is simply added by the caller.
No CPU fallback: CPU tensors raise."""
import torch

from .. import _lib


def _f32c(t):
    return t.float().contiguous()


class _Upsample2x(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _lib.require_cuda(x)
        x = _f32c(x)
        N, C, h, w = x.shape
        out = torch.empty(N, C, 2 * h, 2 * w, device=x.device, dtype=torch.float32)
        _lib.check(_lib.lib().sf_upsample2x_forward(_lib.ptr(x), _lib.ptr(out), N * C, h, w, _lib.stream_ptr()), "upsample2x")
        ctx.shape = (N, C, h, w)
        return out

    @staticmethod
    def backward(ctx, g):
        N, C, h, w = ctx.shape
        g = _f32c(g)
        gin = torch.empty(N, C, h, w, device=g.device, dtype=torch.float32)
        _lib.check(_lib.lib().sf_upsample2x_backward(_lib.ptr(g), _lib.ptr(gin), N * C, h, w, _lib.stream_ptr()), "upsample2x_backward")
        return gin


def upsample2x(x):
    """Bilinear x2 of an NCHW map, align_corners=False (torch's default for F.interpolate(..., mode='bilinear'))."""
    return _Upsample2x.apply(x)


class _RenderLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, sil, rgb, mask, lam, scaling):
        _lib.require_cuda(img, sil, rgb)
        img, sil, rgb = _f32c(img), _f32c(sil), _f32c(rgb)
        mask = _f32c(mask) if mask is not None else None
        if rgb.shape != img.shape or (mask is not None and mask.shape != sil.shape):
            raise RuntimeError("render_loss: targets must have the shapes of the renders")
        lib = _lib.lib()
        part = torch.empty(lib.sf_loss_partial_rows(), 4, device=img.device, dtype=torch.float32)
        _lib.check(lib.sf_render_loss_forward(_lib.ptr(img), _lib.ptr(sil), _lib.ptr(rgb), _lib.ptr(mask), img.numel(), sil.numel(),
                                              scaling, _lib.ptr(part), _lib.stream_ptr()), "render_loss")
        sums = part.sum(0)
        n_img, n_sil = img.numel(), sil.numel()
        ctx.save_for_backward(img, sil, rgb, mask)
        ctx.cfg = (lam, scaling, n_img, n_sil)
        lc, ls, lo, le = lam
        terms = torch.stack((sums[0] / n_img, sums[1] / n_sil, sums[2] / n_sil, sums[3] / n_sil))
        ctx.mark_non_differentiable(terms)
        loss = lc * terms[0] + (ls * terms[1] if mask is not None else 0.0) + lo * terms[2] + le * terms[3]
        return loss, terms

    @staticmethod
    def backward(ctx, g, _g_terms):
        img, sil, rgb, mask = ctx.saved_tensors
        (lc, ls, lo, le), scaling, n_img, n_sil = ctx.cfg
        g = _f32c(g)                      # upstream gradient of the scalar loss: read by the kernel on the device (no host sync)
        g_img, g_sil = torch.empty_like(img), torch.empty_like(sil)
        _lib.check(_lib.lib().sf_render_loss_backward(_lib.ptr(img), _lib.ptr(sil), _lib.ptr(rgb), _lib.ptr(mask), n_img, n_sil, scaling,
                                                      lc / n_img, ls / n_sil, lo / n_sil, le / n_sil, _lib.ptr(g),
                                                      _lib.ptr(g_img), _lib.ptr(g_sil), _lib.stream_ptr()), "render_loss_backward")
        return g_img, g_sil, None, None, None, None


def render_loss(img, sil, target_rgb, target_mask=None, lambda_color=1.0, lambda_sil=1.0, lambda_opacity=1e-3, lambda_entropy=1e-3,
                scaling=0.1, return_terms=False):
    """Stage-A loss of the distillation loop.  `return_terms` adds the detached means (|huber rgb|, |huber sil|, opacity, entropy)."""
    loss, terms = _RenderLoss.apply(img, sil, target_rgb, target_mask, (lambda_color, lambda_sil, lambda_opacity, lambda_entropy), scaling)
    return (loss, terms) if return_terms else loss


class _FusionLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, sil, pred, weight, lam):
        _lib.require_cuda(img, sil, pred, weight)
        img, sil, pred, weight = _f32c(img), _f32c(sil), _f32c(pred), _f32c(weight).reshape(-1)
        V = img.shape[0]
        if pred.shape != img.shape or sil.shape[0] != V or weight.numel() != V:
            raise RuntimeError("fusion_loss: one weight per view; prediction of the render's shape")
        lib = _lib.lib()
        part = torch.empty(lib.sf_loss_partial_rows(), 3, device=img.device, dtype=torch.float32)
        pv_img, pv_sil = img[0].numel(), sil[0].numel()
        _lib.check(lib.sf_fusion_loss_forward(_lib.ptr(img), _lib.ptr(pred), _lib.ptr(weight), _lib.ptr(sil), V, pv_img, pv_sil,
                                              _lib.ptr(part), _lib.stream_ptr()), "fusion_loss")
        sums = part.sum(0)
        ctx.save_for_backward(img, sil, pred, weight)
        ctx.cfg = (lam, V, pv_img, pv_sil)
        terms = torch.stack((sums[0] / (V * pv_img), sums[1] / (V * pv_sil), sums[2] / (V * pv_sil)))
        ctx.mark_non_differentiable(terms)
        return terms[0] + lam[0] * terms[1] + lam[1] * terms[2], terms

    @staticmethod
    def backward(ctx, g, _g_terms):
        img, sil, pred, weight = ctx.saved_tensors
        (lo, le), V, pv_img, pv_sil = ctx.cfg
        g = _f32c(g)
        g_img, g_sil = torch.empty_like(img), torch.empty_like(sil)
        _lib.check(_lib.lib().sf_fusion_loss_backward(_lib.ptr(img), _lib.ptr(pred), _lib.ptr(weight), _lib.ptr(sil), V, pv_img, pv_sil,
                                                      1.0 / (V * pv_img), lo / (V * pv_sil), le / (V * pv_sil), _lib.ptr(g),
                                                      _lib.ptr(g_img), _lib.ptr(g_sil), _lib.stream_ptr()), "fusion_loss_backward")
        return g_img, g_sil, None, None, None


def fusion_loss(img, sil, pred_img, fusion_weight, lambda_opacity=1e-3, lambda_entropy=1e-3, return_terms=False):
    """Stage-B pixel-space loss: fusion_weight = (1 - alpha_cumprod) per view, `pred_img` the decoded x0 prediction (no gradient)."""
    loss, terms = _FusionLoss.apply(img, sil, pred_img.detach(), fusion_weight.detach(), (lambda_opacity, lambda_entropy))
    return (loss, terms) if return_terms else loss

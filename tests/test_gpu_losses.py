"""GPU parity of the loss glue (sparsefusion_amd/utils/losses.py on csrc/loss_ops.hip) against the torch expressions of the
reference loop (sparsefusion/distillation.py:217-241, :287-288, :310-343), values and gradients, fp32 tolerance 1e-5; and
of the evaluation metrics (utils/common_utils.py:44-64) against the oracle's restatement of scikit-image."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _entropy(s):
    a = s.clamp(1e-5, 1 - 1e-5)
    return (-a * torch.log2(a) - (1 - a) * torch.log2(1 - a)).mean()


def _huber(x, y, scaling=0.1):
    d = (x - y) ** 2
    return ((1 + d / scaling ** 2).clamp(1e-4).sqrt() - 1) * scaling


@pytest.mark.parametrize("shape", [(1, 3, 128, 128), (2, 1, 7, 5), (3, 4, 1, 9)])
def test_upsample2x_matches_interpolate(shape):
    from sparsefusion_amd.utils.losses import upsample2x
    g = torch.Generator().manual_seed(0)
    x = torch.randn(*shape, generator=g).to(DEV).requires_grad_(True)
    xr = x.detach().clone().requires_grad_(True)
    y, yr = upsample2x(x), F.interpolate(xr, scale_factor=2, mode='bilinear')
    assert y.shape == yr.shape and torch.allclose(y, yr, atol=1e-6)
    w = torch.randn(*y.shape, generator=g).to(DEV)
    (y * w).sum().backward()
    (yr * w).sum().backward()
    assert torch.allclose(x.grad, xr.grad, atol=1e-5)


@pytest.mark.parametrize("with_mask", [True, False])
def test_render_loss_matches_torch(with_mask):
    from sparsefusion_amd.utils.losses import render_loss
    g = torch.Generator().manual_seed(1)
    img = torch.rand(2, 3, 64, 64, generator=g).to(DEV).requires_grad_(True)
    sil = (torch.rand(2, 1, 64, 64, generator=g) * 1.2 - 0.1).to(DEV).requires_grad_(True)      # some values outside (1e-5, 1 - 1e-5)
    rgb, mask = torch.rand(2, 3, 64, 64, generator=g).to(DEV), (torch.rand(2, 1, 64, 64, generator=g) > 0.5).float().to(DEV)
    lam = dict(lambda_color=1.0, lambda_sil=0.7, lambda_opacity=1e-3, lambda_entropy=2e-3)
    loss, terms = render_loss(img, sil, rgb, mask if with_mask else None, return_terms=True, **lam)
    (loss * 3.0).backward()
    i2, s2 = img.detach().clone().requires_grad_(True), sil.detach().clone().requires_grad_(True)
    ref = lam["lambda_color"] * _huber(i2, rgb).abs().mean() + (lam["lambda_sil"] * _huber(s2, mask).abs().mean() if with_mask else 0) \
        + lam["lambda_opacity"] * torch.sqrt(s2 ** 2 + .01).mean() + lam["lambda_entropy"] * _entropy(s2)
    (ref * 3.0).backward()
    assert abs(float(loss) - float(ref)) < 1e-5 * max(1.0, abs(float(ref)))
    assert torch.allclose(img.grad, i2.grad, atol=2e-7, rtol=1e-4) and torch.allclose(sil.grad, s2.grad, atol=2e-7, rtol=1e-4)   # grads ~1e-4: fp32 rounding of either chain is ~5e-8
    assert terms.shape == (4,) and abs(float(terms[2]) - float(torch.sqrt(s2 ** 2 + .01).mean())) < 1e-5


def test_fusion_loss_matches_torch():
    from sparsefusion_amd.utils.losses import fusion_loss
    g = torch.Generator().manual_seed(2)
    V = 3
    img = torch.rand(V, 3, 64, 64, generator=g).to(DEV).requires_grad_(True)
    sil = torch.rand(V, 1, 64, 64, generator=g).to(DEV).requires_grad_(True)
    pred, w = torch.rand(V, 3, 64, 64, generator=g).to(DEV), torch.rand(V, generator=g).to(DEV)
    loss = fusion_loss(img, sil, pred, w, lambda_opacity=1e-3, lambda_entropy=1e-3)
    loss.backward()
    i2, s2 = img.detach().clone().requires_grad_(True), sil.detach().clone().requires_grad_(True)
    ref = (w.view(-1, 1, 1, 1) * (i2 - pred).abs()).mean() + 1e-3 * torch.sqrt(s2 ** 2 + .01).mean() + 1e-3 * _entropy(s2)
    ref.backward()
    assert abs(float(loss) - float(ref)) < 1e-6 and torch.allclose(img.grad, i2.grad, atol=1e-8, rtol=1e-5)
    assert torch.allclose(sil.grad, s2.grad, atol=2e-8, rtol=1e-4)
    with pytest.raises(RuntimeError):
        fusion_loss(img.cpu(), sil.cpu(), pred.cpu(), w.cpu())                     # no CPU path


def test_metrics_match_oracle():
    from oracle import metrics_ref
    from sparsefusion_amd.utils.common_utils import get_metrics, huber, normalize, unnormalize
    rng = np.random.default_rng(3)
    gt = rng.random((64, 48, 3))
    pred = np.clip(gt + 0.05 * rng.standard_normal(gt.shape), 0, 1)
    s, p = get_metrics(pred, gt, device=DEV)
    assert abs(s - metrics_ref.ssim(pred, gt)) < 1e-9 and abs(p - metrics_ref.psnr(pred, gt)) < 1e-9
    x = torch.rand(4, 5)
    assert torch.allclose(unnormalize(normalize(x)), x, atol=1e-6) and torch.allclose(huber(x, x), torch.zeros_like(x))

"""Drop-ins for the pieces of the reference's utils/common_utils.py the distillation loop touches: `normalize`, `unnormalize`,
`huber` (:9-19, :183-190) and the evaluation metrics `get_metrics` (:44-64, used at sparsefusion/distillation.py:428).
The reference computes SSIM / PSNR with scikit-image (absent here, unpinned): the published algorithm of
`skimage.metrics.structural_similarity` with its defaults is restated on the GPU -- uniform 7x7 window, sample covariance
(N / (N - 1)), K1 = 0.01, K2 = 0.03, statistics over the window-valid region, mean over pixels and channels."""
import numpy as np
import torch
import torch.nn.functional as F


def normalize(x):
    """[0, 1] -> [-1, 1]"""
    return torch.clip(x * 2 - 1.0, -1.0, 1.0)


def unnormalize(x):
    """[-1, 1] -> [0, 1]"""
    return torch.clip((x + 1.0) / 2.0, 0.0, 1.0)


def huber(x, y, scaling=0.1):
    """Smooth-L1 of the reference (elementwise; the fused loss kernels of utils/losses.py use the same definition)."""
    diff_sq = (x - y) ** 2
    return ((1 + diff_sq / (scaling ** 2)).clamp(1e-4).sqrt() - 1) * float(scaling)


def _as_chw(img, device):
    t = torch.as_tensor(np.asarray(img) if not torch.is_tensor(img) else img, dtype=torch.float64)
    if t.dim() != 3 or t.shape[-1] not in (1, 3):
        raise ValueError("get_metrics expects (H, W, 3) images in [0, 1]")
    return t.to(device).permute(2, 0, 1).contiguous()


def psnr(pred, gt, data_range=1.0):
    mse = torch.mean((pred - gt) ** 2)
    return 10.0 * torch.log10(data_range ** 2 / mse)


def ssim(pred, gt, data_range=1.0, win_size=7, K1=0.01, K2=0.03):
    """pred, gt [C, H, W] float64 -> mean structural similarity (skimage defaults, channel_axis=-1)."""
    if min(pred.shape[-2:]) < win_size:
        raise ValueError("win_size exceeds image extent")
    NP = win_size * win_size
    cov_norm = NP / (NP - 1.0)
    x, y = pred[:, None], gt[:, None]
    mu = lambda t: F.avg_pool2d(t, win_size, stride=1)
    ux, uy = mu(x), mu(y)
    vx = cov_norm * (mu(x * x) - ux * ux)
    vy = cov_norm * (mu(y * y) - uy * uy)
    vxy = cov_norm * (mu(x * y) - ux * uy)
    C1, C2 = (K1 * data_range) ** 2, (K2 * data_range) ** 2
    S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux * ux + uy * uy + C1) * (vx + vy + C2))
    return S.mean()


def get_metrics(pred, gt, use_lpips=False, loss_fn_vgg=None, device=None):
    """Reference signature: (H, W, 3) arrays in [0, 1] -> (ssim, psnr[, lpips]) as Python floats."""
    dev = torch.device(device) if device is not None else torch.device("cuda:0" if torch.cuda.is_available() else "cpu")
    p, g = _as_chw(pred, dev), _as_chw(gt, dev)
    s, q = float(ssim(p, g)), float(psnr(p, g))
    if use_lpips:
        if loss_fn_vgg is None:
            from ..lpips import LPIPS
            loss_fn_vgg = LPIPS(net='vgg').to(dev)
        with torch.no_grad():
            lp = float(loss_fn_vgg(g.float()[None] * 2 - 1.0, p.float()[None] * 2 - 1.0).reshape(-1)[0])
        return s, q, lp
    return s, q

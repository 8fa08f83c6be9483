"""CPU restatement of the evaluation metrics the reference takes from scikit-image (utils/common_utils.py:44-64:
`skimage.metrics.structural_similarity(pred, gt, channel_axis=-1, data_range=1)` and `peak_signal_noise_ratio`).
TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED: scikit-image is absent from this image and from /root/reference (unpinned in
requirements); the published algorithm of its defaults is restated with scipy.ndimage.uniform_filter exactly as the
library does (uniform 7x7 window, reflect borders, crop of (win - 1) // 2, sample covariance, K1 = 0.01, K2 = 0.03)."""
import numpy as np
from scipy.ndimage import uniform_filter


def psnr(pred, gt, data_range=1.0):
    mse = np.mean((np.asarray(pred, np.float64) - np.asarray(gt, np.float64)) ** 2)
    return 10 * np.log10(data_range ** 2 / mse)


def ssim(pred, gt, data_range=1.0, win_size=7, K1=0.01, K2=0.03):
    pred, gt = np.asarray(pred, np.float64), np.asarray(gt, np.float64)
    vals = []
    for ch in range(pred.shape[-1]):
        x, y = pred[..., ch], gt[..., ch]
        NP = win_size ** 2
        cov_norm = NP / (NP - 1)
        ux, uy = uniform_filter(x, size=win_size), uniform_filter(y, size=win_size)
        uxx, uyy, uxy = uniform_filter(x * x, size=win_size), uniform_filter(y * y, size=win_size), uniform_filter(x * y, size=win_size)
        vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
        C1, C2 = (K1 * data_range) ** 2, (K2 * data_range) ** 2
        S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2))
        pad = (win_size - 1) // 2
        vals.append(S[pad:-pad, pad:-pad].mean())
    return float(np.mean(vals))

"""Shared helpers for the NGP parity tests."""
import numpy as np
import torch

from oracle import ngp_ref

BOUND = 4


def params_from_cfg(cfg):
    return ngp_ref.init_params(bound=BOUND, seed=cfg["seed"], table_std=cfg["table_std"], sigma_bias=cfg["sigma_bias"])


def grad_leaf(p):
    return {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 and "aabb" not in k else v)
            for k, v in p.items()}


def log2_scale():
    return float(np.log2(ngp_ref.per_level_scale(BOUND)))


def psnr(a, b):
    mse = torch.mean((a.double() - b.double()) ** 2).item()
    return 200.0 if mse == 0 else -10.0 * np.log10(mse)


def live_rays(mask):
    return mask.bool()

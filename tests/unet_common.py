"""Shared helpers for the UNet / PLMS parity tests."""
import json
import os

import torch

from oracle import unet_ref

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CONFIGS = {"canonical": unet_ref.CANONICAL, "small": unet_ref.SMALL, "medium": unet_ref.MEDIUM}


def spec(name):
    return [(k, tuple(s)) for k, s in json.load(open(os.path.join(GOLD, f"unet_keys_{name}.json")))]


def state(name, seed=0):
    return unet_ref.init_state(spec(name), seed=seed)


def inputs(cfg, B, seed):
    """Must match tests/golden/make_golden_unet.py::inputs."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, cfg["channels"], 32, 32, generator=g)
    cond = torch.randn(B, cfg["cond_images_channels"], 32, 32, generator=g)
    t = torch.tensor([0.37, 0.05, 0.9, 0.6][:B])
    return x, unet_ref.log_snr(t), cond


def rel_err(a, b):
    return ((a - b).norm() / b.norm()).item()


def cosine(a, b):
    return torch.nn.functional.cosine_similarity(a.flatten().double(), b.flatten().double(), dim=0).item()

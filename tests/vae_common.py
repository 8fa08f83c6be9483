"""Shared helpers for the SD-VAE parity tests."""
import json
import os

import torch

from oracle import vae_ref

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CONFIGS = {"canonical": vae_ref.CANONICAL, "small": vae_ref.SMALL}


def spec(name):
    return [(k, tuple(s)) for k, s in json.load(open(os.path.join(GOLD, f"vae_keys_{name}.json")))]


def state(name, seed=0):
    return vae_ref.init_state(spec(name), seed=seed)


def inputs(cfg, B, seed):
    """Must match tests/golden/make_golden_vae.py::inputs."""
    g = torch.Generator().manual_seed(seed)
    R = cfg["resolution"]
    f = 2 ** (len(cfg["ch_mult"]) - 1)
    img = torch.rand(B, cfg["in_channels"], R, R, generator=g) * 2 - 1
    z = torch.randn(B, cfg["z_channels"], R // f, R // f, generator=g)
    return img, z


def ddconfig(cfg):
    return {k: v for k, v in cfg.items() if k != "embed_dim"}


def rel_err(a, b):
    return ((a - b).norm() / b.norm()).item()


def cosine(a, b):
    return torch.nn.functional.cosine_similarity(a.flatten().double(), b.flatten().double(), dim=0).item()

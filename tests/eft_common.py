"""Shared scene builder for the EFT (row E1) tests; must match tests/golden/make_golden_eft.py::scene."""
import json
import math
import os

import torch

from oracle import eft_ref
from oracle.ref_loader import PinholeCameras          # plain-torch camera stand-in (no reference import involved)

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def spec():
    return [(k, tuple(s)) for k, s in json.load(open(os.path.join(GOLD, "eft_keys.json")))]


def state(seed=0):
    return eft_ref.init_state(spec(), seed=seed)


def scene(NC, R, N, D, seed):
    g = torch.Generator().manual_seed(seed)
    Rs, Ts = [], []
    for i in range(NC):
        a = 0.45 * i - 0.3
        c, s = math.cos(a), math.sin(a)
        Rs.append(torch.tensor([[c, 0, -s], [0, 1, 0], [s, 0, c]], dtype=torch.float32))
        Ts.append(torch.tensor([0.05 * i, -0.02 * i, 4.0]))
    cams = PinholeCameras(torch.stack(Rs), torch.stack(Ts), torch.full((NC, 2), 2.2))
    images = torch.rand(NC, 3, R, R, generator=g)
    o = torch.tensor([[0.3, 0.1, -4.0]]).expand(N, 3).contiguous()
    d = torch.randn(N, 3, generator=g) * 0.12 + torch.tensor([0.0, 0.0, 1.0])
    d = d * 1.4
    lengths = (torch.linspace(1.8, 4.0, D)[None] + 0.05 * torch.rand(N, 1, generator=g)).contiguous()
    return cams, images, o, d.contiguous(), lengths


def rel_err(a, b):
    return ((a - b).norm() / b.norm()).item()

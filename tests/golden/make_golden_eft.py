"""EFT (row E1) golden vectors from the REAL reference module (dev container only; see make_golden.py).
Weights are rebuilt on both sides from oracle.eft_ref.init_state(seed); only key names/shapes and outputs are committed."""
import json
import math
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import eft_ref, ref_loader  # noqa: E402


def scene(NC, R, N, D, seed):
    """NC input views on a circle looking at the origin, N query rays from another viewpoint, D depths per ray."""
    g = torch.Generator().manual_seed(seed)
    Rs, Ts = [], []
    for i in range(NC):
        a = 0.45 * i - 0.3
        c, s = math.cos(a), math.sin(a)
        Rs.append(torch.tensor([[c, 0, -s], [0, 1, 0], [s, 0, c]], dtype=torch.float32))
        Ts.append(torch.tensor([0.05 * i, -0.02 * i, 4.0]))
    cams = ref_loader.PinholeCameras(torch.stack(Rs), torch.stack(Ts), torch.full((NC, 2), 2.2))
    images = torch.rand(NC, 3, R, R, generator=g)
    o = torch.tensor([[0.3, 0.1, -4.0]]).expand(N, 3).contiguous()
    d = torch.randn(N, 3, generator=g) * 0.12 + torch.tensor([0.0, 0.0, 1.0])
    d = d * 1.4                                                      # non-unit directions, as the pytorch3d ray bundles
    lengths = (torch.linspace(1.8, 4.0, D)[None] + 0.05 * torch.rand(N, 1, generator=g)).contiguous()
    return cams, images, o, d.contiguous(), lengths


CASES = {"small": dict(NC=3, R=64, N=48, D=20, seed=1), "six_views": dict(NC=6, R=128, N=40, D=20, seed=2)}


def main():
    eft, RayBundle = ref_loader.reference_eft()
    eft.eval()
    spec = [(k, list(v.shape)) for k, v in eft.state_dict().items()]
    json.dump(spec, open(os.path.join(HERE, "eft_keys.json"), "w"))
    sd = eft_ref.init_state([(k, tuple(s)) for k, s in spec], seed=0)
    eft.load_state_dict(sd, strict=True)
    out = {}
    for name, c in CASES.items():
        cams, images, o, d, lengths = scene(**c)
        with torch.no_grad():
            rgb, f3, _ = eft(RayBundle(o, d, lengths, None), input_cameras=cams, input_rgb=images)
            # the chunked entry point the pre-pass uses (distillation.py:106: n_batches=16) must agree
            rgb_b, f3_b, _ = eft.batched_forward(RayBundle(o[None], d[None], lengths[None], None), n_batches=4)
        assert torch.allclose(rgb_b[0], rgb, atol=1e-6) and torch.allclose(f3_b[0], f3, atol=1e-5)
        out[name] = dict(cfg=c, rgb=rgb.clone(), f3=f3.clone())
        print(name, "rgb std %.4f f3 std %.4f" % (rgb.std().item(), f3.std().item()))
    torch.save(out, os.path.join(HERE, "eft_forward.pt"))
    print("params %.2fM, %d keys" % (sum(torch.Size(s).numel() for _, s in spec) / 1e6, len(spec)))


if __name__ == "__main__":
    main()

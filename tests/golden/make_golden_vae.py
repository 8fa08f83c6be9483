"""SD-VAE encode / decode golden vectors from the REAL reference classes (dev container only; see make_golden.py).

Weights are not stored: both sides rebuild them from oracle.vae_ref.init_state(seed).  Only the reference's
parameter names/shapes (vae_keys_*.json) and the outputs are committed."""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ref_loader, vae_ref  # noqa: E402

CONFIGS = {"canonical": vae_ref.CANONICAL, "small": vae_ref.SMALL}


def inputs(cfg, B, seed):
    g = torch.Generator().manual_seed(seed)
    R = cfg["resolution"]
    f = 2 ** (len(cfg["ch_mult"]) - 1)
    img = torch.rand(B, cfg["in_channels"], R, R, generator=g) * 2 - 1
    z = torch.randn(B, cfg["z_channels"], R // f, R // f, generator=g)
    return img, z


def main():
    out = {}
    for name, cfg in CONFIGS.items():
        net = ref_loader.reference_vae(cfg).eval()
        spec = [(k, list(v.shape)) for k, v in net.state_dict().items()]
        json.dump(spec, open(os.path.join(HERE, f"vae_keys_{name}.json"), "w"))
        sd = vae_ref.init_state([(k, tuple(s)) for k, s in spec], seed=0)
        net.load_state_dict(sd, strict=True)
        B = 1 if name == "canonical" else 2
        img, z = inputs(cfg, B, seed=7)
        with torch.no_grad():
            lat = net.encode_mode(img)
            dec = net.decode(z)
        # the canonical decode is [1,3,256,256]: keep a strided sample + moments to bound the fixture size
        out[name] = dict(B=B, input_seed=7, state_seed=0, latents=lat.clone(),
                         decoded=(dec.clone() if name == "small" else dec[:, :, ::4, ::4].clone()),
                         decoded_mean=dec.mean().item(), decoded_std=dec.std().item())
        print(name, "params %.2fM" % (sum(v.numel() for v in sd.values()) / 1e6), "latent std %.4f" % lat.std().item(),
              "decoded std %.4f" % dec.std().item())
    torch.save(out, os.path.join(HERE, "vae_forward.pt"))


if __name__ == "__main__":
    main()

"""SD-VAE on the HIP plan vs the CPU oracle (oracle/vae_ref.py, itself pinned to the reference classes) and vs
the committed reference goldens.  bf16 MFMA operands / fp32 accumulation against an fp32 reference: the stated
tolerance is relative L2 <= 2e-2 and cosine >= 0.9995 per tensor (measured values are printed)."""
import pytest
import torch

from oracle import vae_ref
from vae_common import CONFIGS, GOLD, cosine, ddconfig, inputs, rel_err, state

pytestmark = pytest.mark.gpu

REL, COS = 2e-2, 0.9995


def _vae(name):
    from sparsefusion_amd.vae import AutoencoderKL
    cfg = CONFIGS[name]
    net = AutoencoderKL(ddconfig=ddconfig(cfg), embed_dim=cfg["embed_dim"])
    missing = net.load_state_dict(state(name), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return net.cuda(), cfg


@pytest.mark.parametrize("name", ["small", "canonical"])
def test_encode_decode_match_reference_golden(name):
    g = torch.load(f"{GOLD}/vae_forward.pt")[name]
    net, cfg = _vae(name)
    img, z = inputs(cfg, g["B"], g["input_seed"])
    lat = net.encode(img.cuda()).mode().cpu()
    dec = net.decode(z.cuda()).cpu()
    got = dec if name == "small" else dec[:, :, ::4, ::4]
    print(f"{name}: latents rel {rel_err(lat, g['latents']):.2e} cos {cosine(lat, g['latents']):.6f}; "
          f"decoded rel {rel_err(got, g['decoded']):.2e} cos {cosine(got, g['decoded']):.6f}")
    assert lat.shape == g["latents"].shape and torch.isfinite(lat).all() and torch.isfinite(dec).all()
    assert rel_err(lat, g["latents"]) < REL and cosine(lat, g["latents"]) > COS
    assert rel_err(got, g["decoded"]) < REL and cosine(got, g["decoded"]) > COS
    assert abs(dec.mean().item() - g["decoded_mean"]) < 5e-3 and abs(dec.std().item() - g["decoded_std"]) < 5e-3


def test_small_matches_oracle_other_seed_and_batch():
    """Fresh weights / inputs (not the golden ones), B = 3; per-sample results do not depend on the batch beyond rounding."""
    from sparsefusion_amd.vae import AutoencoderKL
    cfg = CONFIGS["small"]
    sd = state("small", seed=3)
    net = AutoencoderKL(ddconfig=ddconfig(cfg), embed_dim=cfg["embed_dim"])
    net.load_state_dict(sd)
    net = net.cuda()
    img, z = inputs(cfg, 3, seed=21)
    with torch.no_grad():
        lat_ref, dec_ref = vae_ref.encode_mode(sd, cfg, img), vae_ref.decode(sd, cfg, z)
    post = net.encode(img.cuda())
    lat, dec = post.mode().cpu(), net.decode(z.cuda()).cpu()
    assert rel_err(lat, lat_ref) < REL and rel_err(dec, dec_ref) < REL
    assert post.sample().shape == lat.shape and post.logvar.max() <= 20.0
    # a sample evaluated alone takes other tile shapes / kernels (the planner switches to the LDS-tiled conv by tile
    # count): different fp32 summation order in front of every bf16 rounding, so it agrees to the bf16 tolerance
    one = net.encode(img[1:2].cuda()).mode().cpu()
    assert rel_err(one, lat[1:2]) < REL and rel_err(one, lat_ref[1:2]) < REL
    one = net.decode(z[2:3].cuda()).cpu()
    assert rel_err(one, dec[2:3]) < REL and rel_err(one, dec_ref[2:3]) < REL


def test_vae_rejects_unsupported():
    from sparsefusion_amd.vae import AutoencoderKL
    with pytest.raises(NotImplementedError):
        AutoencoderKL(ddconfig=dict(ddconfig(CONFIGS["small"]), attn_resolutions=[16]))
    with pytest.raises(NotImplementedError):
        AutoencoderKL(ddconfig=dict(ddconfig(CONFIGS["small"]), ch=64))
    net = AutoencoderKL(ddconfig=ddconfig(CONFIGS["small"]))
    with pytest.raises((RuntimeError, AssertionError)):
        net.encode(torch.zeros(1, 3, 32, 32))                # CPU tensor: no fallback

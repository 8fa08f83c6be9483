"""Pin the SD-VAE oracle to the REAL reference: golden outputs produced by tests/golden/make_golden_vae.py
(reference Encoder / Decoder classes + quant convs on CPU)."""
import pytest
import torch

from oracle import vae_ref
from vae_common import CONFIGS, GOLD, ddconfig, inputs, spec, state


@pytest.mark.parametrize("name", ["small", "canonical"])
def test_vae_restatement_matches_reference(name):
    g = torch.load(f"{GOLD}/vae_forward.pt")[name]
    cfg = CONFIGS[name]
    sd = state(name, g["state_seed"])
    img, z = inputs(cfg, g["B"], g["input_seed"])
    with torch.no_grad():
        lat = vae_ref.encode_mode(sd, cfg, img)
        dec = vae_ref.decode(sd, cfg, z)
    assert lat.abs().max() > 0.3 and dec.abs().max() > 0.3
    assert torch.allclose(lat, g["latents"], rtol=1e-4, atol=2e-5), (lat - g["latents"]).abs().max()
    want = g["decoded"] if name == "small" else dec[:, :, ::4, ::4] * 0 + g["decoded"]
    got = dec if name == "small" else dec[:, :, ::4, ::4]
    assert torch.allclose(got, want, rtol=1e-4, atol=5e-5), (got - want).abs().max()
    assert abs(dec.mean().item() - g["decoded_mean"]) < 1e-5 and abs(dec.std().item() - g["decoded_std"]) < 1e-5


def test_param_spec_matches_reference_keys():
    from sparsefusion_amd.vae import vae_param_spec
    for name, cfg in CONFIGS.items():
        ref = spec(name)
        assert [k for k, _ in vae_ref.vae_param_spec(cfg)] == [k for k, _ in ref]        # same names, same order
        assert dict(vae_ref.vae_param_spec(cfg)) == dict(ref)
        mine = vae_param_spec(**ddconfig(cfg), embed_dim=cfg["embed_dim"])
        assert [k for k, _ in mine] == [k for k, _ in ref] and dict(mine) == dict(ref)
    n = sum(torch.Size(s).numel() for s in dict(spec("canonical")).values())
    assert n == 83653863                                            # 34.16 M encoder + 49.49 M decoder + quant convs

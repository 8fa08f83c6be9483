"""Pin the UNet / PLMS oracle to the REAL reference: golden outputs produced by
tests/golden/make_golden_unet.py (reference Unet.forward_with_cond_scale / PLMSSampler.sample on CPU)."""
import pytest
import torch

from oracle import unet_ref
from unet_common import CONFIGS, GOLD, inputs, spec, state


@pytest.mark.parametrize("name", ["small", "canonical"])
def test_unet_restatement_matches_reference(name):
    g = torch.load(f"{GOLD}/unet_forward.pt")[name]
    sd = state(name, g["state_seed"])
    x, ls, cond = inputs(CONFIGS[name], g["B"], g["input_seed"])
    with torch.no_grad():
        y = unet_ref.unet_forward(sd, x, ls, cond)
    assert y.abs().max() > 0.5                                # final_conv is not zero: the test is not vacuous
    assert torch.allclose(y, g["y"], rtol=1e-4, atol=2e-5), (y - g["y"]).abs().max()


def test_unet_and_plms_restatement_match_reference_medium():
    """dim 128 (oracle/unet_ref.MEDIUM: the smallest configuration the HIP planner runs entirely on the canonical plan's fused kernels):
    the reference's forward and its 51-eval PLMS trajectory (tests/golden/make_golden_unet.py --medium)."""
    G = torch.load(f"{GOLD}/unet_medium.pt")
    g = G["forward"]
    sd = state("medium", g["state_seed"])
    x, ls, cond = inputs(CONFIGS["medium"], g["B"], g["input_seed"])
    with torch.no_grad():
        y = unet_ref.unet_forward(sd, x, ls, cond)
    assert y.abs().max() > 0.5 and torch.allclose(y, g["y"], rtol=1e-4, atol=2e-5), (y - g["y"]).abs().max()
    r = G["plms"]
    gg = torch.Generator().manual_seed(r["input_seed"])
    lat = 0.5 * torch.randn(2, 4, 32, 32, generator=gg)
    cond = torch.randn(2, CONFIGS["medium"]["cond_images_channels"], 32, 32, generator=gg)
    torch.manual_seed(r["noise_seed"])
    noises = [torch.randn(2, 4, 32, 32) for _ in range(unet_ref.plms_noise_count(r["max_thres"]))]
    with torch.no_grad():
        img, xn, nz, acp, ev = unet_ref.plms_sample(lambda a, b: unet_ref.unet_forward(sd, a, b, cond), lat, r["max_thres"], noises)
    assert ev == 51 and torch.equal(nz, r["noise"]) and torch.allclose(xn, r["x_noisy"], atol=1e-6)
    assert torch.allclose(img, r["img"], rtol=1e-3, atol=2e-4), (img - r["img"]).abs().max()


def test_param_spec_matches_reference_keys():
    from sparsefusion_amd.unet import unet_param_spec
    for name, cfg in CONFIGS.items():
        mine = dict(unet_param_spec(**cfg))
        ref = dict(spec(name))
        assert mine == ref and len(ref) == 477
    assert sum(torch.Size(s).numel() for s in dict(spec("canonical")).values()) == 400675357


@pytest.mark.parametrize("max_thres,evals", [(0.005, 0), (0.06, 7), (0.995, 51)])
def test_plms_restatement_matches_reference(max_thres, evals):
    r = torch.load(f"{GOLD}/plms_sample.pt")[max_thres]
    sd = state("small")
    gg = torch.Generator().manual_seed(r["input_seed"])
    lat = 0.5 * torch.randn(2, 4, 32, 32, generator=gg)
    cond = torch.randn(2, 60, 32, 32, generator=gg)
    torch.manual_seed(r["noise_seed"])
    noises = [torch.randn(2, 4, 32, 32) for _ in range(unet_ref.plms_noise_count(max_thres))]
    with torch.no_grad():
        img, xn, nz, acp, ev = unet_ref.plms_sample(lambda x, ls: unet_ref.unet_forward(sd, x, ls, cond), lat, max_thres, noises)
    assert ev == evals                                        # n_steps + 1 UNet evaluations, 0 when n_steps == 0
    assert torch.equal(nz, r["noise"]) and torch.allclose(xn, r["x_noisy"], atol=1e-6)
    assert torch.allclose(acp, r["alpha_cumprod"], atol=1e-7)
    assert torch.allclose(img, r["img"], rtol=1e-3, atol=2e-4), (img - r["img"]).abs().max()


def test_schedule_scalars_match_oracle():
    from sparsefusion_amd.plms import step_coefficients
    for t, tn in ((0.5, 0.49), (0.99, 0.97), (0.02, 0.0)):
        c = step_coefficients(t, tn, 10.0)
        a, s = unet_ref.alpha_sigma(unet_ref.log_snr(torch.tensor(t)))
        an, sn = unet_ref.alpha_sigma(unet_ref.log_snr(torch.tensor(tn)))
        assert abs(c[0] - a.item()) < 1e-7 and abs(c[1] - s.item()) < 1e-7 and abs(c[2] - an.item()) < 1e-7
        assert (c[4] == 0.0) == (tn == 0.0)

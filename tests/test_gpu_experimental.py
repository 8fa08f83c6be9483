"""GPU parity of the planner switches of the SD-VAE and LPIPS plans that the default suite would otherwise never run (r04: they
are plain module attributes -- the SF_* environment switches of the r01-r03 A/B runs are retired -- and every one of them has a
parity case here, in the regular `pytest -m gpu` run; the UNet's are in tests/test_gpu_unet.py::test_plan_switches_match_oracle,
the NGP field cache in tests/test_gpu_ngp.py::test_field_cache_equals_regather)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(a, b):
    return float((a - b).norm() / b.norm())


@pytest.mark.parametrize("attr", ["gn_epilogue", "conv_twin"])
def test_vae_plan_switches_match_the_default_plan_and_the_oracle(attr):
    """`gn_epilogue = False`: GroupNorm statistics from a pass over the tensor (k_gn_stats_px) instead of the producing conv's epilogue
    sums (k_conv_lds_gn / halo + k_gn_finalize); `conv_twin = False`: Upsample / Downsample / nin_shortcut convs read the fp32 block
    output instead of the operand-type twin.  Same results up to the bf16 decorrelation of the path (a last-ulp difference in a
    statistic re-rounds operands downstream: measured 4.8e-3), and both within the reference tolerance."""
    from oracle import vae_ref
    from vae_common import CONFIGS, ddconfig, inputs, state
    from sparsefusion_amd.vae import AutoencoderKL
    cfg = CONFIGS["canonical"]
    sd = state("canonical")
    img, z = inputs(cfg, 1, 5)
    outs = {}
    for on in (True, False):
        net = AutoencoderKL(ddconfig=ddconfig(cfg), embed_dim=cfg["embed_dim"])
        net.load_state_dict(sd)
        setattr(net, attr, on)
        net = net.to(DEV)
        outs[on] = (net.encode(img.to(DEV)).mode().cpu(), net.decode(z.to(DEV)).cpu())
    with torch.no_grad():
        lat_ref, dec_ref = vae_ref.encode_mode(sd, cfg, img), vae_ref.decode(sd, cfg, z)
    for k, ref in ((0, lat_ref), (1, dec_ref)):
        assert torch.isfinite(outs[False][k]).all() and _rel(outs[False][k], outs[True][k]) < 1e-2
        assert _rel(outs[False][k], ref) < 2e-2 and _rel(outs[True][k], ref) < 2e-2


def test_lpips_fp32_links_match_the_operand_type_twins():
    """`conv_twin = False`: the conv -> conv links inside a VGG slice read the fp32 ReLU output instead of its operand-type twin
    (both are rounded to the operand type before the MFMA: same value, other kernels)."""
    from oracle import lpips_ref
    from sparsefusion_amd.lpips import LPIPS
    sd = lpips_ref.init_state(seed=0)
    g = torch.Generator().manual_seed(3)
    a, b = torch.rand(1, 3, 64, 64, generator=g), torch.rand(1, 3, 64, 64, generator=g)
    res = {}
    for on in (True, False):
        net = LPIPS(net='vgg')
        net.load_state_dict(sd, strict=True)
        net.conv_twin = on
        net = net.to(DEV)
        p = a.to(DEV).requires_grad_(True)
        d = net(p, b.to(DEV), normalize=True)
        d.sum().backward()
        res[on] = (d.detach().cpu(), p.grad.cpu())
    assert _rel(res[False][0], res[True][0]) < 2e-3 and _rel(res[False][1], res[True][1]) < 3e-2

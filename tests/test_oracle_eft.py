"""Pin the EFT restatement (oracle/eft_ref.py) to the REAL reference module: goldens from tests/golden/make_golden_eft.py
(reference sparsefusion/eft.py on CPU; only its third-party imports -- torchvision resnet18, pytorch3d ray/camera types --
are stand-ins, see oracle/ref_loader.reference_eft)."""
import pytest
import torch

from oracle import eft_ref
from eft_common import GOLD, scene, spec, state


@pytest.mark.parametrize("name", ["small", "six_views"])
def test_eft_restatement_matches_reference(name):
    g = torch.load(f"{GOLD}/eft_forward.pt")[name]
    sd = state(0)
    cams, images, o, d, lengths = scene(**g["cfg"])
    with torch.no_grad():
        rgb, f3 = eft_ref.eft_forward(sd, cams, images, o, d, lengths)
    assert rgb.shape == g["rgb"].shape and f3.shape == g["f3"].shape and f3.abs().max() > 1
    assert torch.allclose(rgb, g["rgb"], atol=2e-6), (rgb - g["rgb"]).abs().max()
    assert torch.allclose(f3, g["f3"], rtol=1e-4, atol=3e-5), (f3 - g["f3"]).abs().max()


def test_eft_spec_and_harmonic_layout():
    s = dict(spec())
    assert len(s) == 278 and s["t1.pre.0.weight"] == (256, 606) and s["t2.pre.0.weight"] == (256, 425)
    assert s["t3.pre.0.weight"] == (256, 412) and s["encoder_model.layer3.1.conv2.weight"] == (256, 256, 3, 3)
    x = torch.tensor([[0.5, -1.0]])
    h = eft_ref.harmonic(x)                                    # sin block (dim-major, 6 octaves), cos block, then x
    assert h.shape == (1, 26)
    assert torch.allclose(h[0, :6], torch.sin(0.5 * 2.0 ** torch.arange(6))) and torch.allclose(h[0, 12:18], torch.cos(0.5 * 2.0 ** torch.arange(6)))
    assert torch.equal(h[0, 24:], x[0])

"""Sanity of the LPIPS restatement (oracle/lpips_ref.py).  The `lpips` package is absent: nothing to pin against
(PARITY UNPINNED); these are the properties the published definition guarantees."""
import torch

from oracle import lpips_ref


def test_lpips_restatement_properties():
    spec = lpips_ref.lpips_param_spec()
    assert len(spec) == 13 * 2 + 5 and spec[0][0] == "net.slice1.0.weight" and spec[-1] == ("lin4.model.1.weight", (1, 512, 1, 1))
    assert sum(torch.Size(s).numel() for _, s in spec) == 14714688 + 1472          # VGG16 conv stack + the five lin layers
    sd = lpips_ref.init_state(0)
    g = torch.Generator().manual_seed(0)
    a, b = torch.rand(2, 3, 32, 32, generator=g), torch.rand(2, 3, 32, 32, generator=g)
    with torch.no_grad():
        dab, dba, daa = lpips_ref.lpips(sd, a, b), lpips_ref.lpips(sd, b, a), lpips_ref.lpips(sd, a, a)
        mid = lpips_ref.lpips(sd, a, 0.5 * (a + b))
    assert dab.shape == (2, 1, 1, 1) and (dab > 0).all()
    assert torch.allclose(dab, dba, rtol=1e-5) and float(daa.abs().max()) == 0.0     # symmetric, zero on the diagonal
    assert (mid < dab).all()                                                        # closer image, smaller distance
    # normalize=True is exactly the [0,1] -> [-1,1] map of external_utils.py:37-39
    with torch.no_grad():
        assert torch.allclose(lpips_ref.lpips(sd, a, b, normalize=True), lpips_ref.lpips(sd, 2 * a - 1, 2 * b - 1, normalize=False))

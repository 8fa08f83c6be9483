"""LPIPS(net='vgg') on the HIP plan vs the torch-fp32 restatement of the published algorithm (oracle/lpips_ref.py).
The `lpips` package and the pretrained VGG16 are not available (parity unpinned, see the oracle header): both sides use
the same seeded synthetic weights.  r06 option set_operand("f16"): IEEE-half MFMA operands / fp32 accumulate with a normalised upstream
gradient: distance within 2e-3 (measured 3e-5), gradient within 3e-2 / cosine > 0.9995 (measured 1.7e-2 / 0.99985).  bf16 operands (the
default): distance within 1e-2 relative (measured
3e-4), gradient w.r.t. the rendered image within 8e-2 relative L2 and cosine > 0.997 (measured 5e-2 / 0.9988: 13 conv
layers forward and 13 backward in bf16, plus ReLU masks that flip for pre-activations within bf16 rounding of zero; r04: the
error is localised per feature tap and attributed to operand rounding by the last test of this file)."""
import pytest
import torch

from oracle import lpips_ref

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cos(a, b):
    return torch.nn.functional.cosine_similarity(a.flatten().double(), b.flatten().double(), dim=0).item()


@pytest.mark.parametrize("operand", ["f16", "bf16"])
@pytest.mark.parametrize("B,R", [(1, 64), (2, 32), (1, 256)])
def test_lpips_distance_and_gradient(B, R, operand):
    """r06: set_operand("f16") = the IEEE-half operand build with the upstream gradient normalised to max |g| = 4096 (gradient within 3e-2 of
    the fp32 oracle, measured 1.5-1.7e-2; distance 2-6e-5); "bf16" = the default (8e-2, measured 5e-2)."""
    from sparsefusion_amd.lpips import LPIPS, lpips_param_spec
    assert [k for k, _ in lpips_param_spec()] == [k for k, _ in lpips_ref.lpips_param_spec()]
    sd = lpips_ref.init_state(seed=0)
    net = LPIPS(net='vgg')
    missing = net.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    net = net.to(DEV)
    assert net.operand is None and net.grad_scale == 1.0                       # the default stays bf16 (see LPIPS.__init__ for the measurement)
    if operand == "f16":
        net.set_operand("f16")
        assert net.grad_scale == 4096.0
    g = torch.Generator().manual_seed(R + B)
    base = torch.rand(B, 3, R, R, generator=g)
    pred = (base + 0.15 * torch.randn(B, 3, R, R, generator=g)).clamp(0, 1)
    target = base
    p_ref = pred.clone().requires_grad_(True)
    d_ref = lpips_ref.lpips(sd, p_ref, target, normalize=True)
    w = (torch.rand(B, 1, 1, 1, generator=g) + 0.5) * (0.1 / B)                  # the loop's weighting: 0.1 * mean (distillation.py:312-314)
    (d_ref * w).sum().backward()
    p = pred.to(DEV).requires_grad_(True)
    d = net(p, target.to(DEV), normalize=True)
    assert d.shape == (B, 1, 1, 1)
    (d * w.to(DEV)).sum().backward()
    rel_d = float(((d.detach().cpu() - d_ref.detach()).abs() / d_ref.detach().abs()).max())
    rel_g = float((p.grad.cpu() - p_ref.grad).norm() / p_ref.grad.norm())
    print(f"B={B} R={R}: d={d.flatten().tolist()} ref={d_ref.flatten().tolist()} rel {rel_d:.2e}; grad rel L2 {rel_g:.2e} "
          f"cos {_cos(p.grad.cpu(), p_ref.grad):.5f}")
    assert float(d_ref.detach().min()) > 1e-3                                   # a non-trivial distance
    assert rel_d < (2e-3 if operand == "f16" else 1e-2)
    assert (rel_g < 3e-2 and _cos(p.grad.cpu(), p_ref.grad) > 0.9995) if operand == "f16" else (rel_g < 8e-2 and _cos(p.grad.cpu(), p_ref.grad) > 0.997)
    # identical images: zero distance (and a finite, ~zero gradient)
    q = target.to(DEV).requires_grad_(True)
    z = net(q, target.to(DEV), normalize=True)
    z.sum().backward()
    assert float(z.detach().abs().max()) < 1e-6 and torch.isfinite(q.grad).all()


def test_lpips_gradient_error_is_operand_rounding_and_where_it_comes_from():
    """The gradient bound of the test above (8e-2 against the fp32 oracle, measured 5-6e-2) is the loosest number of the suite.
    This test (a) localises it -- one feature tap at a time (the other four lin layers zeroed on both sides): the error grows with
    the depth of the VGG slice the gradient comes back through -- and (b) attributes it: against the SAME oracle with conv operands
    rounded to bf16 on both passes (oracle/lpips_ref._ConvRounded: what any bf16-operand MFMA path computes) the HIP gradient is
    within 3e-2, i.e. the remaining distance to fp32 is operand rounding (plus ReLU masks that flip within that rounding), not a
    defect of a kernel."""
    from sparsefusion_amd.lpips import LPIPS
    B, R = 1, 64
    sd = lpips_ref.init_state(seed=0)
    g = torch.Generator().manual_seed(R + B)
    base = torch.rand(B, 3, R, R, generator=g)
    pred = (base + 0.15 * torch.randn(B, 3, R, R, generator=g)).clamp(0, 1)

    def grads(state, heads):
        pr = pred.clone().requires_grad_(True)
        lpips_ref.lpips(state, pr, base, normalize=True, heads=heads).sum().backward()
        pe = pred.clone().requires_grad_(True)
        lpips_ref.lpips(state, pe, base, normalize=True, heads=heads, operand_dtype=torch.bfloat16).sum().backward()
        net = LPIPS(net='vgg')
        net.load_state_dict(state, strict=True)
        net = net.to(DEV)
        p = pred.to(DEV).requires_grad_(True)
        net(p, base.to(DEV), normalize=True).sum().backward()
        return p.grad.cpu(), pr.grad, pe.grad

    rel = lambda a, b: float((a - b).norm() / b.norm())
    per_head = []
    for k in range(5):
        st = {n: (torch.zeros_like(t) if n.startswith("lin") and not n.startswith(f"lin{k}.") else t) for n, t in sd.items()}
        gh, g32, gbf = grads(st, range(5))
        per_head.append((rel(gh, g32), rel(gh, gbf), rel(gbf, g32)))
        print(f"tap relu{k + 1}: HIP vs fp32 oracle {per_head[-1][0]:.2e}   HIP vs bf16-operand oracle {per_head[-1][1]:.2e}   "
              f"bf16-operand oracle vs fp32 oracle {per_head[-1][2]:.2e}")
    gh, g32, gbf = grads(sd, range(5))
    print(f"all taps:   HIP vs fp32 oracle {rel(gh, g32):.2e}   HIP vs bf16-operand oracle {rel(gh, gbf):.2e}   bf16-operand oracle vs fp32 {rel(gbf, g32):.2e}")
    # measured on MI355X (r04):        HIP vs fp32    HIP vs bf16-operand oracle    bf16-operand oracle vs fp32
    #   tap relu1_2                      2.4e-2           4.7e-5                          2.4e-2
    #   tap relu2_2                      7.6e-2           1.2e-3                          7.6e-2
    #   tap relu3_3                      1.6e-1           1.8e-2                          1.6e-1
    #   tap relu4_3                      2.2e-1           6.6e-2                          2.2e-1
    #   tap relu5_3                      2.9e-1           1.8e-1                          3.1e-1
    #   all five (LPIPS)                 5.0e-2           1.7e-2                          5.1e-2
    # i.e. at every tap the HIP gradient is exactly as far from fp32 as ANY bf16-operand evaluation of the same network is, the
    # distance grows with the number of bf16 conv layers (and ReLU masks) the gradient crosses, and the two bf16 evaluations agree
    # to 5e-5 on the first slice (same roundings) and decorrelate through mask flips deeper down.  The sum is dominated by the
    # shallow taps (largest gradients), hence 5e-2 overall.
    assert rel(gh, g32) < 8e-2 and rel(gh, gbf) < 3e-2
    for (e32, ebf, eref) in per_head:
        assert e32 < 1.15 * eref + 1e-3                      # never worse than operand rounding alone explains
    assert per_head[0][1] < 1e-3 and per_head[1][1] < 1e-2   # shallow slices: the two bf16 evaluations round the same values
    assert all(per_head[k][0] < per_head[k + 1][0] for k in range(4))      # depth owns the error


def test_perceptual_loss_wrapper_and_errors():
    from sparsefusion_amd.lpips import LPIPS, PerceptualLoss
    loss = PerceptualLoss('vgg', device=DEV)
    a, b = torch.rand(1, 64, 64, 3, device=DEV), torch.rand(1, 64, 64, 3, device=DEV)      # channels-last is permuted (:33-35)
    d = loss(a, b, normalize=True)
    assert d.shape == (1, 1, 1, 1) and float(d) > 0
    with pytest.raises(NotImplementedError):
        LPIPS(net='alex')
    with pytest.raises(RuntimeError):
        loss.model(torch.rand(1, 3, 64, 64), torch.rand(1, 3, 64, 64))          # CPU tensors: no fallback
    p = torch.rand(1, 3, 64, 64, device=DEV, requires_grad=True)
    d1 = loss.model(p, b.permute(0, 3, 1, 2))
    _ = loss.model(p, b.permute(0, 3, 1, 2))
    with pytest.raises(RuntimeError):
        d1.sum().backward()                                           # stale: a later forward reused the arena

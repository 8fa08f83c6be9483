"""FusedAdam (one HIP launch per step) vs torch.optim.Adam on the same device: the plain-PyTorch reference of the op."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_fused_adam_matches_torch_adam_with_groups_and_scheduler():
    from sparsefusion_amd.optim import FusedAdam
    g = torch.Generator().manual_seed(0)
    shapes = [(929336, 2), (64, 32), (64,), (64, 64), (64,), (4, 64), (4,)]          # the NGP parameter set
    a = [torch.nn.Parameter((torch.randn(s, generator=g) * 0.1).to(DEV)) for s in shapes]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]

    def groups(ps):
        return [{'params': ps[:1], 'lr': 5e-3}, {'params': ps[1:], 'lr': 5e-4}]       # get_params: table at 10x lr

    ref, fused = torch.optim.Adam(groups(a)), FusedAdam(groups(b))
    s_ref = torch.optim.lr_scheduler.StepLR(ref, step_size=3, gamma=0.2)
    s_fused = torch.optim.lr_scheduler.StepLR(fused, step_size=3, gamma=0.2)
    for it in range(8):
        for pa, pb in zip(a, b):
            grad = torch.randn(pa.shape, generator=g).to(DEV) * (10.0 ** (it % 3 - 1))
            if it == 2 and pa.dim() == 1:
                grad = torch.zeros_like(grad)                                           # exact zeros: v stays tiny, eps dominates
            pa.grad, pb.grad = grad.clone(), grad.clone()
        if it == 5:
            b[3].grad = None                                                            # a parameter without gradient is skipped
            a[3].grad = None
        ref.step(); fused.step()
        s_ref.step(); s_fused.step()
        for pa, pb in zip(a, b):
            assert torch.allclose(pa, pb, rtol=2e-6, atol=3e-7), (it, (pa - pb).abs().max())     # a few ulp of |p| ~ 0.5
    sa, sb = ref.state_dict(), fused.state_dict()
    assert sa['param_groups'][0]['lr'] == sb['param_groups'][0]['lr']
    for k in sa['state']:
        assert int(sa['state'][k]['step']) == int(sb['state'][k]['step'])
        assert torch.allclose(sa['state'][k]['exp_avg'], sb['state'][k]['exp_avg'], rtol=1e-5, atol=1e-6)
        assert torch.allclose(sa['state'][k]['exp_avg_sq'], sb['state'][k]['exp_avg_sq'], rtol=1e-5, atol=1e-8)
    # the torch optimizer can resume from the fused state and vice versa
    ref2 = torch.optim.Adam(groups(a)); ref2.load_state_dict(sb)
    fused2 = FusedAdam(groups(b)); fused2.load_state_dict(sa)


def test_fused_adam_rejects_unsupported():
    from sparsefusion_amd.optim import FusedAdam
    p = torch.nn.Parameter(torch.zeros(4))
    with pytest.raises(NotImplementedError):
        FusedAdam([p], weight_decay=0.1)
    with pytest.raises(NotImplementedError):
        FusedAdam([p], amsgrad=True)
    opt = FusedAdam([p])
    p.grad = torch.ones(4)
    with pytest.raises(RuntimeError):
        opt.step()                                                                      # CPU parameter: no fallback

"""Adam for the NGP field on one HIP launch per step (csrc/adam.hip).

Drop-in for `torch.optim.Adam(ngp_network.get_params(lr=5e-4))` (sparsefusion/distillation.py:165): same constructor
arguments, param groups (per-group `lr`, so `torch.optim.lr_scheduler.StepLR` of :166 works unchanged), state keys
(`step`, `exp_avg`, `exp_avg_sq`) and arithmetic; amsgrad / weight_decay / maximize are not on the reference path."""
import math

import torch

from . import _lib


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, **unsupported):
        if weight_decay != 0 or amsgrad or any(unsupported.get(k) for k in ("maximize", "capturable", "differentiable")):
            raise NotImplementedError("FusedAdam covers the reference configuration only: Adam(params, lr) with defaults")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        # tensors that share (betas, eps, step count) go into one launch, at most SF_ADAM_MAX_TENSORS at a time
        batches = {}
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.grad.is_sparse or p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("FusedAdam: dense contiguous float32 parameters only")
                _lib.require_cuda(p, p.grad)
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                t = int(st["step"])
                batches.setdefault((b1, b2, group["eps"], t), []).append((p, st, group["lr"] / (1 - b1 ** t)))
        lib = _lib.lib()
        for (b1, b2, eps, t), items in batches.items():
            for k in range(0, len(items), _lib.SF_ADAM_MAX_TENSORS):
                chunk = items[k:k + _lib.SF_ADAM_MAX_TENSORS]
                a = _lib.SfAdamArgs()
                a.n_tensors, a.beta1, a.beta2, a.eps = len(chunk), b1, b2, eps
                a.bias_correction2_sqrt = math.sqrt(1 - b2 ** t)
                a.one_minus_beta1, a.one_minus_beta2 = 1 - b1, 1 - b2
                keep = []
                for j, (p, st, step_size) in enumerate(chunk):
                    g = p.grad.contiguous()
                    keep.append(g)
                    a.t[j].param, a.t[j].grad = p.data_ptr(), g.data_ptr()
                    a.t[j].exp_avg, a.t[j].exp_avg_sq = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
                    a.t[j].n, a.t[j].step_size = p.numel(), step_size
                _lib.check(lib.sf_adam_multi(a, _lib.stream_ptr()), "adam_multi")
        return loss

"""Instant-NGP field (tiled grid 16x2 -> MLP 32-64-64-4) on the HIP backend.

Same constructor, state-dict keys and optimiser grouping as the reference's NeRFNetwork
(external/nerf/network_grid.py:36-234): `encoder.embeddings [929336,2]`, `encoder.offsets [17]`,
`sigma_net.net.{0,1,2}.{weight,bias}`, `aabb_train`, `aabb_infer`.  sigma = trunc_exp(h0 +
5 exp(-|x|^2/0.08)), albedo = sigmoid(h1..3) (:69-88).

Two execution routes, both on the GPU library:
  * `render` / `run` (inherited): the fused render node (renderer.py);
  * `common_forward` / `density` / `forward` on arbitrary points: fused sf_ngp_density when no
    gradient is needed, otherwise HIP grid-encode op + torch linears (differentiable)."""
import ctypes as C

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib
from ..gridencoder import GridEncoder
from .renderer import NeRFRenderer, _FieldHandle


class MLP(nn.Module):
    def __init__(self, dim_in, dim_out, dim_hidden, num_layers, bias=True):
        super().__init__()
        self.dim_in, self.dim_out, self.dim_hidden, self.num_layers = dim_in, dim_out, dim_hidden, num_layers
        widths = [dim_in] + [dim_hidden] * (num_layers - 1) + [dim_out]
        self.net = nn.ModuleList(nn.Linear(a, b, bias=bias) for a, b in zip(widths[:-1], widths[1:]))

    def forward(self, x):
        for i, layer in enumerate(self.net):
            x = layer(x)
            if i + 1 < self.num_layers:
                x = F.relu(x)
        return x


class _TruncExp(torch.autograd.Function):
    """exp forward, gradient clamped to exp(clamp(x, -15, 15)) (external/ngp_activation.py:10-23)."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * torch.exp(x.clamp(-15, 15))


trunc_exp = _TruncExp.apply


class NeRFNetwork(NeRFRenderer):
    def __init__(self, opt, num_layers=3, hidden_dim=64, num_layers_bg=2, hidden_dim_bg=64):
        super().__init__(opt)
        if num_layers != 3 or hidden_dim != 64:
            raise NotImplementedError("the fused kernels are built for the reference 32-64-64-4 MLP")
        self.num_layers, self.hidden_dim = num_layers, hidden_dim
        # get_encoder('tiledgrid', input_dim=3, log2_hashmap_size=16, desired_resolution=2048*bound)
        # (network_grid.py:50, ngp_encoder.py:50-79)
        self.encoder = GridEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=16,
                                   desired_resolution=2048 * self.bound, gridtype='tiled', align_corners=False)
        self.in_dim = self.encoder.output_dim
        self.sigma_net = MLP(self.in_dim, 4, hidden_dim, num_layers, bias=True)
        self.bg_net = None
        self._handle = None

    # ---- renderer hooks
    def _field_handle(self):
        if self._handle is None:
            self._handle = _FieldHandle(self)
        return self._handle

    def _field_params(self):
        lin = self.sigma_net.net
        return (self.encoder.embeddings, lin[0].weight, lin[0].bias, lin[1].weight, lin[1].bias, lin[2].weight,
                lin[2].bias)

    # ---- point queries
    def gaussian(self, x):
        d = (x ** 2).sum(-1)
        return 5 * torch.exp(-d / (2 * 0.2 ** 2))

    def common_forward(self, x):
        """x [N,3] in [-bound,bound] -> sigma [N], albedo [N,3]."""
        needs_grad = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))
        if not needs_grad:
            _lib.require_cuda(x)
            xs = x.detach().reshape(-1, 3).float().contiguous()
            sigma = torch.empty(xs.shape[0], dtype=torch.float32, device=xs.device)
            albedo = torch.empty(xs.shape[0], 3, dtype=torch.float32, device=xs.device)
            params = [p.detach().contiguous() for p in self._field_params()]
            f = self._field_handle().struct(params)
            rc = _lib.lib().sf_ngp_density(C.byref(f), _lib.ptr(xs), xs.shape[0], _lib.ptr(sigma), _lib.ptr(albedo),
                                           _lib.stream_ptr())
            _lib.check(rc, "ngp_density")
            return sigma.view(x.shape[:-1]), albedo.view(*x.shape[:-1], 3)
        h = self.sigma_net(self.encoder(x, bound=self.bound))
        return trunc_exp(h[..., 0] + self.gaussian(x)), torch.sigmoid(h[..., 1:])

    def forward(self, x, d, l=None, ratio=1, shading='albedo'):
        if shading != 'albedo':
            raise NotImplementedError("only shading='albedo' is on the distillation path")
        sigma, color = self.common_forward(x)
        return sigma, color, None

    def density(self, x):
        sigma, albedo = self.common_forward(x)
        return {'sigma': sigma, 'albedo': albedo}

    def get_params(self, lr):
        """Adam groups of network_grid.py:223-234: table at 10x lr, MLP at lr."""
        return [{'params': self.encoder.parameters(), 'lr': lr * 10},
                {'params': self.sigma_net.parameters(), 'lr': lr}]

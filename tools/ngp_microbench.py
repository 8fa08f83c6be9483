"""Time the fused NGP render (forward, forward+backward) at the BASELINE size: 128x128 rays, 64+64 samples."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ngp_ref  # noqa: E402  (parameter initialiser only)
from sparsefusion_amd.nerf import NeRFNetwork, get_default_torch_ngp_opt  # noqa: E402

dev = "cuda:0"
p = ngp_ref.init_params(bound=4, seed=1, table_std=0.5, sigma_bias=-3.0)
net = NeRFNetwork(get_default_torch_ngp_opt())
net.load_state_dict({k: p[k] for k in net.state_dict().keys()})
net = net.to(dev).train()
o, d = ngp_ref.circle_rays(128, view=7)
o, d = o[None].to(dev), d[None].to(dev)
kw = dict(staged=False, perturb=True, bg_color=0, shading='albedo', **vars(net.opt))


def run(backward, iters):
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(iters):
        net.zero_grad(set_to_none=True)
        with torch.set_grad_enabled(backward):
            r = net.render(o, d, **kw)
            if backward:
                (r["image"].mean() + r["weights_sum"].mean()).backward()
    torch.cuda.synchronize()
    return (time.time() - t) / iters * 1e3


run(True, 3)
print(f"render fwd      : {run(False, 20):.3f} ms")
print(f"render fwd+bwd  : {run(True, 20):.3f} ms")

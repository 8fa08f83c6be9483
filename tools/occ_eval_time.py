"""Time the occupancy-grid EVALUATION render (cuda_ray=True, eval mode) at 256 x 256 rays: one fused launch
(sf_ngp_render_occ_eval) instead of the reference's host loop of march / network / composite rounds."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import ngp_ref
from sparsefusion_amd.nerf import NeRFNetwork, get_default_torch_ngp_opt
dev = "cuda:0"
opt = get_default_torch_ngp_opt()
opt.cuda_ray = True
p = ngp_ref.init_params(bound=4, seed=1, table_std=0.5, sigma_bias=-3.0)
net = NeRFNetwork(opt)
sd = net.state_dict()
sd.update({k: p[k] for k in p if k in sd})
net.load_state_dict(sd)
net = net.to(dev).train()
for _ in range(3):
    net.update_extra_state()
net.eval()
o, d = ngp_ref.circle_rays(256, view=3)
o, d = o[None].to(dev), d[None].to(dev)
with torch.no_grad():
    for _ in range(3):
        r = net.render(o, d, staged=False, perturb=False, bg_color=0, shading='albedo', **vars(opt))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        r = net.render(o, d, staged=False, perturb=False, bg_color=0, shading='albedo', **vars(opt))
    torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
print(f"occupancy-grid eval render 256x256 rays: {ms:.3f} ms per view (one launch + near/far), mean opacity {float(r['weights_sum'].mean()):.4f}, "
      f"occupied cells {int(sum(bin(int(b)).count('1') for b in net.density_bitfield.cpu().tolist()))}")

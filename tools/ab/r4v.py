"""(r04, measured slower and NOT kept: profiles/r04_unet_fconv_workgroups_ab.log; the attribute this script sets has to be re-added to
_Plan._fused_geometry -- `S = first d with MT * n_frags * d >= u.fconv_min_workgroups` -- before it does anything.)
A/B of Unet.fconv_min_workgroups (input-channel slices of the 4x4 convs: 256 = one workgroup per CU, 512 = two) at batch B:
values against the default plan, eval time in the sampler path.   usage: r4v.py B"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sparsefusion_amd.unet import Unet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
net = Unet(channels=4, dim=256, dim_mults=(1, 2, 4, 4), num_resnet_blocks=(2, 2, 2, 2), layer_attns=(False, False, False, True),
           layer_cross_attns=(False,) * 4, cond_images_channels=256, attn_pool_text=False).to(dev)
g = torch.Generator().manual_seed(0)
x, cond = torch.randn(B, 4, 32, 32, generator=g).to(dev), torch.randn(B, 256, 32, 32, generator=g).to(dev)
ls = torch.linspace(-3, 3, 4, device=dev)
ref = None
for wgs in (256, 512, 1024, 256):
    net.fconv_min_workgroups = wgs
    net.drop_plans()
    ctx = net.begin_sampling(cond, ls)
    y = net.eval_prepared(ctx, x, 0).clone()
    if ref is None:
        ref = y
    for _ in range(20):
        net.eval_prepared(ctx, x, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        net.eval_prepared(ctx, x, 0)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 200 * 1e3
    print(f"B={B} fconv_min_workgroups={wgs}: eval {ms:.4f} ms, body ops {ctx['plan'].n_body_ops}, rel diff vs default {float((y - ref).norm() / ref.norm()):.2e}")

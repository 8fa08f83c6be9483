"""LPIPS distance / gradient error against the fp32 oracle for the bf16 and the IEEE-half operand build, over gradient scales (r06)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import lpips_ref
from sparsefusion_amd.lpips import LPIPS
DEV = "cuda:0"
sd = lpips_ref.init_state(seed=0)
for R, B in ((256, 1), (64, 2)):
    g = torch.Generator().manual_seed(R + B)
    base = torch.rand(B, 3, R, R, generator=g)
    pred = (base + 0.15 * torch.randn(B, 3, R, R, generator=g)).clamp(0, 1)
    p_ref = pred.clone().requires_grad_(True)
    d_ref = lpips_ref.lpips(sd, p_ref, base, normalize=True)
    d_ref.sum().backward()
    print(f"R={R} B={B}: |grad| max {float(p_ref.grad.abs().max()):.3e} median {float(p_ref.grad.abs().median()):.3e}")
    for operand, scales in ((None, (1.0,)), ("f16", (1.0, 16.0, 256.0, 4096.0, 65536.0))):
        for sc in scales:
            net = LPIPS(net='vgg')
            net.load_state_dict(sd, strict=True)
            net = net.to(DEV).set_operand(operand, sc)
            p = pred.to(DEV).requires_grad_(True)
            d = net(p, base.to(DEV), normalize=True)
            d.sum().backward()
            rel_d = float(((d.detach().cpu() - d_ref.detach()).abs() / d_ref.detach().abs()).max())
            gr = p.grad.cpu()
            rel_g = float((gr - p_ref.grad).norm() / p_ref.grad.norm())
            cos = torch.nn.functional.cosine_similarity(gr.flatten().double(), p_ref.grad.flatten().double(), dim=0).item()
            print(f"  operand {operand or 'bf16'} grad_scale {sc:8.0f}: distance rel {rel_d:.2e}  gradient rel L2 {rel_g:.3e} cos {cos:.6f} finite {bool(torch.isfinite(gr).all())}")

"""Generate tests/golden/*.pt by running the REAL reference (read-only /root/reference) on CPU.

Run in the dev container only:  python tests/golden/make_golden.py [ngp] [unet] [plms]
Parameters come from the deterministic oracle initialisers (seeded torch CPU generators), so the
fixtures only need to hold inputs that are not re-derivable plus the reference's outputs."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ngp_ref, ref_loader  # noqa: E402


def make_ngp():
    """Reference NeRFNetwork.render (renderer_df.py:643 -> run :310) on 256 synthetic rays, fwd + bwd."""
    opt = ref_loader.ngp_opt()
    out = {}
    for name, (seed, std, sbias, view, unit) in {
            "teacher": (1, 0.5, -3.0, 3, False), "default_init": (2, 1e-4, None, 11, True)}.items():
        p = ngp_ref.init_params(bound=4, seed=seed, table_std=std, sigma_bias=sbias)
        net = ref_loader.reference_ngp().train()
        net.load_state_dict({k: p[k] for k in net.state_dict().keys()})
        o, d = ngp_ref.circle_rays(16, view=view, unit_dir=unit)
        o[5] = torch.tensor([20.0, 20.0, 20.0]); d[5] = torch.tensor([1.0, 0.0, 0.0])     # a ray that misses the box
        N = o.shape[0]
        noise_seed = 1000 + seed
        torch.manual_seed(noise_seed)
        _ = torch.randn(3); uc = torch.rand(N, 64); uf = torch.rand(N, 64)
        torch.manual_seed(noise_seed)                   # the reference draws the same three tensors in this order
        r = net.render(o[None], d[None], staged=False, perturb=True, bg_color=0, ambient_ratio=1.0, shading='albedo',
                       force_all_rays=True, **vars(opt))
        g = torch.Generator().manual_seed(7)
        gI, gW = torch.randn(N, 3, generator=g), torch.randn(N, generator=g)
        ((r['image'][0] * gI).sum() + (r['weights_sum'] * gW).sum()).backward()
        ge = net.encoder.embeddings.grad
        idx = torch.randperm(ge.shape[0], generator=g)[:4096].clone()
        offs = p["encoder.offsets"].tolist()
        out[name] = dict(
            cfg=dict(seed=seed, table_std=std, sigma_bias=sbias, view=view, unit_dir=unit, noise_seed=noise_seed),
            rays_o=o, rays_d=d, u_coarse=uc, u_fine=uf, g_image=gI, g_ws=gW,
            image=r['image'][0].detach(), weights_sum=r['weights_sum'].detach(), depth=r['depth'][0].detach(),
            mask=r['mask'][0],
            grad_mlp={k: v.grad.clone() for k, v in net.sigma_net.named_parameters()},
            grad_table_rows=idx, grad_table_vals=ge[idx].clone(),
            grad_table_level_abs=torch.stack([ge[offs[l]:offs[l + 1]].abs().sum() for l in range(16)]),
            grad_table_norm=ge.norm())
        # eval-mode render (det sampling, no perturb) of the same rays, as render_batched does
        net.eval()
        with torch.no_grad():
            re = net.render(o[None], d[None], staged=False, perturb=False, bg_color=1, ambient_ratio=1.0,
                            shading='albedo', force_all_rays=True, **vars(opt))
        out[name]["eval_image"] = re['image'][0]
        out[name]["eval_weights_sum"] = re['weights_sum']
    torch.save(out, os.path.join(HERE, "ngp_render.pt"))
    print("wrote ngp_render.pt", {k: float(v['weights_sum'].mean()) for k, v in out.items()})


if __name__ == "__main__":
    what = sys.argv[1:] or ["ngp"]
    if "ngp" in what:
        make_ngp()
    if "unet" in what or "plms" in what:
        from make_golden_unet import make_unet, make_plms
        if "unet" in what:
            make_unet()
        if "plms" in what:
            make_plms()

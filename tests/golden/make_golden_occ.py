"""Golden vectors of the occupancy-grid (cuda_ray=True) path from the REAL reference renderer (dev container only).

The reference's own `NeRFRenderer.update_extra_state` / `run_cuda` (external/nerf/renderer_df.py:471-638) run on CPU;
only the native entry points underneath are the C oracle (oracle/ref_loader.py stubs).  Every random draw is seeded
through torch's CPU generator; the consumer re-creates the same streams (see tests/test_gpu_occ_render.py)."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ngp_ref, ref_loader  # noqa: E402

CFG = dict(seed=1, table_std=3.0, sigma_bias=-1.0, view=3, n_side=32, grid_noise_seed=77, march_noise_seed=78)


def bitfield_checksum(bits):
    w = (torch.arange(bits.numel()) % 251 + 1).long()
    return int((bits.long() * w).sum())


def main():
    opt = ref_loader.ngp_opt()
    opt.cuda_ray = True
    p = ngp_ref.init_params(bound=4, seed=CFG["seed"], table_std=CFG["table_std"], sigma_bias=CFG["sigma_bias"])
    net = ref_loader.reference_ngp(opt).train()
    sd = net.state_dict()
    sd.update({k: p[k] for k in p if k in sd})
    net.load_state_dict(sd)
    o, d = ngp_ref.circle_rays(CFG["n_side"], view=CFG["view"])
    N = o.shape[0]
    g_grid = torch.Generator().manual_seed(CFG["grid_noise_seed"])
    real_rand_like = torch.rand_like
    torch.rand_like = lambda t, **kw: torch.rand(t.shape, generator=g_grid)          # update_extra_state's jitter (:618)
    out = dict(cfg=CFG)
    try:
        net.update_extra_state()
        out["mean_density_1"] = net.mean_density
        out["popcount_1"] = int(sum(bin(int(b)).count("1") for b in net.density_bitfield.tolist()))
        # training render 1: all rays, seeded perturbation
        torch.manual_seed(CFG["march_noise_seed"])
        r1 = net.render(o[None], d[None], staged=False, perturb=True, bg_color=0, ambient_ratio=1.0, shading='albedo',
                        force_all_rays=True, **vars(opt))
        g = torch.Generator().manual_seed(7)
        gI, gW = torch.randn(N, 3, generator=g), torch.randn(N, generator=g)
        ((r1['image'][0] * gI).sum() + (r1['weights_sum'][0] * gW).sum()).backward()
        ge = net.encoder.embeddings.grad
        idx = torch.randperm(ge.shape[0], generator=g)[:4096].clone()
        out.update(image_1=r1['image'][0].detach(), weights_sum_1=r1['weights_sum'][0].detach(), depth_1=r1['depth'][0].detach(),
                   step_counter_1=net.step_counter[0].clone(), grad_table_rows=idx, grad_table_vals=ge[idx].clone(),
                   grad_table_norm=ge.norm(), grad_mlp={k: v.grad.clone() for k, v in net.sigma_net.named_parameters()})
        net.update_extra_state()
        out["mean_density_2"], out["mean_count_2"], out["iter_density"] = net.mean_density, net.mean_count, net.iter_density
        out["popcount_2"] = int(sum(bin(int(b)).count("1") for b in net.density_bitfield.tolist()))
        out["bitfield_checksum_2"] = bitfield_checksum(net.density_bitfield)
        gi = torch.randperm(net.density_grid.numel(), generator=g)[:4096].clone()
        out["grid_rows"], out["grid_vals"] = gi, net.density_grid.view(-1)[gi].clone()
        # training render 2: budgeted by mean_count (force_all_rays=False), no perturbation
        with torch.no_grad():
            r2 = net.render(o[None], d[None], staged=False, perturb=False, bg_color=0, ambient_ratio=1.0, shading='albedo',
                            force_all_rays=False, **vars(opt))
        out.update(image_2=r2['image'][0], weights_sum_2=r2['weights_sum'][0], step_counter_2=net.step_counter[0].clone())
        net.eval()
        with torch.no_grad():
            re = net.render(o[None], d[None], staged=False, perturb=False, bg_color=1, ambient_ratio=1.0, shading='albedo',
                            force_all_rays=True, **vars(opt))
        out.update(eval_image=re['image'][0], eval_weights_sum=re['weights_sum'][0], eval_depth=re['depth'][0])
    finally:
        torch.rand_like = real_rand_like
    torch.save(out, os.path.join(HERE, "ngp_occ_render.pt"))
    print({k: v for k, v in out.items() if not torch.is_tensor(v) and k not in ("grad_mlp",)})
    print("weights_sum mean", float(out["weights_sum_1"].mean()), float(out["eval_weights_sum"].mean()),
          "points", out["step_counter_1"].tolist(), out["step_counter_2"].tolist())


if __name__ == "__main__":
    main()

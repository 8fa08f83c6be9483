"""End-to-end parity of the distillation inner loop (sparsefusion/distillation.py:174-352): K optimiser iterations of the
hot path -- input-view NGP render fwd/bwd + Adam, novel-view render -> x2 bilinear -> SD-VAE encode -> PLMS (UNet) ->
SD-VAE decode -> (1 - alpha_bar) L1 + lambda LPIPS + opacity and entropy regularisers -> bwd + Adam -- run once on the
HIP path and once through the CPU oracle (oracle/ngp_ref + unet_ref + vae_ref + lpips_ref) with the SAME injected
noise, then compared the way BASELINE.json's north_star states the quality bar: the renders of the two trained fields
agree (PSNR between them >= 40 dB) and their PSNR against a common target differs by <= 0.1 dB.
Two sizes: the small configurations (32 x 32 rays, dim-64 UNet, 2-level VAE, 7-eval PLMS, 10 steps: the oracle side takes
well under a minute on the GPU box's host cores) and the configuration the benchmark times (128 x 128 rays, the canonical
400 M-parameter UNet and 83.65 M-parameter SD-VAE, 51-eval PLMS, 6 steps: ~3 minutes of oracle time on 32 cores)."""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import lpips_ref, ngp_ref, unet_ref, vae_ref
from unet_common import CONFIGS, state

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
T, Z_SCALE, LAMBDA_PERCEP, LR = 64, 0.18215, 0.1, 5e-4
K_STEPS, SIDE, MAX_THRES, UNET_CFG, COND_CH = 10, 32, 0.06, "small", 60          # set per test by _configure()


def _configure(name):
    global K_STEPS, SIDE, MAX_THRES, UNET_CFG, COND_CH
    K_STEPS, SIDE, MAX_THRES, UNET_CFG, COND_CH = {"small": (10, 32, 0.06, "small", 60), "canonical": (6, 128, 0.5, "canonical", 256)}[name]


def huber(x, y, scaling=0.1):
    return ((1 + (x - y) ** 2 / scaling ** 2).clamp(1e-4).sqrt() - 1) * scaling


def entropy(sil):
    """opacity entropy regulariser (distillation.py:237-241, :337-341)."""
    a = sil.clamp(1e-5, 1 - 1e-5)
    return (-a * torch.log2(a) - (1 - a) * torch.log2(1 - a)).mean()


def psnr(a, b):
    return -10.0 * math.log10(float(((a - b) ** 2).mean()) + 1e-20)


class Scene:
    def __init__(self):
        g = torch.Generator().manual_seed(123)
        self.rays_in = ngp_ref.circle_rays(SIDE, view=0)
        self.rays_nv = ngp_ref.circle_rays(SIDE, view=5)
        self.target_rgb = torch.rand(1, 3, SIDE, SIDE, generator=g)
        self.target_mask = (torch.rand(1, 1, SIDE, SIDE, generator=g) > 0.5).float()
        self.features = torch.randn(1, COND_CH, 32, 32, generator=g)
        n = SIDE * SIDE
        self.noise = []
        for _ in range(K_STEPS):
            self.noise.append(dict(
                uc_a=torch.rand(n, T, generator=g), uf_a=torch.rand(n, T, generator=g),
                uc_b=torch.rand(n, T, generator=g), uf_b=torch.rand(n, T, generator=g),
                plms=[torch.randn(1, 4, 32, 32, generator=g) for _ in range(unet_ref.plms_noise_count(MAX_THRES))]))
        self.ngp = ngp_ref.init_params(bound=4, seed=1, table_std=0.5, sigma_bias=-3.0)
        self.unet_sd = state(UNET_CFG)
        self.vae_cfg = vae_ref.SMALL if UNET_CFG == "small" else vae_ref.CANONICAL
        self.vae_sd = vae_ref.init_state(vae_ref.vae_param_spec(self.vae_cfg), seed=0)
        self.lpips_sd = lpips_ref.init_state(0)


def to_img(x, ch):
    return x.reshape(1, SIDE, SIDE, ch).permute(0, 3, 1, 2).contiguous()


def losses_a(img, sil, sc, dev="cpu"):
    return (huber(img, sc.target_rgb.to(dev)).abs().mean() + huber(sil, sc.target_mask.to(dev)).abs().mean()
            + 1e-3 * torch.sqrt(sil ** 2 + .01).mean() + 1e-3 * entropy(sil))


def run_oracle(sc):
    p = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 and "aabb" not in k else v) for k, v in sc.ngp.items()}
    table = [p["encoder.embeddings"]]
    mlp = [p[k] for k in p if k.startswith("sigma_net")]
    opt = torch.optim.Adam([{"params": table, "lr": LR * 10}, {"params": mlp, "lr": LR}])

    def render(rays, uc, uf):
        r = ngp_ref.render_run(p, rays[0], rays[1], u_coarse=uc, u_fine=uf, bg_color=0.0, training=True)
        return to_img(r["image"], 3), to_img(r["weights_sum"], 1)

    for k in range(K_STEPS):
        nz = sc.noise[k]
        img, sil = render(sc.rays_in, nz["uc_a"], nz["uf_a"])
        opt.zero_grad()
        losses_a(img, sil, sc).backward()
        opt.step()
        opt.zero_grad()
        img, sil = render(sc.rays_nv, nz["uc_b"], nz["uf_b"])
        img2 = F.interpolate(img, scale_factor=2, mode="bilinear")
        sil2 = F.interpolate(sil, scale_factor=2, mode="bilinear")
        with torch.no_grad():
            lat = vae_ref.encode_mode(sc.vae_sd, sc.vae_cfg, img2 * 2 - 1) * Z_SCALE
            x0, _, _, acp, _ = unet_ref.plms_sample(lambda a, b: unet_ref.unet_forward(sc.unet_sd, a, b, sc.features), lat, MAX_THRES,
                                                    nz["plms"])
            pred = ((vae_ref.decode(sc.vae_sd, sc.vae_cfg, x0 / Z_SCALE) + 1) * 0.5).clip(0.0, 1.0)
        loss = ((1 - acp).view(-1, 1, 1, 1) * (img2 - pred).abs()).mean() \
            + LAMBDA_PERCEP * lpips_ref.lpips(sc.lpips_sd, img2, pred, normalize=True).mean() \
            + 1e-3 * torch.sqrt(sil2 ** 2 + .01).mean() + 1e-3 * entropy(sil2)
        loss.backward()
        opt.step()
    with torch.no_grad():
        r = ngp_ref.render_run(p, sc.rays_nv[0], sc.rays_nv[1], u_coarse=None, u_fine=None, bg_color=0.0, training=False)
        a = ngp_ref.render_run(p, sc.rays_in[0], sc.rays_in[1], u_coarse=None, u_fine=None, bg_color=0.0, training=False)
    return to_img(r["image"], 3), to_img(a["image"], 3)


def run_gpu(sc):
    from sparsefusion_amd.lpips import LPIPS
    from sparsefusion_amd.nerf import NeRFNetwork, get_default_torch_ngp_opt
    from sparsefusion_amd.optim import FusedAdam
    from sparsefusion_amd.plms import PLMSSampler
    from sparsefusion_amd.unet import Unet
    from sparsefusion_amd.vae import AutoencoderKL
    from sparsefusion_amd.vldm import DDPM
    from sparsefusion_amd.utils.losses import fusion_loss, render_loss, upsample2x
    opt_cfg = get_default_torch_ngp_opt()
    ngp = NeRFNetwork(opt_cfg)
    ngp.load_state_dict({k: sc.ngp[k] for k in ngp.state_dict().keys()})
    ngp = ngp.to(DEV).train()
    opt = FusedAdam(ngp.get_params(lr=LR))
    unet = Unet(**CONFIGS[UNET_CFG], layer_cross_attns=(False,) * 4, attn_pool_text=False)
    unet.load_state_dict(sc.unet_sd, strict=True)
    vldm = DDPM(channels=4, unets=(unet,), image_sizes=(32,), timesteps=500, conditional=False, clip_output=True,
                dynamic_thresholding=False, clip_value=10).to(DEV)
    plms = PLMSSampler(vldm, 50)
    ddconfig = {k: v for k, v in sc.vae_cfg.items() if k != "embed_dim"}
    ddconfig["resolution"] = 2 * SIDE                            # the x2-upsampled render (the oracle's functional VAE is size-free)
    vae = AutoencoderKL(ddconfig=ddconfig, embed_dim=sc.vae_cfg["embed_dim"])
    vae.load_state_dict(sc.vae_sd, strict=True)
    vae = vae.to(DEV)
    lp = LPIPS()
    lp.load_state_dict(sc.lpips_sd)
    lp = lp.to(DEV)
    feats = sc.features.to(DEV)
    dv = lambda rays: (rays[0][None].to(DEV), rays[1][None].to(DEV))
    rin, rnv = dv(sc.rays_in), dv(sc.rays_nv)

    def render(rays, uc, uf, train=True):
        out = ngp.render(rays[0], rays[1], staged=False, perturb=train, bg_color=0, shading='albedo',
                         noise=dict(u_coarse=uc.to(DEV), u_fine=uf.to(DEV)) if train else None, **vars(opt_cfg))
        return to_img(out["image"], 3), to_img(out["weights_sum"], 1)

    for k in range(K_STEPS):
        nz = sc.noise[k]
        img, sil = render(rin, nz["uc_a"], nz["uf_a"])
        opt.zero_grad()
        render_loss(img, sil, sc.target_rgb.to(DEV), sc.target_mask.to(DEV), 1.0, 1.0, 1e-3, 1e-3).backward()   # the product's loss glue
        opt.step()
        opt.zero_grad()
        img, sil = render(rnv, nz["uc_b"], nz["uf_b"])
        img2, sil2 = upsample2x(img), upsample2x(sil)
        with torch.no_grad():
            lat = vae.encode(img2 * 2 - 1).mode() * Z_SCALE
            x0, _, _, acp = plms.sample(lat, cond_images=feats, use_tqdm=False, return_noise=True, max_thres=MAX_THRES,
                                        noises=[t.to(DEV) for t in nz["plms"]])
            pred = ((vae.decode(x0 / Z_SCALE) + 1) * 0.5).clip(0.0, 1.0)
        loss = fusion_loss(img2, sil2, pred, 1 - acp, 1e-3, 1e-3) + LAMBDA_PERCEP * lp(img2, pred, normalize=True).mean()
        loss.backward()
        opt.step()
    ngp.eval()
    with torch.no_grad():
        a, _ = render(rin, None, None, train=False)
        r, _ = render(rnv, None, None, train=False)
    return r.cpu(), a.cpu()


@pytest.mark.parametrize("size", ["small", "canonical"])
def test_k_distillation_steps_match_the_oracle(size):
    _configure(size)
    if size == "canonical":
        torch.set_num_threads(min(32, torch.get_num_threads()))
    sc = Scene()
    nv_ref, in_ref = run_oracle(sc)
    nv_gpu, in_gpu = run_gpu(sc)
    # the fields start from the same parameters: make sure the K steps moved them (the comparison is not vacuous)
    with torch.no_grad():
        r0 = ngp_ref.render_run(sc.ngp, sc.rays_nv[0], sc.rays_nv[1], u_coarse=None, u_fine=None, bg_color=0.0, training=False)
    moved = psnr(to_img(r0["image"], 3), nv_ref)
    between_nv, between_in = psnr(nv_gpu, nv_ref), psnr(in_gpu, in_ref)
    d_psnr = abs(psnr(in_gpu, sc.target_rgb) - psnr(in_ref, sc.target_rgb))
    print(f"after {K_STEPS} steps: PSNR(gpu, oracle) novel {between_nv:.1f} dB / input {between_in:.1f} dB; "
          f"|dPSNR vs target| {d_psnr:.4f} dB; PSNR(initial, trained) {moved:.1f} dB")
    # not vacuous: the K steps changed the render by far more than the two paths differ (r03 ran 2 steps at the benchmark's size: 46.7 dB moved vs a
    # 113 dB agreement; r04 runs 6)
    assert moved < (40.0 if size == "small" else 60.0) and between_nv - moved >= 30.0, "the optimisation did not move the field: vacuous comparison"
    assert between_nv >= 40.0 and between_in >= 40.0
    assert d_psnr <= 0.1
